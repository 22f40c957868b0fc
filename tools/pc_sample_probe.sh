cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
timeout 60 rocprofv3 --list-avail > /tmp/la.txt 2>&1; echo rc=$?; wc -l /tmp/la.txt; grep -n -i "sampl\|agent\|gpu:" /tmp/la.txt | head -40; head -5 /tmp/la.txt
timeout 60 rocprofv3-avail -h 2>&1 | head -30
timeout 60 rocprofv3-avail list --pc-sampling 2>&1 | head -40

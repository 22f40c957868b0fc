# round 6: BASELINE configs 0 and 4 (one GPU) re-measured on the final tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python tools/config0.py > gpurun_out/r06_config0_transcripts_1000.json 2> gpurun_out/r06_config0.log; tail -c 700 gpurun_out/r06_config0_transcripts_1000.json; echo
timeout 1500 python tools/scale_test.py > gpurun_out/r06_scale_config5_1gpu.json 2> gpurun_out/r06_scale.log; tail -c 900 gpurun_out/r06_scale_config5_1gpu.json; echo

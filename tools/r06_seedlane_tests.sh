# round 6: the GPU suite with the lane-per-read seeder in the tree (forced in the "lane" kernel variant), then the A/B bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_gpu_tests_seedlane.txt 2>&1; tail -5 gpurun_out/r06_gpu_tests_seedlane.txt
bash tools/r06_ab5_seed_lane.sh 2>&1 | tail -6

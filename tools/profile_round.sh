set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_bench.sh r3s --no-other-workloads
bash tools/profile_bench.sh r3t --workload train
bash tools/profile_bench.sh r3o --workload online
bash tools/profile_bench.sh r3v --workload voxel
timeout -k 10 300 python bench.py > gpurun_out/bench_r3_full.json 2> gpurun_out/bench_r3_full.err

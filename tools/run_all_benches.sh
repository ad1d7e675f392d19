# Every bench configuration once, on the current tree; lines go to gpurun_out/final_*.json
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
run() { name=$1; shift; timeout -k 10 300 python $R/bench.py "$@" > $O/final_$name.json 2> $O/final_$name.err || { tail -20 $O/final_$name.err; exit 1; }; cat $O/final_$name.json; }
run score_bf16
run score_fp32 --precision fp32
run p500_b8192 --points 500 --batch 8192 --no-cpu-baseline
run p500_b65536 --points 500 --batch 65536 --steps 50 --warmup 10 --no-cpu-baseline
run b64 --batch 64 --no-cpu-baseline
run train --workload train
run online --workload online
run voxel --workload voxel
timeout -k 10 300 python -c "import sys; sys.path.insert(0, '$R'); import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1 || { tail -20 $O/final_smoke.log; exit 1; }
tail -2 $O/final_smoke.log

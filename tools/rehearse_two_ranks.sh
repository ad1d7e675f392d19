# Two ranks on ONE GPU over gloo: rehearses the multi-process flow of bench.py (the driver's runs use RCCL, one rank
# per GPU).  Throughput numbers of this rehearsal mean nothing: the ranks share the device.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
export PNGPD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
for wl in score train online; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    $R/bench.py --gpus 2 --steps 20 --warmup 5 --workload $wl > $O/rehearse_$wl.log 2>&1 || { tail -30 $O/rehearse_$wl.log; exit 1; }
  grep '"metric"' $O/rehearse_$wl.log | cut -c1-260
done

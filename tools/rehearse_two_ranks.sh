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
# the strong-scaling switch: one global batch cut over the ranks (config 3's shape)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 \
  $R/bench.py --gpus 2 --steps 10 --warmup 3 --global-batch 16384 --points 500 --no-other-workloads > $O/rehearse_strong.log 2>&1 || { tail -30 $O/rehearse_strong.log; exit 1; }
grep '"metric"' $O/rehearse_strong.log | cut -c1-400
# One rank over RCCL (a one-GPU box cannot hold two): process-group creation on the device, device-side barrier and
# the MAX all-reduce of the timing, as in the driver's multi-GPU runs.
export PNGPD_BENCH_BACKEND=nccl PNGPD_BENCH_FORCE_GROUP=1
for wl in score train online; do
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 \
    $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --workload $wl > $O/rehearse_rccl_$wl.log 2>&1 || { tail -30 $O/rehearse_rccl_$wl.log; exit 1; }
  grep '"metric"' $O/rehearse_rccl_$wl.log | cut -c1-200
done

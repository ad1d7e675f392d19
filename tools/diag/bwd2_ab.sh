# Training step with the fused passes over the per-point layers (default) and with a kernel per step
# (PNGPD_TRAIN_BWD2=split PNGPD_TRAIN_FWD12=split),
# alternating on one box.
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in fused split; do
    if [ $v = split ]; then export PNGPD_TRAIN_BWD2=split PNGPD_TRAIN_FWD12=split; else unset PNGPD_TRAIN_BWD2 PNGPD_TRAIN_FWD12; fi
    echo "== $v"
    timeout -k 10 120 python -u bench.py --workload train --steps 40 --warmup 10 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
  done
done

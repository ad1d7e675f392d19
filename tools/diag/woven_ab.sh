# A/B of the two forms of the bf16 conv stack on one box, alternating, three rounds (same library, env switch).
set -u
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in phase woven; do
    PNGPD_CONV_STACK_BF16=$v timeout -k 10 200 python $R/bench.py --no-cpu-baseline --no-other-workloads "$@" | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('$v', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['frac'])" || exit 1
  done
done

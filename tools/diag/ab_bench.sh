# A/B of two builds of the library on one box: tools/diag/_obj/libpngpd_{prev,cur}.so, alternating
set -u
R=$GRAFT_REPO_ROOT
for v in cur prev cur prev cur prev; do
  cp $R/tools/diag/_obj/libpngpd_$v.so $R/pointnetgpd_amd/libpngpd.so
  timeout -k 10 200 python $R/bench.py --no-cpu-baseline "$@" | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('$v', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['frac'])" || exit 1
done
cp $R/tools/diag/_obj/libpngpd_cur.so $R/pointnetgpd_amd/libpngpd.so

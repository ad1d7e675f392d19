# A/B of builds of the library on one box, alternating: tools/diag/_obj/libpngpd_<name>.so for the names in
# $AB_NAMES (default "cur prev"), three rounds; bench arguments are passed through.
set -u
R=$GRAFT_REPO_ROOT
NAMES=${AB_NAMES:-"cur prev"}
cp $R/pointnetgpd_amd/libpngpd.so /tmp/libpngpd_saved.so
for round in 1 2 3; do
  for v in $NAMES; do
    cp $R/tools/diag/_obj/libpngpd_$v.so $R/pointnetgpd_amd/libpngpd.so
    timeout -k 10 200 python $R/bench.py --no-cpu-baseline "$@" | python3 -c "import sys,json; l=json.loads(sys.stdin.read()); print('$v', l['value'], l['ms_per_step'], l['roofline']['kernel_ms'], l['roofline']['frac'])" || exit 1
  done
done
cp /tmp/libpngpd_saved.so $R/pointnetgpd_amd/libpngpd.so

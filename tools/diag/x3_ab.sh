# alternate the variant libraries named on the command line (tools/diag/<name>.so), three rounds each;
# PREC selects the scoring mode timed (default bf16x3)
set -u
for round in 1 2 3; do
  for v in "$@"; do
    timeout -k 10 120 python -u tools/diag/x3_time.py --lib $v --precision ${PREC:-bf16x3} || exit 1
  done
done

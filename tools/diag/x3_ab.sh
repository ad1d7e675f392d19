# alternate the variant libraries named on the command line, three rounds each
set -u
for round in 1 2 3; do
  for v in "$@"; do
    timeout -k 10 120 python -u tools/diag/x3_time.py --lib $v || exit 1
  done
done

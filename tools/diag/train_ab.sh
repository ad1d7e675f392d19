set -u
for round in 1 2 3; do for v in "$@"; do timeout -k 10 120 python -u tools/diag/train_time.py --lib $v || exit 1; done; done

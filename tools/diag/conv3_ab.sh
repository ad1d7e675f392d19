# Kernel time of the training step's conv3 product/max kernel for the libraries tools/diag/<name>.so given, alternating.
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "$@"; do timeout -k 10 120 python -u tools/diag/conv3_time.py --lib $v || exit 1; done; done

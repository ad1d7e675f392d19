set -u
cd $GRAFT_REPO_ROOT
for r in 1 2; do for mode in x6 h3; do
  PNGPD_TRAIN_CONV3=$mode timeout -k 10 200 python tools/diag/train_time.py 2>&1 | grep "training step" | sed "s/product/$mode/" || exit 1
done; done
timeout -k 10 600 python -u -m pytest tests/test_train_gpu.py tests/test_dist_gpu.py tests/test_train_loop.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -4

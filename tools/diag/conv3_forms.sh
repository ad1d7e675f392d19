# Training-step time and kernel time with the three forms of the conv3 product (PNGPD_TRAIN_CONV3: f32, x6, unset = three
# products of float16 pieces), alternating on one box.
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for mode in f32 x6 default; do
  if [ $mode = default ]; then unset PNGPD_TRAIN_CONV3; else export PNGPD_TRAIN_CONV3=$mode; fi
  timeout -k 10 200 python tools/diag/train_time.py 2>&1 | grep "training step" | sed "s/product/$mode/" || exit 1
  timeout -k 10 200 python tools/diag/conv3_time.py 2>&1 | grep "kernel" || exit 1
done; done

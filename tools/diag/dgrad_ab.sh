# Training step with the dense part of conv3's input gradient on split-bfloat16 products (default) and on the float32
# matrix instruction (PNGPD_TRAIN_DGRAD=f32), alternating on one box.
set -u
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in x3 f32; do
    if [ $v = f32 ]; then export PNGPD_TRAIN_DGRAD=f32; else unset PNGPD_TRAIN_DGRAD; fi
    echo -n "$v: "; timeout -k 10 120 python -u tools/diag/train_time.py 2>&1 | grep -v amdgpu.ids || exit 1
  done
done

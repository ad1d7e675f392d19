// Pre-activations that the training step computes in one kernel and recomputes in another.
//
// conv1's output y1 = W1 x + b1 is never stored (the backward pass rebuilds it from the 12 bytes of the point), and
// the backward kernels of the two per-point BatchNorm + ReLU layers take the ReLU mask from the sign of the
// recomputed pre-activation instead of reading the stored activation (98 MB per pass at 192 000 x 128).  Both only
// work if forward and backward get the same bits, so the expressions are written once, here, with explicit fused
// multiply-adds: nothing is left to the compiler's contraction choices in different kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace pngpd {

__device__ __forceinline__ float conv1_pre(float w0, float w1, float w2, float b, float x0, float x1, float x2) {
    return __builtin_fmaf(w2, x2, __builtin_fmaf(w1, x1, __builtin_fmaf(w0, x0, b)));
}

// (y - mean) * rstd * gamma + beta; the activation is max(., 0) and its ReLU mask is (. > 0)
__device__ __forceinline__ float bn_pre(float y, float mu, float rs, float ga, float be) {
    return __builtin_fmaf((y - mu) * rs, ga, be);
}

}  // namespace pngpd

// Fully connected layers of the STN regressor and of the classifier head, exact f32 on MFMA.
//   fc_f32:  Y = act(X W^T + b) with eval-mode BatchNorm folded into W/b
//            (model/pointnet.py:36-37 and :187-188).
//   fc_tail: the last, tiny layer (256 -> 9 plus the flattened identity, model/pointnet.py:38-44;
//            256 -> k plus log_softmax, model/pointnet.py:189-190).
//   score:   softmax of the log-probabilities and first-index arg-max (main_test.py:80-81).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int FC_BM = 64, FC_BN = 64, FC_BK = 32, FC_LD = FC_BK + 1;

// 64x64 output tile per workgroup, 4 waves as 2x2, one 32x32 MFMA accumulator per wave.
// A operand lane map: row = lane&31, k = lane>>5; LDS rows padded to 33 floats so the 32 rows
// read by one half-wave fall on 32 different banks.
__global__ __launch_bounds__(256) void fc_f32_kernel(const float* __restrict__ X,
                                                     const float* __restrict__ W,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ Y, int M, int N, int K,
                                                     int relu) {
    __shared__ float sA[FC_BM * FC_LD];
    __shared__ float sB[FC_BN * FC_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FC_BM, n0 = blockIdx.x * FC_BN;

    // global -> LDS: each thread moves two float4 of A and two of B per k tile
    const int lrow = tid >> 3;        // 0..31 (+32 for the second)
    const int lcol = (tid & 7) * 4;   // 0..28

    f32x16 acc = {0};
    for (int k0 = 0; k0 < K; k0 += FC_BK) {
        float4 va[2], vb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = lrow + 32 * i;
            const int gm = m0 + row;
            va[i] = gm < M ? *(const float4*)(X + (size_t)gm * K + k0 + lcol) : make_float4(0, 0, 0, 0);
            vb[i] = *(const float4*)(W + (size_t)(n0 + row) * K + k0 + lcol);
        }
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* pa = sA + (lrow + 32 * i) * FC_LD + lcol;
            float* pb = sB + (lrow + 32 * i) * FC_LD + lcol;
            pa[0] = va[i].x; pa[1] = va[i].y; pa[2] = va[i].z; pa[3] = va[i].w;
            pb[0] = vb[i].x; pb[1] = vb[i].y; pb[2] = vb[i].z; pb[3] = vb[i].w;
        }
        __syncthreads();
        const float* ra = sA + (wm * 32 + r) * FC_LD + h;
        const float* rb = sB + (wn * 32 + r) * FC_LD + h;
#pragma unroll
        for (int s = 0; s < FC_BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[2 * s], rb[2 * s], acc, 0, 0, 0);
    }
    const int col = n0 + wn * 32 + r;
    const float bv = bias[col];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < M) {
            float v = acc[i] + bv;
            if (relu) v = __builtin_fmaxf(v, 0.f);
            Y[(size_t)row * N + col] = v;
        }
    }
}

// One wave per row; lane owns k = 4*lane + 256*c .. +3.  N <= 16 outputs, each a wave reduction.
__global__ __launch_bounds__(256) void fc_tail_kernel(const float* __restrict__ X,
                                                      const float* __restrict__ W,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ Y, int M, int N, int K,
                                                      int tail) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float out[16];
#pragma unroll
    for (int n = 0; n < 16; ++n) out[n] = 0.f;
    for (int kc = 0; kc < K; kc += 256) {
        const float4 x = *(const float4*)(X + (size_t)row * K + kc + 4 * lane);
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            if (n < N) {
                const float4 w = *(const float4*)(W + (size_t)n * K + kc + 4 * lane);
                out[n] = fmaf(x.x, w.x, fmaf(x.y, w.y, fmaf(x.z, w.z, fmaf(x.w, w.w, out[n]))));
            }
        }
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) out[n] += __shfl_xor(out[n], off);
        if (n < N) out[n] += bias[n];
    }
    if (tail == FC_TAIL_ADD_IDENTITY) {
#pragma unroll
        for (int n = 0; n < 16; ++n)
            if (n == 0 || n == 4 || n == 8) out[n] += 1.0f;
    } else if (tail == FC_TAIL_LOG_SOFTMAX) {
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < 16; ++n)
            if (n < N) mx = __builtin_fmaxf(mx, out[n]);
        float sum = 0.f;
#pragma unroll
        for (int n = 0; n < 16; ++n)
            if (n < N) sum += expf(out[n] - mx);
        const float lse = mx + logf(sum);
#pragma unroll
        for (int n = 0; n < 16; ++n) out[n] -= lse;
    }
    if (lane == 0) {
#pragma unroll
        for (int n = 0; n < 16; ++n)
            if (n < N) Y[(size_t)row * N + n] = out[n];
    }
}

__global__ void score_kernel(const float* __restrict__ logp, int B, int k, float* __restrict__ probs,
                             int64_t* __restrict__ pred) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float* lp = logp + (size_t)b * k;
    float mx = lp[0];
    int arg = 0;
    for (int j = 1; j < k; ++j)
        if (lp[j] > mx) { mx = lp[j]; arg = j; }
    float sum = 0.f;
    for (int j = 0; j < k; ++j) sum += expf(lp[j] - mx);
    for (int j = 0; j < k; ++j) probs[(size_t)b * k + j] = expf(lp[j] - mx) / sum;
    pred[b] = arg;
}

hipError_t launch_fc_f32(const float* X, const float* W, const float* b, float* Y, int M, int N,
                         int K, int relu, hipStream_t s) {
    if (!X || !W || !b || !Y || M < 1 || N % FC_BN != 0 || K % FC_BK != 0) return hipErrorInvalidValue;
    dim3 grid(N / FC_BN, (M + FC_BM - 1) / FC_BM);
    hipLaunchKernelGGL(fc_f32_kernel, grid, dim3(256), 0, s, X, W, b, Y, M, N, K, relu);
    return hipGetLastError();
}

hipError_t launch_fc_tail(const float* X, const float* W, const float* b, float* Y, int M, int N,
                          int K, int tail, hipStream_t s) {
    if (!X || !W || !b || !Y || M < 1 || N < 1 || N > 16 || K % 256 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fc_tail_kernel, dim3((M + 3) / 4), dim3(256), 0, s, X, W, b, Y, M, N, K, tail);
    return hipGetLastError();
}

hipError_t launch_score(const float* logp, int B, int k, float* probs, int64_t* pred,
                        hipStream_t s) {
    if (!logp || !probs || !pred || B < 1 || k < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(score_kernel, dim3((B + 255) / 256), dim3(256), 0, s, logp, B, k, probs, pred);
    return hipGetLastError();
}

}  // namespace pngpd

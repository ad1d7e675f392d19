// Weight preparation on the device: the same eval-mode BatchNorm folding (float64) and the same fragment images as
// pack.cpp builds on the host, computed by kernels straight from parameter tensors that already live in device
// memory (a model that has just been trained there).  Every output element is one thread's work: it inverts the
// image's index, reads its weight, folds and rounds exactly as the host code does - same operations in the same
// order, floating-point contraction off - so the two loaders give bit-identical models.
// (torch.nn.BatchNorm1d eval semantics, eps = 1e-5; model/pointnet.py:21-25,129-131,182-183.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "pack_device.h"

namespace pngpd {

namespace {

#pragma clang fp contract(off)

__device__ __forceinline__ double fold_scale(const UnitDev& u, int o) {
    return u.gamma != nullptr ? (double)u.gamma[o] / sqrt((double)u.var[o] + 1e-5) : 1.0;
}

__device__ __forceinline__ float folded_w(const UnitDev& u, int o, int i, int n_in) {
    return (float)((double)u.W[(size_t)o * n_in + i] * fold_scale(u, o));
}

__device__ __forceinline__ float folded_b(const UnitDev& u, int o) {
    double s = 1.0, shift = 0.0, mu = 0.0;
    if (u.gamma != nullptr) {
        s = fold_scale(u, o);
        shift = (double)u.beta[o];
        mu = (double)u.mean[o];
    }
    const double d = (double)u.b[o] - mu;
    const double p = d * s;
    return (float)(p + shift);
}

__device__ __forceinline__ uint16_t bf16_rne(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // keep NaN a NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ float bf16_value(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

__device__ __forceinline__ int acc_row_dev(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

}  // namespace

// One thread per element of the largest image (conv3: 1024 x 128); the smaller images are filled by the first
// threads.  Index arithmetic mirrors pack_stack().
__global__ __launch_bounds__(256) void k_pack_stack_device(UnitDev u1, UnitDev u2, UnitDev u3, StackImages o) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < C1) {
        o.w1tab[4 * idx + 0] = folded_w(u1, idx, 0, 3);
        o.w1tab[4 * idx + 1] = folded_w(u1, idx, 1, 3);
        o.w1tab[4 * idx + 2] = folded_w(u1, idx, 2, 3);
        o.w1tab[4 * idx + 3] = folded_b(u1, idx);
    }
    if (idx < B2P_FLOATS) {
        const int t = idx >> 5, h = (idx >> 4) & 1, i = idx & 15;
        o.b2p[idx] = folded_b(u2, 32 * t + acc_row_dev(i, h));
    }
    if (idx < C2) o.b2[idx] = folded_b(u2, idx);
    if (idx < C3) o.b3[idx] = folded_b(u3, idx);
    if (idx < C2 * C1) {
        {   // [T8][s2][lane64][j8]: channel 16T + l%16, k = 32s + 8(l/16) + j
            const int j = idx & 7, l = (idx >> 3) & 63, s = (idx >> 9) & 1, T = idx >> 10;
            const float w = folded_w(u2, 16 * T + (l & 15), 32 * s + 8 * (l >> 4) + j, C1);
            const uint16_t hi = bf16_rne(w);
            o.w2_bf16[idx] = hi;
            o.w2_lo[idx] = bf16_rne(w - bf16_value(hi));
        }
        {   // [t4][q8][lane64][e4]: channel 32t + l%32, k = 2(4q + e) + l/32
            const int e = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) & 7, t = idx >> 11;
            o.w2_f32[idx] = folded_w(u2, 32 * t + (l & 31), 2 * (4 * q + e) + (l >> 5), C1);
        }
    }
    if (idx < C3 * C2) {
        {   // [NT64][ks4][lane64][j8]: the k slot (quarter l/16, element j) of step ks is register j&3 of conv2's row
            // tile 2ks + (j>>2)
            const int j = idx & 7, l = (idx >> 3) & 63, ks = (idx >> 9) & 3, NT = idx >> 11;
            const int ch3 = 16 * NT + (l & 15);
            const int ch2 = 16 * (2 * ks + (j >> 2)) + 4 * (l >> 4) + (j & 3);
            const float w = folded_w(u3, ch3, ch2, C2);
            const uint16_t hi = bf16_rne(w);
            o.w3_bf16[idx] = hi;
            o.w3_lo[idx] = bf16_rne(w - bf16_value(hi));
        }
        {   // [nt32][q16][lane64][e4]
            const int e = idx & 3, l = (idx >> 2) & 63, q = (idx >> 8) & 15, nt = idx >> 12;
            const int ch3 = 32 * nt + (l & 31);
            const int ch2 = 32 * (q >> 2) + acc_row_dev(4 * (q & 3) + e, l >> 5);
            o.w3_f32[idx] = folded_w(u3, ch3, ch2, C2);
        }
    }
}

// A linear layer [N,K]: folded float32 weights and bias and, when wsplit is given (N % 32 == 0, K % 16 == 0), the
// split-bf16 image [nt][ks][plane][lane][j] of pack_fc_split().  One thread per weight.
__global__ __launch_bounds__(256) void k_pack_fc_device(UnitDev u, int N, int K, float* __restrict__ w,
                                                        float* __restrict__ b, uint16_t* __restrict__ wsplit) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx < N) b[idx] = folded_b(u, idx);
    if (idx >= N * K) return;
    w[idx] = folded_w(u, idx / K, idx % K, K);
    if (wsplit == nullptr) return;
    const int ksteps = K / 16;
    const int j = idx & 7, l = (idx >> 3) & 63, ks = (idx >> 9) % ksteps, nt = (idx >> 9) / ksteps;
    const float v = folded_w(u, 32 * nt + (l & 31), 16 * ks + 8 * (l >> 5) + j, K);
    const uint16_t hi = bf16_rne(v);
    const size_t base = (((size_t)nt * ksteps + ks) * 2) * 512 + (size_t)l * 8 + j;
    wsplit[base] = hi;
    wsplit[base + 512] = bf16_rne(v - bf16_value(hi));
}

hipError_t launch_pack_stack_device(const UnitDev& u1, const UnitDev& u2, const UnitDev& u3, const StackImages& o,
                                    hipStream_t s) {
    hipLaunchKernelGGL(k_pack_stack_device, dim3(C3 * C2 / 256), dim3(256), 0, s, u1, u2, u3, o);
    return hipGetLastError();
}

hipError_t launch_pack_fc_device(const UnitDev& u, int N, int K, float* w, float* b, uint16_t* wsplit, hipStream_t s) {
    if (N < 1 || K < 1 || (wsplit != nullptr && (N % 32 != 0 || K % 16 != 0))) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_pack_fc_device, dim3((N * K + 255) / 256), dim3(256), 0, s, u, N, K, w, b, wsplit);
    return hipGetLastError();
}

}  // namespace pngpd

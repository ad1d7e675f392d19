// Test-only probes for the gfx950 MFMA lane maps that the fused kernels rely on.
// Built into tests/_probe.so; never linked into the product library.
//
// Each probe runs ONE wave and moves data with the same lane->element formulas the
// product kernels use, so a wrong formula shows up as a mismatch against numpy.
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ int acc_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

// D[32x32] = A[32x16] * B[16x32], bf16 inputs rounded from the given floats.
__global__ void k_mfma_bf16(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[r * 16 + 8 * h + j];
        b[j] = (__bf16)B[(8 * h + j) * 32 + r];
    }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[acc_row(i, h) * 32 + r] = acc[i];
}

// D[32x32] = A[32x2] * B[2x32], exact f32.
__global__ void k_mfma_f32(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * 2 + h], B[h * 32 + r], acc, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[acc_row(i, h) * 32 + r] = acc[i];
}

// Two chained products with the first accumulator reused as the A operand of the second:
//   X[32x32] = W[32x16] * Hm[16x32]          (rows of X = "channels", cols = "points")
//   Z[32x32] = relu(X)^T * V[32x32]          (sum over X's row index)
// f32 variant: register i of X is the A operand of one 32x32x2 step whose two k slots
// (lane halves) are X rows acc_row(i,0) and acc_row(i,1).
__global__ void k_chain_f32(const float* W, const float* Hm, const float* V, float* Z) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 x = {0};
    for (int s = 0; s < 8; ++s)
        x = __builtin_amdgcn_mfma_f32_32x32x2f32(W[r * 16 + 2 * s + h], Hm[(2 * s + h) * 32 + r], x, 0, 0, 0);
    f32x16 z = {0};
    for (int i = 0; i < 16; ++i)
        z = __builtin_amdgcn_mfma_f32_32x32x2f32(fmaxf(x[i], 0.f), V[acc_row(i, h) * 32 + r], z, 0, 0, 0);
    for (int i = 0; i < 16; ++i) Z[acc_row(i, h) * 32 + r] = z[i];
}

// 16x16x32 shape: D[16x16] = A[16x32] * B[32x16].
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k_mfma_bf16_16(const float* A, const float* B, float* D) {
    const int l = threadIdx.x, c = l & 15, q = l >> 4;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)A[c * 32 + 8 * q + j];
        b[j] = (__bf16)B[(8 * q + j) * 16 + c];
    }
    f32x4 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * q + i) * 16 + c] = acc[i];
}

// Accumulators of two 16x16x32 products reused as the A operand of a third:
//   X[32x16] = W[32x32] * Hm[32x16]   (two row tiles T = 0,1; rows = "channels", cols = "points")
//   Z[16x16] = relu(X)^T * V[32x16]
// k slot (quarter q, elem j) of the second product <-> X row 16(j>>2) + 4q + (j&3).
__global__ void k_chain_bf16_16(const float* W, const float* Hm, const float* V, float* Z) {
    const int l = threadIdx.x, c = l & 15, q = l >> 4;
    bf16x8 b;
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)Hm[(8 * q + j) * 16 + c];
    f32x4 x[2];
    for (int T = 0; T < 2; ++T) {
        bf16x8 a;
        for (int j = 0; j < 8; ++j) a[j] = (__bf16)W[(16 * T + c) * 32 + 8 * q + j];
        x[T] = f32x4{0};
        x[T] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, x[T], 0, 0, 0);
    }
    bf16x8 xa, vb;
    for (int j = 0; j < 8; ++j) {
        xa[j] = (__bf16)fmaxf(x[j >> 2][j & 3], 0.f);
        const int k = 16 * (j >> 2) + 4 * q + (j & 3);
        vb[j] = (__bf16)V[k * 16 + c];
    }
    f32x4 z = {0};
    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, vb, z, 0, 0, 0);
    for (int i = 0; i < 4; ++i) Z[(4 * q + i) * 16 + c] = z[i];
}

__global__ void k_axpy(float* x, float a, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = a * x[i] + 1.0f;
}

extern "C" {
int probe_mfma_bf16(const float* A, const float* B, float* D, void* stream) {
    hipLaunchKernelGGL(k_mfma_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    return (int)hipGetLastError();
}
int probe_mfma_f32(const float* A, const float* B, float* D, void* stream) {
    hipLaunchKernelGGL(k_mfma_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    return (int)hipGetLastError();
}
int probe_chain_f32(const float* W, const float* Hm, const float* V, float* Z, void* stream) {
    hipLaunchKernelGGL(k_chain_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, W, Hm, V, Z);
    return (int)hipGetLastError();
}
int probe_mfma_bf16_16(const float* A, const float* B, float* D, void* stream) {
    hipLaunchKernelGGL(k_mfma_bf16_16, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D);
    return (int)hipGetLastError();
}
int probe_chain_bf16_16(const float* W, const float* Hm, const float* V, float* Z, void* stream) {
    hipLaunchKernelGGL(k_chain_bf16_16, dim3(1), dim3(64), 0, (hipStream_t)stream, W, Hm, V, Z);
    return (int)hipGetLastError();
}
int probe_axpy(float* x, float a, int n, void* stream) {
    hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, a, n);
    return (int)hipGetLastError();
}
}

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
// bf16 variant: X is packed to bf16 in registers, k order permuted as
//   slot (step s, half h, elem j) <-> X row 16s + 8(j>>2) + 4h + (j&3).
__global__ void k_chain_bf16(const float* W, const float* Hm, const float* V, float* Z) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)W[r * 16 + 8 * h + j];
        b[j] = (__bf16)Hm[(8 * h + j) * 32 + r];
    }
    f32x16 x = {0};
    x = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, x, 0, 0, 0);
    f32x16 z = {0};
    for (int s = 0; s < 2; ++s) {
        bf16x8 xa, vb;
        for (int j = 0; j < 8; ++j) {
            xa[j] = (__bf16)fmaxf(x[8 * s + j], 0.f);
            const int k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            vb[j] = (__bf16)V[k * 32 + r];
        }
        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa, vb, z, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) Z[acc_row(i, h) * 32 + r] = z[i];
}

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
int probe_chain_bf16(const float* W, const float* Hm, const float* V, float* Z, void* stream) {
    hipLaunchKernelGGL(k_chain_bf16, dim3(1), dim3(64), 0, (hipStream_t)stream, W, Hm, V, Z);
    return (int)hipGetLastError();
}
int probe_chain_f32(const float* W, const float* Hm, const float* V, float* Z, void* stream) {
    hipLaunchKernelGGL(k_chain_f32, dim3(1), dim3(64), 0, (hipStream_t)stream, W, Hm, V, Z);
    return (int)hipGetLastError();
}
int probe_axpy(float* x, float a, int n, void* stream) {
    hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, a, n);
    return (int)hipGetLastError();
}
}

// Bare bf16 MFMA loops on random register operands: what the chip sustains under load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(256, 2) void k32(const bf16x8* in, float* out, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) % 4096]; b[i] = in[(threadIdx.x * 16 + 8 + i) % 4096]; }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x16{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k], b[(k + n) & 7], acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k16(const bf16x8* in, float* out, int iters) {
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(threadIdx.x * 16 + i) % 4096]; b[i] = in[(threadIdx.x * 16 + 8 + i) % 4096]; }
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) acc[n] = f32x4{0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k], b[(k + n) & 7], acc[n], 0, 0, 0);
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += acc[n][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    bf16x8* d; float* o;
    hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4096 * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 512, iters = 4000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 2; ++which) {
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k32<2>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            else hipLaunchKernelGGL(k16<8>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = which == 0 ? (double)blocks * 4 * iters * 2 * 8 * 32768.0 : (double)blocks * 4 * iters * 8 * 8 * 16384.0;
            printf("%s: %.3f ms  %.1f TFLOP/s\n", which == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
        }
    }
    return 0;
}

// Lone-wave issue rate of a 32-MFMA stream shaped like the conv-stack consume loop
// (v_mfma_f32_16x16x32_bf16, 4 A fragments x 32 B fragments, 8 accumulation chains).
// One wave per SIMD; prints shader cycles per MFMA for several operand arrangements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V>
__global__ __launch_bounds__(512) void k_stream(const bf16x8* wsrc, const bf16x8* asrc, float* out,
                                                unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 bw[8][4];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) bw[n][k] = wsrc[(n * 4 + k) * 64 + lane];
    bf16x8 af[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) af[k] = asrc[k * 64 + lane];
    f32x4 acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float rm = 0.f, rm2 = (float)lane, rm3 = 1.f, rm4 = 2.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {           // accumulate in place, ks-major over all 8 chains
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], acc[n], 0, 0, 0);
        } else if (V == 1) {    // weights as the A operand instead of B
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[n][k], af[k], acc[n], 0, 0, 0);
        } else if (V == 2) {    // a single B fragment
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[0][0], acc[n], 0, 0, 0);
        } else if (V == 3) {    // 4 chains at a time (two groups), as the product kernel
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[4 * g + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[4 * g + n][k], acc[4 * g + n], 0, 0, 0);
        } else if (V == 4) {    // single chain of 32 dependent MFMAs
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], acc[0], 0, 0, 0);
        } else if (V == 5) {    // chain-major: 4 dependent MFMAs per chain, 8 chains in turn
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], acc[n], 0, 0, 0);
        } else if (V == 6) {    // two chains alternating
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[n & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], acc[n & 1], 0, 0, 0);
        } else if (V == 7) {    // four chains alternating
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[n & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], acc[n & 3], 0, 0, 0);
        } else if (V == 8) {    // chain-major with a fresh accumulator per chain (C = 0), max-reduced
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], z, 0, 0, 0);
                acc[n] = z;
            }
        } else if (V == 14 || V == 15) {   // V8 + two VALU per chain: unrelated (14) / reading a chain that retired 8 chains ago (15)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 old = acc[n];
#pragma unroll
                for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], z, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (V == 14) { rm2 = fmaxf(fmaxf(rm2, rm3), rm4); rm3 = fmaxf(fmaxf(rm3, rm2), rm4); }
                if (V == 15) rm = fmaxf(fmaxf(rm, fmaxf(fmaxf(old[0], old[1]), old[2])), old[3]);
                __builtin_amdgcn_sched_barrier(0);
                acc[n] = z;
            }
        } else if (V == 9) {    // chain-major, pairs of chains interleaved (distance 2)
#pragma unroll
            for (int n = 0; n < 8; n += 2)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], acc[n], 0, 0, 0);
                    acc[n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n + 1][k], acc[n + 1], 0, 0, 0);
                }
        } else if (V >= 10 && V <= 13) {
            // chain-major with the max-reduction of the previous chain (2 VALU) per chain
            f32x4 a0 = acc[0], a1;
#pragma unroll
            for (int n = 0; n < 8; n += 2) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4& cur = half == 0 ? a1 : a0;
                    f32x4& old = half == 0 ? a0 : a1;
                    f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    if (V == 12) {
                        z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bw[n + half][0], z, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        rm = fmaxf(fmaxf(rm, fmaxf(fmaxf(old[0], old[1]), old[2])), old[3]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 1; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n + half][k], z, 0, 0, 0);
                        cur = z;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n + half][k], z, 0, 0, 0);
                        cur = z;
                        __builtin_amdgcn_sched_barrier(0);
                        if (V == 10) rm = fmaxf(fmaxf(rm, fmaxf(fmaxf(old[0], old[1]), old[2])), old[3]);
                        if (V == 13) { rm2 = fmaxf(fmaxf(rm2, rm3), rm4); rm3 = fmaxf(fmaxf(rm3, rm2), rm4); }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (V == 11) rm = fmaxf(rm, a0[0] + a1[1]);
            acc[0] = a0;
            acc[1] = a1;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int n = 0; n < 8; ++n) rm += acc[n][0] + acc[n][1] + acc[n][2] + acc[n][3];
    rm += rm2 + rm3;
    out[blockIdx.x * 256 + threadIdx.x] = rm;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// The consume loop of the conv-stack kernel, lone wave: per tile 4 ds_read_b128 (next tile's A fragments,
// alternating register sets), 8 in-place chains of 4 MFMAs, 2 v_max3 per chain after the following chain.
//   V = 0: as in the kernel   1: without the LDS reads   2: without the max3   3: neither
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

__device__ __forceinline__ f32x4 chain4(const bf16x8 (&af)[4], const bf16x8 (&b)[4]) {
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], b[k], z, 0, 0, 0);
    return z;
}

template <int V>
__device__ __forceinline__ void consume(const bf16x8 (&af)[4], const bf16x8 (&bw)[8][4], f32x4& carry, float (&rm)[8],
                                        bf16x8 (&afn)[4], const bf16x8* next = nullptr) {
    f32x4 accA = carry, accB;
#pragma unroll
    for (int n = 0; n < 8; n += 2) {
        if (V == 4) {            // one fragment of the next tile per chain pair, placed inside the first chain
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0], bw[n][0], z, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            afn[n >> 1] = next[(n >> 1) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], bw[n][k], z, 0, 0, 0);
            accB = z;
        } else {
            accB = chain4(af, bw[n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int pn = (n + 7) & 7;
        if (!(V & 2)) rm[pn] = max3f(rm[pn], max3f(accA[0], accA[1], accA[2]), accA[3]);
        else rm[pn] += accA[0];
        __builtin_amdgcn_sched_barrier(0);
        accA = chain4(af, bw[n + 1]);
        __builtin_amdgcn_sched_barrier(0);
        if (!(V & 2)) rm[n] = max3f(rm[n], max3f(accB[0], accB[1], accB[2]), accB[3]);
        else rm[n] += accB[0];
        __builtin_amdgcn_sched_barrier(0);
    }
    carry = accA;
}

// PR: how the two waves of a SIMD share the matrix pipe.  0: nothing  1: waves 4-7 at priority 3
//     2: the favoured wave changes with bit 11 of the shader clock  3: the wave that is behind its partner (progress
//     words in LDS, written and read once per tile pair) gets priority
template <int V, int PR = 0>
__global__ __launch_bounds__(512) void k_loop(const bf16x8* wsrc, const bf16x8* asrc, float* out,
                                              unsigned long long* cyc, int iters) {
    __shared__ bf16x8 s_a[16 * 4 * 64];        // 16 tiles of A fragments
    __shared__ int s_prog[8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < 8) s_prog[threadIdx.x] = 0;
    if (PR == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    int prog = 0;
    for (int i = threadIdx.x; i < 16 * 4 * 64; i += blockDim.x) s_a[i] = asrc[i & 255];
    __syncthreads();
    bf16x8 bw[8][4];
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) bw[n][k] = wsrc[(n * 4 + k) * 64 + lane];
    float rm[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) rm[n] = -1e30f;
    f32x4 carry = {0.f, 0.f, 0.f, 0.f};
    bf16x8 afA[4], afB[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) afA[k] = s_a[k * 64 + lane], afB[k] = s_a[(4 + k) * 64 + lane];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < 8; ++i) {
            if (PR == 2) {
                const int fav = (int)((__builtin_amdgcn_s_memtime() >> 11) & 1) ^ (wave >> 2);
                if (fav) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            }
            if (PR == 3) {
                ++prog;
                const int other = __builtin_amdgcn_readfirstlane(((volatile int*)s_prog)[wave ^ 4]);
                ((volatile int*)s_prog)[wave] = prog;
                if (other > prog || (other == prog && wave >= 4)) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            }
            if (!(V & 1) && V != 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) afB[k] = s_a[((2 * i + 1) * 4 + k) * 64 + lane];
            }
            if (V == 4) {
                consume<V>(afA, bw, carry, rm, afB, s_a + ((2 * i + 1) * 4) * 64 + lane);
                consume<V>(afB, bw, carry, rm, afA, s_a + ((2 * i + 2 > 15 ? 15 : 2 * i + 2) * 4) * 64 + lane);
                continue;
            }
            consume<V>(afA, bw, carry, rm, afB);
            if (!(V & 1)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) afA[k] = s_a[((2 * i + 2 > 15 ? 15 : 2 * i + 2) * 4 + k) * 64 + lane];
            }
            consume<V>(afB, bw, carry, rm, afA);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = carry[0];
#pragma unroll
    for (int n = 0; n < 8; ++n) r += rm[n];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V, int PR = 0>
static void run_loop(const char* name, const bf16x8* w, const bf16x8* a, float* out, unsigned long long* cyc, int blocks,
                     int threads = 256) {
    const int iters = 250;                    // x 16 tiles x 32 MFMAs
    const int waves = threads / 64;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k_loop<V, PR>), dim3(blocks), dim3(threads), 0, 0, w, a, out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 8);
        hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
        double lo = 0, hi = 0;                // waves 0-3 / waves 4-7 (the second wave of each SIMD)
        for (int b = 0; b < blocks; ++b)
            for (int wv = 0; wv < waves; ++wv) (wv < 4 ? lo : hi) += (double)h[b * 8 + wv];
        lo /= blocks * 4;
        hi /= blocks * (waves > 4 ? waves - 4 : 1);
        const double per_simd = (waves > 4 ? (hi > lo ? hi : lo) / 2 : lo) / (iters * 16 * 32.0);
        printf("%-36s %d waves rep %d: waves 0-3 %.2f, waves 4-7 %.2f ticks per own MFMA; %.2f ticks/MFMA per SIMD\n", name,
               waves, rep, lo / (iters * 16 * 32.0), hi / (iters * 16 * 32.0), per_simd);
    }
}

template <int V>
static void run(const char* name, const bf16x8* w, const bf16x8* a, float* out, unsigned long long* cyc, int blocks) {
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_stream<V>, dim3(blocks), dim3(256), 0, 0, w, a, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks);
        hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (auto v : h) mean += (double)v;
        mean /= blocks;
        const double flop = (double)blocks * 4 * iters * 32 * 16 * 16 * 32 * 2;
        printf("%-28s rep %d: %.2f ticks/MFMA, %.3f ms, %.0f TFLOP/s\n", name, rep, mean / (iters * 32.0), ms,
               flop / (ms * 1e-3) / 1e12);
    }
}

int main() {
    const int blocks = 256;
    std::vector<unsigned short> hw(32 * 64 * 8), ha(4 * 64 * 8);
    srand(1);
    for (auto& v : hw) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));   // random bf16 around 0.01..
    for (auto& v : ha) v = (unsigned short)(0x3c00 + (rand() & 0x3ff));
    bf16x8 *w, *a;
    float* out;
    unsigned long long* cyc;
    hipMalloc(&w, hw.size() * 2);
    hipMalloc(&a, ha.size() * 2);
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
    run<0>("8 chains, B = 32 fragments", w, a, out, cyc, blocks);
    run<1>("weights as A operand", w, a, out, cyc, blocks);
    run<2>("single B fragment", w, a, out, cyc, blocks);
    run<3>("two groups of 4 chains", w, a, out, cyc, blocks);
    run<4>("one dependent chain", w, a, out, cyc, blocks);
    run<5>("chain-major, 8 x 4", w, a, out, cyc, blocks);
    run<10>("chains + max3 of previous", w, a, out, cyc, blocks);
    run<14>("chains + 2 unrelated VALU", w, a, out, cyc, blocks);
    run<15>("chains + max3 of 8 chains ago", w, a, out, cyc, blocks);
    run<12>("chains + max3 after 1st MFMA", w, a, out, cyc, blocks);
    run<6>("two chains alternating", w, a, out, cyc, blocks);
    run<7>("four chains alternating", w, a, out, cyc, blocks);
    run<8>("chain-major, C = 0 starts", w, a, out, cyc, blocks);
    run<9>("chain-major, pairs", w, a, out, cyc, blocks);
    run_loop<0>("consume loop as in the kernel", w, a, out, cyc, blocks);
    run_loop<1>("consume loop, no LDS reads", w, a, out, cyc, blocks);
    run_loop<4>("consume loop, LDS read inside chain", w, a, out, cyc, blocks);
    run_loop<2>("consume loop, no max3", w, a, out, cyc, blocks);
    run_loop<3>("consume loop, no max3, no LDS", w, a, out, cyc, blocks);
    run_loop<0>("consume loop as in the kernel", w, a, out, cyc, blocks, 512);
    run_loop<1>("consume loop, no LDS reads", w, a, out, cyc, blocks, 512);
    run_loop<4>("consume loop, LDS read inside chain", w, a, out, cyc, blocks, 512);
    run_loop<0, 1>("kernel loop, waves 4-7 prio 3", w, a, out, cyc, blocks, 512);
    run_loop<0, 2>("kernel loop, prio by clock bit 11", w, a, out, cyc, blocks, 512);
    run_loop<0, 3>("kernel loop, prio to the one behind", w, a, out, cyc, blocks, 512);
    return 0;
}

// Fused per-point MLP + max-pool at float32 grade on the bf16 matrix pipe ("bf16x3" precision):
//
//     pooled[b, :] = act( max_p  W3 * relu(W2 * relu(W1 * T_b^T x[b,p] + b1) + b2) ) + b3
//
// (STN3d.forward, model/pointnet.py:30-34, and PointNetfeat.forward, model/pointnet.py:138-147, with eval-mode
// BatchNorm folded into W/b), same structure as conv_stack_bf16_kernel (conv_stack.hip): layer 2 transposed so that its
// accumulators are layer 3's A operand, layer-3 weights stationary in registers, A fragments through a double buffer
// in LDS, per-lane running maxima.  What differs is the arithmetic.  Every operand of layers 2 and 3 is TWO bf16
// numbers, v = hi + lo + r with hi = bf16(v), lo = bf16(v - hi) (both round-to-nearest, so |r| <= 2^-18 |v|), and a
// product is three MFMAs accumulating in float32:
//
//     x*w  ~  xh*wl + xl*wh + xh*wh          (what is dropped, xl*wl and the two r terms, is <= 3 * 2^-18 |x*w|)
//
// Piece products are exact in the float32 accumulator, so a layer's result differs from a float32 evaluation by the
// representation error of its operands only: about 4e-6 relative per term (rms), against 2e-3 for plain bf16.
// Layer 1 already runs at float32 accuracy on the matrix pipe (three-piece split, conv_stack.hip).
//
// Cost: 3 MFMAs per product, and the stationary weights of a wave are hi + lo images, so a wave holds 64 channels
// (4 tiles of 16 x 4 k-steps x 2 planes = 128 VGPRs) and a workgroup of 8 waves 512: a candidate is covered by TWO
// workgroups (grid.y = 2), each repeating layers 1-2 (13 % of a workgroup's MFMAs).  An A tile of 16 points is 8 KiB
// (two planes), a batch is 8 tiles (one produced per wave), the double buffer 128 KiB; conv2's hi weight image lives
// in LDS (16 KiB) and its lo image is read from memory (16 KiB, L1-resident) because LDS has no room for both.
#include <hip/hip_runtime.h>

#include "conv_stack_common.h"
#include "kernels.h"
#include "pngpd.h"

namespace pngpd {

#ifndef X3_PRIO_PRODUCE
#define X3_PRIO_PRODUCE 0
#define X3_PRIO_CONSUME 3
#endif
constexpr int X3_WAVES = 8;
constexpr int X3_THREADS = X3_WAVES * 64;
constexpr int X3_PLANE_BYTES = 4096;                       // [ks4][lane64][16 B]
constexpr int X3_TILE_BYTES = 2 * X3_PLANE_BYTES;          // hi plane, lo plane
constexpr int X3_BATCH_TILES = X3_WAVES;                   // one tile produced per wave and batch
constexpr int X3_BATCH_BYTES = X3_BATCH_TILES * X3_TILE_BYTES;
constexpr int X3_OFF_W2 = 2 * X3_BATCH_BYTES;              // conv2 weights, hi image
constexpr int X3_OFF_B2 = X3_OFF_W2 + (int)W2P_BF16_BYTES;
constexpr int X3_OFF_W1B = X3_OFF_B2 + C2 * 4;             // BN-folded conv1 table
constexpr int X3_OFF_W1 = X3_OFF_W1B + C1 * 16;            // layer-1 A fragments of two candidates
constexpr int X3_OFF_B3 = X3_OFF_W1 + 2 * W1F_BYTES;       // conv3 bias
constexpr int X3_LDS_BYTES = X3_OFF_B3 + C3 * 4;
static_assert(X3_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

// max(v, 0) as a signed-integer maximum of the bit pattern: one instruction (a float maximum canonicalises its
// operands first); negative zero and negative NaNs become +0.
__device__ __forceinline__ float relu_bits(float v) {
    const int i = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, i > 0 ? i : 0);
}

// relu, then the two bf16 pieces of each value, packed in pairs (element 0 in the low half).
__device__ __forceinline__ void relu_split2(float a, float b, unsigned int& hi, unsigned int& lo) {
    a = relu_bits(a);
    b = relu_bits(b);
    hi = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{a, b}, bf16x2));
    const float ra = a - u2f(hi << 16), rb = b - u2f(hi & 0xffff0000u);      // exact
    lo = __builtin_bit_cast(unsigned int, __builtin_convertvector(f32x2{ra, rb}, bf16x2));
}

__device__ __forceinline__ void relu_split(const f32x4v& v0, const f32x4v& v1, bf16x8& hi, bf16x8& lo) {
    unsigned int h[4], l[4];
    relu_split2(v0[0], v0[1], h[0], l[0]);
    relu_split2(v0[2], v0[3], h[1], l[1]);
    relu_split2(v1[0], v1[1], h[2], l[2]);
    relu_split2(v1[2], v1[3], h[3], l[3]);
    hi = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    lo = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
}

// Coordinate q of point c of the tile a wave produces next (lanes with q = 3 hold 0).
__device__ __forceinline__ float load_tile_point(const ConvStackArgs& a, int b, int tile, int c, int q) {
    int pt = tile * 16 + c;
    pt = pt < a.P ? pt : a.P - 1;
    const size_t at = a.layout == PNGPD_LAYOUT_C3P ? ((size_t)b * 3 + q) * a.P + pt : ((size_t)b * a.P + pt) * 3 + q;
    return q < 3 ? a.x[at] : 0.f;
}

// conv2 fragments of one pair of row tiles (2ks, 2ks+1): [row tile][k-step]
struct X3W2 {
    bf16x8 w[2][2];
};

__device__ __forceinline__ void load_w2_pair(X3W2& f, int ks, const char* img, int lane) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) f.w[tt][s] = *(const bf16x8*)(img + (((2 * ks + tt) * 2 + s) * 64 + lane) * 16);
}

// The same fragments from memory, as buffer loads: scalar resource + one 32-bit lane offset + a constant, so that the
// sixteen loads of a produce step share one address register (as flat loads the compiler keeps sixteen 64-bit
// addresses alive across the whole kernel).
__device__ __forceinline__ void load_w2_pair_global(X3W2& f, int ks, __amdgpu_buffer_rsrc_t img, unsigned int lane16) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s)
            f.w[tt][s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(img, lane16, ((2 * ks + tt) * 2 + s) * 1024, 0));
}
#ifdef X3_ABL_W2LO_LDS   // timing only: the lo image read from where the hi image lies
#define load_w2_pair_global(f, ks, img, lane16) load_w2_pair(f, ks, s_w2h, lane)
#endif

// Layers 1-2 of one tile of 16 points; the two planes of its layer-3 A fragments go to dst, dst + X3_PLANE_BYTES.
__device__ __forceinline__ void produce_tile_x3(float px, const char* s_w1f, unsigned int ones, const char* s_w2h,
                                                __amdgpu_buffer_rsrc_t g_w2l, const float* s_b2, char* dst, int lane) {
    const int q = lane >> 4;
    bf16x8 w1f[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w1f[t] = *(const bf16x8*)(s_w1f + (t * 64 + lane) * 16);
    X3W2 lo_a, lo_b;
    const unsigned int lane16 = (unsigned int)lane * 16u;
    load_w2_pair_global(lo_a, 0, g_w2l, lane16);
    // layer 1: the lane's eight k slots are made from one coordinate of its point (conv_stack.hip)
    bf16x8 xb;
    {
        const unsigned int x = f2u(px);
        const float r1 = px - u2f(x & 0xffff0000u);
        const unsigned int m = f2u(r1);
        const unsigned int l = f2u(r1 - u2f(m & 0xffff0000u));
        u32x4 v;
        v[0] = __builtin_amdgcn_perm(x, x, 0x03020302u);      // xh, xh
        v[1] = __builtin_amdgcn_perm(x, m, 0x07060302u);      // xm, xh
        v[2] = __builtin_amdgcn_perm(l, m, 0x07060302u);      // xm, xl
        v[3] = ones;
        xb = __builtin_bit_cast(bf16x8, v);
    }
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
    bf16x8 hbh[2], hbl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4v h1a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * s], xb, zero, 0, 0, 0);
        const f32x4v h1b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * s + 1], xb, zero, 0, 0, 0);
        relu_split(h1a, h1b, hbh[s], hbl[s]);
    }
    // layer 2, transposed: per pair of row tiles six MFMAs per tile, small terms first
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        X3W2& lo_cur = (ks & 1) ? lo_b : lo_a;
        X3W2& lo_nxt = (ks & 1) ? lo_a : lo_b;
        if (ks + 1 < 4) load_w2_pair_global(lo_nxt, ks + 1, g_w2l, lane16);
        X3W2 hi;
        load_w2_pair(hi, ks, s_w2h, lane);
        f32x4v acc[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = *(const f32x4v*)(s_b2 + 16 * (2 * ks + tt) + 4 * q);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo_cur.w[tt][s], hbh[s], acc[tt], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi.w[tt][s], hbl[s], acc[tt], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi.w[tt][s], hbh[s], acc[tt], 0, 0, 0);
        }
        bf16x8 H, L;
        relu_split(acc[0], acc[1], H, L);
        *(bf16x8*)(dst + (ks * 64 + lane) * 16) = H;
        *(bf16x8*)(dst + X3_PLANE_BYTES + (ks * 64 + lane) * 16) = L;
    }
}

struct X3AFrag {
    bf16x8 h[4], l[4];
};

__device__ __forceinline__ void load_afrags_x3(X3AFrag& f, const char* tile, int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.h[ks] = *(const bf16x8*)(tile + (ks * 64 + lane) * 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.l[ks] = *(const bf16x8*)(tile + X3_PLANE_BYTES + (ks * 64 + lane) * 16);
}

// One channel tile against one point tile: a chain of 12 MFMAs accumulating in place (a dependent chain of this
// instruction issues at the pipe's full rate, conv_stack.hip), the two small piece products of every k-step first.
__device__ __forceinline__ f32x4v mfma_chain12(const X3AFrag& f, const bf16x8 (&wh)[4], const bf16x8 (&wl)[4]) {
    f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.h[k], wl[k], z, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.l[k], wh[k], z, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.h[k], wh[k], z, 0, 0, 0);
    return z;
}

// One tile of 16 points against the wave's NT channel tiles; the max-reduction of a chain is issued behind the
// following chain, `carry` hands the last chain of a tile over to the next call.
template <int NT>
__device__ __forceinline__ void consume_tile_x3(const X3AFrag& f, const bf16x8 (&bwh)[NT][4], const bf16x8 (&bwl)[NT][4],
                                                f32x4v& carry, float (&rm)[NT]) {
    f32x4v accA = carry, accB;
#pragma unroll
    for (int n = 0; n < NT; n += 2) {
        accB = mfma_chain12(f, bwh[n], bwl[n]);
        __builtin_amdgcn_sched_barrier(0);
        const int pn = (n + NT - 1) & (NT - 1);
        rm[pn] = max3(rm[pn], max3(accA[0], accA[1], accA[2]), accA[3]);
        __builtin_amdgcn_sched_barrier(0);
        accA = mfma_chain12(f, bwh[n + 1], bwl[n + 1]);
        __builtin_amdgcn_sched_barrier(0);
        rm[n] = max3(rm[n], max3(accB[0], accB[1], accB[2]), accB[3]);
        __builtin_amdgcn_sched_barrier(0);
    }
    carry = accA;
}

// Diagnostic build only (-DPNGPD_STAMPS, tools/diag): shader-clock stamps of workgroup 0 go to a buffer of their own;
// no output value depends on them.  The product build compiles none of it.
#ifdef PNGPD_STAMPS
constexpr int X3_STAMP_FIRST = 7, X3_STAMP_BATCHES = 6, X3_STAMP_EVENTS = 5;
__device__ unsigned long long g_stamps_x3[X3_WAVES][X3_STAMP_BATCHES * X3_STAMP_EVENTS];
__device__ unsigned long long g_stamp_clock_x3[4];
#define X3_STAMP(g, ev)                                                                                   \
    do {                                                                                                  \
        if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && (g) >= X3_STAMP_FIRST &&                    \
            (g) < X3_STAMP_FIRST + X3_STAMP_BATCHES)                                                      \
            g_stamps_x3[wave][((g)-X3_STAMP_FIRST) * X3_STAMP_EVENTS + (ev)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define X3_STAMP(g, ev)
#endif

// NT = channel tiles of 16 per wave.  4 is the throughput shape: a workgroup covers 512 channels, grid.y = 2.  A batch
// too small to give every CU a workgroup runs NT = 2 with grid.y = 4.  The arithmetic of a channel does not depend on
// the cut, so a candidate's result does not depend on the batch it is scored in.
template <int NT>
__global__ __launch_bounds__(X3_THREADS) void conv_stack_x3_kernel(ConvStackArgs a) {
    static_assert(NT == 4 || NT == 2, "channel tiles per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar branches and addresses
    const int c = lane & 15, q = lane >> 4;

    char* s_a = smem;
    char* s_w2 = smem + X3_OFF_W2;
    float* s_b2 = (float*)(smem + X3_OFF_B2);
    float4* s_w1base = (float4*)(smem + X3_OFF_W1B);
    char* s_w1f = smem + X3_OFF_W1;
    float* s_b3 = (float*)(smem + X3_OFF_B3);
    // conv2's lo image as a raw buffer (gfx950: 32-bit elements, no swizzle), W2P_BF16_BYTES long
    const __amdgpu_buffer_rsrc_t g_w2l = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2lo, 0, (int)W2P_BF16_BYTES, 0x00020000);
    const unsigned int ones = q == 0 ? 0x3f803f80u : q == 1 ? 0x00003f80u : 0u;

    const int ct0 = ((int)blockIdx.y * X3_WAVES + wave) * NT;      // first channel tile of this wave
    bf16x8 bwh[NT][4], bwl[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const size_t at = (size_t)((((ct0 + nt) * 4 + ks) * 64) + lane) * 16;
            bwh[nt][ks] = *(const bf16x8*)((const char*)a.w3p + at);
            bwl[nt][ks] = *(const bf16x8*)((const char*)a.w3lo + at);
        }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
    }
    if (tid < C2) s_b2[tid] = a.b2p[tid];
    if (tid < C1) s_w1base[tid] = a.w1tab[tid];
    for (int i = tid; i < C3; i += X3_THREADS) s_b3[i] = a.b3[i];
    if (tid < 2 * 4 * 16)                          // k slots 24..31 of both layer-1 tables stay zero for the whole kernel
        *(u32x4*)(s_w1f + (tid >> 6) * W1F_BYTES + ((((tid >> 4) & 3) * 4 + 3) * 16 + (tid & 15)) * 16) = u32x4{0u, 0u, 0u, 0u};
    lds_dma_wait();
    __syncthreads();

    const int ntiles = (a.P + 15) >> 4;
    const int nbatch = (ntiles + X3_BATCH_TILES - 1) / X3_BATCH_TILES;
    const int stride = gridDim.x;
    const int ncand = (a.B - (int)blockIdx.x + stride - 1) / stride;

    // Produce cursor: the (candidate, batch) whose tile `wave` this wave produces next; the point is loaded one step
    // ahead.  The layer-1 table of a candidate is written by wave 0 when its cursor reaches that candidate, into the
    // table the previous candidate does not use: the first produce step that reads it lies behind the barrier that
    // ends the batch, and the table it replaces was last read a batch earlier.
    int pb = blockIdx.x, pbt = 0, ptab = 0;
    bool ptile = wave < ntiles;
    float px = 0.f;
    if (wave == 0) fold_w1_frags(a, pb, s_w1base, s_w1f, lane);
    if (ptile) px = load_tile_point(a, pb, wave, c, q);
    __syncthreads();
    auto produce_step = [&](char* batch_base) {
        if (ptile) produce_tile_x3(px, s_w1f + ptab * W1F_BYTES, ones, s_w2, g_w2l, s_b2, batch_base + wave * X3_TILE_BYTES, lane);
        if (++pbt == nbatch) {
            pbt = 0;
            pb += stride;
            ptab ^= 1;
            if (wave == 0 && pb < a.B) fold_w1_frags(a, pb, s_w1base, s_w1f + ptab * W1F_BYTES, lane);
        }
        const int tile = pbt * X3_BATCH_TILES + wave;
        ptile = pb < a.B && tile < ntiles;
        if (ptile) px = load_tile_point(a, pb, tile, c, q);
    };
    produce_step(s_a);
    __syncthreads();

#ifdef PNGPD_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        g_stamp_clock_x3[0] = __builtin_amdgcn_s_memtime();
        g_stamp_clock_x3[1] = __builtin_amdgcn_s_memrealtime();
    }
    int g = 0;
#endif
    int buf = 0;
    for (int ci = 0; ci < ncand; ++ci) {
        const int b = blockIdx.x + ci * stride;
        float rm[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rm[nt] = -INFINITY;
        for (int bt = 0; bt < nbatch; ++bt, buf ^= 1) {
            const int nt_b = min(X3_BATCH_TILES, ntiles - X3_BATCH_TILES * bt);
            const int npairs = (nt_b + 1) >> 1;
            // pair of tiles in front of which this wave's produce step sits: the two waves of a SIMD (w, w + 4) are
            // two pairs apart
#ifdef X3_STEP_LO
            const int step0 = min(wave < 4 ? X3_STEP_LO : X3_STEP_HI, npairs - 1);
#else
            const int step0 = min(wave >> 1, npairs - 1);
#endif
            const char* cur = s_a + buf * X3_BATCH_BYTES;
            char* nxt = s_a + (buf ^ 1) * X3_BATCH_BYTES;
            // Software pipeline over the tiles: the fragments of the next tile are in flight during the MFMAs of the
            // current one.  Behind a produce step the current tile is loaded again, so that no A fragment is live
            // across the step (it needs the registers).
            X3_STAMP(g, 0);
            X3AFrag fA, fB;
            load_afrags_x3(fA, cur, lane);
            f32x4v carry = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int i = 0; i < (nt_b >> 1); ++i) {
                if (i == step0) {
                    X3_STAMP(g, 1);
                    __builtin_amdgcn_s_setprio(X3_PRIO_PRODUCE);
                    produce_step(nxt);
                    __builtin_amdgcn_s_setprio(X3_PRIO_CONSUME);
                    X3_STAMP(g, 2);
                    load_afrags_x3(fA, cur + (2 * i) * X3_TILE_BYTES, lane);
                }
                load_afrags_x3(fB, cur + (2 * i + 1) * X3_TILE_BYTES, lane);
                consume_tile_x3<NT>(fA, bwh, bwl, carry, rm);
                load_afrags_x3(fA, cur + min(2 * i + 2, nt_b - 1) * X3_TILE_BYTES, lane);
                consume_tile_x3<NT>(fB, bwh, bwl, carry, rm);
            }
            if (nt_b & 1) {                       // odd tail: fA already holds tile nt_b - 1
                if ((nt_b >> 1) == step0) {
                    __builtin_amdgcn_s_setprio(X3_PRIO_PRODUCE);
                    produce_step(nxt);
                    __builtin_amdgcn_s_setprio(X3_PRIO_CONSUME);
                    load_afrags_x3(fA, cur + (nt_b - 1) * X3_TILE_BYTES, lane);
                }
                consume_tile_x3<NT>(fA, bwh, bwl, carry, rm);
            }
            rm[NT - 1] = max3(rm[NT - 1], max3(carry[0], carry[1], carry[2]), carry[3]);
            X3_STAMP(g, 3);
            lds_barrier();          // orders the A-fragment buffers only: the point load of the next produce step stays in flight
            X3_STAMP(g, 4);
#ifdef PNGPD_STAMPS
            ++g;
#endif
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ch = ((int)blockIdx.y * X3_WAVES + wave) * (16 * NT) + nt * 16 + c;
            float m = __builtin_fmaxf(rm[nt], __shfl_xor(rm[nt], 16));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32)) + s_b3[ch];
            if (a.relu_out) m = __builtin_fmaxf(m, 0.f);
            if (q == 0) a.pooled[(size_t)b * C3 + ch] = m;
        }
    }
#ifdef PNGPD_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
        g_stamp_clock_x3[2] = __builtin_amdgcn_s_memtime();
        g_stamp_clock_x3[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef PNGPD_STAMPS
extern "C" int pngpd_debug_read_stamps_x3(unsigned long long* stamps, unsigned long long* clock) {
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_stamps_x3), sizeof(g_stamps_x3));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(clock, HIP_SYMBOL(g_stamp_clock_x3), sizeof(g_stamp_clock_x3));
}
#endif

hipError_t launch_conv_stack_x3(const ConvStackArgs& a, hipStream_t s) {
    if (a.B < 1 || a.P < 1 || !a.x || !a.w1tab || !a.w2p || !a.w2lo || !a.b2p || !a.w3p || !a.w3lo || !a.b3 || !a.pooled)
        return hipErrorInvalidValue;
    int n_cu = 0;
    hipError_t e = device_cu_count(&n_cu);
    if (e != hipSuccess) return e;
    auto launch = [&](auto kern, int pieces) {
        hipError_t er = set_max_lds_once((const void*)kern, X3_LDS_BYTES);
        if (er != hipSuccess) return er;
        const int per_piece = n_cu / pieces > 0 ? n_cu / pieces : 1;
        hipLaunchKernelGGL(kern, dim3(a.B < per_piece ? a.B : per_piece, pieces), dim3(X3_THREADS), X3_LDS_BYTES, s, a);
        return hipGetLastError();
    };
    if (a.B * 4 <= n_cu) return launch(conv_stack_x3_kernel<2>, 4);
    return launch(conv_stack_x3_kernel<4>, 2);
}

}  // namespace pngpd

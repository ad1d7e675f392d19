// Fused per-point MLP + symmetric max-pool for gfx950:
//
//     pooled[b, :] = act( max_p  W3 * relu(W2 * relu(W1 * T_b^T x[b,p] + b1) + b2) ) + b3
//
// which is conv1->bn->relu->conv2->bn->relu->conv3->bn(->relu)->MaxPool1d(num_points) of
// STN3d.forward (model/pointnet.py:30-34) and PointNetfeat.forward (model/pointnet.py:138-147)
// with eval-mode BatchNorm folded into W/b.  The [B,1024,P] activation of the reference never
// exists: a tile of points goes from its input coordinates to a per-channel maximum on chip.
//
// Common to both precisions:
//   * layer 1 (K = 3) lands straight in the MFMA B-operand layout of layer 2, with the candidate's 3x3 input
//     transform folded into the 64x3 weight table (f32 path: on the VALU; bf16 path: on the matrix pipe at float32
//     accuracy, see produce_tiles);
//   * layer 2 is computed TRANSPOSED, X[ch2][point] = W2-fragments * h1-fragments, so that the
//     accumulator (point on the lane, channels in the registers) is, after ReLU, directly the A
//     operand of layer 3: no transposition through LDS, no cross-lane traffic.  The k order this
//     implies for layer 3 is baked into the packed W3 image (pack.cpp);
//   * layer 3 has the channel on the lane and points in the registers, so the max over points
//     is a per-lane register reduction.
// Points beyond P are clamped to point P-1: duplicates cannot change a maximum.
//
// bf16 path: weight-stationary persistent kernel on v_mfma_f32_16x16x32_bf16 (below).
// f32 path:  one workgroup per candidate, W3 streamed through LDS, v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>

#include "conv_stack_common.h"
#include "kernels.h"
#include "pngpd.h"

namespace pngpd {


constexpr int WAVES = 4;
constexpr int THREADS = WAVES * 64;


__device__ __forceinline__ float max16(const f32x16& v) {
    float m0 = max3(v[0], v[1], v[2]);
    float m1 = max3(v[3], v[4], v[5]);
    float m2 = max3(v[6], v[7], v[8]);
    float m3 = max3(v[9], v[10], v[11]);
    float m4 = max3(v[12], v[13], v[14]);
    return max3(max3(m0, m1, m2), max3(m3, m4, v[15]), m0);
}

__device__ __forceinline__ void load_point(const ConvStackArgs& a, int b, int pt, float& x0,
                                           float& x1, float& x2) {
    if (a.layout == PNGPD_LAYOUT_C3P) {
        const float* base = a.x + (size_t)b * 3 * a.P;
        x0 = base[pt];
        x1 = base[a.P + pt];
        x2 = base[2 * a.P + pt];
    } else {
        const float* base = a.x + ((size_t)b * a.P + pt) * 3;
        x0 = base[0];
        x1 = base[1];
        x2 = base[2];
    }
}

// BN-folded conv1 row, with the candidate's 3x3 `trans` folded in:
// conv1(trans^T x)[o] = sum_j x[j] * (sum_c W1[o][c] * trans[j][c])   (model/pointnet.py:138-142).
__device__ __forceinline__ float4 fold_w1(const ConvStackArgs& a, int b, int o) {
    float4 w = a.w1tab[o];
    if (a.trans != nullptr) {
        const float* T = a.trans + (size_t)b * 9;
        float4 f;
        f.x = w.x * T[0] + w.y * T[1] + w.z * T[2];
        f.y = w.x * T[3] + w.y * T[4] + w.z * T[5];
        f.z = w.x * T[6] + w.y * T[7] + w.z * T[8];
        f.w = w.w;
        return f;
    }
    return w;
}

// Merge this wave's per-channel maxima into its private LDS row (channel = 32*nt + lane&31).
__device__ __forceinline__ void merge_runmax(float* s_red_wave, int nt, float m, int r, int h) {
    m = __builtin_fmaxf(m, __shfl_xor(m, 32));
    if (h == 0) {
        float* p = s_red_wave + 32 * nt + r;
        *p = __builtin_fmaxf(*p, m);
    }
}

__device__ __forceinline__ void write_pooled(const ConvStackArgs& a, int b, const float* s_red,
                                             int tid, int c_begin, int c_end) {
    for (int c = c_begin + tid; c < c_end; c += THREADS) {
        float m = __builtin_fmaxf(__builtin_fmaxf(s_red[c], s_red[C3 + c]),
                                  __builtin_fmaxf(s_red[2 * C3 + c], s_red[3 * C3 + c]));
        m += a.b3[c];
        if (a.relu_out) m = __builtin_fmaxf(m, 0.f);
        a.pooled[(size_t)b * C3 + c] = m;
    }
}

// ------------------------------------------------------------------------------------------
// bf16 path: weight-stationary, persistent, v_mfma_f32_16x16x32_bf16 (f32 accumulate).
//
// One workgroup of 8 waves per CU walks over candidates b = blockIdx.x, +gridDim.x, ...
// Wave w keeps the layer-3 B fragments of channels [128w, 128w+128) in 128 VGPRs for the whole
// kernel (8 channel tiles of 16 x 4 k-steps of 32): W3 is read from memory once per workgroup,
// and since a wave sees every point of a candidate for its channels the per-channel running
// maximum is a register - no LDS merge, no cross-wave reduction.
// What moves through LDS is the layer-3 A operand.  Points are handled in tiles of 16 and
// batches of 16 tiles: wave w computes layers 1-2 of tiles 2w, 2w+1 of the NEXT batch and stores
// their A fragments (2 x 4 KiB) into the other half of a double buffer; every wave then
// multiplies every tile of the current batch with its own channels.  The produce step of wave w
// sits in front of tile 2w of the consume loop, so the two waves of a SIMD (w and w+4) are never
// both away from the matrix pipe.  One barrier per batch (256 points); the first batch of the
// next candidate is produced during the last batch of the current one.
//
// Lane l = (c = l%16, q = l/16).  The transposed layer-2 accumulator register i of row tile T
// holds channel 16T + 4q + i of point c, so the registers of tiles 2ks and 2ks+1 are the 8
// k slots of layer-3 step ks.  This shape is used instead of 32x32x16 because the chip holds a
// higher clock on it under sustained load (same cycles per flop, ~12 % more flop/s).
//
// LDS: [0,128K) two batches of A fragments | [128K,144K) W2 fragments | conv2 bias |
//      BN-folded conv1 table | two candidates' layer-1 A fragments (trans folded in) | conv3 bias.
// ------------------------------------------------------------------------------------------
constexpr int WS_WAVES = 8;
constexpr int WS_THREADS = WS_WAVES * 64;
constexpr int W16_TILE_BYTES = 4096;                       // [ks4][lane64][16 B]
constexpr int W16_BATCH_TILES = 2 * WS_WAVES;
// Wave priority while producing / consuming: the wave that feeds the matrix pipe goes first.  Measured with two
// builds alternating on one box (tools/diag/ab_bench.sh): consume above produce +0.7..0.9 % over the reverse,
// levels 1, 2, 3 alike.
#ifndef PNGPD_PRIO_PRODUCE
#define PNGPD_PRIO_PRODUCE 0
#define PNGPD_PRIO_CONSUME 3
#endif
#ifndef PNGPD_PRODUCER_WAVES
#define PNGPD_PRODUCER_WAVES 8
#endif
constexpr int PRODUCER_WAVES = PNGPD_PRODUCER_WAVES;       // waves 0..PRODUCER_WAVES-1 produce
constexpr int PRODUCE_STEPS = W16_BATCH_TILES / (2 * PRODUCER_WAVES);   // steps of 2 tiles per batch
constexpr int WS_BATCH_BYTES = W16_BATCH_TILES * W16_TILE_BYTES;
constexpr int WS_OFF_W2 = 2 * WS_BATCH_BYTES;
constexpr int WS_OFF_B2 = WS_OFF_W2 + (int)W2P_BF16_BYTES;
constexpr int WS_OFF_W1B = WS_OFF_B2 + C2 * 4;             // BN-folded conv1 table, shared
constexpr int WS_OFF_W1 = WS_OFF_W1B + C1 * 16;            // layer-1 A fragments of two candidates (2 x W1F_BYTES)
constexpr int WS_OFF_B3 = WS_OFF_W1 + 2 * 4096;            // conv3 bias: read once per candidate by every wave
constexpr int WS_LDS_BYTES = WS_OFF_B3 + C3 * 4;

// Layer 1 (3 -> 64, K = 3 plus the bias) also runs on the matrix pipe, as ONE v_mfma_f32_16x16x32_bf16 per tile of
// 16 channels x 16 points, at float32 accuracy: every float is cut into three bf16 pieces by truncation,
// v = vh + vm + vl exactly (3 x 8 significand bits), and the 32 k slots of the instruction hold, for each of the
// three coordinates, the six piece products xh*wh, xh*wm, xm*wh, xh*wl, xm*wm, xl*wh (what is left out is below
// 2^-23 of the product), plus the bias pieces against 1.0:
//
//     k slot      0   1   2   3   4   5   6   7  | 8..13   14  15 | 16..21  22 23 | 24..31
//     B (points)  xh  xh  xm  xh  xm  xl  1   1  | same for y  1   0 | same for z 0  0 |  0
//     A (weights) wh  wm  wh  wl  wm  wh  bh  bm | same for y  bl  0 | same for z 0  0 |  0
//
// so lane (c, q) of the B operand needs nothing but coordinate q of point c (q = 3: zeros).  On the VALU this
// layer was 48 packed fmas and 16 LDS reads per produce step (of two tiles); here it is 14 VALU instructions,
// 4 LDS reads and 8 MFMAs, and the step - during which the other wave of the SIMD has the matrix pipe to itself and
// cannot fill it - is shorter: +1.9 % candidates/s in a one-box A/B.
// The rows of an A tile are a permutation of the channels chosen so that the accumulators of tiles (s, 0) and
// (s, 1) are, after ReLU and packing, lane (c, q)'s eight k slots 32s + 8q .. + 7 of layer 2's B operand.
static_assert(WS_OFF_B3 - WS_OFF_W1 == 2 * W1F_BYTES, "LDS room for two layer-1 tables");



// Layers 1-2 of two tiles of 16 points; their layer-3 A fragments go to dst, dst + 4 KiB.
// The LDS reads are issued in groups ahead of their use (layer 2 double-buffers one k-step of
// W2 fragments and bias), with fences so the scheduler keeps them there: left alone it emits
// read -> wait -> two MFMAs thirty times over, and a produce step then costs thirty LDS latencies.
struct L2Frags {
    bf16x8 wf[2][2];     // [row tile of the pair][k-step of conv2]
    f32x4v bias[2];
};

__device__ __forceinline__ void load_l2frags(L2Frags& f, int ks, const char* s_w2, const float* s_b2, int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int T = 2 * ks + tt;
        f.bias[tt] = *(const f32x4v*)(s_b2 + 16 * T + 4 * q);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) f.wf[tt][s2] = *(const bf16x8*)(s_w2 + ((T * 2 + s2) * 64 + lane) * 16);
    }
}

__device__ __forceinline__ void produce_tiles(const TilePoints& p, const char* s_w1f, unsigned int ones, const char* s_w2,
                                              const float* s_b2, char* dst, int lane) {
    bf16x8 hb[2][2];
    bf16x8 xb[2];
    // every LDS read of layer 1 and the first k-step of layer 2 is issued before anything waits for one
    bf16x8 w1f[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w1f[t] = *(const bf16x8*)(s_w1f + (t * 64 + lane) * 16);
    L2Frags fa, fb;
    load_l2frags(fa, 0, s_w2, s_b2, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const unsigned int x = f2u(p.x[u]);
        const float r1 = p.x[u] - u2f(x & 0xffff0000u);
        const unsigned int m = f2u(r1);
        const unsigned int l = f2u(r1 - u2f(m & 0xffff0000u));
        u32x4 v;
        v[0] = __builtin_amdgcn_perm(x, x, 0x03020302u);      // xh, xh
        v[1] = __builtin_amdgcn_perm(x, m, 0x07060302u);      // xm, xh
        v[2] = __builtin_amdgcn_perm(l, m, 0x07060302u);      // xm, xl
        v[3] = ones;
        xb[u] = __builtin_bit_cast(bf16x8, v);
    }
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4v h1[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            h1[u][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * s], xb[u], zero, 0, 0, 0);
            h1[u][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * s + 1], xb[u], zero, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) hb[u][s] = relu_pack(h1[u][0], h1[u][1]);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        L2Frags& cur = (ks & 1) ? fb : fa;
        L2Frags& nxt = (ks & 1) ? fa : fb;
        if (ks + 1 < 4) load_l2frags(nxt, ks + 1, s_w2, s_b2, lane);
        __builtin_amdgcn_sched_barrier(0);
        f32x4v acc[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.wf[tt][0], hb[u][0], cur.bias[tt], 0, 0, 0);
                acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.wf[tt][1], hb[u][1], acc[u][tt], 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            *(bf16x8*)(dst + u * W16_TILE_BYTES + (ks * 64 + lane) * 16) = relu_pack(acc[u][0], acc[u][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void load_afrags(bf16x8 (&af)[4], const char* tile, int lane) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) af[ks] = *(const bf16x8*)(tile + (ks * 64 + lane) * 16);
}

// One tile of 16 points against the wave's 8 channel tiles.
// Each channel tile is one chain of 4 MFMAs accumulating IN PLACE, issued back to back: on this
// chip a dependent chain of v_mfma_f32_16x16x32_bf16 issues every 16.75 cycles, while the same
// MFMAs interleaved over 8 accumulators take 24 cycles each (tools/ubench/mfma_stream.hip), so
// the usual "interleave independent accumulators" ordering is the slow one here.
// The max-reduction of a chain (2 v_max3) is issued after the FOLLOWING chain's four MFMAs, when
// its result has long retired (no wait states) and never splits a chain; `carry` hands the last
// chain of a tile over to the next call.
__device__ __forceinline__ f32x4v mfma_chain4(const bf16x8 (&af)[4], const bf16x8 (&b)[4]) {
    f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[k], b[k], z, 0, 0, 0);
    return z;
}

template <int NT>
__device__ __forceinline__ void consume_tile(const bf16x8 (&af)[4], const bf16x8 (&bw)[NT][4], f32x4v& carry,
                                             float (&rm)[NT]) {
    f32x4v accA = carry, accB;
#pragma unroll
    for (int n = 0; n < NT; n += 2) {
        accB = mfma_chain4(af, bw[n]);
        __builtin_amdgcn_sched_barrier(0);
        const int pn = (n + NT - 1) & (NT - 1);
        rm[pn] = max3(rm[pn], max3(accA[0], accA[1], accA[2]), accA[3]);
        __builtin_amdgcn_sched_barrier(0);
        accA = mfma_chain4(af, bw[n + 1]);
        __builtin_amdgcn_sched_barrier(0);
        rm[n] = max3(rm[n], max3(accB[0], accB[1], accB[2]), accB[3]);
        __builtin_amdgcn_sched_barrier(0);
    }
    carry = accA;
}

// Diagnostic build only (-DPNGPD_STAMPS, tools/diag): shader-clock stamps of workgroup 0 go to a
// buffer of their own; no output value depends on them.  The product build compiles none of it.
#ifdef PNGPD_STAMPS
constexpr int STAMP_FIRST_BATCH = 6, STAMP_BATCHES = 6, STAMP_EVENTS = 5;
__device__ unsigned long long g_stamps[WS_WAVES][STAMP_BATCHES * STAMP_EVENTS];
__device__ unsigned long long g_stamp_clock[4];
#define STAMP(g, ev)                                                                          \
    do {                                                                                      \
        if (blockIdx.x == 0 && lane == 0 && (g) >= STAMP_FIRST_BATCH &&                        \
            (g) < STAMP_FIRST_BATCH + STAMP_BATCHES)                                          \
            g_stamps[wave][((g)-STAMP_FIRST_BATCH) * STAMP_EVENTS + (ev)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define STAMP(g, ev)
#endif

// which quarter / half of the channels a workgroup covers; the full-width kernel has a single piece
template <int NT>
__device__ __forceinline__ int piece_of() {
    return NT == 8 ? 0 : (int)blockIdx.y;
}

// NT = channel tiles of 16 per wave: 8 is the throughput shape (a workgroup covers all 1024 channels, grid.y = 1).
// A batch too small to give every CU a candidate runs NT = 2 with grid.y = 4: workgroup (x, y) covers channels
// [256y, 256y + 256) and repeats layers 1-2; the arithmetic of a channel does not depend on the cut.
// SPLIT (small batches only): grid.z cuts the POINTS of a candidate into groups of a.tiles_per_group tiles as well;
// workgroup (x, y, z) writes the maxima over its group to pooled + z * B * 1024 (bias and ReLU applied: both commute
// with a maximum, bit for bit), and the layer that reads them takes the maximum over the groups.
template <int NT, bool SPLIT = false>
__global__ __launch_bounds__(WS_THREADS) void conv_stack_bf16_kernel(ConvStackArgs a) {
    static_assert(NT == 8 || NT == 4 || NT == 2, "channel tiles per wave");
    static_assert(!(SPLIT && NT == 8), "the throughput shape is never cut over points");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform, and said so: branches on it become scalar branches and its address arithmetic scalar registers
    // (as `tid >> 6` the compiler must treat it as per-lane: 8 spilled registers instead of 2, conv stack 0.4615 ->
    // 0.4468 ms in a one-box A/B)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;

    char* s_a = smem;
    char* s_w2 = smem + WS_OFF_W2;
    float* s_b2 = (float*)(smem + WS_OFF_B2);
    float4* s_w1base = (float4*)(smem + WS_OFF_W1B);
    char* s_w1f = smem + WS_OFF_W1;               // layer-1 A fragments: two candidates' tables of W1F_BYTES
    // slots 6, 7 of this lane's eight layer-1 k slots: the ones that meet the bias pieces
    const unsigned int ones = q == 0 ? 0x3f803f80u : q == 1 ? 0x00003f80u : 0u;

    const int ct0 = (piece_of<NT>() * WS_WAVES + wave) * NT;      // first channel tile of this wave
    bf16x8 bw[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bw[nt][ks] = *(const bf16x8*)((const char*)a.w3p + ((((ct0 + nt) * 4 + ks) * 64) + lane) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
    }
    if (tid < C2) s_b2[tid] = a.b2p[tid];
    if (tid < C1) s_w1base[tid] = a.w1tab[tid];
    float* s_b3 = (float*)(smem + WS_OFF_B3);
    for (int i = tid; i < C3; i += WS_THREADS) s_b3[i] = a.b3[i];
    if (tid < 2 * 4 * 16)                          // k slots 24..31 of both tables stay zero for the whole kernel
        *(u32x4*)(s_w1f + (tid >> 6) * W1F_BYTES + ((((tid >> 4) & 3) * 4 + 3) * 16 + (tid & 15)) * 16) = u32x4{0u, 0u, 0u, 0u};
    lds_dma_wait();
    __syncthreads();

    // this workgroup's tiles: [tile_base, tile_base + ntiles)
    const int tile_base = SPLIT ? (int)blockIdx.z * a.tiles_per_group : 0;
    const int ntiles = SPLIT ? min(a.tiles_per_group, ((a.P + 15) >> 4) - tile_base) : (a.P + 15) >> 4;
    const int nbatch = (ntiles + W16_BATCH_TILES - 1) / W16_BATCH_TILES;
    const int stride = gridDim.x;
    const int ncand = (a.B - (int)blockIdx.x + stride - 1) / stride;
    float* pooled = SPLIT ? a.pooled + (size_t)blockIdx.z * a.B * C3 : a.pooled;

    // Produce cursor: the (candidate, batch, half) this wave produces next; every wave's cursor moves to the next
    // candidate in the same batch.  The points are loaded one step ahead.  The layer-1 table of a candidate is
    // written by wave 0 when its cursor reaches that candidate, into the table the previous candidate does not
    // use; the first produce step that reads it lies behind the barrier that ends the batch, and the table it
    // replaces was last read a batch earlier.
    static_assert(PRODUCE_STEPS == 1 && PRODUCER_WAVES == WS_WAVES, "one produce step per wave and batch");
    int pb = blockIdx.x, pbt = 0, ptab = 0;
    const int first_tile = 2 * wave;                       // of this wave, within a batch
    bool ptile = first_tile < ntiles;
    TilePoints px;
    if (wave == 0) fold_w1_frags(a, pb, s_w1base, s_w1f, lane);
    if (ptile) load_tile_points(a, pb, tile_base + first_tile, c, q, px);
    __syncthreads();
    auto produce_step = [&](char* batch_base) {
        if (ptile) produce_tiles(px, s_w1f + ptab * W1F_BYTES, ones, s_w2, s_b2, batch_base + first_tile * W16_TILE_BYTES, lane);
        if (++pbt == nbatch) {
            pbt = 0;
            pb += stride;
            ptab ^= 1;
            if (wave == 0 && pb < a.B) fold_w1_frags(a, pb, s_w1base, s_w1f + ptab * W1F_BYTES, lane);
        }
        const int tile0 = pbt * W16_BATCH_TILES + first_tile;
        ptile = pb < a.B && tile0 < ntiles;
        if (ptile) load_tile_points(a, pb, tile_base + tile0, c, q, px);
    };
    produce_step(s_a);
    __syncthreads();

#ifdef PNGPD_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        g_stamp_clock[0] = __builtin_amdgcn_s_memtime();
        g_stamp_clock[1] = __builtin_amdgcn_s_memrealtime();
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
            const int nt_b = min(W16_BATCH_TILES, ntiles - W16_BATCH_TILES * bt);
            const int npairs = (nt_b + 1) >> 1;
            // pair index in front of which this wave's produce step sits
            const int step0 = min(wave, npairs - 1);
            const char* cur = s_a + buf * WS_BATCH_BYTES;
            char* nxt = s_a + (buf ^ 1) * WS_BATCH_BYTES;
            STAMP(g, 0);
            // Software pipeline over the tiles of the batch: the A fragments of the next tile are
            // in flight during the MFMAs of the current one (consume_tile overlaps the
            // max-reductions with the MFMAs).
            bf16x8 afA[4], afB[4];
            load_afrags(afA, cur, lane);
            f32x4v carry = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            for (int i = 0; i < (nt_b >> 1); ++i) {
                if (i == step0) {
                    STAMP(g, 1);
                    __builtin_amdgcn_s_setprio(PNGPD_PRIO_PRODUCE);
#ifndef PNGPD_ABLATE_PRODUCE
                    produce_step(nxt);
#endif
                    __builtin_amdgcn_s_setprio(PNGPD_PRIO_CONSUME);
                    STAMP(g, 2);
                }
#ifdef PNGPD_ABLATE_HALF
                if (wave >= 4) continue;
#endif
                // Each tile's MFMAs run while the next tile's fragments load into the other set.
                // No branches around the loads: the compiler then tracks the LDS counter exactly.
                load_afrags(afB, cur + (2 * i + 1) * W16_TILE_BYTES, lane);
                consume_tile(afA, bw, carry, rm);
                load_afrags(afA, cur + min(2 * i + 2, nt_b - 1) * W16_TILE_BYTES, lane);
                consume_tile(afB, bw, carry, rm);
            }
            if (nt_b & 1) {                       // odd tail: afA already holds tile nt_b-1
                const int i = nt_b >> 1;
                if (i == step0) {
                    __builtin_amdgcn_s_setprio(PNGPD_PRIO_PRODUCE);
                    produce_step(nxt);
                    __builtin_amdgcn_s_setprio(PNGPD_PRIO_CONSUME);
                }
                consume_tile(afA, bw, carry, rm);
            }
            rm[NT - 1] = max3(rm[NT - 1], max3(carry[0], carry[1], carry[2]), carry[3]);
            STAMP(g, 3);
            lds_barrier();          // orders the A-fragment buffers only: the point loads of the next produce step stay in flight
            STAMP(g, 4);
#ifdef PNGPD_STAMPS
            ++g;
#endif
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            // recomputed here: a value kept alive across the loops costs this kernel a register it does not have
            const int ch = (piece_of<NT>() * WS_WAVES + wave) * (16 * NT) + nt * 16 + c;
            float m = __builtin_fmaxf(rm[nt], __shfl_xor(rm[nt], 16));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32)) + s_b3[ch];
            if (a.relu_out) m = __builtin_fmaxf(m, 0.f);
            if (q == 0) pooled[(size_t)b * C3 + ch] = m;
        }
    }
#ifdef PNGPD_STAMPS
    if (blockIdx.x == 0 && tid == 0) {
        g_stamp_clock[2] = __builtin_amdgcn_s_memtime();
        g_stamp_clock[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

#ifdef PNGPD_STAMPS
extern "C" int pngpd_debug_read_stamps(unsigned long long* stamps, unsigned long long* clock) {
    hipError_t e = hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemcpyFromSymbol(clock, HIP_SYMBOL(g_stamp_clock), sizeof(g_stamp_clock));
}
#endif

// ------------------------------------------------------------------------------------------
// exact-f32 MFMA path (v_mfma_f32_32x32x2_f32): every product is an f32 fma chain.
// LDS: [0,16K) two W3 half-tile buffers (8 KiB) | [16K,48K) W2 fragments | conv1 table |
//      conv2 bias | per-wave running maxima.
// ------------------------------------------------------------------------------------------
constexpr int F32_CHUNK_BYTES = 8192;
constexpr int F32_OFF_W2 = 2 * F32_CHUNK_BYTES;
constexpr int F32_OFF_W1 = F32_OFF_W2 + (int)W2P_F32_BYTES;
constexpr int F32_OFF_B2 = F32_OFF_W1 + C1 * 16;
constexpr int F32_OFF_RED = F32_OFF_B2 + B2P_FLOATS * 4;
constexpr int F32_LDS_BYTES = F32_OFF_RED + WAVES * C3 * 4;

// Grid (B, pieces): workgroup (b, y) computes channel tiles [y * nt_per, (y + 1) * nt_per) of candidate b.  One
// piece per candidate is the throughput shape; a small batch is cut into pieces (each repeats layers 1-2, 6 % of
// the work per piece) so that it still covers the chip.  The arithmetic of a channel does not depend on the cut.
template <int MT>
__global__ __launch_bounds__(THREADS, 2) void conv_stack_f32_kernel(ConvStackArgs a, int nt_per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;

    char* s_w3 = smem;
    char* s_w2 = smem + F32_OFF_W2;
    float4* s_w1 = (float4*)(smem + F32_OFF_W1);
    float* s_b2 = (float*)(smem + F32_OFF_B2);
    float* s_red = (float*)(smem + F32_OFF_RED);
    const char* g_w3 = (const char*)a.w3p;
    const int nt0 = blockIdx.y * nt_per, nt1 = nt0 + nt_per;
    const int ck_first = 2 * nt0, ck_last = 2 * nt1 - 1;        // this workgroup's W3 chunks

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int piece = wave * 8 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        dma_piece(g_w3 + (size_t)ck_first * F32_CHUNK_BYTES + piece * 1024, s_w3 + piece * 1024, lane);
    }
    if (tid < C1) s_w1[tid] = fold_w1(a, b, tid);
    if (tid < B2P_FLOATS) s_b2[tid] = a.b2p[tid];
    for (int c = tid; c < WAVES * C3; c += THREADS) s_red[c] = -INFINITY;
    lds_dma_wait();
    __syncthreads();

    float* s_red_wave = s_red + wave * C3;
    const int pts_per_pass = WAVES * 32 * MT;
    const int npass = (a.P + pts_per_pass - 1) / pts_per_pass;

    for (int pass = 0; pass < npass; ++pass) {
        float x0[MT], x1[MT], x2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int pt = pass * pts_per_pass + (wave * MT + mt) * 32 + r;
            pt = pt < a.P ? pt : a.P - 1;
            load_point(a, b, pt, x0[mt], x1[mt], x2[mt]);
        }
        // layer 1: lane half h computes channels 2s+h, the k slots of step s of layer 2
        float hb[MT][32];
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float4 w = s_w1[2 * s + h];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float v = fmaf(w.x, x0[mt], fmaf(w.y, x1[mt], fmaf(w.z, x2[mt], w.w)));
                hb[mt][s] = __builtin_fmaxf(v, 0.f);
            }
        }
        // layer 2 transposed; X[mt][t] register i holds channel 32t + acc_row(i, h) of point r
        f32x16 X[MT][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            {
                const f32x4* bq = (const f32x4*)(s_b2 + (t * 2 + h) * 16);
                f32x16 binit;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = bq[q];
                    binit[4 * q + 0] = v[0];
                    binit[4 * q + 1] = v[1];
                    binit[4 * q + 2] = v[2];
                    binit[4 * q + 3] = v[3];
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) X[mt][t] = binit;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 wq = *(const f32x4*)(s_w2 + ((t * 8 + q) * 64 + lane) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        X[mt][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[e], hb[mt][4 * q + e],
                                                                       X[mt][t], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) X[mt][t][i] = __builtin_fmaxf(X[mt][t][i], 0.f);
        }
        // layer 3: each W3 chunk is half a channel tile (k rows of conv2 tiles 2*half, 2*half+1)
        const bool last_pass = (pass == npass - 1);
        for (int nt = nt0; nt < nt1; ++nt) {
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x16{0};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ck = 2 * nt + half;
                if (!(last_pass && ck == ck_last)) {
                    const int nxt = ck == ck_last ? ck_first : ck + 1;      // chunk ck sits in buffer ck & 1
                    const char* src = g_w3 + (size_t)nxt * F32_CHUNK_BYTES;
                    char* dst = s_w3 + ((half + 1) & 1) * F32_CHUNK_BYTES;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int piece = wave * 2 + i;
                        dma_piece(src + piece * 1024, dst + piece * 1024, lane);
                    }
                }
                const char* buf = s_w3 + half * F32_CHUNK_BYTES;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 bw = *(const f32x4*)(buf + (q * 64 + lane) * 16);
                    const int t = 2 * half + (q >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                X[mt][t][4 * (q & 3) + e], bw[e], acc[mt], 0, 0, 0);
                }
                lds_dma_wait();         // the chunk issued at the top of this half: long landed behind the MFMAs
                __syncthreads();
            }
            float m = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) m = __builtin_fmaxf(m, max16(acc[mt]));
            merge_runmax(s_red_wave, nt, m, r, h);
        }
    }
    __syncthreads();
    write_pooled(a, b, s_red, tid, 32 * nt0, 32 * nt1);
}

static hipError_t check_args(const ConvStackArgs& a) {
    if (a.B < 1 || a.P < 1 || !a.x || !a.w1tab || !a.w2p || !a.b2p || !a.w3p || !a.b3 || !a.pooled)
        return hipErrorInvalidValue;
    return hipSuccess;
}

hipError_t launch_conv_stack_bf16(const ConvStackArgs& a, hipStream_t s) {
    hipError_t e = check_args(a);
    if (e != hipSuccess) return e;
    static_assert(WS_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    int n_cu = 0;
    e = device_cu_count(&n_cu);
    if (e != hipSuccess) return e;
    // persistent: one workgroup per CU, each strides over the candidates; a batch of fewer candidates than half
    // the CUs is cut into channel pieces (2 or 4 workgroups per candidate)
    auto launch = [&](auto kern, int pieces) {
        hipError_t er = set_max_lds_once((const void*)kern, WS_LDS_BYTES);
        if (er != hipSuccess) return er;
        const int per_piece = n_cu / pieces > 0 ? n_cu / pieces : 1;
        hipLaunchKernelGGL(kern, dim3(a.B < per_piece ? a.B : per_piece, pieces, a.point_groups), dim3(WS_THREADS),
                           WS_LDS_BYTES, s, a);
        return hipGetLastError();
    };
    if (a.point_groups > 1) {
        // the caller sized the groups with conv_stack_point_groups(): only batches that run the 4-piece cut
        if (a.B * 4 > n_cu || a.tiles_per_group < 1 || (a.point_groups - 1) * a.tiles_per_group >= ((a.P + 15) >> 4))
            return hipErrorInvalidValue;
        return launch(conv_stack_bf16_kernel<2, true>, 4);
    }
    if (a.B * 4 <= n_cu) return launch(conv_stack_bf16_kernel<2>, 4);
    if (a.B * 2 <= n_cu) return launch(conv_stack_bf16_kernel<4>, 2);
    return launch(conv_stack_bf16_kernel<8>, 1);
}

hipError_t launch_conv_stack_f32(const ConvStackArgs& a, hipStream_t s) {
    hipError_t e = check_args(a);
    if (e != hipSuccess) return e;
    static_assert(F32_LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    auto kern = conv_stack_f32_kernel<2>;
    e = set_max_lds_once((const void*)kern, F32_LDS_BYTES);
    if (e != hipSuccess) return e;
    int n_cu = 0;
    e = device_cu_count(&n_cu);
    if (e != hipSuccess) return e;
    // cut the 32 channel tiles of a candidate into pieces until there is a workgroup for every CU (measured at
    // B = 16 / 64: 187 / 329 us per forward against 231 / 366 us with two workgroups per CU; 860 us uncut)
    int pieces = 1;
    while (pieces < 32 && (long long)a.B * pieces < (long long)n_cu) pieces *= 2;
    hipLaunchKernelGGL(kern, dim3(a.B, pieces), dim3(THREADS), F32_LDS_BYTES, s, a, 32 / pieces);
    return hipGetLastError();
}

}  // namespace pngpd

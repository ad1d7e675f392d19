// Fused per-point MLP + max-pool at float32 grade on the bf16 matrix pipe ("bf16x3" precision):
//
//     pooled[b, :] = act( max_p  W3 * relu(W2 * relu(W1 * T_b^T x[b,p] + b1) + b2) ) + b3
//
// (STN3d.forward, model/pointnet.py:30-34, and PointNetfeat.forward, model/pointnet.py:138-147, with eval-mode
// BatchNorm folded into W/b), with the data flow of conv_stack_bf16_kernel (conv_stack.hip): layer 2 transposed so that
// its accumulators are layer 3's A operand, layer-3 weights stationary in registers, A fragments through a double
// buffer in LDS, per-lane running maxima.  What differs is the arithmetic (below) and the instruction stream (woven).  Every operand of layers 2 and 3 is TWO bf16
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
// The lo plane of the activations goes first: its registers are free after a third of the chain, the hi plane's
// only one by one at the end, which is the order in which the woven kernel refills them for the next tile.
__device__ __forceinline__ f32x4v mfma_chain12(const X3AFrag& f, const bf16x8 (&wh)[4], const bf16x8 (&wl)[4]) {
    f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.l[k], wh[k], z, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.h[k], wl[k], z, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.h[k], wh[k], z, 0, 0, 0);
    return z;
}

// ------------------------------------------------------------------------------------------------------------------
// The produce work is WOVEN into the consume stream.
//
// A first form of this kernel followed conv_stack_bf16_kernel: once per batch a wave left the consume loop for a
// produce step of ~250 vector and 52 matrix instructions.  Every wave here runs ONE instruction stream instead, in
// which the produce instructions of its tile sit in the issue slots that its own MFMAs leave free: a batch is 8
// tiles, tile step t carries row tile t of layer 2 (a chain of six MFMAs behind the step's first chain, the split of
// its result and the LDS traffic spread at one instruction per MFMA gap), and layer 1 of the tile after next closes
// the batch.  Nothing branches inside a batch: batches are always 8 tiles (rows beyond P repeat the last point, which
// cannot change a maximum) and a workgroup's last batch produces a tile that nobody reads.
// A wave consumes the tiles of a batch starting with its own (the one it produced): those fragments are fetched
// before the barrier that publishes the batch, so no LDS latency is exposed behind it.  Fragment registers are
// refilled in place: the lo plane of the next tile as soon as the last chain of a tile has used the current one (a
// third into the chain), the hi plane fragment by fragment behind the chain's last four MFMAs.
// Measured against the first form at B = 4096, P = 750 (one box, alternating): 1.42 -> 1.40 ms per launch; counters:
// matrix pipe 84 % busy at the 2.15 GHz the chip holds under this load; the same stream without the produce
// instructions runs at 96 %.
// ------------------------------------------------------------------------------------------------------------------
#define X3_SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define X3_SGB_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
#define X3_SGB_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define X3_SGB_DSW(n) __builtin_amdgcn_sched_group_barrier(0x200, n, 0)
#define X3_SGB_VMEM(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
#define X3_FENCE() __builtin_amdgcn_sched_barrier(0)

// conv2 fragments of one row tile: hi image from LDS, lo image from memory, [k-step]
struct X3W2Row {
    bf16x8 hi[2], lo[2];
    f32x4v bias;
};

__device__ __forceinline__ void load_w2_row_lds(X3W2Row& f, int T, const char* s_w2h, const float* s_b2, int lane) {
#pragma unroll
    for (int s = 0; s < 2; ++s) f.hi[s] = *(const bf16x8*)(s_w2h + ((T * 2 + s) * 64 + lane) * 16);
    f.bias = *(const f32x4v*)(s_b2 + 16 * T + 4 * (lane >> 4));
}

__device__ __forceinline__ void load_w2_row_mem(X3W2Row& f, int T, __amdgpu_buffer_rsrc_t g_w2l, unsigned int lane16) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
        f.lo[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(g_w2l, lane16, (T * 2 + s) * 1024, 0));
}

// one row tile of layer 2 for one point tile: bias, then the small piece products, then the large ones
__device__ __forceinline__ f32x4v l2_row_chain(const X3W2Row& w, const bf16x8 (&hbh)[2], const bf16x8 (&hbl)[2]) {
    f32x4v acc = w.bias;
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.lo[s], hbh[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi[s], hbl[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 2; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.hi[s], hbh[s], acc, 0, 0, 0);
    return acc;
}

// layer 1 of one tile: lane's coordinate -> the hi / lo B fragments of layer 2
__device__ __forceinline__ void layer1_x3(float px, const char* s_w1f, unsigned int ones, int lane, bf16x8 (&hbh)[2],
                                          bf16x8 (&hbl)[2]) {
    bf16x8 w1f[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) w1f[t] = *(const bf16x8*)(s_w1f + (t * 64 + lane) * 16);
    const unsigned int x = f2u(px);
    const float r1 = px - u2f(x & 0xffff0000u);
    const unsigned int m = f2u(r1);
    const unsigned int l = f2u(r1 - u2f(m & 0xffff0000u));
    u32x4 v;
    v[0] = __builtin_amdgcn_perm(x, x, 0x03020302u);      // xh, xh
    v[1] = __builtin_amdgcn_perm(x, m, 0x07060302u);      // xm, xh
    v[2] = __builtin_amdgcn_perm(l, m, 0x07060302u);      // xm, xl
    v[3] = ones;
    const bf16x8 xb = __builtin_bit_cast(bf16x8, v);
    const f32x4v zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const f32x4v h1a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * s], xb, zero, 0, 0, 0);
        const f32x4v h1b = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[2 * s + 1], xb, zero, 0, 0, 0);
        relu_split(h1a, h1b, hbh[s], hbl[s]);
    }
}

// Layers 1-2 of one tile of 16 points outright (the prologue's first batch); the two planes of its layer-3 A fragments
// go to dst, dst + X3_PLANE_BYTES.  Same arithmetic, instruction for instruction, as the woven steps of the kernel.
__device__ __forceinline__ void produce_tile_x3(float px, const char* s_w1f, unsigned int ones, const char* s_w2h,
                                                __amdgpu_buffer_rsrc_t g_w2l, const float* s_b2, char* dst, int lane) {
    bf16x8 hbh[2], hbl[2];
    layer1_x3(px, s_w1f, ones, lane, hbh, hbl);
    const unsigned int lane16 = (unsigned int)lane * 16u;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        unsigned int h[4], l[4];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            X3W2Row w;
            load_w2_row_lds(w, 2 * ks + tt, s_w2h, s_b2, lane);
            load_w2_row_mem(w, 2 * ks + tt, g_w2l, lane16);
            const f32x4v acc = l2_row_chain(w, hbh, hbl);
            relu_split2(acc[0], acc[1], h[2 * tt], l[2 * tt]);
            relu_split2(acc[2], acc[3], h[2 * tt + 1], l[2 * tt + 1]);
        }
        *(u32x4*)(dst + (ks * 64 + lane) * 16) = u32x4{h[0], h[1], h[2], h[3]};
        *(u32x4*)(dst + X3_PLANE_BYTES + (ks * 64 + lane) * 16) = u32x4{l[0], l[1], l[2], l[3]};
    }
}

// NT = channel tiles of 16 per wave.  4 is the throughput shape: a workgroup covers 512 channels, grid.y = 2.  A batch
// too small to give every CU a workgroup runs NT = 2 with grid.y = 4, or NT = 1 with grid.y = 8 (a single candidate,
// the reference's own call shape).  The arithmetic of a channel does not depend on the cut, so a candidate's result
// does not depend on the batch it is scored in.
// SPLIT (small batches only): grid.z also cuts the points of a candidate into groups of a.tiles_per_group tiles, each
// group writing its maxima to pooled + z * B * 1024 (conv_stack.hip, conv_stack_bf16_kernel).
template <int NT, bool SPLIT = false>
__global__ __launch_bounds__(X3_THREADS) void conv_stack_x3_kernel(ConvStackArgs a) {
    static_assert(NT == 4 || NT == 2 || NT == 1, "channel tiles per wave");
    static_assert(!SPLIT || NT == 1, "only the single-candidate shape is cut over points");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar branches and addresses
    const int c = lane & 15, q = lane >> 4;
    const unsigned int lane16 = (unsigned int)lane * 16u;

    char* s_a = smem;
    char* s_w2 = smem + X3_OFF_W2;
    float* s_b2 = (float*)(smem + X3_OFF_B2);
    float4* s_w1base = (float4*)(smem + X3_OFF_W1B);
    char* s_w1f = smem + X3_OFF_W1;
    float* s_b3 = (float*)(smem + X3_OFF_B3);
    const __amdgpu_buffer_rsrc_t g_w2l = __builtin_amdgcn_make_buffer_rsrc((void*)a.w2lo, 0, (int)W2P_BF16_BYTES, 0x00020000);
    const unsigned int ones = q == 0 ? 0x3f803f80u : q == 1 ? 0x00003f80u : 0u;

    const int ct0 = ((int)blockIdx.y * X3_WAVES + wave) * NT;
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
    if (tid < 2 * 4 * 16)
        *(u32x4*)(s_w1f + (tid >> 6) * W1F_BYTES + ((((tid >> 4) & 3) * 4 + 3) * 16 + (tid & 15)) * 16) = u32x4{0u, 0u, 0u, 0u};
    lds_dma_wait();
    __syncthreads();

    const int tile_base = SPLIT ? (int)blockIdx.z * a.tiles_per_group : 0;
    const int ntiles = SPLIT ? min(a.tiles_per_group, ((a.P + 15) >> 4) - tile_base) : (a.P + 15) >> 4;
    const int nbatch = (ntiles + X3_BATCH_TILES - 1) / X3_BATCH_TILES;      // every batch runs all 8 tiles
    float* pooled = SPLIT ? a.pooled + (size_t)blockIdx.z * a.B * C3 : a.pooled;
    const int stride = gridDim.x;
    const int ncand = (a.B - (int)blockIdx.x + stride - 1) / stride;

    // Layer-1 cursor: the (candidate, batch) whose tile `wave` gets its layer 1 NEXT, always two batches ahead of the
    // batch being consumed; px is this lane's coordinate of that tile's point.  Wave 0 writes a candidate's layer-1
    // table when the cursor reaches the candidate, into the slot the candidate before it does not use; a barrier lies
    // between that and the first read, and the slot's previous owner was last read two barriers earlier.
    int lb = blockIdx.x, lbt = 0, ltab = 0;
    float px = 0.f;
    auto fetch_point = [&]() {
        if (lb < a.B) px = load_tile_point(a, lb, tile_base + lbt * X3_BATCH_TILES + wave, c, q);
    };
    auto advance = [&]() {
        if (++lbt == nbatch) {
            lbt = 0;
            lb += stride;
            ltab ^= 1;
            if (wave == 0 && lb < a.B) fold_w1_frags(a, lb, s_w1base, s_w1f + ltab * W1F_BYTES, lane);
        }
        fetch_point();
    };
    bf16x8 hbh[2], hbl[2];
    // prologue: batch 0 is produced outright, layer 1 of batch 1 is left in hbh / hbl, px belongs to batch 2
    if (wave == 0) fold_w1_frags(a, lb, s_w1base, s_w1f, lane);
    fetch_point();
    __syncthreads();
    produce_tile_x3(px, s_w1f + ltab * W1F_BYTES, ones, s_w2, g_w2l, s_b2, s_a + wave * X3_TILE_BYTES, lane);
    advance();
    __syncthreads();
    layer1_x3(px, s_w1f + ltab * W1F_BYTES, ones, lane, hbh, hbl);
    advance();
    __syncthreads();

    // fragments of this wave's first tile (its own) of the first batch, and conv2's first row tile
    X3AFrag cf;
    load_afrags_x3(cf, s_a + wave * X3_TILE_BYTES, lane);
    X3W2Row w2r;
    load_w2_row_mem(w2r, 0, g_w2l, lane16);

    int buf = 0;
    for (int ci = 0; ci < ncand; ++ci) {
        const int b = blockIdx.x + ci * stride;
        float rm[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) rm[nt] = -INFINITY;
        for (int bt = 0; bt < nbatch; ++bt, buf ^= 1) {
            const char* cur = s_a + buf * X3_BATCH_BYTES;
            char* dst = s_a + (buf ^ 1) * X3_BATCH_BYTES + wave * X3_TILE_BYTES;
            f32x4v carry = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            unsigned int keep_h[2] = {0u, 0u}, keep_l[2] = {0u, 0u};       // even row tile of a pair, split and packed
#pragma unroll
            for (int t = 0; t < X3_BATCH_TILES; ++t) {
                // the tile consumed after this one: the wave's own tile of the next batch behind the last step
                const char* nxt_tile = t + 1 < X3_BATCH_TILES ? cur + ((wave + t + 1) & 7) * X3_TILE_BYTES : dst;
                // A: chain 0 | the max of the previous tile's last chain, conv2's hi fragments and bias of row tile t
                X3_FENCE();
                f32x4v acc0 = {0.f, 0.f, 0.f, 0.f};
                if constexpr (NT >= 2) acc0 = mfma_chain12(cf, bwh[0], bwl[0]);
                rm[NT - 1] = pinned_max(rm[NT - 1], carry);
                load_w2_row_lds(w2r, t, s_w2, s_b2, lane);
                if constexpr (NT >= 2) {
                    X3_SGB_MFMA(1); X3_SGB_VALU(1); X3_SGB_MFMA(1); X3_SGB_VALU(1); X3_SGB_MFMA(1); X3_SGB_DSR(1);
                    X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(7);
                }
                X3_FENCE();
                // B: row tile t of layer 2 (a chain of six) | then the lo fragments of the next row tile
                const f32x4v pacc = l2_row_chain(w2r, hbh, hbl);
                load_w2_row_mem(w2r, (t + 1) & 7, g_w2l, lane16);
                X3_SGB_MFMA(6); X3_SGB_VMEM(2);
                X3_FENCE();
                unsigned int ph[2], pl[2];
                if constexpr (NT == 4) {
                    // C: chain 1 | max of chain 0, split of the row tile's first two values
                    f32x4v acc1 = mfma_chain12(cf, bwh[1], bwl[1]);
                    rm[0] = pinned_max(rm[0], acc0);
                    relu_split2(pacc[0], pacc[1], ph[0], pl[0]);
#pragma unroll
                    for (int i = 0; i < 10; ++i) { X3_SGB_MFMA(1); X3_SGB_VALU(1); }
                    X3_SGB_MFMA(2);
                    X3_FENCE();
                    // D: chain 2 | max of chain 1, split of the other two values, the stores of a completed pair
                    acc0 = mfma_chain12(cf, bwh[2], bwl[2]);
                    rm[1] = pinned_max(rm[1], acc1);
                    relu_split2(pacc[2], pacc[3], ph[1], pl[1]);
                    if (t & 1) {
                        *(u32x4*)(dst + ((t >> 1) * 64 + lane) * 16) = u32x4{keep_h[0], keep_h[1], ph[0], ph[1]};
                        *(u32x4*)(dst + X3_PLANE_BYTES + ((t >> 1) * 64 + lane) * 16) = u32x4{keep_l[0], keep_l[1], pl[0], pl[1]};
                    }
#pragma unroll
                    for (int i = 0; i < 10; ++i) { X3_SGB_MFMA(1); X3_SGB_VALU(1); }
                    if (t & 1) { X3_SGB_MFMA(1); X3_SGB_DSW(1); X3_SGB_MFMA(1); X3_SGB_DSW(1); } else { X3_SGB_MFMA(2); }
                    X3_FENCE();
                } else {
                    relu_split2(pacc[0], pacc[1], ph[0], pl[0]);
                    relu_split2(pacc[2], pacc[3], ph[1], pl[1]);
                    if (t & 1) {
                        *(u32x4*)(dst + ((t >> 1) * 64 + lane) * 16) = u32x4{keep_h[0], keep_h[1], ph[0], ph[1]};
                        *(u32x4*)(dst + X3_PLANE_BYTES + ((t >> 1) * 64 + lane) * 16) = u32x4{keep_l[0], keep_l[1], pl[0], pl[1]};
                    }
                    X3_FENCE();
                }
                if (!(t & 1)) { keep_h[0] = ph[0]; keep_h[1] = ph[1]; keep_l[0] = pl[0]; keep_l[1] = pl[1]; }
                // E: the last chain; every fragment register is refilled for the next tile as soon as its last MFMA
                // has been issued (lo plane after four MFMAs, hi plane one by one at the end)
                f32x4v z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf.l[k], bwh[NT - 1][k], z, 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) cf.l[k] = *(const bf16x8*)(nxt_tile + X3_PLANE_BYTES + (k * 64 + lane) * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf.h[k], bwl[NT - 1][k], z, 0, 0, 0);
                if constexpr (NT >= 2) rm[NT - 2] = pinned_max(rm[NT - 2], acc0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cf.h[k], bwh[NT - 1][k], z, 0, 0, 0);
                    cf.h[k] = *(const bf16x8*)(nxt_tile + (k * 64 + lane) * 16);
                }
                X3_SGB_MFMA(4);
                X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1);
                X3_SGB_VALU(2);
                X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1); X3_SGB_MFMA(1); X3_SGB_DSR(1);
                X3_FENCE();
                carry = z;
            }
            rm[NT - 1] = pinned_max(rm[NT - 1], carry);
            // tail of the batch: layer 1 of the tile after next (the fragments of this wave's own tile of the next batch
            // are already on their way)
            layer1_x3(px, s_w1f + ltab * W1F_BYTES, ones, lane, hbh, hbl);
            advance();
            lds_barrier();
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ch = ((int)blockIdx.y * X3_WAVES + wave) * (16 * NT) + nt * 16 + c;
            float m = __builtin_fmaxf(rm[nt], __shfl_xor(rm[nt], 16));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32)) + s_b3[ch];
            if (a.relu_out) m = __builtin_fmaxf(m, 0.f);
            if (q == 0) pooled[(size_t)b * C3 + ch] = m;
        }
    }
}

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
        hipLaunchKernelGGL(kern, dim3(a.B < per_piece ? a.B : per_piece, pieces, a.point_groups), dim3(X3_THREADS),
                           X3_LDS_BYTES, s, a);
        return hipGetLastError();
    };
    if (a.point_groups > 1) {
        // the caller sized the groups with conv_stack_point_groups(): only batches that run the 8-piece cut
        if (a.B * 8 > n_cu || a.tiles_per_group < 1 || (a.point_groups - 1) * a.tiles_per_group >= ((a.P + 15) >> 4))
            return hipErrorInvalidValue;
        return launch(conv_stack_x3_kernel<1, true>, 8);
    }
    if (a.B * 8 <= n_cu) return launch(conv_stack_x3_kernel<1>, 8);
    if (a.B * 4 <= n_cu) return launch(conv_stack_x3_kernel<2>, 4);
    return launch(conv_stack_x3_kernel<4>, 2);
}

hipError_t conv_stack_point_groups(int precision, int B, int P, int* groups, int* tiles_per_group) {
    *groups = 1;
    *tiles_per_group = 0;
    if (B < 1 || P < 1) return hipErrorInvalidValue;
    if (B > CONV_SPLIT_MAX_B || (precision != PNGPD_BF16 && precision != PNGPD_BF16X3)) return hipSuccess;
    int n_cu = 0;
    hipError_t e = device_cu_count(&n_cu);
    if (e != hipSuccess) return e;
    // channel pieces and tiles per batch of the kernels that run such a batch (launch_conv_stack_bf16 / _x3)
    const int pieces = precision == PNGPD_BF16 ? 4 : 8;
    const int batch_tiles = precision == PNGPD_BF16 ? 16 : X3_BATCH_TILES;
    if (B * pieces > n_cu) return hipSuccess;
    int room = n_cu / (B * pieces);                    // groups for which every workgroup still has a CU of its own
    if (room > CONV_SPLIT_MAX_GROUPS) room = CONV_SPLIT_MAX_GROUPS;
    if (room < 2) return hipSuccess;
    const int ntiles = (P + 15) >> 4;
    const int nbatch = (ntiles + batch_tiles - 1) / batch_tiles;
    const int tpg = (nbatch + room - 1) / room * batch_tiles;      // whole batches per group
    const int g = (ntiles + tpg - 1) / tpg;
    if (g < 2) return hipSuccess;
    *groups = g;
    *tiles_per_group = tpg;
    return hipSuccess;
}

}  // namespace pngpd

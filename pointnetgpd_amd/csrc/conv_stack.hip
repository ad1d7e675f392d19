// Fused per-point MLP + symmetric max-pool for gfx950:
//
//     pooled[b, :] = act( max_p  W3 * relu(W2 * relu(W1 * T_b^T x[b,p] + b1) + b2) ) + b3
//
// which is conv1->bn->relu->conv2->bn->relu->conv3->bn(->relu)->MaxPool1d(num_points) of
// STN3d.forward (model/pointnet.py:30-34) and PointNetfeat.forward (model/pointnet.py:138-147)
// with eval-mode BatchNorm folded into W/b.  The [B,1024,P] activation of the reference never
// exists: each wave keeps a tile of points in registers from the input coordinates to the
// per-channel maximum.
//
// Work split (one workgroup = one candidate, 4 waves):
//   * a pass covers 128*MT points; each wave owns MT tiles of 32 points;
//   * layer 1 (K = 3) runs on the VALU, straight into the MFMA B-operand layout of layer 2;
//   * layer 2 is computed TRANSPOSED, X[ch2][point] = W2^T-fragments * h1-fragments, so that the
//     accumulator (point on the lane, channel in the register) is, after ReLU, directly the A
//     operand of layer 3 (no LDS round trip, no cross-lane traffic);
//   * layer 3 computes Z[point][ch3]: channel on the lane, points in the registers, so the
//     max over points is a per-lane register reduction; W3 streams through LDS by LDS-DMA in
//     fragment order (double buffered, one barrier per chunk) and is shared by the 4 waves;
//   * per-wave running maxima live in LDS, merged and written once per candidate.
// Points beyond P are clamped to point P-1: duplicates cannot change a maximum.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "pngpd.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int WAVES = 4;
constexpr int THREADS = WAVES * 64;

// One wave-instruction of LDS-DMA: 64 lanes x 16 B, LDS destination = lds_piece + lane*16.
__device__ __forceinline__ void dma_piece(const char* gsrc_piece, char* lds_piece, int lane) {
    __builtin_amdgcn_global_load_lds((const void*)(gsrc_piece + lane * 16), (lds_void*)lds_piece,
                                     16, 0, 0);
}

__device__ __forceinline__ float max3(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}

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
                                             int tid) {
    for (int c = tid; c < C3; c += THREADS) {
        float m = __builtin_fmaxf(__builtin_fmaxf(s_red[c], s_red[C3 + c]),
                                  __builtin_fmaxf(s_red[2 * C3 + c], s_red[3 * C3 + c]));
        m += a.b3[c];
        if (a.relu_out) m = __builtin_fmaxf(m, 0.f);
        a.pooled[(size_t)b * C3 + c] = m;
    }
}

// ------------------------------------------------------------------------------------------
// bf16 MFMA path (v_mfma_f32_32x32x16_bf16, f32 accumulate).
// LDS: [0,32K) two W3 chunk buffers (2 ch3-tiles = 16 KiB each) | [32K,48K) W2 fragments |
//      [48K,49K) conv1 table | [49K,49.5K) conv2 bias | [49.5K,65.5K) per-wave running maxima.
// ------------------------------------------------------------------------------------------
constexpr int BF_CHUNK_BYTES = 16384;
constexpr int BF_CHUNKS = 16;
constexpr int BF_OFF_W2 = 2 * BF_CHUNK_BYTES;
constexpr int BF_OFF_W1 = BF_OFF_W2 + (int)W2P_BF16_BYTES;
constexpr int BF_OFF_B2 = BF_OFF_W1 + C1 * 16;
constexpr int BF_OFF_RED = BF_OFF_B2 + B2P_FLOATS * 4;
constexpr int BF_LDS_BYTES = BF_OFF_RED + WAVES * C3 * 4;

template <int MT>
__global__ __launch_bounds__(THREADS, 2) void conv_stack_bf16_kernel(ConvStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;

    char* s_w3 = smem;
    char* s_w2 = smem + BF_OFF_W2;
    float4* s_w1 = (float4*)(smem + BF_OFF_W1);
    float* s_b2 = (float*)(smem + BF_OFF_B2);
    float* s_red = (float*)(smem + BF_OFF_RED);
    const char* g_w3 = (const char*)a.w3p;

    // prologue: W2 fragments and W3 chunk 0 by LDS-DMA, small tables by ordinary stores
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
        dma_piece(g_w3 + piece * 1024, s_w3 + piece * 1024, lane);
    }
    if (tid < C1) s_w1[tid] = fold_w1(a, b, tid);
    if (tid < B2P_FLOATS) s_b2[tid] = a.b2p[tid];
    for (int c = tid; c < WAVES * C3; c += THREADS) s_red[c] = -INFINITY;
    __syncthreads();

    float* s_red_wave = s_red + wave * C3;
    const int pts_per_pass = WAVES * 32 * MT;
    const int npass = (a.P + pts_per_pass - 1) / pts_per_pass;

    for (int pass = 0; pass < npass; ++pass) {
        // ---- layer 1 on the VALU, written as the B fragments of layer 2 ---------------------
        float x0[MT], x1[MT], x2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int pt = pass * pts_per_pass + (wave * MT + mt) * 32 + r;
            pt = pt < a.P ? pt : a.P - 1;
            load_point(a, b, pt, x0[mt], x1[mt], x2[mt]);
        }
        bf16x8 hb[MT][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 w = s_w1[16 * s + 8 * h + j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float v = fmaf(w.x, x0[mt], fmaf(w.y, x1[mt], fmaf(w.z, x2[mt], w.w)));
                    hb[mt][s][j] = (__bf16)__builtin_fmaxf(v, 0.f);
                }
            }
        }
        // ---- layer 2, transposed: X[ch2][pt]; ReLU + pack = A fragments of layer 3 -----------
        bf16x8 a3[MT][8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 acc[MT];
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
                for (int mt = 0; mt < MT; ++mt) acc[mt] = binit;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 wf = *(const bf16x8*)(s_w2 + ((t * 4 + s) * 64 + lane) * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hb[mt][s], acc[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        a3[mt][2 * t + sp][j] = (__bf16)__builtin_fmaxf(acc[mt][8 * sp + j], 0.f);
        }
        // ---- layer 3 + max over points; W3 streamed in 16 chunks of 2 channel tiles ----------
        const bool last_pass = (pass == npass - 1);
#pragma unroll 2
        for (int ck = 0; ck < BF_CHUNKS; ++ck) {
            if (!(last_pass && ck == BF_CHUNKS - 1)) {
                const int nxt = (ck + 1) & (BF_CHUNKS - 1);
                const char* src = g_w3 + (size_t)nxt * BF_CHUNK_BYTES;
                char* dst = s_w3 + ((ck + 1) & 1) * BF_CHUNK_BYTES;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int piece = wave * 4 + i;
                    dma_piece(src + piece * 1024, dst + piece * 1024, lane);
                }
            }
            const char* buf = s_w3 + (ck & 1) * BF_CHUNK_BYTES;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                bf16x8 bf[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks)
                    bf[ks] = *(const bf16x8*)(buf + ((u * 8 + ks) * 64 + lane) * 16);
                float m = -INFINITY;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    f32x16 acc = {0};
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[mt][ks], bf[ks], acc, 0, 0, 0);
                    m = __builtin_fmaxf(m, max16(acc));
                }
                merge_runmax(s_red_wave, 2 * ck + u, m, r, h);
            }
            __syncthreads();
        }
    }
    write_pooled(a, b, s_red, tid);
}

// ------------------------------------------------------------------------------------------
// bf16 path, software-pipelined inside each wave.
// The plain kernel above serialises, per channel tile: LDS fragment reads -> 16 MFMAs -> max
// reduction -> LDS merge, so one wave keeps the matrix pipe busy ~60 % of the time.  Here a wave
// keeps two fragment sets and two accumulator sets: while the MFMAs of tile t run it (a) has the
// fragments of tile t+1 in flight and (b) reduces the accumulators of tile t-1 on the VALU.
// The chunk barrier sits in the MIDDLE of the second tile of each chunk: by then every wave holds
// both tiles of the chunk in registers, so the same barrier that publishes chunk c+1 also frees
// chunk c's buffer for the LDS-DMA of chunk c+2 (two buffers suffice).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_frags(bf16x8 (&bf)[8], const char* tile_base, int lane) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bf[ks] = *(const bf16x8*)(tile_base + (ks * 64 + lane) * 16);
}

__device__ __forceinline__ f32x16 mfma_chain(const bf16x8 (&a)[8], const bf16x8 (&b)[8]) {
    f32x16 acc = {0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks], b[ks], acc, 0, 0, 0);
    return acc;
}

template <int MT>
__device__ __forceinline__ void reduce_tile(const f32x16 (&acc)[MT], float* s_red_wave, int nt, int r) {
    float m = max16(acc[0]);
#pragma unroll
    for (int mt = 1; mt < MT; ++mt) m = __builtin_fmaxf(m, max16(acc[mt]));
    m = __builtin_fmaxf(m, __shfl_xor(m, 32));
    float* p = s_red_wave + 32 * nt + r;        // lanes l and l+32 store the same value
    *p = __builtin_fmaxf(*p, m);
}

template <int MT>
__global__ __launch_bounds__(THREADS, 2) void conv_stack_bf16_pipelined_kernel(ConvStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;

    char* s_w3 = smem;
    char* s_w2 = smem + BF_OFF_W2;
    float4* s_w1 = (float4*)(smem + BF_OFF_W1);
    float* s_b2 = (float*)(smem + BF_OFF_B2);
    float* s_red = (float*)(smem + BF_OFF_RED);
    const char* g_w3 = (const char*)a.w3p;

    // prologue: W2 fragments and W3 chunks 0 and 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int piece = wave * 4 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
        dma_piece(g_w3 + piece * 1024, s_w3 + piece * 1024, lane);
        dma_piece(g_w3 + BF_CHUNK_BYTES + piece * 1024, s_w3 + BF_CHUNK_BYTES + piece * 1024, lane);
    }
    if (tid < C1) s_w1[tid] = fold_w1(a, b, tid);
    if (tid < B2P_FLOATS) s_b2[tid] = a.b2p[tid];
    for (int c = tid; c < WAVES * C3; c += THREADS) s_red[c] = -INFINITY;
    __syncthreads();

    float* s_red_wave = s_red + wave * C3;
    const int pts_per_pass = WAVES * 32 * MT;
    const int npass = (a.P + pts_per_pass - 1) / pts_per_pass;

    bf16x8 bfA[8], bfB[8];
    load_frags(bfA, s_w3, lane);                       // tile 0 of pass 0
    bf16x8 a3[MT][8];

    for (int pass = 0; pass < npass; ++pass) {
        // ---- layer 1 (VALU) and layer 2 (transposed MFMA): A fragments of layer 3 ------------
        {
        float x0[MT], x1[MT], x2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int pt = pass * pts_per_pass + (wave * MT + mt) * 32 + r;
            pt = pt < a.P ? pt : a.P - 1;
            load_point(a, b, pt, x0[mt], x1[mt], x2[mt]);
        }
        bf16x8 hb[MT][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 w = s_w1[16 * s + 8 * h + j];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float v = fmaf(w.x, x0[mt], fmaf(w.y, x1[mt], fmaf(w.z, x2[mt], w.w)));
                    hb[mt][s][j] = (__bf16)__builtin_fmaxf(v, 0.f);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 acc[MT];
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
                for (int mt = 0; mt < MT; ++mt) acc[mt] = binit;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8 wf = *(const bf16x8*)(s_w2 + ((t * 4 + s) * 64 + lane) * 16);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, hb[mt][s], acc[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        a3[mt][2 * t + sp][j] = (__bf16)__builtin_fmaxf(acc[mt][8 * sp + j], 0.f);
        }
        }
        // ---- layer 3, two tiles (one chunk) per iteration ------------------------------------
        const bool last_pass = (pass == npass - 1);
        f32x16 acc0[MT], acc1[MT];
        for (int ck = 0; ck < BF_CHUNKS; ++ck) {
            char* bufc = s_w3 + (ck & 1) * BF_CHUNK_BYTES;
            const char* bufn = s_w3 + ((ck + 1) & 1) * BF_CHUNK_BYTES;
            // tile 2ck: fragments already in bfA; fetch tile 2ck+1 of the same chunk
            load_frags(bfB, bufc + BF_CHUNK_BYTES / 2, lane);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc0[mt] = mfma_chain(a3[mt], bfA);
            if (ck > 0) reduce_tile<MT>(acc1, s_red_wave, 2 * ck - 1, r);
            // tile 2ck+1: first chain, then the chunk barrier, then the rest
            acc1[0] = mfma_chain(a3[0], bfB);
            __syncthreads();                                   // chunk ck+1 landed; chunk ck's buffer is free
            if (!(last_pass && ck >= BF_CHUNKS - 2)) {
                const char* src = g_w3 + (size_t)((ck + 2) & (BF_CHUNKS - 1)) * BF_CHUNK_BYTES;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int piece = wave * 4 + i;
                    dma_piece(src + piece * 1024, bufc + piece * 1024, lane);
                }
            }
            if (!(last_pass && ck == BF_CHUNKS - 1)) load_frags(bfA, bufn, lane);   // tile 2ck+2
#pragma unroll
            for (int mt = 1; mt < MT; ++mt) acc1[mt] = mfma_chain(a3[mt], bfB);
            reduce_tile<MT>(acc0, s_red_wave, 2 * ck, r);
        }
        reduce_tile<MT>(acc1, s_red_wave, 2 * BF_CHUNKS - 1, r);
    }
    __syncthreads();
    write_pooled(a, b, s_red, tid);
}

// ------------------------------------------------------------------------------------------
// exact-f32 MFMA path (v_mfma_f32_32x32x2_f32): every product is an f32 fma chain.
// LDS: [0,16K) two W3 half-tile buffers (8 KiB) | [16K,48K) W2 fragments | conv1 table |
//      conv2 bias | per-wave running maxima.
// ------------------------------------------------------------------------------------------
constexpr int F32_CHUNK_BYTES = 8192;
constexpr int F32_CHUNKS = 64;
constexpr int F32_OFF_W2 = 2 * F32_CHUNK_BYTES;
constexpr int F32_OFF_W1 = F32_OFF_W2 + (int)W2P_F32_BYTES;
constexpr int F32_OFF_B2 = F32_OFF_W1 + C1 * 16;
constexpr int F32_OFF_RED = F32_OFF_B2 + B2P_FLOATS * 4;
constexpr int F32_LDS_BYTES = F32_OFF_RED + WAVES * C3 * 4;

template <int MT>
__global__ __launch_bounds__(THREADS, 2) void conv_stack_f32_kernel(ConvStackArgs a) {
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

#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int piece = wave * 8 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        dma_piece(g_w3 + piece * 1024, s_w3 + piece * 1024, lane);
    }
    if (tid < C1) s_w1[tid] = fold_w1(a, b, tid);
    if (tid < B2P_FLOATS) s_b2[tid] = a.b2p[tid];
    for (int c = tid; c < WAVES * C3; c += THREADS) s_red[c] = -INFINITY;
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
        for (int nt = 0; nt < 32; ++nt) {
            f32x16 acc[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x16{0};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ck = 2 * nt + half;
                if (!(last_pass && ck == F32_CHUNKS - 1)) {
                    const int nxt = (ck + 1) & (F32_CHUNKS - 1);
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
                __syncthreads();
            }
            float m = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) m = __builtin_fmaxf(m, max16(acc[mt]));
            merge_runmax(s_red_wave, nt, m, r, h);
        }
    }
    __syncthreads();
    write_pooled(a, b, s_red, tid);
}

static hipError_t check_args(const ConvStackArgs& a) {
    if (a.B < 1 || a.P < 1 || !a.x || !a.w1tab || !a.w2p || !a.b2p || !a.w3p || !a.b3 || !a.pooled)
        return hipErrorInvalidValue;
    return hipSuccess;
}

hipError_t launch_conv_stack_bf16(const ConvStackArgs& a, hipStream_t s) {
    hipError_t e = check_args(a);
    if (e != hipSuccess) return e;
    static_assert(BF_LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    // points per pass = 128 * MT: take the MT with the least padding (ties: fewer passes).
    // MT = 3 runs the plain kernel (the pipelined one would spill), MT = 2 the pipelined one.
    const int pad2 = (a.P + 255) / 256 * 256, pad3 = (a.P + 383) / 384 * 384;
    void (*kern)(ConvStackArgs) = pad3 <= pad2 ? conv_stack_bf16_kernel<3> : conv_stack_bf16_pipelined_kernel<2>;
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, BF_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(THREADS), BF_LDS_BYTES, s, a);
    return hipGetLastError();
}

hipError_t launch_conv_stack_f32(const ConvStackArgs& a, hipStream_t s) {
    hipError_t e = check_args(a);
    if (e != hipSuccess) return e;
    static_assert(F32_LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    auto kern = conv_stack_f32_kernel<2>;
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, F32_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.B), dim3(THREADS), F32_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace pngpd

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
//   * layer 1 (K = 3) runs on the VALU, straight into the MFMA B-operand layout of layer 2, with
//     the candidate's 3x3 input transform folded into the 64x3 weight table;
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
//      per-wave conv1 tables (trans folded in; private to a wave, so no workgroup barrier).
// ------------------------------------------------------------------------------------------
constexpr int WS_WAVES = 8;
constexpr int WS_THREADS = WS_WAVES * 64;
constexpr int W16_TILE_BYTES = 4096;                       // [ks4][lane64][16 B]
constexpr int W16_BATCH_TILES = 2 * WS_WAVES;
constexpr int WS_BATCH_BYTES = W16_BATCH_TILES * W16_TILE_BYTES;
constexpr int WS_OFF_W2 = 2 * WS_BATCH_BYTES;
constexpr int WS_OFF_B2 = WS_OFF_W2 + (int)W2P_BF16_BYTES;
constexpr int WS_OFF_W1 = WS_OFF_B2 + C2 * 4;
constexpr int WS_LDS_BYTES = WS_OFF_W1 + WS_WAVES * C1 * 16;
typedef f32x4 f32x4v;

__device__ __forceinline__ void produce_tiles(const ConvStackArgs& a, int b, int tile0, float4* s_w1w,
                                             const char* s_w2, const float* s_b2, char* dst, int lane) {
    const int c = lane & 15, q = lane >> 4;
    s_w1w[lane] = fold_w1(a, b, lane);
    float x0[2], x1[2], x2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int pt = (tile0 + u) * 16 + c;
        pt = pt < a.P ? pt : a.P - 1;
        load_point(a, b, pt, x0[u], x1[u], x2[u]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    bf16x8 hb[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float4 w = s_w1w[32 * s + 8 * q + j];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float v = fmaf(w.x, x0[u], fmaf(w.y, x1[u], fmaf(w.z, x2[u], w.w)));
                hb[u][s][j] = (__bf16)__builtin_fmaxf(v, 0.f);
            }
        }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        f32x4v acc[2][2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int T = 2 * ks + tt;
            const f32x4v bias = *(const f32x4v*)(s_b2 + 16 * T + 4 * q);
            acc[0][tt] = bias;
            acc[1][tt] = bias;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 wf = *(const bf16x8*)(s_w2 + ((T * 2 + s) * 64 + lane) * 16);
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, hb[u][s], acc[u][tt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            bf16x8 f;
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = (__bf16)__builtin_fmaxf(acc[u][j >> 2][j & 3], 0.f);
            *(bf16x8*)(dst + u * W16_TILE_BYTES + (ks * 64 + lane) * 16) = f;
        }
    }
}

__global__ __launch_bounds__(WS_THREADS) void conv_stack_bf16_kernel(ConvStackArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;

    char* s_a = smem;
    char* s_w2 = smem + WS_OFF_W2;
    float* s_b2 = (float*)(smem + WS_OFF_B2);
    float4* s_w1w = (float4*)(smem + WS_OFF_W1) + wave * C1;

    bf16x8 bw[8][4];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            bw[nt][ks] = *(const bf16x8*)((const char*)a.w3p + ((((wave * 8 + nt) * 4 + ks) * 64) + lane) * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        dma_piece((const char*)a.w2p + piece * 1024, s_w2 + piece * 1024, lane);
    }
    if (tid < C2) s_b2[tid] = a.b2p[tid];
    __syncthreads();

    const int ntiles = (a.P + 15) >> 4;
    const int nbatch = (ntiles + W16_BATCH_TILES - 1) / W16_BATCH_TILES;
    const int stride = gridDim.x;
    const int ncand = (a.B - (int)blockIdx.x + stride - 1) / stride;

    if (2 * wave < ntiles)
        produce_tiles(a, blockIdx.x, 2 * wave, s_w1w, s_w2, s_b2, s_a + wave * 2 * W16_TILE_BYTES, lane);
    __syncthreads();

    int buf = 0;
    for (int ci = 0; ci < ncand; ++ci) {
        const int b = blockIdx.x + ci * stride;
        float rm[8];
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) rm[nt] = -INFINITY;
        for (int bt = 0; bt < nbatch; ++bt, buf ^= 1) {
            const int nt_b = min(W16_BATCH_TILES, ntiles - W16_BATCH_TILES * bt);
            int nb = b, nbt = bt + 1;
            bool have_next = true;
            if (nbt == nbatch) {
                nbt = 0;
                nb = b + stride;
                have_next = ci + 1 < ncand;
            }
            const int my_tile = nbt * W16_BATCH_TILES + 2 * wave;
            have_next = have_next && my_tile < ntiles;
            const int slot = 2 * wave < nt_b ? 2 * wave : 0;
            const char* cur = s_a + buf * WS_BATCH_BYTES;
            char* nxt = s_a + (buf ^ 1) * WS_BATCH_BYTES + wave * 2 * W16_TILE_BYTES;
            for (int t = 0; t < nt_b; ++t) {
                if (t == slot && have_next) produce_tiles(a, nb, my_tile, s_w1w, s_w2, s_b2, nxt, lane);
                bf16x8 af[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    af[ks] = *(const bf16x8*)(cur + t * W16_TILE_BYTES + (ks * 64 + lane) * 16);
                f32x4v acc[8];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt)
                        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks], bw[nt][ks], acc[nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt)
                    rm[nt] = max3(rm[nt], max3(acc[nt][0], acc[nt][1], acc[nt][2]), acc[nt][3]);
            }
            __syncthreads();
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const int ch = wave * 128 + nt * 16 + c;
            float m = __builtin_fmaxf(rm[nt], __shfl_xor(rm[nt], 16));
            m = __builtin_fmaxf(m, __shfl_xor(m, 32)) + a.b3[ch];
            if (a.relu_out) m = __builtin_fmaxf(m, 0.f);
            if (q == 0) a.pooled[(size_t)b * C3 + ch] = m;
        }
    }
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
    static_assert(WS_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
    int dev = 0, n_cu = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    if (n_cu < 1) return hipErrorInvalidDevice;
    e = hipFuncSetAttribute((const void*)conv_stack_bf16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            WS_LDS_BYTES);
    if (e != hipSuccess) return e;
    // persistent: one workgroup per CU, each strides over the candidates
    hipLaunchKernelGGL(conv_stack_bf16_kernel, dim3(a.B < n_cu ? a.B : n_cu), dim3(WS_THREADS), WS_LDS_BYTES, s, a);
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

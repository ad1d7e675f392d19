// Fused passes over the per-point layers of a stack in the training step (conv12_fwd_kernel: forward of the first two
// layers; conv2_bwd_fused_kernel: the backward of the second).
//
// Backward of the second per-point layer of a stack in one pass over its rows:
//
//     dy2 = gamma2 rstd2 (g - m1 - yhat m2),  g = dh2 masked by the ReLU behind BatchNorm 2,  yhat = (y2 - mean2) rstd2
//     dh1 = dy2 W2                                                   ([M,128] x [128,64], exact f32 on the matrix pipe)
//     s1[c] = sum_m g1[m,c],  s2[c] = sum_m g1[m,c] yhat1[m,c]       (column sums that BatchNorm 1's backward needs)
//
// (model/pointnet.py:30-31 / :143-146 run backwards: conv2 = a row-wise linear map, bn2 in training mode, relu.)
// Done as three kernels this moves dh2 four times and y2 twice; here a lane reads its half row of dh2 and y2 once,
// turns it into dy2 in registers, stores it (the weight-gradient product reads it next) and feeds the matrix pipe from
// the same registers.  The product's accumulators are masked with the sign of conv1's recomputed pre-activation
// (three fmas from the row's coordinates) and summed per column before dh1 is stored, so the column reduction over
// dh1 needs no pass of its own.  Structure as tall_gemm_kernel<128,64>: persistent workgroups, W2 resident in LDS, the
// next half tile (128 rows x 64 columns of dh2 and y2) in flight during the MFMAs.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "kernels.h"
#include "train_act.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
constexpr int FK = 128, FN = 64, FNT = FN / 32;
// W2 row k in LDS: every group of 32 rows starts 32 floats later than plain row-major, so the two rows an MFMA step
// reads (k and k + 32, one per half wave) lie on different halves of the banks
constexpr int F_W2_FLOATS = FK * FN + 32 * (FK / 32);
__device__ __forceinline__ int w2_row_offset(int k) { return k * FN + 32 * (k >> 5); }
constexpr int F_CONST = 7;      // per column of layer 2: mean, rstd, gamma, beta, m1, m2, gamma * rstd
constexpr int F_CONST1 = 8;     // per column of layer 1: three weights, bias, mean, rstd, gamma, beta
constexpr size_t F_LDS_BYTES = ((size_t)F_W2_FLOATS + F_CONST * FK + F_CONST1 * FN) * sizeof(float);
constexpr int F_X3_PLANE_BYTES = (FN / 32) * 8 * 64 * 16;      // W2 as B fragments of v_mfma_f32_32x32x16_bf16, one piece
static_assert(2 * F_X3_PLANE_BYTES <= F_W2_FLOATS * 4, "the two bfloat16 planes of W2 fit the float32 image's LDS");

typedef __attribute__((ext_vector_type(8))) __bf16 dg_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 dg_bf16x2;
typedef __attribute__((ext_vector_type(2))) float dg_f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int dg_u32x4;
constexpr int DG_PLANE_BYTES = 4 * 8 * 64 * 16;       // [column tile 4][step 8][lane 64] x 16 B
constexpr size_t DG_LDS_BYTES = 2 * DG_PLANE_BYTES;

// hi and lo pieces of eight floats as A/B fragment registers
__device__ __forceinline__ void dg_split8(const float4& v0, const float4& v1, dg_bf16x8& hi, dg_bf16x8& lo) {
    const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    dg_u32x4 h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned hp = __builtin_bit_cast(unsigned, __builtin_convertvector(dg_f32x2{x[2 * q], x[2 * q + 1]}, dg_bf16x2));
        const float r0 = x[2 * q] - __builtin_bit_cast(float, hp << 16);
        const float r1 = x[2 * q + 1] - __builtin_bit_cast(float, hp & 0xffff0000u);
        h[q] = hp;
        l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(dg_f32x2{r0, r1}, dg_bf16x2));
    }
    hi = __builtin_bit_cast(dg_bf16x8, h);
    lo = __builtin_bit_cast(dg_bf16x8, l);
}
}  // namespace

// SPLIT: the product dy2 W2 as three products of bfloat16 pieces (dg_split8; a gradient term, as in dense_dgrad_x3_kernel
// below) instead of v_mfma_f32_32x32x2_f32.  A lane then owns, per 16-term step ks of a phase, the 8 consecutive floats
// [64p + 16ks + 8h, + 8) of its row; without SPLIT the 32 consecutive floats [64p + 32h, + 32).
template <bool SPLIT>
__global__ __launch_bounds__(256, 2) void conv2_bwd_fused_kernel(
    float* __restrict__ dh2, const float* __restrict__ y2, const float* __restrict__ mean2,
    const float* __restrict__ rstd2, const float* __restrict__ gamma2, const float* __restrict__ beta2,
    const float* __restrict__ m1, const float* __restrict__ m2, const float* __restrict__ W2, float* __restrict__ dh1,
    int M, const float* __restrict__ X3, const float* __restrict__ W1, const float* __restrict__ b1,
    const float* __restrict__ mean1, const float* __restrict__ rstd1, const float* __restrict__ gamma1,
    const float* __restrict__ beta1, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float f_smem[];
    float* sB = f_smem;
    float* sC = f_smem + F_W2_FLOATS;      // [7][128]
    float* sC1 = sC + F_CONST * FK;             // [8][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // float4 q of a lane's phase sits at this column of the row
    auto col_of = [&](int p, int q) { return SPLIT ? 64 * p + 16 * (q >> 1) + 8 * h + 4 * (q & 1) : 64 * p + 32 * h + 4 * q; };
    char* sBh = (char*)f_smem;
    char* sBl = sBh + F_X3_PLANE_BYTES;
    if (SPLIT) {
        // a thread makes whole fragments: rows k = 8 kg .. 8 kg + 7 of column n, both pieces
        for (int item = tid; item < FN * (FK / 8); item += 256) {
            const int n = item & (FN - 1), kg = item / FN;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = W2[(size_t)(8 * kg + j) * FN + n];
            dg_bf16x8 hi, lo;
            dg_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
            const int off = (((n >> 5) * 8 + (kg >> 1)) * 64 + (n & 31) + 32 * (kg & 1)) * 16;
            *(dg_bf16x8*)(sBh + off) = hi;
            *(dg_bf16x8*)(sBl + off) = lo;
        }
    } else {
        for (int i = tid * 4; i < FK * FN; i += 1024)
            *(float4*)(sB + w2_row_offset(i / FN) + (i % FN)) = *(const float4*)(W2 + i);
    }
    if (tid < FK) {
        const float mu = mean2[tid], rs = rstd2[tid], ga = gamma2[tid];
        sC[0 * FK + tid] = mu;
        sC[1 * FK + tid] = rs;
        sC[2 * FK + tid] = ga;
        sC[3 * FK + tid] = beta2[tid];
        sC[4 * FK + tid] = m1[tid];
        sC[5 * FK + tid] = m2[tid];
        sC[6 * FK + tid] = ga * rs;
    } else if (tid < FK + FN) {
        const int c = tid - FK;
        sC1[0 * FN + c] = W1[c * 3];
        sC1[1 * FN + c] = W1[c * 3 + 1];
        sC1[2 * FN + c] = W1[c * 3 + 2];
        sC1[3 * FN + c] = b1[c];
        sC1[4 * FN + c] = mean1[c];
        sC1[5 * FN + c] = rstd1[c];
        sC1[6 * FN + c] = gamma1[c];
        sC1[7 * FN + c] = beta1[c];
    }
    __syncthreads();

    double s1[FNT], s2[FNT];
#pragma unroll
    for (int t = 0; t < FNT; ++t) s1[t] = s2[t] = 0.0;

    const int tiles = (M + 127) / 128;
    // A tile of 128 rows is taken in two phases of 64 columns of dh2 (= 64 of the product's 128 summed terms): lane
    // (r, h) owns the 32 consecutive floats [64p + 32h, 64p + 32h + 32) of its row in phase p, so MFMA step s of a
    // phase pairs term 64p + s (lanes h = 0) with term 64p + 32 + s (lanes h = 1).  The next phase's floats of dh2
    // and y2 are in flight while this phase's MFMAs run.
    constexpr int PQ = 8;                          // float4 per lane and phase
    float4 a[PQ], yv[PQ];
    float xr[3];                                   // coordinates of this lane's row (layer-1 pre-activation of the epilogue)
    auto load_phase = [&](int tile, int p) {
        int row = tile * 128 + wave * 32 + r;
        row = row < M ? row : M - 1;
        const float* sd = dh2 + (size_t)row * FK;
        const float* sy = y2 + (size_t)row * FK;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            a[q] = *(const float4*)(sd + col_of(p, q));
            yv[q] = *(const float4*)(sy + col_of(p, q));
        }
        if (p == 0) {
            xr[0] = X3[(size_t)row * 3];
            xr[1] = X3[(size_t)row * 3 + 1];
            xr[2] = X3[(size_t)row * 3 + 2];
        }
    };
    int tile = blockIdx.x;
    if (tile < tiles) load_phase(tile, 0);
    for (; tile < tiles; tile += gridDim.x) {
        const int rowA = tile * 128 + wave * 32 + r;
        const int next = tile + gridDim.x;
        const float x0 = xr[0], x1 = xr[1], x2 = xr[2];
        f32x16 acc[FNT];
#pragma unroll
        for (int t = 0; t < FNT; ++t) acc[t] = f32x16{0};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            // dy2 of this lane's 32 floats, in registers and (in place) in memory
            float4 d[PQ];
#pragma unroll
            for (int q = 0; q < PQ; ++q) {
                const float* cst = sC + col_of(p, q);
                const float4 mu = *(const float4*)(cst), rs = *(const float4*)(cst + FK), ga = *(const float4*)(cst + 2 * FK),
                             be = *(const float4*)(cst + 3 * FK), a1 = *(const float4*)(cst + 4 * FK),
                             a2 = *(const float4*)(cst + 5 * FK), gr = *(const float4*)(cst + 6 * FK);
                float4 g = a[q];
                const float4 y = yv[q];
                if (!(bn_pre(y.x, mu.x, rs.x, ga.x, be.x) > 0.f)) g.x = 0.f;
                if (!(bn_pre(y.y, mu.y, rs.y, ga.y, be.y) > 0.f)) g.y = 0.f;
                if (!(bn_pre(y.z, mu.z, rs.z, ga.z, be.z) > 0.f)) g.z = 0.f;
                if (!(bn_pre(y.w, mu.w, rs.w, ga.w, be.w) > 0.f)) g.w = 0.f;
                d[q].x = gr.x * (g.x - a1.x - (y.x - mu.x) * rs.x * a2.x);
                d[q].y = gr.y * (g.y - a1.y - (y.y - mu.y) * rs.y * a2.y);
                d[q].z = gr.z * (g.z - a1.z - (y.z - mu.z) * rs.z * a2.z);
                d[q].w = gr.w * (g.w - a1.w - (y.w - mu.w) * rs.w * a2.w);
            }
            if (rowA < M) {
                float* dst = dh2 + (size_t)rowA * FK;
#pragma unroll
                for (int q = 0; q < PQ; ++q) *(float4*)(dst + col_of(p, q)) = d[q];
            }
            if (p == 0)
                load_phase(tile, 1);
            else if (next < tiles)
                load_phase(next, 0);
            if (SPLIT) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    dg_bf16x8 ah, al;
                    dg_split8(d[2 * ks], d[2 * ks + 1], ah, al);
#pragma unroll
                    for (int t = 0; t < FNT; ++t) {
                        const int off = ((t * 8 + 4 * p + ks) * 64 + lane) * 16;
                        const dg_bf16x8 bh = *(const dg_bf16x8*)(sBh + off);
                        const dg_bf16x8 bl = *(const dg_bf16x8*)(sBl + off);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < PQ; ++q) {
                    const float av[4] = {d[q].x, d[q].y, d[q].z, d[q].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* brow = sB + w2_row_offset(64 * p + 32 * h + 4 * q + e) + r;
#pragma unroll
                        for (int t = 0; t < FNT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], brow[32 * t], acc[t], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // accumulator i of a lane: row (i&3) + 8(i>>2) + 4h of the wave's 32 rows, column 32t + r
        const int row0 = tile * 128 + wave * 32;
        float f1[FNT], f2[FNT], w1[FNT][3], bb1[FNT], mu1[FNT], rs1[FNT], ga1[FNT], be1[FNT];
        int c1off = r;
        asm volatile("" : "+v"(c1off));            // read per tile: these sixteen values are not kept across the MFMAs
#pragma unroll
        for (int t = 0; t < FNT; ++t) {
            const float* c1 = sC1 + c1off + 32 * t;
            f1[t] = f2[t] = 0.f;
            w1[t][0] = c1[0];
            w1[t][1] = c1[FN];
            w1[t][2] = c1[2 * FN];
            bb1[t] = c1[3 * FN];
            mu1[t] = c1[4 * FN];
            rs1[t] = c1[5 * FN];
            ga1[t] = c1[6 * FN];
            be1[t] = c1[7 * FN];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int lr = (i & 3) + 8 * (i >> 2) + 4 * h;          // the lane that holds this row's coordinates
            const float c0 = __shfl(x0, lr), c1 = __shfl(x1, lr), c2 = __shfl(x2, lr);
            const int row = row0 + lr;
            if (row < M) {
#pragma unroll
                for (int t = 0; t < FNT; ++t) {
                    const float v = acc[t][i];
                    const float y = conv1_pre(w1[t][0], w1[t][1], w1[t][2], bb1[t], c0, c1, c2);
                    const float g = bn_pre(y, mu1[t], rs1[t], ga1[t], be1[t]) > 0.f ? v : 0.f;
                    f1[t] += g;
                    f2[t] += g * ((y - mu1[t]) * rs1[t]);
                    dh1[(size_t)row * FN + 32 * t + r] = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < FNT; ++t) {
            s1[t] += (double)f1[t];
            s2[t] += (double)f2[t];
        }
    }

    // per lane -> per workgroup: 8 lane sums per column (4 waves x 2 halves), added in a fixed order
    __syncthreads();
    double* red = (double*)f_smem;                  // [8][64][2]
#pragma unroll
    for (int t = 0; t < FNT; ++t) {
        double* p = red + ((size_t)(wave * 2 + h) * FN + 32 * t + r) * 2;
        p[0] = s1[t];
        p[1] = s2[t];
    }
    __syncthreads();
    if (tid < FN) {
        double u = 0.0, v = 0.0;
        for (int k = 0; k < 8; ++k) {
            u += red[((size_t)k * FN + tid) * 2];
            v += red[((size_t)k * FN + tid) * 2 + 1];
        }
        part[((size_t)blockIdx.x * 2 + 0) * FN + tid] = u;
        part[((size_t)blockIdx.x * 2 + 1) * FN + tid] = v;
    }
}

// Forward of the first two per-point layers in one pass: h1 = relu(bn1(W1 x + b1)) is computed from the three
// coordinates of the row in the registers that feed the matrix pipe (and stored once, for the backward pass),
// y2 = h1 W2^T + b2 comes out of the MFMAs together with the column statistics BatchNorm 2 needs (records of
// count, mean, sum of squared deviations per workgroup, as tall_gemm_kernel writes them).
// (model/pointnet.py:30-31 / :143-144: relu(bn1(conv1(x))), conv2.)
// Lane (r, h) owns row r of the wave's 32 rows and the 32 channels [32h, 32h + 32) of h1: MFMA step s pairs term s
// (lanes h = 0) with term 32 + s (lanes h = 1).
namespace {
constexpr int GK = 64, GN = 128, GNT = GN / 32;
constexpr int G_W2_FLOATS = GK * GN + 32;          // rows k >= 32 start 32 floats later (other half of the banks)
constexpr size_t G_LDS_BYTES = ((size_t)G_W2_FLOATS + F_CONST1 * GK) * sizeof(float);
constexpr size_t G_RED_BYTES = (size_t)8 * GN * 3 * sizeof(double);
}  // namespace

__global__ __launch_bounds__(256, 2) void conv12_fwd_kernel(const float* __restrict__ X3, const float* __restrict__ W1,
                                                            const float* __restrict__ b1, const float* __restrict__ mean1,
                                                            const float* __restrict__ rstd1,
                                                            const float* __restrict__ gamma1,
                                                            const float* __restrict__ beta1, const float* __restrict__ W2,
                                                            const float* __restrict__ b2, float* __restrict__ h1,
                                                            float* __restrict__ y2, int M, double* __restrict__ stat_rec) {
    extern __shared__ __attribute__((aligned(16))) float g_smem[];
    float* sB = g_smem;                          // B[k][n] = W2[n][k]
    float* sC1 = g_smem + G_W2_FLOATS;           // [8][64]: three weights, bias, mean, rstd, gamma, beta of layer 1
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // W2 is [n][k]: a thread reads four consecutive k of one n (16 bytes) and stores them to four rows of B; lanes run
    // over n, so the stores of a wave fall on consecutive banks (a scalar transposing copy put all 64 lanes on one bank)
    for (int i = tid; i < GK * GN / 4; i += 256) {
        const int n = i & (GN - 1), k4 = (i >> 7) * 4;
        const float4 w = *(const float4*)(W2 + (size_t)n * GK + k4);
        float* dst = sB + k4 * GN + n + (k4 >= GK / 2 ? 32 : 0);
        dst[0] = w.x;
        dst[GN] = w.y;
        dst[2 * GN] = w.z;
        dst[3 * GN] = w.w;
    }
    if (tid < GK) {
        sC1[0 * GK + tid] = W1[tid * 3];
        sC1[1 * GK + tid] = W1[tid * 3 + 1];
        sC1[2 * GK + tid] = W1[tid * 3 + 2];
        sC1[3 * GK + tid] = b1[tid];
        sC1[4 * GK + tid] = mean1[tid];
        sC1[5 * GK + tid] = rstd1[tid];
        sC1[6 * GK + tid] = gamma1[tid];
        sC1[7 * GK + tid] = beta1[tid];
    }
    __syncthreads();

    float bv[GNT];
#pragma unroll
    for (int t = 0; t < GNT; ++t) bv[t] = b2[32 * t + r];
    double cnt = 0.0, s1[GNT], s2[GNT];
    float a0[GNT];
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
        s1[t] = 0.0;
        s2[t] = 0.0;
        a0[t] = 0.f;
    }

    const int tiles = (M + 127) / 128;
    float xn[3];
    auto load_point = [&](int tile) {
        int row = tile * 128 + wave * 32 + r;
        row = row < M ? row : M - 1;
        xn[0] = X3[(size_t)row * 3];
        xn[1] = X3[(size_t)row * 3 + 1];
        xn[2] = X3[(size_t)row * 3 + 2];
    };
    int tile = blockIdx.x;
    if (tile < tiles) load_point(tile);
    for (; tile < tiles; tile += gridDim.x) {
        const float x0 = xn[0], x1 = xn[1], x2 = xn[2];
        const int next = tile + gridDim.x;
        if (next < tiles) load_point(next);
        const int rowA = tile * 128 + wave * 32 + r;
        float4 a[GK / 8];
#pragma unroll
        for (int q = 0; q < GK / 8; ++q) {
            const float* cst = sC1 + 32 * h + 4 * q;
            const float4 w0 = *(const float4*)(cst), w1 = *(const float4*)(cst + GK), w2 = *(const float4*)(cst + 2 * GK),
                         bb = *(const float4*)(cst + 3 * GK), mu = *(const float4*)(cst + 4 * GK),
                         rs = *(const float4*)(cst + 5 * GK), ga = *(const float4*)(cst + 6 * GK),
                         be = *(const float4*)(cst + 7 * GK);
            a[q].x = __builtin_fmaxf(bn_pre(conv1_pre(w0.x, w1.x, w2.x, bb.x, x0, x1, x2), mu.x, rs.x, ga.x, be.x), 0.f);
            a[q].y = __builtin_fmaxf(bn_pre(conv1_pre(w0.y, w1.y, w2.y, bb.y, x0, x1, x2), mu.y, rs.y, ga.y, be.y), 0.f);
            a[q].z = __builtin_fmaxf(bn_pre(conv1_pre(w0.z, w1.z, w2.z, bb.z, x0, x1, x2), mu.z, rs.z, ga.z, be.z), 0.f);
            a[q].w = __builtin_fmaxf(bn_pre(conv1_pre(w0.w, w1.w, w2.w, bb.w, x0, x1, x2), mu.w, rs.w, ga.w, be.w), 0.f);
        }
        if (rowA < M) {
            float4* dst = (float4*)(h1 + (size_t)rowA * GK + 32 * h);
#pragma unroll
            for (int q = 0; q < GK / 8; ++q) dst[q] = a[q];
        }
        f32x16 acc[GNT];
#pragma unroll
        for (int t = 0; t < GNT; ++t) acc[t] = f32x16{0};
#pragma unroll
        for (int q = 0; q < GK / 8; ++q) {
            const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* brow = sB + (32 * h + 4 * q + e) * GN + 32 * h + r;
#pragma unroll
                for (int t = 0; t < GNT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], brow[32 * t], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const int row0 = tile * 128 + wave * 32 + 4 * h;
        int nvalid = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) nvalid += (row0 + (i & 3) + 8 * (i >> 2)) < M ? 1 : 0;
#pragma unroll
        for (int t = 0; t < GNT; ++t) {
            float f1 = 0.f, f2 = 0.f;
            if (cnt == 0.0 && nvalid > 0) a0[t] = acc[t][0] + bv[t];        // this lane's shift against cancellation
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + (i & 3) + 8 * (i >> 2);
                if (row < M) {
                    const float v = acc[t][i] + bv[t];
                    const float d = v - a0[t];
                    f1 += d;
                    f2 += d * d;
                    y2[(size_t)row * GN + 32 * t + r] = v;
                }
            }
            s1[t] += (double)f1;
            s2[t] += (double)f2;
        }
        cnt += (double)nvalid;
    }

    // per lane -> per workgroup: 8 lane records per column (4 waves x 2 halves), merged in a fixed order
    __syncthreads();
    double* red = (double*)g_smem;                     // [8][128][3]
#pragma unroll
    for (int t = 0; t < GNT; ++t) {
        double* p = red + ((size_t)(wave * 2 + h) * GN + 32 * t + r) * 3;
        const double mean = cnt > 0.0 ? (double)a0[t] + s1[t] / cnt : 0.0;
        const double m2 = cnt > 0.0 ? s2[t] - s1[t] * s1[t] / cnt : 0.0;
        p[0] = cnt;
        p[1] = mean;
        p[2] = m2 > 0.0 ? m2 : 0.0;
    }
    __syncthreads();
    if (tid < GN) {
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int k = 0; k < 8; ++k) {
            const double* p = red + ((size_t)k * GN + tid) * 3;
            const double nb = p[0], meanb = p[1], m2b = p[2];     // Chan et al.
            if (nb == 0.0) continue;
            const double nt = n + nb, d = meanb - mean;
            mean += d * (nb / nt);
            m2 += m2b + d * d * (n * nb / nt);
            n = nt;
        }
        double* o = stat_rec + ((size_t)blockIdx.x * GN + tid) * 3;
        o[0] = n;
        o[1] = mean;
        o[2] = m2;
    }
}

// h1 [M,64] and y2 [M,128] out; stat_rec [tall_gemm_blocks(M)][128][3] doubles for t_stats_from_records.
hipError_t launch_conv12_fwd(const float* X3, const float* W1, const float* b1, const float* mean1, const float* rstd1,
                             const float* gamma1, const float* beta1, const float* W2, const float* b2, float* h1,
                             float* y2, int M, double* stat_rec, hipStream_t s) {
    if (!X3 || !W1 || !b1 || !W2 || !b2 || !h1 || !y2 || !stat_rec || M < 1) return hipErrorInvalidValue;
    const size_t lds = G_LDS_BYTES > G_RED_BYTES ? G_LDS_BYTES : G_RED_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)conv12_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv12_fwd_kernel, dim3(tall_gemm_blocks(M)), dim3(256), lds, s, X3, W1, b1, mean1, rstd1, gamma1,
                       beta1, W2, b2, h1, y2, M, stat_rec);
    return hipGetLastError();
}

// Y[M,128] = X[M,128] G + ub (G [128][128], row k = input channel) with float32-grade products on the bfloat16 matrix
// pipe: every operand is two bfloat16 pieces (hi = bf16(v), lo = bf16(v - hi): 16 significand bits), a product three
// MFMAs accumulating in float32 (lo.hi, hi.lo, hi.hi - the dropped lo.lo and the representation remainders are below
// 2^-16 of the product).  This is the dense part of conv3's input gradient (h2 times the 128x128 matrix that the
// BatchNorm statistics contribute), 6.3 GFLOP per call that take 40 us on the float32 MFMA and 7.5 us this way, so
// that the pass is bound by its rows (98 MB in, 98 MB out) instead.  Structure as conv2_bwd_fused_kernel: persistent
// workgroups, G resident in LDS as B fragments of v_mfma_f32_32x32x16_bf16 (two planes), a tile of 128 rows in two
// phases of 64 summed terms, the next phase's floats in flight during the MFMAs.  Lane (r, h) owns row r of its wave's
// 32 and, per 16-term step, the 8 consecutive floats [16 ks + 8 h, + 8).
__global__ __launch_bounds__(256, 2) void dense_dgrad_x3_kernel(const float* __restrict__ X, const float* __restrict__ G,
                                                                const float* __restrict__ ub, float* __restrict__ Y,
                                                                int M) {
    extern __shared__ __attribute__((aligned(16))) char dg_smem[];
    char* sBh = dg_smem;
    char* sBl = dg_smem + DG_PLANE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // a thread makes whole fragments: the 8 consecutive rows k = 8 kg .. 8 kg + 7 of column n (reads coalesced over n)
    for (int item = tid; item < 128 * 16; item += 256) {
        const int n = item & 127, kg = item >> 7;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = G[(size_t)(8 * kg + j) * 128 + n];
        dg_bf16x8 hi, lo;
        dg_split8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]), hi, lo);
        const int off = (((n >> 5) * 8 + (kg >> 1)) * 64 + (n & 31) + 32 * (kg & 1)) * 16;
        *(dg_bf16x8*)(sBh + off) = hi;
        *(dg_bf16x8*)(sBl + off) = lo;
    }
    __syncthreads();
    float bv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) bv[t] = ub[32 * t + r];

    const int tiles = (M + 127) / 128;
    float4 a[8];
    auto load_phase = [&](int tile, int p) {
        int row = tile * 128 + wave * 32 + r;
        row = row < M ? row : M - 1;
        const float* src = X + (size_t)row * 128 + 64 * p + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a[2 * ks] = *(const float4*)(src + 16 * ks);
            a[2 * ks + 1] = *(const float4*)(src + 16 * ks + 4);
        }
    };
    int tile = blockIdx.x;
    if (tile < tiles) load_phase(tile, 0);
    for (; tile < tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x16{0};
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            dg_bf16x8 ah[4], al[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dg_split8(a[2 * ks], a[2 * ks + 1], ah[ks], al[ks]);
            if (p == 0)
                load_phase(tile, 1);
            else if (next < tiles)
                load_phase(next, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int off = ((t * 8 + 4 * p + ks) * 64 + lane) * 16;
                    const dg_bf16x8 bh = *(const dg_bf16x8*)(sBh + off);
                    const dg_bf16x8 bl = *(const dg_bf16x8*)(sBl + off);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ks], bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ks], bh, acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int row0 = tile * 128 + wave * 32 + 4 * h;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + (i & 3) + 8 * (i >> 2);
                if (row < M) Y[(size_t)row * 128 + 32 * t + r] = acc[t][i] + bv[t];
            }
    }
}

hipError_t launch_dense_dgrad_x3(const float* X, const float* G, const float* ub, float* Y, int M, hipStream_t s) {
    if (!X || !G || !ub || !Y || M < 1) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute((const void*)dense_dgrad_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)DG_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(dense_dgrad_x3_kernel, dim3(tall_gemm_blocks(M)), dim3(256), DG_LDS_BYTES, s, X, G, ub, Y, M);
    return hipGetLastError();
}

// dh2 [M,128]: in dL/dh2 (before the ReLU mask), out dL/dy2.  dh1 [M,64]: dL/dh1 (before conv1's ReLU mask).
// part: [tall_gemm_blocks(M)][2][64] doubles, the partial column sums for BatchNorm 1's backward.
hipError_t launch_conv2_bwd_fused(float* dh2, const float* y2, const float* mean2, const float* rstd2,
                                  const float* gamma2, const float* beta2, const float* m1, const float* m2,
                                  const float* W2, float* dh1, int M, const float* X3, const float* W1, const float* b1,
                                  const float* mean1, const float* rstd1, const float* gamma1, const float* beta1,
                                  double* part, hipStream_t s) {
    if (!dh2 || !y2 || !W2 || !dh1 || !X3 || !part || M < 1) return hipErrorInvalidValue;
    // PNGPD_TRAIN_DGRAD=f32 keeps the product on the float32 matrix instruction (as for dense_dgrad_x3_kernel)
    const char* env = getenv("PNGPD_TRAIN_DGRAD");
    auto kern = env && env[0] == 'f' ? conv2_bwd_fused_kernel<false> : conv2_bwd_fused_kernel<true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)F_LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(tall_gemm_blocks(M)), dim3(256), F_LDS_BYTES, s, dh2, y2, mean2, rstd2, gamma2, beta2, m1,
                       m2, W2, dh1, M, X3, W1, b1, mean1, rstd1, gamma1, beta1, part);
    return hipGetLastError();
}

}  // namespace pngpd

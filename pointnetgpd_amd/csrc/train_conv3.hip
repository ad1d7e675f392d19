// Training step of the 128 -> 1024 layer with its BatchNorm (batch statistics) and the max over
// points (conv3 -> bn3 (-> relu) -> MaxPool1d(num_points), model/pointnet.py:32-34 and :146-147,
// in train() mode, main_1v.py:266-272), without ever holding the [M,1024] activation.
//
// With h[M,128] the input rows, w_c the weight row of channel c, abar the column mean of h and
// Cov = h^T h / M - abar abar^T the 128x128 covariance (one Gram product on the f64 MFMA):
//   forward   mean_c = w_c.abar + b_c,   var_c = w_c^T Cov w_c   (biased, as BatchNorm normalises)
//             pooled[b,c] = act(max_p h[b,p,:].(sc_c w_c) + sh_c),  sc = gamma*rstd,
//             sh = beta - sc*(w_c.abar): the fused product/max kernel of the inference path, on the
//             exact-f32 MFMA, extended by the arg-max row.
//   backward  dy[m,c] = A_c g[b,c] [m == arg(b,c)] + p_c + q_c w_c.(h[m,:] - abar)
//             with g the pooled gradient (masked by the ReLU), m1 = sum g / M, m2 = sum g*yhat_arg / M,
//             A = gamma*rstd, p = -A m1, q = -A rstd m2.  The dense part is linear in h, so
//             dh     = h G + ub  (G = W^T diag(q) W, 128x128)  +  rows arg(b,c) += A_c g[b,c] w_c
//             dW[c,:] = A_c sum_b g[b,c] h[arg(b,c),:] + M (p_c abar + q_c Cov w_c)
//             dgamma = sum g*yhat_arg, dbeta = sum g.
// Small reductions run in double, in a fixed order (reproducible).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "kernels.h"
#include "train_act.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr float BN_EPS3 = 1e-5f;

// ---------------------------------------------------------------------------------------------
// forward: statistics from the Gram matrix, scaled + packed weights
// ---------------------------------------------------------------------------------------------
// Gram matrix S = h^T h of the [M,128] input rows on the f64 MFMA (products of floats are exact in double,
// the sums run in double): partial sums over chunks of t_gram_rows_per_chunk(M) rows.
// grid (3, chunks): 64x64 output blocks (0,0), (0,1), (1,1) - block (1,0) is the transpose of (0,1).
// v_mfma_f64_16x16x4_f64 lane l: A[i = l&15][k = l>>4], B[k = l>>4][j = l&15], D reg r = [row (l>>4) + 4r][col l&15].
typedef __attribute__((ext_vector_type(4))) double f64x4;
constexpr int GR_LD = 80;       // LDS row stride in floats: the 4 k rows one instruction reads fall on distinct banks

// The diagonal blocks also add up the columns they load: colsum[chunk][128].
// With bn_mean != nullptr the input is the PRE-activation y of the previous layer: the rows become
// h = relu((y - mean)*rstd*gamma + beta) on their way into LDS (train-mode BatchNorm + ReLU of conv2,
// model/pointnet.py:31, :145) and the diagonal blocks write them to h_out - the activation is produced by the pass
// that first needs it instead of by a pass of its own.
__device__ __forceinline__ float4 bn_relu4(const float4& y, const float4& mu, const float4& rs, const float4& ga,
                                           const float4& be) {
    float4 v;
    v.x = fmaxf(bn_pre(y.x, mu.x, rs.x, ga.x, be.x), 0.f);
    v.y = fmaxf(bn_pre(y.y, mu.y, rs.y, ga.y, be.y), 0.f);
    v.z = fmaxf(bn_pre(y.z, mu.z, rs.z, ga.z, be.z), 0.f);
    v.w = fmaxf(bn_pre(y.w, mu.w, rs.w, ga.w, be.w), 0.f);
    return v;
}

__global__ __launch_bounds__(256) void gram_f64_kernel(const float* __restrict__ A, int M, int rows_per_chunk,
                                                       double* __restrict__ part, double* __restrict__ colsum,
                                                       const float* __restrict__ bn_mean,
                                                       const float* __restrict__ bn_rstd,
                                                       const float* __restrict__ bn_gamma,
                                                       const float* __restrict__ bn_beta, float* __restrict__ h_out) {
    __shared__ __attribute__((aligned(16))) float sI[32 * GR_LD];
    __shared__ __attribute__((aligned(16))) float sJ[32 * GR_LD];
    __shared__ double s_cs[16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kq = lane >> 4, wi = wave >> 1, wj = wave & 1;
    const int i0 = blockIdx.x == 2 ? 64 : 0, j0 = blockIdx.x == 0 ? 0 : 64;
    const int row_begin = blockIdx.y * rows_per_chunk;
    const int row_end = min(M, row_begin + rows_per_chunk);
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;
    f64x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f64x4{0.0, 0.0, 0.0, 0.0};
    float4 va[2], vb[2];
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    const bool fused = bn_mean != nullptr;
    float4 mu[2], rs[2], ga[2], be[2];          // [0]: columns i0 + lcol .., [1]: columns j0 + lcol ..
    if (fused) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c0 = (t == 0 ? i0 : j0) + lcol;
            mu[t] = *(const float4*)(bn_mean + c0);
            rs[t] = *(const float4*)(bn_rstd + c0);
            // gamma / beta are caller tensors (a view of a flat parameter bucket need not be 16-byte aligned)
            ga[t] = make_float4(bn_gamma[c0], bn_gamma[c0 + 1], bn_gamma[c0 + 2], bn_gamma[c0 + 3]);
            be[t] = make_float4(bn_beta[c0], bn_beta[c0 + 1], bn_beta[c0 + 2], bn_beta[c0 + 3]);
        }
    }
    auto load_tile = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gm = mb + lrow + 16 * i;
            const bool ok = gm < row_end;
            va[i] = ok ? *(const float4*)(A + (size_t)gm * C2 + i0 + lcol) : make_float4(0, 0, 0, 0);
            vb[i] = ok ? *(const float4*)(A + (size_t)gm * C2 + j0 + lcol) : make_float4(0, 0, 0, 0);
        }
    };
    // BatchNorm + ReLU of the rows just loaded (rows past the end stay zero); diagonal blocks store their columns
    auto activate = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gm = mb + lrow + 16 * i;
            if (gm < row_end) {
                va[i] = bn_relu4(va[i], mu[0], rs[0], ga[0], be[0]);
                vb[i] = bn_relu4(vb[i], mu[1], rs[1], ga[1], be[1]);
                if (i0 == j0) *(float4*)(h_out + (size_t)gm * C2 + i0 + lcol) = va[i];
            }
        }
    };
    if (row_begin < row_end) load_tile(row_begin);
    for (int mb = row_begin; mb < row_end; mb += 32) {
        if (fused) activate(mb);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(float4*)(sI + (lrow + 16 * i) * GR_LD + lcol) = va[i];
            *(float4*)(sJ + (lrow + 16 * i) * GR_LD + lcol) = vb[i];
            cs[0] += (double)va[i].x;
            cs[1] += (double)va[i].y;
            cs[2] += (double)va[i].z;
            cs[3] += (double)va[i].w;
        }
        __syncthreads();
        if (mb + 32 < row_end) load_tile(mb + 32);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float* ri = sI + (4 * s + kq) * GR_LD + wi * 32 + c;
            const float* rj = sJ + (4 * s + kq) * GR_LD + wj * 32 + c;
            const double a0 = (double)ri[0], a1 = (double)ri[16];
            const double b0 = (double)rj[0], b1 = (double)rj[16];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    double* out = part + (size_t)blockIdx.y * C2 * C2;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + wi * 32 + 16 * a + kq + 4 * r, j = j0 + wj * 32 + 16 * b + c;
                out[(size_t)i * C2 + j] = acc[a][b][r];
            }
    if (i0 != j0) return;                      // column sums: once per column, from the diagonal blocks
#pragma unroll
    for (int e = 0; e < 4; ++e) s_cs[lrow][lcol + e] = cs[e];
    __syncthreads();
    if (tid < 64) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += s_cs[k][tid];
        colsum[(size_t)blockIdx.y * C2 + i0 + tid] = t;
    }
}

// The same Gram partials on the bf16 matrix pipe.  Every float is three bfloat16 pieces, h = hh + hm + hl to 2^-27
// (each piece the rounded remainder of the one before), a product is the six largest of the nine piece products
// (exact in the float32 accumulator; what is left out is below 2^-26 of the product and of either sign), and the
// float32 accumulators, which only ever hold the sum over 64 rows, are added into double accumulators after every
// second tile - so the sums over the rows run in double as in the kernel above, and a partial is within about 1e-7 of the
// exact one, unbiased.  One workgroup of five waves per chunk of rows computes the ten 32x32 blocks of the 128x128
// matrix that are not a transpose of another (wave w: row block GX_BI[w], column blocks GX_BJ[w][0..1]); waves 0-3
// bring a tile of 32 rows in (BatchNorm + ReLU of the previous layer as above, h_out written for all columns),
// split it and store the pieces TRANSPOSED - [piece][k-step][k half][column] x 8 consecutive rows - so that both
// operands of v_mfma_f32_32x32x16_bf16 (each wants, per lane, 8 consecutive rows of one column) are one 16-byte LDS
// read.  Two tile buffers, one barrier per tile.
typedef __attribute__((ext_vector_type(8))) __bf16 gx_bf16x8;
constexpr int GX_THREADS = 320;
constexpr int GX_PLANE = 2 * 2 * C2 * 16;        // bytes of one piece of a 32-row tile

typedef __attribute__((ext_vector_type(2))) float gx_f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gx_bf16x2;

// the three pieces of two floats, each pair packed as the two halves of a word (x0 in the lower half)
__device__ __forceinline__ void gx_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f32x2{x0, x1}, gx_bf16x2));
    const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f32x2{r0, r1}, gx_bf16x2));
    const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(gx_f32x2{q0, q1}, gx_bf16x2));
}

__global__ __launch_bounds__(GX_THREADS) void gram_x6_kernel(const float* __restrict__ A, int M, int rows_per_chunk,
                                                             double* __restrict__ part, double* __restrict__ colsum,
                                                             const float* __restrict__ bn_mean,
                                                             const float* __restrict__ bn_rstd,
                                                             const float* __restrict__ bn_gamma,
                                                             const float* __restrict__ bn_beta, float* __restrict__ h_out) {
    __shared__ __attribute__((aligned(16))) char s_p[2][3 * GX_PLANE];
    __shared__ double s_cs[8][C2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int row_begin = blockIdx.x * rows_per_chunk;
    const int row_end = min(M, row_begin + rows_per_chunk);
    // staging role (waves 0-3): rows 4rg .. 4rg+3 of the tile, columns cg + 32e
    const bool stager = tid < 256;
    const int cg = tid & 31, rg = (tid >> 5) & 7;
    const int st_off = (((rg >> 2) * 2 + ((rg >> 1) & 1)) * C2) * 16 + (rg & 1) * 8;      // + column * 16
    const bool fused = bn_mean != nullptr;
    float mu[4], rs[4], ga[4], be[4];
    if (fused && stager) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c0 = cg + 32 * e;
            mu[e] = bn_mean[c0];
            rs[e] = bn_rstd[c0];
            ga[e] = bn_gamma[c0];
            be[e] = bn_beta[c0];
        }
    }
    // compute role: row block bi, column blocks bj0 and bj1
    const int bi = wave == 0 || wave == 1 ? 0 : wave == 2 ? 1 : wave == 3 ? 3 : 2;
    const int bj0 = wave == 0 ? 0 : wave == 1 ? 2 : wave == 2 ? 1 : wave == 3 ? 3 : 2;
    const int bj1 = wave == 0 ? 1 : wave == 1 ? 3 : wave == 2 ? 2 : wave == 3 ? 1 : 3;
    f64x4 acc[2][4];                         // [tile][register quad] as doubles: 2 x 16
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = f64x4{0.0, 0.0, 0.0, 0.0};
    double cs[4] = {0.0, 0.0, 0.0, 0.0};
    float v[4][4], vn[4][4];                 // [row][e]: the tile to be staged next and the one after it
    auto load_tile = [&](int mb, float (&dst)[4][4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = mb + 4 * rg + i;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[i][e] = gm < row_end ? A[(size_t)gm * C2 + cg + 32 * e] : 0.f;
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = vn[i][e];
    };
    auto stage = [&](int mb, char* buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = mb + 4 * rg + i;
            if (gm < row_end) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (fused) {
                        v[i][e] = fmaxf(bn_pre(v[i][e], mu[e], rs[e], ga[e], be[e]), 0.f);
                        h_out[(size_t)gm * C2 + cg + 32 * e] = v[i][e];
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // rows past the end are zero; four floats are added in float before they join the double sum
            cs[e] += (double)((v[0][e] + v[1][e]) + (v[2][e] + v[3][e]));
            unsigned h01, m01, l01, h23, m23, l23;
            gx_split2(v[0][e], v[1][e], h01, m01, l01);
            gx_split2(v[2][e], v[3][e], h23, m23, l23);
            char* dst = buf + st_off + (cg + 32 * e) * 16;
            *(uint2*)dst = make_uint2(h01, h23);
            *(uint2*)(dst + GX_PLANE) = make_uint2(m01, m23);
            *(uint2*)(dst + 2 * GX_PLANE) = make_uint2(l01, l23);
        }
    };
    // The pieces of tile t+1 are made and stored while the matrix pipe works on tile t (same basic block: the
    // scheduler places the staging instructions in the shadow of the MFMAs).
    // Rows are requested two tiles ahead: with one workgroup per CU, one tile in flight (16 KiB) keeps too little of
    // the memory system busy.
    if (stager && row_begin < row_end) {
        load_tile(row_begin, v);
        load_tile(row_begin + 32, vn);
        stage(row_begin, s_p[0]);
        advance();
        load_tile(row_begin + 64, vn);
    }
    __syncthreads();
    int b = 0;
    f32x16 d0 = f32x16{0}, d1 = f32x16{0};
    for (int mb = row_begin; mb < row_end; mb += 32, b ^= 1) {
        const char* buf = s_p[b];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const char* base = buf + ((ks * 2 + hh) * C2 + r) * 16;
            gx_bf16x8 a[3], b0[3], b1[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                a[pc] = *(const gx_bf16x8*)(base + pc * GX_PLANE + bi * 512);
                b0[pc] = *(const gx_bf16x8*)(base + pc * GX_PLANE + bj0 * 512);
                b1[pc] = *(const gx_bf16x8*)(base + pc * GX_PLANE + bj1 * 512);
            }
            // smallest products first
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b0[2], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b1[2], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b0[0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b1[0], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b0[1], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b1[1], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b0[1], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b1[1], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b0[0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b1[0], d1, 0, 0, 0);
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b0[0], d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b1[0], d1, 0, 0, 0);
        }
        if (stager && mb + 32 < row_end) {
            stage(mb + 32, s_p[b ^ 1]);
            advance();
            load_tile(mb + 96, vn);
        }
        if (b == 1 || mb + 32 >= row_end) {       // the float accumulators hold two tiles (64 rows) at most
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0][u][q] += (double)d0[4 * u + q];
                    acc[1][u][q] += (double)d1[4 * u + q];
                }
            d0 = f32x16{0};
            d1 = f32x16{0};
        }
        __syncthreads();
    }
    // D register t of a 32x32 block: row (t & 3) + 8 (t >> 2) + 4 hh, column r
    double* out = part + (size_t)blockIdx.x * C2 * C2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int bj = t == 0 ? bj0 : bj1;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = bi * 32 + q + 8 * u + 4 * hh, j = bj * 32 + r;
                out[(size_t)i * C2 + j] = acc[t][u][q];
            }
    }
    if (stager) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s_cs[rg][cg + 32 * e] = cs[e];
    }
    __syncthreads();
    if (tid < C2) {
        double t = 0.0;
        for (int k = 0; k < 8; ++k) t += s_cs[k][tid];
        colsum[(size_t)blockIdx.x * C2 + tid] = t;
    }
}

// abar[i] = sum_chunks colsum[chunk][i] / M; block = 16 columns x 16 chunk lanes
__global__ __launch_bounds__(256) void k_gram_mean(const double* __restrict__ colsum, int chunks, int M,
                                                   double* __restrict__ abar_d) {
    __shared__ double sh[16][16];
    const int el = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + el;
    double s = 0.0;
    for (int c = rl; c < chunks; c += 16) s += colsum[(size_t)c * C2 + i];
    sh[rl][el] = s;
    __syncthreads();
    if (rl != 0) return;
    s = 0.0;
    for (int k = 0; k < 16; ++k) s += sh[k][el];
    abar_d[i] = s / M;
}

// cov[i][j] = sum_chunks S_chunk[i][j] / M - abar_i abar_j.  `present` bit 4 (i / 32) + j / 32: that 32x32 block of the
// partials was computed (the others are transposes of computed ones).  grid (32 rows, computed blocks): a workgroup
// owns one row of one computed block - 32 consecutive doubles of every partial - as 32 elements x 8 chunk lanes; a
// lane sums every 8th chunk, the lane sums are added in a fixed order, and the result goes to cov[i][j] and, for a
// block off the diagonal whose mirror was not computed, to cov[j][i] as well.
__global__ __launch_bounds__(256) void k_gram_finalize(const double* __restrict__ part, int chunks,
                                                       const double* __restrict__ abar_d, int M,
                                                       double* __restrict__ cov, unsigned present) {
    __shared__ double sh[8][32];
    const int el = threadIdx.x & 31, rl = threadIdx.x >> 5;
    // the blockIdx.y-th set bit of `present`
    int blk = 0;
    for (int seen = -1; blk < 16; ++blk)
        if ((present >> blk) & 1u)
            if (++seen == (int)blockIdx.y) break;
    const int bi = blk >> 2, bj = blk & 3;
    const int i = 32 * bi + (int)blockIdx.x, j = 32 * bj + el;
    const size_t src = (size_t)i * C2 + j;
    double s = 0.0;
#pragma unroll 8
    for (int c = rl; c < chunks; c += 8) s += part[(size_t)c * C2 * C2 + src];
    sh[rl][el] = s;
    __syncthreads();
    if (rl != 0) return;
    s = 0.0;
    for (int k = 0; k < 8; ++k) s += sh[k][el];
    const double v = s / M - abar_d[i] * abar_d[j];
    cov[(size_t)i * C2 + j] = v;
    if (bi != bj && !((present >> (4 * bj + bi)) & 1u)) cov[(size_t)j * C2 + i] = v;
}

// Index of W3[c][k] in the packed image the product/max kernel streams (32x32x2 fragments, same element
// order as the inference image): lane half hh of step (t, g, e) holds k = 32t + 8g + 4hh + e.
__device__ __forceinline__ int w3p_index(int c, int k) {
    const int nt = c >> 5, r = c & 31, t = k >> 5, rem = k & 31;
    const int g = rem >> 3, hh = (rem >> 2) & 1, e = rem & 3;
    return (((nt * 16 + 4 * t + g) * 64) + r + 32 * hh) * 4 + e;
}

// Index of piece 0 (hi) of W3[c][k] in the split-bfloat16 image: [nt 32][ks 8][piece 3][lane 64][8], the B fragments
// of v_mfma_f32_32x32x16_bf16 (lane l holds channel 32nt + l%32, k = 16ks + 8(l/32) + j).  Pieces are 512 values apart.
__device__ __forceinline__ int w3s_index(int c, int k) {
    const int nt = c >> 5, r = c & 31, ks = k >> 4, hh = (k >> 3) & 1, j = k & 7;
    return (((nt * 8 + ks) * 3) * 64 + r + 32 * hh) * 8 + j;
}

// Index of piece 0 (hi) of W3[c][k] in the split-float16 image: [ct 64][ks 4][piece 2][lane 64][8], the B fragments of
// v_mfma_f32_16x16x32_f16; lane l holds channel 16ct + l%16 and k = 32ks + 8(l/16) + j.  The two pieces are 512 values apart.
__device__ __forceinline__ int w3h16_index(int c, int k) {
    const int ct = c >> 4, cl = c & 15, ks = k >> 5, q = (k >> 3) & 3, j = k & 7;
    return (((ct * 4 + ks) * 2) * 64 + cl + 16 * q) * 8 + j;
}

// The power of two that moves m (>= 0) into [2^13, 2^14); 1 for m = 0 or not finite.  Clamped so that the factor and
// its reciprocal stay normal floats.
__device__ __forceinline__ float pow2_scale_to_2p13(float m) {
    const int e = (int)((__builtin_bit_cast(unsigned int, m) >> 23) & 0xffu);      // biased exponent of m
    if (e == 0xff || m == 0.f) return 1.0f;
    int sx = 13 - (e - 127);
    sx = sx > 120 ? 120 : (sx < -120 ? -120 : sx);
    return __builtin_bit_cast(float, (unsigned int)(127 + sx) << 23);
}

// one block per channel: 128 columns x 4 groups of 32 rows of Cov for the product Cov w (one batch of loads per thread),
// the first 128 threads do the rest
__global__ __launch_bounds__(512) void k_conv3_stats_pack(const float* __restrict__ W3, const float* __restrict__ b3,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const double* __restrict__ cov,
                                                          const double* __restrict__ abar_d, int M, float momentum,
                                                          float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ w3p,
                                                          unsigned short* __restrict__ w3s,
                                                          float* __restrict__ sh, float* __restrict__ wsc) {
    __shared__ double s_w[C2], s_q[2], s_m[2], s_t[3][C2];
    __shared__ float s_sc;
    __shared__ float s_max[2];
    const int c = blockIdx.x, i = threadIdx.x & (C2 - 1), jg = threadIdx.x >> 7;
    const float wf = W3[(size_t)c * C2 + i];
    if (jg == 0) s_w[i] = (double)wf;
    __syncthreads();
    double t = 0.0;
#pragma unroll
    for (int j = 32 * jg; j < 32 * jg + 32; ++j) t += cov[(size_t)j * C2 + i] * s_w[j];    // Cov is symmetric: coalesced over i
    if (jg != 0) s_t[jg - 1][i] = t;
    __syncthreads();
    const bool lead = jg == 0;                   // the first 128 threads do the rest; the others only keep the barriers
    // var = sum_i w_i (Cov w)_i and wa = sum_i abar_i w_i: butterfly sums inside each of the two leading waves (every
    // lane ends with the same total), then the two wave totals in a fixed order
    double vq = 0.0, vm = 0.0;
    if (lead) {
        t = ((t + s_t[0][i]) + s_t[1][i]) + s_t[2][i];
        vq = t * (double)wf;
        vm = abar_d[i] * (double)wf;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        vq += __shfl_xor(vq, off);
        vm += __shfl_xor(vm, off);
    }
    if (lead && (i & 63) == 0) {
        s_q[i >> 6] = vq;
        s_m[i >> 6] = vm;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double var = s_q[0] + s_q[1];
        const double wa = s_m[0] + s_m[1];
        if (var < 0.0) var = 0.0;
        const double mu = wa + (double)b3[c];
        const double rs = 1.0 / sqrt(var + (double)BN_EPS3);
        mean[c] = (float)mu;
        rstd[c] = (float)rs;
        if (running_mean != nullptr) {
            const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
        }
        const float sc = (float)rs * gamma[c];
        s_sc = sc;
        sh[c] = (float)((double)beta[c] - (double)sc * wa);
    }
    __syncthreads();
    const float wv = wf * s_sc;
    if (lead) w3p[w3p_index(c, i)] = wv;
    if (w3s != nullptr && wsc != nullptr) {
        // The same weight as TWO float16 pieces (conv3_pool_train_ws_kernel): the channel's row is first scaled by a
        // power of two that puts its largest entry in [2^13, 2^14) - float16 has 11 significand bits but only 5
        // exponent bits - so that hi + lo carries 22 bits of every entry within 2^17 of the largest; wsc[c] undoes it.
        float m = fabsf(wv);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lead && (i & 63) == 0) s_max[i >> 6] = m;
        __syncthreads();
        if (!lead) return;
        m = fmaxf(s_max[0], s_max[1]);
        const float up = pow2_scale_to_2p13(m);
        if (i == 0) wsc[c] = 1.0f / up;                       // a power of two: exact
        const float ws = wv * up;
        const _Float16 hi = (_Float16)ws;
        const _Float16 lo = (_Float16)(ws - (float)hi);
        const int base = w3h16_index(c, i);
        w3s[base] = __builtin_bit_cast(unsigned short, hi);
        w3s[base + 512] = __builtin_bit_cast(unsigned short, lo);
    } else if (w3s != nullptr && lead) {
        // the same weight as three bfloat16 pieces (wv = hi + mid + lo to 2^-27), in 32x32x16 fragment order
        const __bf16 hi = (__bf16)wv;
        const float r1 = wv - (float)hi;
        const __bf16 mid = (__bf16)r1;
        const __bf16 lo = (__bf16)(r1 - (float)mid);
        const int base = w3s_index(c, i);
        w3s[base] = __builtin_bit_cast(unsigned short, hi);
        w3s[base + 512] = __builtin_bit_cast(unsigned short, mid);
        w3s[base + 1024] = __builtin_bit_cast(unsigned short, lo);
    }
}

// ---------------------------------------------------------------------------------------------
// forward: pooled[b,c] = act(max_p h[b,p,:].w'_c + sh_c), arg = first p attaining the maximum
// grid (B, CS): a workgroup of 4 waves handles 1024/CS channels of one candidate; every wave takes
// 64 points per pass (two 32-point tiles as MFMA A operands, read straight from the rows of h) and
// sweeps the block's channel tiles, whose weights stream through a double-buffered 8 KiB LDS chunk.
// ---------------------------------------------------------------------------------------------
constexpr int TC_WAVES = 4, TC_THREADS = 256, TC_MT = 2;
constexpr int TC_CHUNK_BYTES = 8192;     // half a 32-channel tile: 64 k values x 32 channels

__device__ __forceinline__ void tc_dma_chunk(const char* src, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        __builtin_amdgcn_global_load_lds((const void*)(src + piece * 1024 + lane * 16), (lds_void*)(dst + piece * 1024),
                                         16, 0, 0);
    }
}

__global__ __launch_bounds__(TC_THREADS, 2) void conv3_pool_train_kernel(const float* __restrict__ h,
                                                                         const float* __restrict__ w3p,
                                                                         const float* __restrict__ sh, int P, int relu,
                                                                         int ntb, float* __restrict__ pooled,
                                                                         int* __restrict__ arg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, nt0 = blockIdx.y * ntb;
    const int nch = 32 * ntb, nchunks = 2 * ntb;
    char* s_w3 = smem;
    float* s_red = (float*)(smem + 2 * TC_CHUNK_BYTES);      // [4][nch]
    int* s_arg = (int*)(s_red + TC_WAVES * nch);             // [4][nch]
    const char* g_w3 = (const char*)w3p + (size_t)(2 * nt0) * TC_CHUNK_BYTES;

    tc_dma_chunk(g_w3, s_w3, wave, lane);
    for (int c = tid; c < TC_WAVES * nch; c += TC_THREADS) {
        s_red[c] = -INFINITY;
        s_arg[c] = 0;
    }
    lds_dma_wait();
    __syncthreads();

    float* s_red_wave = s_red + wave * nch;
    int* s_arg_wave = s_arg + wave * nch;
    const int pts_per_pass = TC_WAVES * 32 * TC_MT;
    const int npass = (P + pts_per_pass - 1) / pts_per_pass;

    for (int pass = 0; pass < npass; ++pass) {
        // A operands: X[mt][t] register 4g+e holds k = 32t + 8g + 4hh + e of point r of tile mt
        f32x16 X[TC_MT][4];
        const int pt_base = pass * pts_per_pass + wave * TC_MT * 32;
#pragma unroll
        for (int mt = 0; mt < TC_MT; ++mt) {
            int pt = pt_base + mt * 32 + r;
            pt = pt < P ? pt : P - 1;                       // duplicates of the last point cannot change a maximum
            const float* hp = h + ((size_t)b * P + pt) * C2 + 4 * hh;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *(const f32x4*)(hp + 32 * t + 8 * g);
                    X[mt][t][4 * g + 0] = v[0];
                    X[mt][t][4 * g + 1] = v[1];
                    X[mt][t][4 * g + 2] = v[2];
                    X[mt][t][4 * g + 3] = v[3];
                }
        }
        const bool last_pass = (pass == npass - 1);
        for (int nt = 0; nt < ntb; ++nt) {
            f32x16 acc[TC_MT];
#pragma unroll
            for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = f32x16{0};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ck = 2 * nt + half;
                if (!(last_pass && ck == nchunks - 1)) {
                    const int nxt = ck + 1 < nchunks ? ck + 1 : 0;
                    tc_dma_chunk(g_w3 + (size_t)nxt * TC_CHUNK_BYTES, s_w3 + ((half + 1) & 1) * TC_CHUNK_BYTES, wave,
                                 lane);
                }
                const char* buf = s_w3 + half * TC_CHUNK_BYTES;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const f32x4 bw = *(const f32x4*)(buf + (q * 64 + lane) * 16);
                    const int t = 2 * half + (q >> 2);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int mt = 0; mt < TC_MT; ++mt)
                            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(X[mt][t][4 * (q & 3) + e], bw[e], acc[mt], 0,
                                                                          0, 0);
                }
                lds_dma_wait();         // the chunk issued at the top of this half landed long ago, behind the MFMAs
                __syncthreads();
            }
            // accumulator register i of lane half hh is point (i&3) + 8(i>>2) + 4hh of the tile, channel 32nt + r:
            // ascending i is ascending point index, so a strict compare keeps the first maximum
            float bv = -INFINITY;
            int bi = 0;
#pragma unroll
            for (int mt = 0; mt < TC_MT; ++mt) {
                const int base = pt_base + mt * 32 + 4 * hh;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = acc[mt][i];
                    const int idx = base + (i & 3) + 8 * (i >> 2);
                    if (v > bv) {
                        bv = v;
                        bi = idx;
                    }
                }
            }
            bi = bi < P ? bi : P - 1;
            const float ov = __shfl_xor(bv, 32);
            const int oi = __shfl_xor(bi, 32);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
            if (hh == 0 && bv > s_red_wave[32 * nt + r]) {   // later passes hold later points: strict compare
                s_red_wave[32 * nt + r] = bv;
                s_arg_wave[32 * nt + r] = bi;
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += TC_THREADS) {
        float bv = s_red[c];
        int bi = s_arg[c];
#pragma unroll
        for (int w = 1; w < TC_WAVES; ++w) {
            const float v = s_red[w * nch + c];
            const int i = s_arg[w * nch + c];
            if (v > bv || (v == bv && i < bi)) {
                bv = v;
                bi = i;
            }
        }
        const int ch = 32 * nt0 + c;
        bv += sh[ch];
        pooled[(size_t)b * C3 + ch] = relu ? __builtin_fmaxf(bv, 0.f) : bv;
        arg[(size_t)b * C3 + ch] = bi;
    }
}

// The same product on the bfloat16 matrix pipe at float32 accuracy.  Each operand is split into three bfloat16
// pieces, x = xh + xm + xl exactly to 2^-27 |x| (each piece is the rounded remainder of the one before), and the
// product keeps the six of the nine piece products that are larger than 2^-27 of the whole:
//     x w ~ xh wh + (xh wm + xm wh) + (xh wl + xm wm + xl wh)
// Piece products are exact in the float32 accumulator, so the result differs from an exact-float32 dot product by a
// few 2^-24 - the rounding of the accumulation itself - while six v_mfma_f32_32x32x16_bf16 (32 cycles each for
// k = 16) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each): 2.7 times the rate.
// (Two pieces, three products, were tried in this step and left 1.5e-5 relative: too much for BatchNorm over a
// handful of samples.)  Same grid, tiles and epilogue as the kernel above; the weight chunks are 12 KiB.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
constexpr int TX_CHUNK_BYTES = 12288;    // half a 32-channel tile: 4 k-steps x 3 pieces x 1 KiB

__device__ __forceinline__ void tx_dma_chunk(const char* src, char* dst, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int piece = wave * 3 + i;
        __builtin_amdgcn_global_load_lds((const void*)(src + piece * 1024 + lane * 16), (lds_void*)(dst + piece * 1024),
                                         16, 0, 0);
    }
}

__global__ __launch_bounds__(TC_THREADS, 2) void conv3_pool_train_x6_kernel(const float* __restrict__ h,
                                                                            const unsigned short* __restrict__ w3s,
                                                                            const float* __restrict__ sh, int P,
                                                                            int relu, int ntb,
                                                                            float* __restrict__ pooled,
                                                                            int* __restrict__ arg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, nt0 = blockIdx.y * ntb;
    const int nch = 32 * ntb, nchunks = 2 * ntb;
    char* s_w3 = smem;
    float* s_red = (float*)(smem + 2 * TX_CHUNK_BYTES);      // [4][nch]
    int* s_arg = (int*)(s_red + TC_WAVES * nch);             // [4][nch]
    const char* g_w3 = (const char*)w3s + (size_t)(2 * nt0) * TX_CHUNK_BYTES;

    tx_dma_chunk(g_w3, s_w3, wave, lane);
    for (int c = tid; c < TC_WAVES * nch; c += TC_THREADS) {
        s_red[c] = -INFINITY;
        s_arg[c] = 0;
    }
    lds_dma_wait();
    __syncthreads();

    float* s_red_wave = s_red + wave * nch;
    int* s_arg_wave = s_arg + wave * nch;
    const int pts_per_pass = TC_WAVES * 32 * TC_MT;
    const int npass = (P + pts_per_pass - 1) / pts_per_pass;

    for (int pass = 0; pass < npass; ++pass) {
        // A operands: piece [mt][ks] of lane (r, hh) holds k = 16ks + 8hh + j of point r of tile mt
        bf16x8 Ah[TC_MT][8], Am[TC_MT][8], Al[TC_MT][8];
        const int pt_base = pass * pts_per_pass + wave * TC_MT * 32;
#pragma unroll
        for (int mt = 0; mt < TC_MT; ++mt) {
            int pt = pt_base + mt * 32 + r;
            pt = pt < P ? pt : P - 1;                       // duplicates of the last point cannot change a maximum
            const float* hp = h + ((size_t)b * P + pt) * C2 + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const f32x4 v0 = *(const f32x4*)(hp + 16 * ks);
                const f32x4 v1 = *(const f32x4*)(hp + 16 * ks + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float x = j < 4 ? v0[j & 3] : v1[j & 3];
                    const __bf16 xh = (__bf16)x;
                    const float r1 = x - (float)xh;
                    const __bf16 xm = (__bf16)r1;
                    Ah[mt][ks][j] = xh;
                    Am[mt][ks][j] = xm;
                    Al[mt][ks][j] = (__bf16)(r1 - (float)xm);
                }
            }
        }
        const bool last_pass = (pass == npass - 1);
        for (int nt = 0; nt < ntb; ++nt) {
            f32x16 acc[TC_MT];
#pragma unroll
            for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = f32x16{0};
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ck = 2 * nt + half;
                if (!(last_pass && ck == nchunks - 1)) {
                    const int nxt = ck + 1 < nchunks ? ck + 1 : 0;
                    tx_dma_chunk(g_w3 + (size_t)nxt * TX_CHUNK_BYTES, s_w3 + ((half + 1) & 1) * TX_CHUNK_BYTES, wave,
                                 lane);
                }
                const char* buf = s_w3 + half * TX_CHUNK_BYTES;
#pragma unroll
                for (int ksl = 0; ksl < 4; ++ksl) {
                    const int ks = 4 * half + ksl;
                    const bf16x8 bh = *(const bf16x8*)(buf + ((ksl * 3 + 0) * 64 + lane) * 16);
                    const bf16x8 bm = *(const bf16x8*)(buf + ((ksl * 3 + 1) * 64 + lane) * 16);
                    const bf16x8 bl = *(const bf16x8*)(buf + ((ksl * 3 + 2) * 64 + lane) * 16);
                    // smallest products first
#pragma unroll
                    for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[mt][ks], bh, acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[mt][ks], bm, acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[mt][ks], bl, acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am[mt][ks], bh, acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[mt][ks], bm, acc[mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < TC_MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[mt][ks], bh, acc[mt], 0, 0, 0);
                }
                lds_dma_wait();         // the chunk issued at the top of this half landed long ago, behind the MFMAs
                __syncthreads();
            }
            // accumulator register i of lane half hh is point (i&3) + 8(i>>2) + 4hh of the tile, channel 32nt + r:
            // ascending i is ascending point index, so a strict compare keeps the first maximum
            float bv = -INFINITY;
            int bi = 0;
#pragma unroll
            for (int mt = 0; mt < TC_MT; ++mt) {
                const int base = pt_base + mt * 32 + 4 * hh;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = acc[mt][i];
                    const int idx = base + (i & 3) + 8 * (i >> 2);
                    if (v > bv) {
                        bv = v;
                        bi = idx;
                    }
                }
            }
            bi = bi < P ? bi : P - 1;
            const float ov = __shfl_xor(bv, 32);
            const int oi = __shfl_xor(bi, 32);
            if (ov > bv || (ov == bv && oi < bi)) {
                bv = ov;
                bi = oi;
            }
            if (hh == 0 && bv > s_red_wave[32 * nt + r]) {   // later passes hold later points: strict compare
                s_red_wave[32 * nt + r] = bv;
                s_arg_wave[32 * nt + r] = bi;
            }
        }
    }
    __syncthreads();
    for (int c = tid; c < nch; c += TC_THREADS) {
        float bv = s_red[c];
        int bi = s_arg[c];
#pragma unroll
        for (int w = 1; w < TC_WAVES; ++w) {
            const float v = s_red[w * nch + c];
            const int i = s_arg[w * nch + c];
            if (v > bv || (v == bv && i < bi)) {
                bv = v;
                bi = i;
            }
        }
        const int ch = 32 * nt0 + c;
        bv += sh[ch];
        pooled[(size_t)b * C3 + ch] = relu ? __builtin_fmaxf(bv, 0.f) : bv;
        arg[(size_t)b * C3 + ch] = bi;
    }
}

// The same product with every operand as TWO float16 pieces and THREE piece products:
//     x w ~ xl wh + xh wl + xh wh
// float16 carries 11 significand bits, so hi + lo holds 22 bits of an operand (the bfloat16 split above needs three
// pieces for 24) and what is dropped - xl wl and the two representation remainders - is below 3 * 2^-22 of the product:
// the size of the float32 accumulation's own rounding over 128 terms.  Half the MFMAs of the six-product kernel.
// float16 has only 5 exponent bits, so both operands are scaled by powers of two first (exact): a weight row by the
// factor that k_conv3_stats_pack chose for its channel (largest entry in [2^13, 2^14)), a tile of 16 points by the
// factor that does the same for its largest activation.  Entries within 2^17 of the largest keep all 22 bits; smaller
// ones lose low bits whose weight is below 2^-39 of the largest entry.  A product of two scaled operands is below 2^28
// and a sum of 128 below 2^35.  A block's maximum is found in the scaled domain and descaled (exactly) before it
// meets the running maximum.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// The kernel keeps the WEIGHTS stationary (the scoring kernels' structure, conv_stack_x3.hip): a
// workgroup of 8 waves owns one candidate and 512 channels; wave w keeps both float16 pieces of channels
// [64w, 64w + 64) of that half in 128 registers for the whole kernel (4 tiles of 16 channels x 4 k-steps of 32 x 2
// pieces, v_mfma_f32_16x16x32_f16), so no weight ever passes through LDS and no barrier paces the MFMAs.  What moves
// through LDS is the points: tiles of 16 rows of h, scaled by the power of two that puts the tile's largest entry in
// [2^13, 2^14), split into the two pieces and stored in A-fragment order (8 KiB a tile, 8 tiles a batch, two batches
// in LDS); wave w makes tile w of the NEXT batch in the second half of the consume loop - its rows were requested at the
// top of the batch - two waves of different SIMDs per tile step.  One barrier per 128 points.
// Every wave multiplies every tile with its channels (a chain of 12 MFMAs per 16 x 16 block accumulating in place),
// finds the block's largest value and its row, descales it (tile factor x channel factor: exact) and compares it with
// the lane's running maximum; tiles come in ascending order of their points, so a strict compare keeps the first
// maximum, as max_pool1d does.
constexpr int TW_WAVES = 8, TW_THREADS = TW_WAVES * 64;
constexpr int TW_PLANE_BYTES = 4096;                       // [ks 4][lane 64][16 B]
constexpr int TW_TILE_BYTES = 2 * TW_PLANE_BYTES;
constexpr int TW_BATCH_TILES = TW_WAVES;
constexpr int TW_BATCH_BYTES = TW_BATCH_TILES * TW_TILE_BYTES;
constexpr int TW_OFF_SCALE = 2 * TW_BATCH_BYTES;           // [2][8] floats: the factor that undoes a tile's scaling
constexpr int TW_LDS_BYTES = TW_OFF_SCALE + 2 * TW_BATCH_TILES * 4;
constexpr int TW_NT = 4;                                   // channel tiles of 16 per wave

// Diagnostic build only (-DPNGPD_STAMPS, tools/diag/tw_stamps.py): shader-clock stamps of workgroup 0; no output value
// depends on them and the product build compiles none of it.
#ifdef PNGPD_STAMPS
constexpr int TW_STAMP_BATCHES = 6, TW_STAMP_EVENTS = 11;    // batch start, behind each of 8 tiles, barrier, behind it
__device__ unsigned long long g_tw_stamps[TW_WAVES][TW_STAMP_BATCHES * TW_STAMP_EVENTS];
extern "C" int pngpd_debug_read_tw_stamps(unsigned long long* stamps) {
    return (int)hipMemcpyFromSymbol(stamps, HIP_SYMBOL(g_tw_stamps), sizeof(g_tw_stamps));
}
#define TW_STAMP(bt, ev)                                                                                   \
    do {                                                                                                   \
        if (blockIdx.x == 0 && lane == 0 && (bt) < TW_STAMP_BATCHES)                                       \
            g_tw_stamps[wave][(bt) * TW_STAMP_EVENTS + (ev)] = __builtin_amdgcn_s_memtime();               \
    } while (0)
#else
#define TW_STAMP(bt, ev)
#endif

__global__ __launch_bounds__(TW_THREADS) void conv3_pool_train_ws_kernel(const float* __restrict__ h,
                                                                         const unsigned short* __restrict__ w3h,
                                                                         const float* __restrict__ wsc,
                                                                         const float* __restrict__ sh, int B, int P,
                                                                         int relu, float* __restrict__ pooled,
                                                                         int* __restrict__ arg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, q = lane >> 4;
    // Workgroups are handed to the 8 XCDs in turn: the two channel halves of a candidate are workgroups L and L + 8, so
    // they run on the same XCD one after the other and the second finds the candidate's rows in that XCD's L2
    // (as (candidate, half) = (blockIdx.x, blockIdx.y) they were 256 workgroups apart: 201 MB fetched per launch for
    // 98 MB of rows)
    const int L = blockIdx.x;
    const int b = (L >> 4) * 8 + (L & 7), half = (L >> 3) & 1;
    if (b >= B) return;
    float* s_scale = (float*)(smem + TW_OFF_SCALE);

    const int ct0 = (half * TW_WAVES + wave) * TW_NT;                 // first channel tile of this wave
    f16x8 bwh[TW_NT][4], bwl[TW_NT][4];
    float cdescale[TW_NT];
#pragma unroll
    for (int nt = 0; nt < TW_NT; ++nt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const size_t at = (size_t)((((ct0 + nt) * 4 + ks) * 2) * 64 + lane) * 16;
            bwh[nt][ks] = *(const f16x8*)((const char*)w3h + at);
            bwl[nt][ks] = *(const f16x8*)((const char*)w3h + at + 1024);
        }
        cdescale[nt] = wsc[16 * (ct0 + nt) + c];
    }

    const int ntiles = (P + 15) >> 4;
    const int nbatch = (ntiles + TW_BATCH_TILES - 1) / TW_BATCH_TILES;
    const float* hb = h + (size_t)b * P * C2;

    // rows of the tile this wave makes next: lane (c, q) holds k = 32ks + 8q .. + 7 of row c
    f32x4 st[4][2];
    auto fetch_tile = [&](int tile) {
        int pt = tile * 16 + c;
        pt = pt < P ? pt : P - 1;                               // rows beyond P repeat the last point
        const float* hp = hb + (size_t)pt * C2 + 8 * q;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            st[ks][0] = *(const f32x4*)(hp + 32 * ks);
            st[ks][1] = *(const f32x4*)(hp + 32 * ks + 4);
        }
    };
    auto make_tile = [&](char* dst, float* scale_slot) {
        float m = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) m = __builtin_fmaxf(m, __builtin_fabsf(st[ks][j >> 2][j & 3]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off));
        const float up = pow2_scale_to_2p13(m);
        if (lane == 0) *scale_slot = 1.0f / up;                  // a power of two: exact
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f16x8 xh, xl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = st[ks][j >> 2][j & 3] * up;
                const _Float16 hi = (_Float16)x;
                xh[j] = hi;
                xl[j] = (_Float16)(x - (float)hi);
            }
            *(f16x8*)(dst + (ks * 64 + lane) * 16) = xh;
            *(f16x8*)(dst + TW_PLANE_BYTES + (ks * 64 + lane) * 16) = xl;
        }
    };

    fetch_tile(wave);
    make_tile(smem + wave * TW_TILE_BYTES, s_scale + wave);
    __syncthreads();

    float bv[TW_NT];
    int bi[TW_NT];
#pragma unroll
    for (int nt = 0; nt < TW_NT; ++nt) {
        bv[nt] = -INFINITY;
        bi[nt] = 0;
    }
    // A fragments of the tile being multiplied, refilled IN PLACE for the next tile inside the last chain of a tile:
    // the lo plane when its four MFMAs have been issued, each hi fragment behind its last MFMA (a second set of 32
    // registers does not fit beside 128 of weights and 32 of rows in flight)
    f16x8 ah[4], al[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ah[ks] = *(const f16x8*)(smem + (ks * 64 + lane) * 16);
        al[ks] = *(const f16x8*)(smem + TW_PLANE_BYTES + (ks * 64 + lane) * 16);
    }
    auto search = [&](const f32x4& z, int nt, float descale, int row0) {
        // register i is row 4q + i of the tile, channel c of the channel tile: ascending i is ascending point
        float m = z[0];
        int mi = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
            if (z[i] > m) {
                m = z[i];
                mi = i;
            }
        m *= descale * cdescale[nt];                             // powers of two: exact
        if (m > bv[nt]) {                                        // later tiles hold later points: strict compare
            bv[nt] = m;
            bi[nt] = row0 + mi;
        }
        // opaque to the optimiser: in the unrolled batch it otherwise merges the compare chains of all tiles into one
        // tree at the end of the batch and keeps every accumulator alive (in scratch memory) until then
        asm volatile("" : "+v"(bv[nt]), "+v"(bi[nt]));
    };
    f32x4 zp = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};     // before the first block: never beats a maximum
    float zp_descale = 1.0f;
    int zp_row0 = 0;
    for (int bt = 0; bt < nbatch; ++bt) {
        const int buf = bt & 1;
        const char* cur = smem + buf * TW_BATCH_BYTES;
        const bool more = bt + 1 < nbatch;
        TW_STAMP(bt, 0);
        if (more) fetch_tile((bt + 1) * TW_BATCH_TILES + wave);
#pragma unroll
        for (int t = 0; t < TW_BATCH_TILES; ++t) {
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) TW_STAMP(bt, t);
            // waves 2j and 2j + 1 (different SIMDs) make their tiles in front of tile 4 + j: the rows were requested at
            // the top of the batch, four tiles of MFMAs earlier (made in front of tile `wave`, wave 0 waited for memory
            // at the start of every batch and the other seven for wave 0 at its end)
            if (more && t == 4 + (wave >> 1))
                make_tile(smem + (buf ^ 1) * TW_BATCH_BYTES + wave * TW_TILE_BYTES, s_scale + (buf ^ 1) * TW_BATCH_TILES + wave);
            // the tile multiplied after this one: tile 0 of the next batch behind the last (its buffer is complete only
            // behind the barrier, so that refill waits for it - see below)
            const char* nxt = t + 1 < TW_BATCH_TILES ? cur + (t + 1) * TW_TILE_BYTES : nullptr;
            const float tdescale = s_scale[buf * TW_BATCH_TILES + t];
            const int row0 = (bt * TW_BATCH_TILES + t) * 16 + 4 * q;
            __builtin_amdgcn_sched_barrier(0);
            // The search of a block (17 vector instructions that read the chain's result) goes with the MFMAs of the
            // FOLLOWING block, when that result has long been written, and in one scheduling region with them, so that
            // the scheduler fills the hazard slots between a compare and its select with MFMAs (fenced off behind the
            // chain: +1.4 % kernel time; pinned one MFMA : two vector instructions: no gain over the scheduler's own
            // order).  zp / zp_descale / zp_row0 carry the previous block.
#pragma unroll
            for (int nt = 0; nt < TW_NT - 1; ++nt) {
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                // smallest products first
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bwh[nt][ks], z, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bwl[nt][ks], z, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bwh[nt][ks], z, 0, 0, 0);
                search(zp, (nt + TW_NT - 1) % TW_NT, nt == 0 ? zp_descale : tdescale, nt == 0 ? zp_row0 : row0);
                __builtin_amdgcn_sched_barrier(0);
                zp = z;
            }
            {
                constexpr int nt = TW_NT - 1;
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) z = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ks], bwh[nt][ks], z, 0, 0, 0);
                if (nxt != nullptr) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) al[ks] = *(const f16x8*)(nxt + TW_PLANE_BYTES + (ks * 64 + lane) * 16);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bwl[nt][ks], z, 0, 0, 0);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    z = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ks], bwh[nt][ks], z, 0, 0, 0);
                    if (nxt != nullptr) ah[ks] = *(const f16x8*)(nxt + (ks * 64 + lane) * 16);
                }
                search(zp, nt - 1, tdescale, row0);
                __builtin_amdgcn_sched_barrier(0);
                zp = z;
                zp_descale = tdescale;
                zp_row0 = row0;
            }
        }
        TW_STAMP(bt, 8);
        __syncthreads();
        TW_STAMP(bt, 9);
        if (more) {
            const char* first = smem + (buf ^ 1) * TW_BATCH_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                ah[ks] = *(const f16x8*)(first + (ks * 64 + lane) * 16);
                al[ks] = *(const f16x8*)(first + TW_PLANE_BYTES + (ks * 64 + lane) * 16);
            }
        }
    }
    search(zp, TW_NT - 1, zp_descale, zp_row0);
    // the four row groups q of a channel; the lowest point wins a tie
#pragma unroll
    for (int nt = 0; nt < TW_NT; ++nt) {
        float v = bv[nt];
        int i = bi[nt] < P ? bi[nt] : P - 1;
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(v, off);
            const int oi = __shfl_xor(i, off);
            if (ov > v || (ov == v && oi < i)) {
                v = ov;
                i = oi;
            }
        }
        const int ch = 16 * (ct0 + nt) + c;
        v += sh[ch];
        if (q == 0) {
            pooled[(size_t)b * C3 + ch] = relu ? __builtin_fmaxf(v, 0.f) : v;
            arg[(size_t)b * C3 + ch] = i;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// m1 = sum_b g / M, m2 = sum_b g*yhat / M, dgamma = sum g*yhat, dbeta = sum g, and the masked pooled gradient g.
// yhat at the arg-max row follows from the pooled value itself: pooled (before the ReLU) = gamma*yhat + beta; where
// the ReLU clipped it g is 0 and yhat is not needed.  Only a channel with gamma == 0 (or tiny against beta)
// recomputes the 128-term product at its arg-max row.  block = 16 channels x 16 candidate lanes.
__global__ __launch_bounds__(256) void k_conv3_bwd_reduce(const float* __restrict__ h, const float* __restrict__ W3,
                                                          const float* __restrict__ b3,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const float* __restrict__ dpooled,
                                                          const float* __restrict__ pooled, const int* __restrict__ arg,
                                                          int B, int P, int relu, float* __restrict__ gmask,
                                                          float* __restrict__ m1, float* __restrict__ m2,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ dbias_zero) {
    __shared__ double shm[2][16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const float ga = gamma[c], be = beta[c];
    // (pooled - beta) / gamma loses digits when |gamma| << |beta|: such channels take the product path as well
    const bool direct = ga != 0.f && fabsf(ga) > 1e-2f * fabsf(be);
    const float inv_ga = direct ? 1.f / ga : 0.f;
    double s1 = 0.0, s2 = 0.0;
    for (int b = rl; b < B; b += 16) {
        const size_t o = (size_t)b * C3 + c;
        const float pv = pooled[o];
        float g = dpooled[o];
        if (relu && !(pv > 0.f)) g = 0.f;
        gmask[o] = g;
        float yh;
        if (direct) {
            yh = (pv - be) * inv_ga;
        } else {
            const float* hr = h + ((size_t)b * P + arg[o]) * C2;
            const float* wr = W3 + (size_t)c * C2;
            float d = 0.f;
            for (int k = 0; k < C2; ++k) d = fmaf(hr[k], wr[k], d);
            yh = (d + b3[c] - mean[c]) * rstd[c];
        }
        s1 += (double)g;
        s2 += (double)(g * yh);
    }
    shm[0][rl][cl] = s1;
    shm[1][rl][cl] = s2;
    __syncthreads();
    if (rl != 0) return;
    s1 = 0.0;
    s2 = 0.0;
    for (int k = 0; k < 16; ++k) {
        s1 += shm[0][k][cl];
        s2 += shm[1][k][cl];
    }
    const double M = (double)B * P;
    m1[c] = (float)(s1 / M);
    m2[c] = (float)(s2 / M);
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    if (dbias_zero != nullptr) dbias_zero[c] = 0.f;      // the conv bias in front of the BatchNorm: exactly 0
}

// G[i][k] = sum_c q_c W[c,i] W[c,k] (blocks 0..127 = i), ub[k] = sum_c (p_c - q_c (mean_c - b_c)) W[c,k] (block 128)
// 512 threads = 128 columns k x 4 groups of 256 channels; the group sums are added in a fixed order.
__global__ __launch_bounds__(512) void k_conv3_bwd_G(const float* __restrict__ W3, const float* __restrict__ b3,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* __restrict__ m1,
                                                     const float* __restrict__ m2, float* __restrict__ G,
                                                     float* __restrict__ ub) {
    __shared__ double s_f[C3];
    __shared__ double s_acc[4][C2];
    const int i = blockIdx.x, k = threadIdx.x & (C2 - 1), cg = threadIdx.x >> 7;
    for (int c = threadIdx.x; c < C3; c += 512) {
        const double A = (double)gamma[c] * (double)rstd[c];
        const double q = -A * (double)rstd[c] * (double)m2[c];
        if (i < C2)
            s_f[c] = q * (double)W3[(size_t)c * C2 + i];
        else
            s_f[c] = -A * (double)m1[c] - q * ((double)mean[c] - (double)b3[c]);
    }
    __syncthreads();
    double acc = 0.0;
    const int c0 = cg * (C3 / 4);
#pragma unroll 8
    for (int c = c0; c < c0 + C3 / 4; ++c) acc += s_f[c] * (double)W3[(size_t)c * C2 + k];
    s_acc[cg][k] = acc;
    __syncthreads();
    if (cg != 0) return;
    acc = s_acc[0][k] + s_acc[1][k] + s_acc[2][k] + s_acc[3][k];
    if (i < C2)
        G[(size_t)i * C2 + k] = (float)acc;
    else
        ub[k] = (float)acc;
}

// Rows that hold a maximum: dh[b*P + arg(b,c), :] += A_c g[b,c] W[c,:].  One workgroup of 16 waves per
// candidate: the 1024 channels are sorted by (arg row, channel) and every wave adds up a 64-entry piece of the
// sorted list, run by run, in a fixed order.
__global__ __launch_bounds__(1024) void k_conv3_sparse_dgrad(const float* __restrict__ W3, const float* __restrict__ gamma,
                                                             const float* __restrict__ rstd,
                                                             const float* __restrict__ gmask,
                                                             const int* __restrict__ arg, int P,
                                                             float* __restrict__ dh) {
    __shared__ int s_key[C3];
    __shared__ float s_coef[C3];
    const int b = blockIdx.x, c = threadIdx.x, lane = c & 63, wave = c >> 6;
    // keys (row << 10 | channel) sorted ascending by a bitonic network: order by row, then by channel
    s_key[c] = (arg[(size_t)b * C3 + c] << 10) | c;
    s_coef[c] = gamma[c] * rstd[c] * gmask[(size_t)b * C3 + c];
    __syncthreads();
    for (int k = 2; k <= C3; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int other = c ^ j;
            if (other > c) {
                const int x = s_key[c], y = s_key[other];
                if ((x > y) == ((c & k) == 0)) {
                    s_key[c] = y;
                    s_key[other] = x;
                }
            }
            __syncthreads();
        }
    // Wave w walks positions [64w, 64w+64) of the sorted list and adds up runs of equal rows.  A run that lies
    // inside the wave's piece is added to its row of dh directly - one add per address, so the hardware float
    // atomic is order-free there and spares the read half of a read-modify-write.  The first and the last run of
    // a piece may continue in a neighbouring piece: they go to an LDS edge list that wave 0 combines in order.
    __shared__ float s_edge[32][C2];
    __shared__ int s_edge_row[32];
    float* dhb = dh + (size_t)b * P * C2 + lane;      // lane owns columns lane and lane + 64: every add instruction covers 256 consecutive bytes
    const int begin = wave * 64, end = begin + 64;
    int cur = -1, runs = 0;
    float2 acc = make_float2(0.f, 0.f);
    auto close_run = [&](bool last) {
        if (cur < 0) return;
        if (runs == 1 || last) {
            const int e = 2 * wave + (runs == 1 ? 0 : 1);
            s_edge[e][lane] = acc.x;
            s_edge[e][64 + lane] = acc.y;
            if (lane == 0) s_edge_row[e] = cur;
        } else {
            unsafeAtomicAdd(dhb + (size_t)cur * C2, acc.x);
            unsafeAtomicAdd(dhb + (size_t)cur * C2 + 64, acc.y);
        }
    };
    if (lane == 0) s_edge_row[2 * wave + 1] = -1;     // stays empty when the piece is a single run
    constexpr int DEPTH = 8;                          // weight rows in flight
    for (int base = begin; base < end; base += DEPTH) {
        int row[DEPTH];
        float cf[DEPTH];
        float2 w[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int key = s_key[base + u];
            const int ch = key & (C3 - 1);
            row[u] = key >> 10;
            cf[u] = s_coef[ch];
            w[u] = make_float2(W3[(size_t)ch * C2 + lane], W3[(size_t)ch * C2 + 64 + lane]);
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (row[u] != cur) {
                close_run(false);
                cur = row[u];
                ++runs;
                acc = make_float2(0.f, 0.f);
            }
            acc.x = fmaf(cf[u], w[u].x, acc.x);
            acc.y = fmaf(cf[u], w[u].y, acc.y);
        }
    }
    close_run(true);
    __syncthreads();
    if (wave != 0) return;
    cur = -1;
    acc = make_float2(0.f, 0.f);
    for (int e = 0; e < 32; ++e) {
        const int r = s_edge_row[e];
        if (r < 0) continue;
        if (r != cur) {
            if (cur >= 0) {
                unsafeAtomicAdd(dhb + (size_t)cur * C2, acc.x);
                unsafeAtomicAdd(dhb + (size_t)cur * C2 + 64, acc.y);
            }
            cur = r;
            acc = make_float2(0.f, 0.f);
        }
        acc.x += s_edge[e][lane];
        acc.y += s_edge[e][64 + lane];
    }
    if (cur >= 0) {
        unsafeAtomicAdd(dhb + (size_t)cur * C2, acc.x);
        unsafeAtomicAdd(dhb + (size_t)cur * C2 + 64, acc.y);
    }
}

// dW[c,j] = A_c sum_b g[b,c] h[arg(b,c), j] + M (p_c abar_j + q_c sum_i W[c,i] Cov[i][j]); 128 threads per channel,
// two channels (2 blk, 2 blk + 1) per workgroup of 256
__device__ __forceinline__ void conv3_wgrad_block(const float* __restrict__ h, const float* __restrict__ W3,
                                                  const float* __restrict__ gamma, const float* __restrict__ rstd,
                                                  const float* __restrict__ m1, const float* __restrict__ m2,
                                                  const float* __restrict__ gmask, const int* __restrict__ arg,
                                                  const double* __restrict__ cov, const double* __restrict__ abar_d,
                                                  int B, int P, float* __restrict__ dW, int blk) {
    __shared__ float s_g[2][128];
    __shared__ int s_r[2][128];
    __shared__ double s_w[2][C2];
    const int half = threadIdx.x >> 7, j = threadIdx.x & 127, c = 2 * blk + half;
    s_w[half][j] = (double)W3[(size_t)c * C2 + j];
    double sparse = 0.0;
    for (int b0 = 0; b0 < B; b0 += 128) {
        __syncthreads();
        const int bb = b0 + j;
        if (bb < B) {
            s_g[half][j] = gmask[(size_t)bb * C3 + c];
            s_r[half][j] = bb * P + arg[(size_t)bb * C3 + c];
        }
        __syncthreads();
        const int n = min(128, B - b0);
        float acc = 0.f;
#pragma unroll 8
        for (int t = 0; t < n; ++t) acc = fmaf(s_g[half][t], h[(size_t)s_r[half][t] * C2 + j], acc);
        sparse += (double)acc;
    }
    __syncthreads();
    double dense = 0.0;
#pragma unroll 8
    for (int i = 0; i < C2; ++i) dense += s_w[half][i] * cov[(size_t)i * C2 + j];
    const double M = (double)B * P;
    const double A = (double)gamma[c] * (double)rstd[c];
    const double p = -A * (double)m1[c], q = -A * (double)rstd[c] * (double)m2[c];
    dW[(size_t)c * C2 + j] = (float)(A * sparse + M * (p * abar_d[j] + q * dense));
}

// Column sums for the backward of the BatchNorm in FRONT of conv3 (128 channels, ReLU behind it), over the rows
// [blk rows_per_block, ...): s1 = sum g, s2 = sum g yhat with g = dh masked by the sign of the recomputed
// pre-activation and yhat = (y - mean) rstd.  part [nblk][2][128] doubles, as k_colreduce writes them.
// 256 threads = 32 column quads x 8 row lanes.
__device__ __forceinline__ void bn2_bwd_sums_block(const float* __restrict__ dh, const float* __restrict__ y,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta, int M,
                                                   int rows_per_block, double* __restrict__ part, int blk) {
    __shared__ double red[2][8][C2];
    const int ql = threadIdx.x & 31, rl = threadIdx.x >> 5, col = ql * 4;
    const int r0 = blk * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const float4 mu = *(const float4*)(mean + col), rs = *(const float4*)(rstd + col);
    const float4 ga = make_float4(gamma[col], gamma[col + 1], gamma[col + 2], gamma[col + 3]);
    const float4 be = make_float4(beta[col], beta[col + 1], beta[col + 2], beta[col + 3]);
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 8) {
        float4 g = *(const float4*)(dh + (size_t)r * C2 + col);
        const float4 yv = *(const float4*)(y + (size_t)r * C2 + col);
        if (!(bn_pre(yv.x, mu.x, rs.x, ga.x, be.x) > 0.f)) g.x = 0.f;
        if (!(bn_pre(yv.y, mu.y, rs.y, ga.y, be.y) > 0.f)) g.y = 0.f;
        if (!(bn_pre(yv.z, mu.z, rs.z, ga.z, be.z) > 0.f)) g.z = 0.f;
        if (!(bn_pre(yv.w, mu.w, rs.w, ga.w, be.w) > 0.f)) g.w = 0.f;
        s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
        s2.x += g.x * ((yv.x - mu.x) * rs.x);
        s2.y += g.y * ((yv.y - mu.y) * rs.y);
        s2.z += g.z * ((yv.z - mu.z) * rs.z);
        s2.w += g.w * ((yv.w - mu.w) * rs.w);
    }
    double* d1 = &red[0][rl][col];
    double* d2 = &red[1][rl][col];
    d1[0] = (double)s1.x; d1[1] = (double)s1.y; d1[2] = (double)s1.z; d1[3] = (double)s1.w;
    d2[0] = (double)s2.x; d2[1] = (double)s2.y; d2[2] = (double)s2.z; d2[3] = (double)s2.w;
    __syncthreads();
    if (threadIdx.x < C2) {
        double a = 0.0, b = 0.0;
        for (int k = 0; k < 8; ++k) {          // fixed order: reproducible
            a += red[0][k][threadIdx.x];
            b += red[1][k][threadIdx.x];
        }
        part[((size_t)blk * 2 + 0) * C2 + threadIdx.x] = a;
        part[((size_t)blk * 2 + 1) * C2 + threadIdx.x] = b;
    }
}

// conv3's weight gradient and - independent of it, and bound by memory where the former is bound by the latency of
// its row gathers - the column sums of the BatchNorm backward that follows, in one launch: workgroups [0, C3/2) take
// two channels of dW each, the rest the row blocks of the sums (bn_y == nullptr: no sums, grid = C3/2).
__global__ __launch_bounds__(256) void k_conv3_wgrad_bn2sums(const float* __restrict__ h, const float* __restrict__ W3,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ rstd, const float* __restrict__ m1,
                                                             const float* __restrict__ m2,
                                                             const float* __restrict__ gmask, const int* __restrict__ arg,
                                                             const double* __restrict__ cov,
                                                             const double* __restrict__ abar_d, int B, int P,
                                                             float* __restrict__ dW, const float* __restrict__ bn_dh,
                                                             const float* __restrict__ bn_y,
                                                             const float* __restrict__ bn_mean,
                                                             const float* __restrict__ bn_rstd,
                                                             const float* __restrict__ bn_gamma,
                                                             const float* __restrict__ bn_beta, int rows_per_block,
                                                             double* __restrict__ bn_part) {
    if (blockIdx.x < C3 / 2)
        conv3_wgrad_block(h, W3, gamma, rstd, m1, m2, gmask, arg, cov, abar_d, B, P, dW, blockIdx.x);
    else
        bn2_bwd_sums_block(bn_dh, bn_y, bn_mean, bn_rstd, bn_gamma, bn_beta, B * P, rows_per_block, bn_part,
                           blockIdx.x - C3 / 2);
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
hipError_t t_conv3_fwd(float* h2, const float* W3, const float* b3, const float* gamma, const float* beta, int B,
                       int P, int relu_out, float momentum, double* gram_part, const Conv3Ws& w, float* mean, float* rstd,
                       float* running_mean, float* running_var, float* pooled, int* arg, hipStream_t s,
                       const float* y2_pre, const float* bn2_mean, const float* bn2_rstd, const float* bn2_gamma,
                       const float* bn2_beta) {
    const int M = B * P;
    hipError_t e;
    const int chunks = t_gram_chunks(M);
    double* colsum = gram_part + (size_t)chunks * C2 * C2;
    // y2_pre given: h2 = relu(bn2(y2_pre)) is produced by the Gram pass (and stored for the kernels after it)
    if (y2_pre != nullptr && (!bn2_mean || !bn2_rstd || !bn2_gamma || !bn2_beta)) return hipErrorInvalidValue;
    // PNGPD_TRAIN_GRAM=f64 keeps the Gram product on the float64 matrix instruction (the split-bfloat16 form is the default)
    // (read at every call: a test compares the two forms within one process)
    const char* gram_env = getenv("PNGPD_TRAIN_GRAM");
    const bool gram_f64 = gram_env && gram_env[0] == 'f';
    const float* gram_in = y2_pre != nullptr ? y2_pre : (const float*)h2;
    const float* gram_mean = y2_pre != nullptr ? bn2_mean : (const float*)nullptr;
    if (gram_f64)
        hipLaunchKernelGGL(gram_f64_kernel, dim3(3, chunks), dim3(256), 0, s, gram_in, M, t_gram_rows_per_chunk(M), gram_part,
                           colsum, gram_mean, bn2_rstd, bn2_gamma, bn2_beta, h2);
    else
        hipLaunchKernelGGL(gram_x6_kernel, dim3(chunks), dim3(GX_THREADS), 0, s, gram_in, M, t_gram_rows_per_chunk(M),
                           gram_part, colsum, gram_mean, bn2_rstd, bn2_gamma, bn2_beta, h2);
    hipLaunchKernelGGL(k_gram_mean, dim3(C2 / 16), dim3(256), 0, s, (const double*)colsum, chunks, M, w.abar_d);
    // 32x32 blocks of a partial that its kernel wrote (bit 4 i + j)
    const unsigned present = gram_f64 ? 0xCCFFu : 0xAC6Fu;
    hipLaunchKernelGGL(k_gram_finalize, dim3(32, __builtin_popcount(present)), dim3(256), 0, s, (const double*)gram_part,
                       chunks, (const double*)w.abar_d, M, w.cov, present);
    // PNGPD_TRAIN_CONV3 selects the product/max kernel: f32 (v_mfma_f32_32x32x2_f32), x6 (six products of bfloat16 pieces),
    // default: three products of float16 pieces with the weights stationary in registers; read at every call, like
    // PNGPD_TRAIN_GRAM above
    const char* conv3_env = getenv("PNGPD_TRAIN_CONV3");
    const bool use_f32 = (conv3_env && conv3_env[0] == 'f') || w.w3s == nullptr;
    const bool use_x6 = !use_f32 && ((conv3_env && conv3_env[0] == 'x') || w.wsc == nullptr);
    hipLaunchKernelGGL(k_conv3_stats_pack, dim3(C3), dim3(512), 0, s, W3, b3, gamma, beta, (const double*)w.cov,
                       (const double*)w.abar_d, M, momentum, mean, rstd, running_mean, running_var, w.w3p, w.w3s, w.sh,
                       use_x6 ? (float*)nullptr : w.wsc);
    // enough workgroups for two per CU: split the channels of a candidate when the batch is small
    // (B = 256 on 256 CUs: 444 / 408 / 422 / 458 us with 1 / 2 / 4 / 8 pieces)
    int dev = 0, n_cu = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    int cs = 1;
    while (cs < 8 && B * cs < 2 * n_cu) cs *= 2;
    const int ntb = 32 / cs;
    if (use_f32) {
        const size_t lds = 2 * TC_CHUNK_BYTES + (size_t)TC_WAVES * 32 * ntb * 8;
        e = hipFuncSetAttribute((const void*)conv3_pool_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        if (w.ev_begin != nullptr && (e = hipEventRecord(w.ev_begin, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(conv3_pool_train_kernel, dim3(B, cs), dim3(TC_THREADS), lds, s, h2, (const float*)w.w3p,
                           (const float*)w.sh, P, relu_out, ntb, pooled, arg);
    } else if (!use_x6) {
        e = hipFuncSetAttribute((const void*)conv3_pool_train_ws_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                TW_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (w.ev_begin != nullptr && (e = hipEventRecord(w.ev_begin, s)) != hipSuccess) return e;
        static_assert(C3 / (TW_WAVES * TW_NT * 16) == 2, "two channel halves per candidate");
        hipLaunchKernelGGL(conv3_pool_train_ws_kernel, dim3((B + 7) / 8 * 16), dim3(TW_THREADS), TW_LDS_BYTES, s, h2,
                           (const unsigned short*)w.w3s, (const float*)w.wsc, (const float*)w.sh, B, P, relu_out, pooled,
                           arg);
    } else {
        const size_t lds = 2 * TX_CHUNK_BYTES + (size_t)TC_WAVES * 32 * ntb * 8;
        e = hipFuncSetAttribute((const void*)conv3_pool_train_x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
        if (e != hipSuccess) return e;
        if (w.ev_begin != nullptr && (e = hipEventRecord(w.ev_begin, s)) != hipSuccess) return e;
        hipLaunchKernelGGL(conv3_pool_train_x6_kernel, dim3(B, cs), dim3(TC_THREADS), lds, s, h2,
                           (const unsigned short*)w.w3s, (const float*)w.sh, P, relu_out, ntb, pooled, arg);
    }
    if (w.ev_end != nullptr && (e = hipEventRecord(w.ev_end, s)) != hipSuccess) return e;
    return hipGetLastError();
}

hipError_t t_conv3_bwd(const float* h2, const float* W3, const float* b3, const float* gamma, const float* beta,
                       const float* dpooled,
                       const float* pooled, const int* arg, const float* mean, const float* rstd, int B, int P,
                       int relu_out, const Conv3Ws& w, float* m1, float* m2, float* dW3, float* db3_zero, float* dgamma,
                       float* dbeta, float* dh2, hipStream_t s, const float* bn2_y, const float* bn2_mean,
                       const float* bn2_rstd, const float* bn2_gamma, const float* bn2_beta, double* bn2_part) {
    const int M = B * P;
    hipLaunchKernelGGL(k_conv3_bwd_reduce, dim3(C3 / 16), dim3(256), 0, s, h2, W3, b3, gamma, beta, mean, rstd, dpooled,
                       pooled, arg, B, P, relu_out, w.gmask, m1, m2, dgamma, dbeta, db3_zero);
    hipLaunchKernelGGL(k_conv3_bwd_G, dim3(C2 + 1), dim3(512), 0, s, W3, b3, gamma, mean, rstd, (const float*)m1,
                       (const float*)m2, w.G, w.ub);
    // the dense part of the input gradient, h2 G + ub: split-bfloat16 products by default, PNGPD_TRAIN_DGRAD=f32 keeps it
    // on the float32 matrix instruction (read at every call, like the switches of t_conv3_fwd)
    const char* dgrad_env = getenv("PNGPD_TRAIN_DGRAD");
    hipError_t e = dgrad_env && dgrad_env[0] == 'f' ? launch_tall_gemm(h2, w.G, 1, w.ub, dh2, M, C2, C2, 0, nullptr, s)
                                                    : launch_dense_dgrad_x3(h2, w.G, w.ub, dh2, M, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_conv3_sparse_dgrad, dim3(B), dim3(1024), 0, s, W3, gamma, rstd, (const float*)w.gmask, arg, P,
                       dh2);
    // bn2_part given: the launch also leaves the partial column sums of the backward of the BatchNorm in front of
    // conv3 there ([t_colreduce_blocks(M)][2][128] doubles, for the finalize step of t_bn_bwd)
    if (bn2_part != nullptr && (!bn2_y || !bn2_mean || !bn2_rstd || !bn2_gamma || !bn2_beta)) return hipErrorInvalidValue;
    const int sum_blocks = bn2_part != nullptr ? t_colreduce_blocks(M) : 0;
    hipLaunchKernelGGL(k_conv3_wgrad_bn2sums, dim3(C3 / 2 + sum_blocks), dim3(256), 0, s, h2, W3, gamma, rstd,
                       (const float*)m1, (const float*)m2, (const float*)w.gmask, arg, (const double*)w.cov,
                       (const double*)w.abar_d, B, P, dW3, (const float*)dh2, bn2_y, bn2_mean, bn2_rstd, bn2_gamma, bn2_beta,
                       t_colreduce_rows(M), bn2_part);
    return hipGetLastError();
}

}  // namespace pngpd

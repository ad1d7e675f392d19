// Training step kernels (train-mode BatchNorm, backward, Adam) for the path of
// main_1v.py:262-274: output = model(data); nll_loss; backward; Adam(lr, betas .9/.999, eps 1e-8).
//
// Layout: activations are "rows" matrices [M, C] float32, M = B*P rows ordered (candidate, point)
// for the per-point layers and M = B for the fully connected layers.  All matrix products run on
// the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) in three shapes:
//     NT  Y[M,N]  = X[M,K]  W[N,K]^T + b      forward            (fc_f32_kernel in fc.hip)
//     NN  dX[M,K] = dY[M,N] W[N,K]            input gradient     (gemm_nn_block)
//     TN  dW[N,K] = dY[M,N]^T H[M,K]          weight gradient    (gemm_tn_block, split over M)
// (the two of one layer share a launch, gemm_tn_nn_kernel)
// BatchNorm uses batch statistics over the M rows (biased variance for the normalisation,
// unbiased for running_var, momentum 0.1, eps 1e-5: torch.nn.BatchNorm1d defaults, which the
// reference keeps, model/pointnet.py:21-25).  Its backward is
//     dy = gamma*rstd*(g - mean(g) - yhat*mean(g*yhat)),   g = dh*(h > 0)
// The 1024-channel layer with its BatchNorm and max-pool never materialises its [M,1024] output
// (train_conv3.hip).  The bias of a layer that feeds a BatchNorm has an exactly zero gradient
// (sum_m dy = 0) and is written as 0.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "train_act.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr float BN_EPS = 1e-5f;

// ---------------------------------------------------------------------------------------------
// points -> rows, first layer
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// conv1 + train-mode BatchNorm + ReLU in one pass over the points (model/pointnet.py:30, :143).
// y1 = W1 x + b1 is linear in the 3 coordinates, so its batch statistics follow from the first and second
// moments of x:  mean_c = w_c . xbar + b_c,  var_c = w_c^T Cov(x) w_c  (double) - no pass over [M,64] for them.
// ---------------------------------------------------------------------------------------------
constexpr int X3M_ROWS = 1024;      // rows per block of the moment sums

// X3[m, c] = sum_j x[b, j, p] * T[b, j, c]   (T = identity when trans == nullptr); m = b*P + p, and the moment sums
// of the rows: part[blk][9] = sums over the block's X3M_ROWS rows of x, y, z, xx, xy, xz, yy, yz, zz (double)
__global__ __launch_bounds__(256) void k_points_rows_moments(const float* __restrict__ x, const float* __restrict__ trans,
                                                             float* __restrict__ X3, int B, int P,
                                                             double* __restrict__ part) {
    __shared__ double sh[4][9];
    const int M = B * P;
    const int r0 = blockIdx.x * X3M_ROWS, r1 = min(M, r0 + X3M_ROWS);
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int m = r0 + threadIdx.x; m < r1; m += 256) {
        const int b = m / P, p = m - b * P;
        const float* xb = x + (size_t)b * 3 * P;
        const float x0 = xb[p], x1 = xb[P + p], x2 = xb[2 * P + p];
        float o0 = x0, o1 = x1, o2 = x2;
        if (trans != nullptr) {
            const float* T = trans + (size_t)b * 9;
            o0 = x0 * T[0] + x1 * T[3] + x2 * T[6];
            o1 = x0 * T[1] + x1 * T[4] + x2 * T[7];
            o2 = x0 * T[2] + x1 * T[5] + x2 * T[8];
        }
        X3[(size_t)m * 3 + 0] = o0;
        X3[(size_t)m * 3 + 1] = o1;
        X3[(size_t)m * 3 + 2] = o2;
        const double dx = o0, dy = o1, dz = o2;
        a[0] += dx; a[1] += dy; a[2] += dz;
        a[3] += dx * dx; a[4] += dx * dy; a[5] += dx * dz;
        a[6] += dy * dy; a[7] += dy * dz; a[8] += dz * dz;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a[k] += __shfl_xor(a[k], off);
        if (lane == 0) sh[wave][k] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 9)
        part[(size_t)blockIdx.x * 9 + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// one block of 576 threads: wave k adds up moment k over the blocks (fixed order), then 64 threads finish the
// channels: mean, rstd (biased variance, eps 1e-5) and the running update (momentum, unbiased variance)
__global__ __launch_bounds__(576) void k_conv1_stats(const double* __restrict__ part, int nblk, int M,
                                                     const float* __restrict__ W1, const float* __restrict__ b1,
                                                     float momentum, float* __restrict__ mean,
                                                     float* __restrict__ rstd, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var) {
    __shared__ double mom[9];
    const int lane = threadIdx.x & 63, k = threadIdx.x >> 6;
    double a = 0.0;
    for (int b = lane; b < nblk; b += 64) a += part[(size_t)b * 9 + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) mom[k] = a / M;
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int c = threadIdx.x;
    const double mx = mom[0], my = mom[1], mz = mom[2];
    const double cxx = mom[3] - mx * mx, cxy = mom[4] - mx * my, cxz = mom[5] - mx * mz;
    const double cyy = mom[6] - my * my, cyz = mom[7] - my * mz, czz = mom[8] - mz * mz;
    const double wx = W1[3 * c], wy = W1[3 * c + 1], wz = W1[3 * c + 2];
    const double mu = wx * mx + wy * my + wz * mz + (double)b1[c];
    double var = wx * (cxx * wx + cxy * wy + cxz * wz) + wy * (cxy * wx + cyy * wy + cyz * wz) +
                 wz * (cxz * wx + cyz * wy + czz * wz);
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    if (running_mean != nullptr) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// h1 = relu((y1 - mean)*rstd*gamma + beta) with y1[m, c] = W1[c,:] . X3[m,:] + b1[c] (64 channels), which is not
// stored: the backward pass recomputes it from the points; 4 rows per 256-thread block
__global__ __launch_bounds__(256) void k_conv1_bn_fwd(const float* __restrict__ X3, const float* __restrict__ W1,
                                                      const float* __restrict__ b1, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float* __restrict__ h1,
                                                      int M) {
    const int c = threadIdx.x & 63;
    const float w0 = W1[3 * c], w1 = W1[3 * c + 1], w2 = W1[3 * c + 2], bb = b1[c];
    const float mu = mean[c], rs = rstd[c], ga = gamma[c], be = beta[c];
    for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += gridDim.x * 4) {
        const float* xr = X3 + (size_t)m * 3;
        const float y = conv1_pre(w0, w1, w2, bb, xr[0], xr[1], xr[2]);
        h1[(size_t)m * 64 + c] = fmaxf(bn_pre(y, mu, rs, ga, be), 0.f);
    }
}

// ---------------------------------------------------------------------------------------------
// column reductions over the rows of [M, C] matrices (two-stage, double accumulation of partials)
//   mode 0 (statistics): s1 = sum(a - a0), s2 = sum((a - a0)^2), a0 = A[0][c] (shift against cancellation)
//   mode 1 (bn backward): s1 = sum(g), s2 = sum(g*yhat); yhat = (Y-mean)*rstd; g = A masked by the ReLU that
//                         follows the BatchNorm: (H > 0) with the stored activation H, or, with gamma/beta given,
//                         the sign of the recomputed pre-activation (H is then not read)
// part: [nblk][2][C] doubles
// ---------------------------------------------------------------------------------------------
constexpr int RED_ROWS = 256;   // rows per block

// block = 256 threads = QB column quads (float4) x RL row lanes, QB = min(C/4, 64); grid = (ceil(C/4/QB), ceil(M/RED_ROWS))
__global__ __launch_bounds__(256) void k_colreduce(const float* __restrict__ A, const float* __restrict__ H,
                                                   const float* __restrict__ Y, const float* __restrict__ mean,
                                                   const float* __restrict__ rstd, int M, int C, int mode,
                                                   int rows_per_block, double* __restrict__ part,
                                                   const float* __restrict__ X3, const float* __restrict__ W1,
                                                   const float* __restrict__ b1, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta) {
    // X3 != nullptr (mode 1, C = 64): Y is conv1's output and is recomputed from the three coordinates of the row
    // (y = W1 x + b1) instead of being read: 12 bytes per row instead of 256
    __shared__ double red[2][256 * 4];       // [s1|s2][row lane][column of the block]
    const int cq = C >> 2;
    const int QB = cq < 64 ? cq : 64, RL = 256 / QB;
    const int ql = threadIdx.x % QB, rl = threadIdx.x / QB;
    const int q = blockIdx.x * QB + ql;      // column quad
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
    if (q < cq && rl < RL) {
        const size_t col = (size_t)q * 4;
        if (mode == 0) {
            const float4 a0 = *(const float4*)(A + col);
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += RL) {
                const float4 a = *(const float4*)(A + (size_t)r * C + col);
                const float dx = a.x - a0.x, dy = a.y - a0.y, dz = a.z - a0.z, dw = a.w - a0.w;
                s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
                s2.x += dx * dx; s2.y += dy * dy; s2.z += dz * dz; s2.w += dw * dw;
            }
        } else {
            const float4 mu = *(const float4*)(mean + col), rs = *(const float4*)(rstd + col);
            const bool remask = gamma != nullptr;         // gamma / beta are caller tensors: scalar loads
            const float4 ga = remask ? make_float4(gamma[col], gamma[col + 1], gamma[col + 2], gamma[col + 3]) : mu;
            const float4 be = remask ? make_float4(beta[col], beta[col + 1], beta[col + 2], beta[col + 3]) : mu;
            float wq[4][3] = {}, bq[4] = {};
            if (X3 != nullptr)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    wq[e][0] = W1[(col + e) * 3];
                    wq[e][1] = W1[(col + e) * 3 + 1];
                    wq[e][2] = W1[(col + e) * 3 + 2];
                    bq[e] = b1[col + e];
                }
#pragma unroll 4
            for (int r = r0 + rl; r < r1; r += RL) {
                float4 g = *(const float4*)(A + (size_t)r * C + col);
                float4 y;
                if (X3 != nullptr) {
                    const float x0 = X3[(size_t)r * 3], x1 = X3[(size_t)r * 3 + 1], x2 = X3[(size_t)r * 3 + 2];
                    y.x = conv1_pre(wq[0][0], wq[0][1], wq[0][2], bq[0], x0, x1, x2);
                    y.y = conv1_pre(wq[1][0], wq[1][1], wq[1][2], bq[1], x0, x1, x2);
                    y.z = conv1_pre(wq[2][0], wq[2][1], wq[2][2], bq[2], x0, x1, x2);
                    y.w = conv1_pre(wq[3][0], wq[3][1], wq[3][2], bq[3], x0, x1, x2);
                } else {
                    y = *(const float4*)(Y + (size_t)r * C + col);
                }
                if (remask) {
                    if (!(bn_pre(y.x, mu.x, rs.x, ga.x, be.x) > 0.f)) g.x = 0.f;
                    if (!(bn_pre(y.y, mu.y, rs.y, ga.y, be.y) > 0.f)) g.y = 0.f;
                    if (!(bn_pre(y.z, mu.z, rs.z, ga.z, be.z) > 0.f)) g.z = 0.f;
                    if (!(bn_pre(y.w, mu.w, rs.w, ga.w, be.w) > 0.f)) g.w = 0.f;
                } else if (H != nullptr) {
                    const float4 hv = *(const float4*)(H + (size_t)r * C + col);
                    if (!(hv.x > 0.f)) g.x = 0.f;
                    if (!(hv.y > 0.f)) g.y = 0.f;
                    if (!(hv.z > 0.f)) g.z = 0.f;
                    if (!(hv.w > 0.f)) g.w = 0.f;
                }
                s1.x += g.x; s1.y += g.y; s1.z += g.z; s1.w += g.w;
                s2.x += g.x * ((y.x - mu.x) * rs.x);
                s2.y += g.y * ((y.y - mu.y) * rs.y);
                s2.z += g.z * ((y.z - mu.z) * rs.z);
                s2.w += g.w * ((y.w - mu.w) * rs.w);
            }
        }
    }
    if (rl < RL) {
        double* d1 = red[0] + (rl * QB + ql) * 4;
        double* d2 = red[1] + (rl * QB + ql) * 4;
        d1[0] = (double)s1.x; d1[1] = (double)s1.y; d1[2] = (double)s1.z; d1[3] = (double)s1.w;
        d2[0] = (double)s2.x; d2[1] = (double)s2.y; d2[2] = (double)s2.z; d2[3] = (double)s2.w;
    }
    __syncthreads();
    const int ncol = QB * 4;
    for (int t = threadIdx.x; t < ncol; t += 256) {
        const int c = blockIdx.x * ncol + t;
        if (c >= C) continue;
        double a = 0.0, b = 0.0;
        for (int k = 0; k < RL; ++k) {          // fixed order: reproducible
            a += red[0][k * ncol + t];
            b += red[1][k * ncol + t];
        }
        part[((size_t)blockIdx.y * 2 + 0) * C + c] = a;
        part[((size_t)blockIdx.y * 2 + 1) * C + c] = b;
    }
}

// The finalize kernels sum nblk partials per column.  Block = 4 columns x 64 row lanes: every row
// lane sums every 64th partial, the 64 lane sums are then added in a fixed order (reproducible).
constexpr int FIN_COLS = 4, FIN_LANES = 64;

// v summed over the 64 row lanes of a column (thread = row lane * FIN_COLS + column): a butterfly over the 16 row
// lanes a wave holds, then the four wave totals in a fixed order.  The result is valid in row lane 0.
__device__ __forceinline__ double fin_row_lane_sum(double v, double (*sh)[FIN_COLS]) {
#pragma unroll
    for (int off = FIN_COLS; off < 64; off <<= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < FIN_COLS) sh[wave][lane] = v;
    __syncthreads();
    const int cl = threadIdx.x & (FIN_COLS - 1);
    return ((sh[0][cl] + sh[1][cl]) + sh[2][cl]) + sh[3][cl];
}

__device__ __forceinline__ void sum_partials2(const double* __restrict__ part, int nblk, int C, int c, int rl,
                                              double& s1, double& s2) {
    __shared__ double sh[2][4][FIN_COLS];
    double a = 0.0, b = 0.0;
    if (c < C)
#pragma unroll 4
        for (int i = rl; i < nblk; i += FIN_LANES) {
            a += part[((size_t)i * 2 + 0) * C + c];
            b += part[((size_t)i * 2 + 1) * C + c];
        }
    s1 = fin_row_lane_sum(a, sh[0]);
    s2 = fin_row_lane_sum(b, sh[1]);
}

// Finalise batch statistics: mean, rstd; update running stats (momentum, unbiased variance).
__global__ __launch_bounds__(256) void k_stats_finalize(const double* __restrict__ part, int nblk,
                                                        const float* __restrict__ A, int M, int C, float momentum,
                                                        float* __restrict__ mean, float* __restrict__ rstd,
                                                        float* __restrict__ running_mean,
                                                        float* __restrict__ running_var) {
    const int c = blockIdx.x * FIN_COLS + (threadIdx.x & (FIN_COLS - 1)), rl = threadIdx.x / FIN_COLS;
    double s1, s2;
    sum_partials2(part, nblk, C, c, rl, s1, s2);
    if (rl != 0 || c >= C) return;
    const double dm = s1 / M;                      // mean of the shifted data
    const double mu = (double)A[c] + dm;
    double var = s2 / M - dm * dm;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    if (running_mean != nullptr) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// m1 = s1/M, m2 = s2/M; dgamma = s2, dbeta = s1; the bias of the layer in front of the BatchNorm gets its exact 0
__device__ __forceinline__ void bnbwd_finalize_block(const double* __restrict__ part, int nblk, int M, int C,
                                                     float* __restrict__ m1, float* __restrict__ m2,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                     float* __restrict__ dbias_zero, int blk) {
    const int c = blk * FIN_COLS + (threadIdx.x & (FIN_COLS - 1)), rl = threadIdx.x / FIN_COLS;
    double s1, s2;
    sum_partials2(part, nblk, C, c, rl, s1, s2);
    if (rl != 0 || c >= C) return;
    m1[c] = (float)(s1 / M);
    m2[c] = (float)(s2 / M);
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    if (dbias_zero != nullptr) dbias_zero[c] = 0.f;
}

__global__ __launch_bounds__(256) void k_bnbwd_finalize(const double* __restrict__ part, int nblk, int M, int C,
                                                        float* __restrict__ m1, float* __restrict__ m2,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                        float* __restrict__ dbias_zero) {
    bnbwd_finalize_block(part, nblk, M, C, m1, m2, dgamma, dbeta, dbias_zero, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm of a matrix of few rows (the FC layers: one row per candidate) in ONE kernel per direction.
// A workgroup owns 16 columns and all M rows (16 row lanes), so the column sums stay inside it: no partial
// buffers, no finalize kernel, no separate normalisation pass - at this size each of those was a launch of its own
// that cost more than its work.  Sums in double, added in a fixed order.
// ---------------------------------------------------------------------------------------------
constexpr int FEW_ROWS_MAX = 1024;

__device__ __forceinline__ double few_rows_colsum(double v, double (*sh)[16], int cl, int rl) {
    __syncthreads();
    sh[rl][cl] = v;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += sh[k][cl];
    return t;
}

// forward: batch statistics of z, a = relu((z - mean)*rstd*gamma + beta); writes a, mean, rstd and the running
// update.  RPT rows per thread (M <= 16*RPT) live in registers: every load is issued before the first use.
template <int RPT>
__global__ __launch_bounds__(256) void k_few_rows_bn_fwd(const float* __restrict__ z, int M, int C,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, float momentum,
                                                         float* __restrict__ a, float* __restrict__ mean,
                                                         float* __restrict__ rstd, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var) {
    __shared__ double sh[16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float v[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = rl + 16 * j;
        v[j] = r < M ? z[(size_t)r * C + c] : 0.f;
    }
    double s1 = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (rl + 16 * j < M) s1 += (double)v[j];
    const double mu = few_rows_colsum(s1, sh, cl, rl) / M;
    double s2 = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (rl + 16 * j < M) {
            const double d = (double)v[j] - mu;
            s2 += d * d;
        }
    const double var = few_rows_colsum(s2, sh, cl, rl) / M;
    const float mf = (float)mu, rs = (float)(1.0 / sqrt(var + (double)BN_EPS));
    const float ga = gamma[c], be = beta[c];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = rl + 16 * j;
        if (r < M) a[(size_t)r * C + c] = fmaxf((v[j] - mf) * rs * ga + be, 0.f);
    }
    if (rl != 0) return;
    mean[c] = mf;
    rstd[c] = rs;
    if (running_mean != nullptr) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mu);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

// backward, in place over dh: g = dh*(h > 0), dh <- gamma*rstd*(g - mean(g) - yhat*mean(g*yhat)); dgamma, dbeta,
// and the exact 0 of the bias in front of the BatchNorm
template <int RPT>
__global__ __launch_bounds__(256) void k_few_rows_bn_bwd(float* __restrict__ dh, const float* __restrict__ h,
                                                         const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const float* __restrict__ gamma, int M, int C,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                         float* __restrict__ dbias_zero) {
    __shared__ double sh[16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const float mu = mean[c], rs = rstd[c];
    float g[RPT], yh[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = rl + 16 * j;
        const size_t i = (size_t)(r < M ? r : 0) * C + c;
        const float gv = dh[i], hv = h[i], yv = y[i];
        g[j] = (r < M && hv > 0.f) ? gv : 0.f;
        yh[j] = (yv - mu) * rs;
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (rl + 16 * j < M) {
            s1 += (double)g[j];
            s2 += (double)(g[j] * yh[j]);
        }
    s1 = few_rows_colsum(s1, sh, cl, rl);
    s2 = few_rows_colsum(s2, sh, cl, rl);
    const float m1 = (float)(s1 / M), m2 = (float)(s2 / M), gr = gamma[c] * rs;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = rl + 16 * j;
        if (r < M) dh[(size_t)r * C + c] = gr * (g[j] - m1 - yh[j] * m2);
    }
    if (rl != 0) return;
    dgamma[c] = (float)s2;
    dbeta[c] = (float)s1;
    if (dbias_zero != nullptr) dbias_zero[c] = 0.f;
}

// h = relu((y - mean)*rstd*gamma + beta)   (relu optional); four consecutive channels per thread (C % 4 == 0)
__global__ __launch_bounds__(256) void k_bn_act_fwd(const float* __restrict__ y, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ h, size_t n4,
                                                    int C, int relu) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % (size_t)C);
    const float4 yv = ((const float4*)y)[i];
    const float4 mv = *(const float4*)(mean + c), rv = *(const float4*)(rstd + c);
    // gamma / beta are caller tensors (views of a flat parameter bucket need not be 16-byte aligned): scalar loads
    const float4 gv = make_float4(gamma[c], gamma[c + 1], gamma[c + 2], gamma[c + 3]);
    const float4 bv = make_float4(beta[c], beta[c + 1], beta[c + 2], beta[c + 3]);
    float4 v;
    v.x = (yv.x - mv.x) * rv.x * gv.x + bv.x;
    v.y = (yv.y - mv.y) * rv.y * gv.y + bv.y;
    v.z = (yv.z - mv.z) * rv.z * gv.z + bv.z;
    v.w = (yv.w - mv.w) * rv.w * gv.w + bv.w;
    if (relu) {
        v.x = fmaxf(v.x, 0.f);
        v.y = fmaxf(v.y, 0.f);
        v.z = fmaxf(v.z, 0.f);
        v.w = fmaxf(v.w, 0.f);
    }
    ((float4*)h)[i] = v;
}

// dense BN backward, in place over dh:  dh <- gamma*rstd*(g - m1 - yhat*m2), g = dh*(h>0); four channels per thread
__global__ __launch_bounds__(256) void k_bn_bwd_dense(float* __restrict__ dh, const float* __restrict__ h,
                                                      const float* __restrict__ y, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                      const float* __restrict__ m1, const float* __restrict__ m2,
                                                      size_t n4, int C, const float* __restrict__ beta) {
    // beta != nullptr: the ReLU mask is the sign of the recomputed pre-activation and h is not read
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % (size_t)C);
    float4 gv = ((const float4*)dh)[i];
    const float4 yv = ((const float4*)y)[i];
    const float4 mv = *(const float4*)(mean + c), rv = *(const float4*)(rstd + c);
    const float4 ga = make_float4(gamma[c], gamma[c + 1], gamma[c + 2], gamma[c + 3]);     // caller tensor: scalar loads
    if (beta != nullptr) {
        if (!(bn_pre(yv.x, mv.x, rv.x, ga.x, beta[c]) > 0.f)) gv.x = 0.f;
        if (!(bn_pre(yv.y, mv.y, rv.y, ga.y, beta[c + 1]) > 0.f)) gv.y = 0.f;
        if (!(bn_pre(yv.z, mv.z, rv.z, ga.z, beta[c + 2]) > 0.f)) gv.z = 0.f;
        if (!(bn_pre(yv.w, mv.w, rv.w, ga.w, beta[c + 3]) > 0.f)) gv.w = 0.f;
    } else if (h != nullptr) {
        const float4 hv = ((const float4*)h)[i];
        if (!(hv.x > 0.f)) gv.x = 0.f;
        if (!(hv.y > 0.f)) gv.y = 0.f;
        if (!(hv.z > 0.f)) gv.z = 0.f;
        if (!(hv.w > 0.f)) gv.w = 0.f;
    }
    const float4 a1 = *(const float4*)(m1 + c), a2 = *(const float4*)(m2 + c);
    float4 o;
    o.x = ga.x * rv.x * (gv.x - a1.x - (yv.x - mv.x) * rv.x * a2.x);
    o.y = ga.y * rv.y * (gv.y - a1.y - (yv.y - mv.y) * rv.y * a2.y);
    o.z = ga.z * rv.z * (gv.z - a1.z - (yv.z - mv.z) * rv.z * a2.z);
    o.w = ga.w * rv.w * (gv.w - a1.w - (yv.w - mv.w) * rv.w * a2.w);
    ((float4*)dh)[i] = o;
}

// k_bn_bwd_dense for conv1's BatchNorm (C = 64) with y = W1 x + b1 recomputed from the row's coordinates.
// block = 16 column quads x 16 row lanes over 256 rows: a thread keeps the weights of its four channels.
__global__ __launch_bounds__(256) void k_conv1_bn_bwd_dense(float* __restrict__ dh, const float* __restrict__ h,
                                                            const float* __restrict__ X3,
                                                            const float* __restrict__ W1,
                                                            const float* __restrict__ b1,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ m1,
                                                            const float* __restrict__ m2, int M,
                                                            const float* __restrict__ beta) {
    // the ReLU mask is the sign of the recomputed pre-activation: h is not read
    const int col = (threadIdx.x & 15) * 4, rl = threadIdx.x >> 4;
    float wq[4][3], bq[4], mu[4], rs[4], gr[4], a1[4], a2[4], ga[4], be[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wq[e][0] = W1[(col + e) * 3];
        wq[e][1] = W1[(col + e) * 3 + 1];
        wq[e][2] = W1[(col + e) * 3 + 2];
        bq[e] = b1[col + e];
        mu[e] = mean[col + e];
        rs[e] = rstd[col + e];
        ga[e] = gamma[col + e];
        be[e] = beta[col + e];
        gr[e] = ga[e] * rs[e];
        a1[e] = m1[col + e];
        a2[e] = m2[col + e];
    }
    const int r0 = blockIdx.x * 256;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + rl + 16 * j;
        if (r >= M) break;
        const float x0 = X3[(size_t)r * 3], x1 = X3[(size_t)r * 3 + 1], x2 = X3[(size_t)r * 3 + 2];
        const float4 gv = *(const float4*)(dh + (size_t)r * 64 + col);
        const float gin[4] = {gv.x, gv.y, gv.z, gv.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = conv1_pre(wq[e][0], wq[e][1], wq[e][2], bq[e], x0, x1, x2);
            const float g = bn_pre(y, mu[e], rs[e], ga[e], be[e]) > 0.f ? gin[e] : 0.f;
            o[e] = gr[e] * (g - a1[e] - (y - mu[e]) * rs[e] * a2[e]);
        }
        *(float4*)(dh + (size_t)r * 64 + col) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Everything behind conv1's BatchNorm in one pass over dh1: dy1 as in k_conv1_bn_bwd_dense, used where it is made and
// never stored - partial sums of dW1[c,j] = sum_m dy1[m,c] X3[m,j] (part [gridDim.x][64*3] doubles, for
// k_reduce_double) and, when dX3 is given, dX3[m,j] = sum_c dy1[m,c] W1[c,j].
// block = 16 column quads x 16 row lanes over 256 rows.
__device__ __forceinline__ void conv1_bwd_all_block(const float* __restrict__ dh, const float* __restrict__ X3,
                                                    const float* __restrict__ W1, const float* __restrict__ b1,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ m1, const float* __restrict__ m2, int M,
                                                    double* __restrict__ part, float* __restrict__ dX3, int blk) {
    __shared__ float sw[16][64][3];
    const int cq = threadIdx.x & 15, col = cq * 4, rl = threadIdx.x >> 4;
    float wq[4][3], bq[4], mu[4], rs[4], gr[4], a1[4], a2[4], ga[4], be[4], acc[4][3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        wq[e][0] = W1[(col + e) * 3];
        wq[e][1] = W1[(col + e) * 3 + 1];
        wq[e][2] = W1[(col + e) * 3 + 2];
        bq[e] = b1[col + e];
        mu[e] = mean[col + e];
        rs[e] = rstd[col + e];
        ga[e] = gamma[col + e];
        be[e] = beta[col + e];
        gr[e] = ga[e] * rs[e];
        a1[e] = m1[col + e];
        a2[e] = m2[col + e];
        acc[e][0] = acc[e][1] = acc[e][2] = 0.f;
    }
    const int r0 = blk * 256;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + rl + 16 * j;
        const bool ok = r < M;                      // uniform over the 16 lanes that share a row
        const int rr = ok ? r : M - 1;
        const float x0 = X3[(size_t)rr * 3], x1 = X3[(size_t)rr * 3 + 1], x2 = X3[(size_t)rr * 3 + 2];
        const float4 gv = *(const float4*)(dh + (size_t)rr * 64 + col);
        const float gin[4] = {gv.x, gv.y, gv.z, gv.w};
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = conv1_pre(wq[e][0], wq[e][1], wq[e][2], bq[e], x0, x1, x2);
            const float g = bn_pre(y, mu[e], rs[e], ga[e], be[e]) > 0.f ? gin[e] : 0.f;
            const float o = ok ? gr[e] * (g - a1[e] - (y - mu[e]) * rs[e] * a2[e]) : 0.f;
            acc[e][0] += o * x0;
            acc[e][1] += o * x1;
            acc[e][2] += o * x2;
            d0 += o * wq[e][0];
            d1 += o * wq[e][1];
            d2 += o * wq[e][2];
        }
        if (dX3 != nullptr) {
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                d0 += __shfl_xor(d0, off);
                d1 += __shfl_xor(d1, off);
                d2 += __shfl_xor(d2, off);
            }
            if (cq == 0 && ok) {
                dX3[(size_t)r * 3 + 0] = d0;
                dX3[(size_t)r * 3 + 1] = d1;
                dX3[(size_t)r * 3 + 2] = d2;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 3; ++j) sw[rl][col + e][j] = acc[e][j];
    __syncthreads();
    if (threadIdx.x < 192) {
        const int c = threadIdx.x / 3, j = threadIdx.x - 3 * c;
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += (double)sw[k][c][j];      // fixed order: reproducible
        part[(size_t)blk * 192 + threadIdx.x] = t;
    }
}

__device__ __forceinline__ void reduce_chunks_block(const float* __restrict__ part, int chunks, size_t n,
                                                    float* __restrict__ out, int blk);      // defined below

// workgroups [0, n_rows) as above; the rest add up the chunks of another product's partial sums (reduce_chunks_block:
// the weight gradient of the layer behind, which this pass does not depend on) - one launch instead of two
__global__ __launch_bounds__(256) void k_conv1_bwd_all(const float* __restrict__ dh, const float* __restrict__ X3,
                                                       const float* __restrict__ W1, const float* __restrict__ b1,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ m1, const float* __restrict__ m2, int M,
                                                       double* __restrict__ part, float* __restrict__ dX3, int n_rows,
                                                       const float* __restrict__ chunk_part, int chunks, size_t chunk_n,
                                                       float* __restrict__ chunk_out) {
    if ((int)blockIdx.x < n_rows)
        conv1_bwd_all_block(dh, X3, W1, b1, mean, rstd, gamma, beta, m1, m2, M, part, dX3, blockIdx.x);
    else
        reduce_chunks_block(chunk_part, chunks, chunk_n, chunk_out, blockIdx.x - n_rows);
}

// ---------------------------------------------------------------------------------------------
// exact-f32 MFMA GEMMs for the backward pass
// ---------------------------------------------------------------------------------------------
constexpr int GB_M = 64, GB_N = 64, GB_K = 32;

// dX[M,Kout] = dY[M,N] W[N,Kout];  Kout % 64 == 0, N % 32 == 0.
// optional fused mask: dX *= (Hmask > 0) is NOT applied here (BN backward needs the unmasked sums).
// (bx, by, bz) of (gx, gy, gz): the workgroup's place in the product's grid (Kout/64, ceil(M/64), slices)
__device__ __forceinline__ void gemm_nn_block(const float* __restrict__ dY, const float* __restrict__ W,
                                              float* __restrict__ dX, int M, int N, int Kout, float* __restrict__ part,
                                              int bx, int by, int bz, int gz) {
    __shared__ float sA[GB_M * (GB_K + 1)];
    __shared__ float sB[GB_K * GB_N];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int m0 = by * GB_M, k0 = bx * GB_N;
    const int arow = tid >> 3, acol = (tid & 7) * 4;        // A tile 64x32: 2 float4 per thread
    const int brow = tid >> 4, bcol = (tid & 15) * 4;       // B tile 32x64: 2 float4 per thread
    f32x16 acc = {0};
    float4 va[2], vb[2];
    auto load_tile = [&](int n0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gm = m0 + arow + 32 * i;
            va[i] = gm < M ? *(const float4*)(dY + (size_t)gm * N + n0 + acol) : make_float4(0, 0, 0, 0);
            vb[i] = *(const float4*)(W + (size_t)(n0 + brow + 16 * i) * Kout + k0 + bcol);
        }
    };
    // split over the summed dimension: slice blockIdx.z of gridDim.z writes a partial product (launch_splitk_reduce)
    const int n_len = N / gz, n_begin = bz * n_len, n_end = n_begin + n_len;
    load_tile(n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += GB_K) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* pa = sA + (arow + 32 * i) * (GB_K + 1) + acol;
            pa[0] = va[i].x; pa[1] = va[i].y; pa[2] = va[i].z; pa[3] = va[i].w;
            *(float4*)(sB + (brow + 16 * i) * GB_N + bcol) = vb[i];
        }
        __syncthreads();
        if (n0 + GB_K < n_end) load_tile(n0 + GB_K);       // in flight during this tile's MFMAs
        const float* ra = sA + (wm * 32 + r) * (GB_K + 1) + h;
        const float* rb = sB + h * GB_N + wn * 32 + r;
#pragma unroll
        for (int s = 0; s < GB_K / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[2 * s], rb[2 * s * GB_N], acc, 0, 0, 0);
    }
    const int col = k0 + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = m0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (row < M) (part != nullptr ? part + (size_t)bz * M * Kout : dX)[(size_t)row * Kout + col] = acc[i];
    }
}

// partial dW[chunk][N,K] = dY[rows of chunk, N]^T H[rows of chunk, K];  N % 64 == 0, K % 64 == 0
// grid (K/64, N/64, chunks); rows_per_chunk % 32 == 0
__device__ __forceinline__ void gemm_tn_block(const float* __restrict__ dY, const float* __restrict__ Hm,
                                              float* __restrict__ part, int M, int N, int K, int rows_per_chunk, int bx,
                                              int by, int bz) {
    __shared__ float sA[GB_K * GB_M];   // [m][n] 32 x 64
    __shared__ float sB[GB_K * GB_N];   // [m][k] 32 x 64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    const int k0 = bx * 64, n0 = by * 64;
    const int row_begin = bz * rows_per_chunk;
    const int row_end = min(M, row_begin + rows_per_chunk);
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;        // 32x64 tile: 2 float4 per thread
    f32x16 acc = {0};
    float4 va[2], vb[2];
    auto load_tile = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int gm = mb + lrow + 16 * i;
            const bool ok = gm < row_end;
            va[i] = ok ? *(const float4*)(dY + (size_t)gm * N + n0 + lcol) : make_float4(0, 0, 0, 0);
            vb[i] = ok ? *(const float4*)(Hm + (size_t)gm * K + k0 + lcol) : make_float4(0, 0, 0, 0);
        }
    };
    if (row_begin < row_end) load_tile(row_begin);
    for (int mb = row_begin; mb < row_end; mb += GB_K) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *(float4*)(sA + (lrow + 16 * i) * GB_M + lcol) = va[i];
            *(float4*)(sB + (lrow + 16 * i) * GB_N + lcol) = vb[i];
        }
        __syncthreads();
        if (mb + GB_K < row_end) load_tile(mb + GB_K);  // in flight during this tile's MFMAs
        const float* ra = sA + h * GB_M + wm * 32 + r;      // A[i = n][kk = m]
        const float* rb = sB + h * GB_N + wn * 32 + r;      // B[kk = m][j = k]
#pragma unroll
        for (int s = 0; s < GB_K / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[2 * s * GB_M], rb[2 * s * GB_N], acc, 0, 0, 0);
    }
    float* out = part + (size_t)bz * N * K;
    const int col = k0 + wn * 32 + r;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = n0 + wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
        out[(size_t)row * K + col] = acc[i];
    }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ dY, const float* __restrict__ Hm,
                                                      float* __restrict__ part, int M, int N, int K,
                                                      int rows_per_chunk) {
    gemm_tn_block(dY, Hm, part, M, N, K, rows_per_chunk, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The weight-gradient product with the finalize step of an independent BatchNorm backward riding along: workgroups
// [0, n_fin) are k_bnbwd_finalize's, the rest the product's grid (K/64, N/64, chunks).
__global__ __launch_bounds__(256) void gemm_tn_finalize_kernel(const float* __restrict__ dY, const float* __restrict__ Hm,
                                                               float* __restrict__ tn_out, int M, int N, int K,
                                                               int rows_per_chunk, int n_fin,
                                                               const double* __restrict__ fin_part, int fin_nblk,
                                                               int fin_C, float* __restrict__ m1, float* __restrict__ m2,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ dbias_zero) {
    int b = blockIdx.x;
    if (b < n_fin) {
        bnbwd_finalize_block(fin_part, fin_nblk, M, fin_C, m1, m2, dgamma, dbeta, dbias_zero, b);
    } else {
        b -= n_fin;
        const int kx = K / 64;
        gemm_tn_block(dY, Hm, tn_out, M, N, K, rows_per_chunk, b % kx, (b / kx) % (N / 64), b / (kx * (N / 64)));
    }
}

// Both products of a linear layer's backward in one launch - they share dY and nothing else, and at a few hundred
// rows neither fills the chip: workgroups [0, n_tn) take the weight-gradient product's grid (K/64, N/64, chunks),
// the rest the input-gradient product's (K/64, ceil(M/64), slices).
__global__ __launch_bounds__(256) void gemm_tn_nn_kernel(const float* __restrict__ dY, const float* __restrict__ Hm,
                                                         float* __restrict__ tn_out, const float* __restrict__ W,
                                                         float* __restrict__ dX, float* __restrict__ nn_part, int M,
                                                         int N, int K, int rows_per_chunk, int n_tn, int nn_slices) {
    const int kx = K / 64;
    int b = blockIdx.x;
    if (b < n_tn) {
        const int bx = b % kx, by = (b / kx) % (N / 64), bz = b / (kx * (N / 64));
        gemm_tn_block(dY, Hm, tn_out, M, N, K, rows_per_chunk, bx, by, bz);
    } else {
        b -= n_tn;
        const int my = (M + 63) / 64;
        const int bx = b % kx, by = (b / kx) % my, bz = b / (kx * my);
        gemm_nn_block(dY, W, dX, M, N, K, nn_part, bx, by, bz, nn_slices);
    }
}

// out[i] = sum over chunks of part[chunk][i].  block = 16 elements x 16 chunk lanes: a lane sums every 16th
// chunk, the lane sums are added in a fixed order (reproducible).
__device__ __forceinline__ void reduce_chunks_block(const float* __restrict__ part, int chunks, size_t n,
                                                    float* __restrict__ out, int blk) {
    __shared__ float sh[16][16];
    const int el = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const size_t i = (size_t)blk * 16 + el;
    float s = 0.f;
    if (i < n)
#pragma unroll 4
        for (int c = rl; c < chunks; c += 16) s += part[(size_t)c * n + i];
    sh[rl][el] = s;
    __syncthreads();
    if (rl != 0 || i >= n) return;
    s = 0.f;
    for (int k = 0; k < 16; ++k) s += sh[k][el];
    out[i] = s;
}

__global__ __launch_bounds__(256) void k_reduce_chunks(const float* __restrict__ part, int chunks, size_t n,
                                                       float* __restrict__ out) {
    reduce_chunks_block(part, chunks, n, out, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------
// small layers: conv1 backward, the two 256 -> {9,k} layers, 3x3 transform gradient, loss
// ---------------------------------------------------------------------------------------------
// partial sums of dW1[c,j] = sum_m dy1[m,c]*X3[m,j] over RED_ROWS rows; part [nblk][64*3] doubles
__global__ __launch_bounds__(256) void k_conv1_wgrad(const float* __restrict__ dy1, const float* __restrict__ X3,
                                                     int M, double* __restrict__ part) {
    __shared__ double sp[4][64][3];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int r0 = blockIdx.x * RED_ROWS, r1 = min(M, r0 + RED_ROWS);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) {
        const float g = dy1[(size_t)r * 64 + cl];
        s0 += g * X3[(size_t)r * 3 + 0];
        s1 += g * X3[(size_t)r * 3 + 1];
        s2 += g * X3[(size_t)r * 3 + 2];
    }
    sp[rl][cl][0] = s0; sp[rl][cl][1] = s1; sp[rl][cl][2] = s2;
    __syncthreads();
    if (rl == 0)
        for (int j = 0; j < 3; ++j)
            part[(size_t)blockIdx.x * 192 + cl * 3 + j] = sp[0][cl][j] + sp[1][cl][j] + sp[2][cl][j] + sp[3][cl][j];
}

__global__ __launch_bounds__(256) void k_reduce_double(const double* __restrict__ part, int nblk, int n,
                                                       float* __restrict__ out) {
    __shared__ double sh[4][FIN_COLS];
    const int cl = threadIdx.x & (FIN_COLS - 1), rl = threadIdx.x / FIN_COLS;
    const int i = blockIdx.x * FIN_COLS + cl;
    double s = 0.0;
    if (i < n)
        for (int b = rl; b < nblk; b += FIN_LANES) s += part[(size_t)b * n + i];
    s = fin_row_lane_sum(s, sh);
    if (rl != 0 || i >= n) return;
    out[i] = (float)s;
}

// dX3[m, j] = sum_c dy1[m,c] * W1[c,j]: four lanes per row, 16 channels each (their W1 rows in registers), so a
// wave reads 4 KiB of consecutive memory and a row costs two shuffle steps instead of six
__global__ __launch_bounds__(256) void k_conv1_dgrad(const float* __restrict__ dy1, const float* __restrict__ W1,
                                                     float* __restrict__ dX3, int M) {
    const int cg = threadIdx.x & 3;
    float w[16][3];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) w[i][j] = W1[(16 * cg + i) * 3 + j];
    for (int m = blockIdx.x * 64 + (threadIdx.x >> 2); m < M; m += gridDim.x * 64) {
        const float4* src = (const float4*)(dy1 + (size_t)m * 64 + 16 * cg);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 g = src[q];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 += gv[e] * w[4 * q + e][0];
                a1 += gv[e] * w[4 * q + e][1];
                a2 += gv[e] * w[4 * q + e][2];
            }
        }
#pragma unroll
        for (int off = 2; off >= 1; off >>= 1) {
            a0 += __shfl_xor(a0, off);
            a1 += __shfl_xor(a1, off);
            a2 += __shfl_xor(a2, off);
        }
        if (cg == 0) {
            dX3[(size_t)m * 3 + 0] = a0;
            dX3[(size_t)m * 3 + 1] = a1;
            dX3[(size_t)m * 3 + 2] = a2;
        }
    }
}

// dtrans[b, j, c] (+)= sum_p x[b, j, p] * dX3[b*P + p, c]     (one block per candidate)
__global__ __launch_bounds__(256) void k_trans_grad(const float* __restrict__ x, const float* __restrict__ dX3,
                                                    int P, const float* __restrict__ dtrans_ext,
                                                    float* __restrict__ dtrans) {
    __shared__ float sp[4][9];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float a[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a[i] = 0.f;
    const float* xb = x + (size_t)b * 3 * P;
    for (int p = tid; p < P; p += 256) {
        const float* d = dX3 + ((size_t)b * P + p) * 3;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float xv = xb[j * P + p];
            a[3 * j + 0] += xv * d[0];
            a[3 * j + 1] += xv * d[1];
            a[3 * j + 2] += xv * d[2];
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) a[i] += __shfl_xor(a[i], off);
        if (lane == 0) sp[wave][i] = a[i];
    }
    __syncthreads();
    if (tid < 9) {
        float v = sp[0][tid] + sp[1][tid] + sp[2][tid] + sp[3][tid];
        if (dtrans_ext != nullptr) v += dtrans_ext[(size_t)b * 9 + tid];
        dtrans[(size_t)b * 9 + tid] = v;
    }
}

// tiny-output layer backward (N <= 16, K = 256):  dA[m,k] = sum_n dZ[m,n] W[n,k]
__device__ __forceinline__ void small_dgrad_block(const float* __restrict__ dZ, const float* __restrict__ W,
                                                  float* __restrict__ dA, int M, int N, int K, int block) {
    const size_t i = (size_t)block * 256 + threadIdx.x;
    if (i >= (size_t)M * K) return;
    const int k = (int)(i % K);
    const size_t m = i / K;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dZ[m * N + n] * W[(size_t)n * K + k];
    dA[i] = s;
}

// dW[n,k] = sum_m dZ[m,n] A[m,k];  db[n] = sum_m dZ[m,n]     (M = batch)
// block = 16 k x 16 row lanes, grid (K/16, N): every row lane sums every 16th row, the lane sums are added in a
// fixed order
__device__ __forceinline__ void small_wgrad_block(const float* __restrict__ dZ, const float* __restrict__ A,
                                                  float* __restrict__ dW, float* __restrict__ db, int M, int N, int K,
                                                  int bx, int n) {
    __shared__ float sw[16][16], sb[16];
    const int kl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int k = bx * 16 + kl;
    float s = 0.f, b = 0.f;
    if (k < K) {
#pragma unroll 4
        for (int m = rl; m < M; m += 16) {
            const float g = dZ[(size_t)m * N + n];
            s = fmaf(g, A[(size_t)m * K + k], s);
            b += g;
        }
    }
    sw[rl][kl] = s;
    if (kl == 0) sb[rl] = b;
    __syncthreads();
    if (rl != 0) return;
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sw[i][kl];
    if (k < K) dW[(size_t)n * K + k] = t;
    if (bx == 0 && kl == 0) {
        float u = 0.f;
        for (int i = 0; i < 16; ++i) u += sb[i];
        db[n] = u;
    }
}

// both in one launch: workgroups [0, n_dgrad) make dA, the rest dW and db (grid (K/16, N) of the latter, row-major)
__global__ __launch_bounds__(256) void k_small_bwd(const float* __restrict__ dZ, const float* __restrict__ A,
                                                   const float* __restrict__ W, float* __restrict__ dA,
                                                   float* __restrict__ dW, float* __restrict__ db, int M, int N, int K,
                                                   int n_dgrad) {
    const int b = blockIdx.x;
    if (b < n_dgrad) {
        small_dgrad_block(dZ, W, dA, M, N, K, b);
    } else {
        const int kx = (K + 15) / 16;
        small_wgrad_block(dZ, A, dW, db, M, N, K, (b - n_dgrad) % kx, (b - n_dgrad) / kx);
    }
}

// grad of log_softmax: dlogits = dlogp - softmax * sum_j dlogp_j  (logp are log-probabilities)
__global__ void k_logsoftmax_bwd(const float* __restrict__ logp, const float* __restrict__ dlogp,
                                 float* __restrict__ dlogits, int B, int k) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += dlogp[(size_t)b * k + j];
    for (int j = 0; j < k; ++j) dlogits[(size_t)b * k + j] = dlogp[(size_t)b * k + j] - expf(logp[(size_t)b * k + j]) * s;
}

// F.nll_loss(logp, target) with mean reduction and its gradient wrt logp (main_1v.py:270)
__global__ void k_nll_loss(const float* __restrict__ logp, const long long* __restrict__ target, int B, int k,
                           float* __restrict__ loss, float* __restrict__ dlogp) {
    __shared__ float s[256];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        // a label outside [0, k) (F.nll_loss raises on it) is never used as an index: the loss becomes NaN, which
        // the caller sees without a separate flag, and that row gets no gradient
        const long long tl = target[b];
        const bool ok = tl >= 0 && tl < k;
        const int t = ok ? (int)tl : -1;
        acc -= ok ? logp[(size_t)b * k + t] : __int_as_float(0x7fc00000);
        for (int j = 0; j < k; ++j) dlogp[(size_t)b * k + j] = (j == t) ? -1.0f / B : 0.f;
    }
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (threadIdx.x < off) s[threadIdx.x] += s[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = s[0] / B;
}

// torch.optim.Adam defaults (no amsgrad, no weight decay): one fused pass over the flat buffers.
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, size_t n, float step_size, float b1, float b2, float eps,
                       float bc2_sqrt, float grad_scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i] * grad_scale;
    const float mi = m[i] + (1.f - b1) * (gi - m[i]);                  // lerp form used by torch
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= step_size * (mi / denom);
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
static inline int colreduce_grid_x(int C) { return (C / 4 + 63) / 64; }   // blocks of up to 64 column quads
static inline dim3 grid1(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

int t_x3_blocks(int M) { return (M + X3M_ROWS - 1) / X3M_ROWS; }
// rows of points (through the candidate's transform, if given) and their moment sums in part (t_x3_blocks(B*P) * 9
// doubles), which t_conv1_bn_fwd turns into conv1's batch statistics
hipError_t t_points_to_rows_moments(const float* x, const float* trans, float* X3, int B, int P, double* part,
                                    hipStream_t s) {
    hipLaunchKernelGGL(k_points_rows_moments, dim3(t_x3_blocks(B * P)), dim3(256), 0, s, x, trans, X3, B, P, part);
    return hipGetLastError();
}
// conv1 + BatchNorm + ReLU: batch statistics from the moment sums t_points_to_rows_moments left in part (mean, rstd,
// running statistics updated) and, when h1 is given, the activation itself.
hipError_t t_conv1_bn_fwd(const float* X3, const float* W1, const float* b1, const float* gamma, const float* beta,
                          float momentum, double* part, float* mean, float* rstd, float* running_mean,
                          float* running_var, float* h1, int M, hipStream_t s) {
    const int nblk = t_x3_blocks(M);
    hipLaunchKernelGGL(k_conv1_stats, dim3(1), dim3(576), 0, s, (const double*)part, nblk, M, W1, b1, momentum, mean, rstd,
                       running_mean, running_var);
    if (h1 == nullptr) return hipGetLastError();      // statistics only: the caller makes h1 where it is consumed
    int blocks = (M + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_conv1_bn_fwd, dim3(blocks), dim3(256), 0, s, X3, W1, b1, (const float*)mean, (const float*)rstd,
                       gamma, beta, h1, M);
    return hipGetLastError();
}
int t_red_blocks(int M) { return (M + RED_ROWS - 1) / RED_ROWS; }
// Rows per block of the column reductions: a matrix of a few hundred rows (the FC layers: one row per candidate)
// would otherwise be one or two blocks whose threads walk 64 rows one dependent load after the other.
int t_colreduce_rows(int M) { return M > 16 * RED_ROWS ? RED_ROWS : 32; }
int t_colreduce_blocks(int M) {
    const int rows = t_colreduce_rows(M);
    return (M + rows - 1) / rows;
}

hipError_t t_bn_stats(const float* A, int M, int C, float momentum, double* part, float* mean, float* rstd,
                      float* running_mean, float* running_var, hipStream_t s) {
    const int nblk = t_colreduce_blocks(M);
    hipLaunchKernelGGL(k_colreduce, dim3(colreduce_grid_x(C), nblk), dim3(256), 0, s, A, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, M, C, 0,
                       t_colreduce_rows(M), part, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr);
    hipLaunchKernelGGL(k_stats_finalize, grid1(C, FIN_COLS), dim3(256), 0, s, part, nblk, A, M, C, momentum, mean, rstd,
                       running_mean, running_var);
    return hipGetLastError();
}
// FC layer forward in training: Z = X W^T + b, batch statistics, A = relu(bn(Z)).  Few rows: split-k product (with
// its reduce) and one kernel for the rest; otherwise product, column reduction, finalize, normalisation.
hipError_t t_fc_bn_fwd(const float* X, const float* W, const float* b, const float* gamma, const float* beta, int M,
                       int N, int K, float momentum, double* part, float* splitk_part, float* Z, float* A, float* mean,
                       float* rstd, float* running_mean, float* running_var, hipStream_t s) {
    hipError_t e;
    if (M <= FEW_ROWS_MAX && N % 64 == 0 && splitk_part != nullptr) {
        e = launch_fc_f32_splitk(X, W, b, Z, M, N, K, 0, splitk_part, s);
        if (e != hipSuccess) return e;
        if (M <= 256)
            hipLaunchKernelGGL(k_few_rows_bn_fwd<16>, dim3(N / 16), dim3(256), 0, s, (const float*)Z, M, N, gamma, beta,
                               momentum, A, mean, rstd, running_mean, running_var);
        else
            hipLaunchKernelGGL(k_few_rows_bn_fwd<FEW_ROWS_MAX / 16>, dim3(N / 16), dim3(256), 0, s, (const float*)Z, M, N,
                               gamma, beta, momentum, A, mean, rstd, running_mean, running_var);
        return hipGetLastError();
    }
    e = launch_fc_f32(X, W, b, Z, M, N, K, 0, s);
    if (e != hipSuccess) return e;
    e = t_bn_stats(Z, M, N, momentum, part, mean, rstd, running_mean, running_var, s);
    if (e != hipSuccess) return e;
    return t_bn_act_fwd(Z, mean, rstd, gamma, beta, A, M, N, 1, s);
}
hipError_t t_bn_act_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        float* h, int M, int C, int relu, hipStream_t s) {
    if (C % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_bn_act_fwd, grid1((size_t)M * C / 4), dim3(256), 0, s, y, mean, rstd, gamma, beta, h,
                       (size_t)M * C / 4, C, relu);
    return hipGetLastError();
}
// dh (in place) <- dy ; also dgamma, dbeta
// conv1_x3 / conv1_w / conv1_b given (C = 64): y is conv1's output and is recomputed from the points instead of read
hipError_t t_bn_bwd(float* dh, const float* h, const float* y, const float* mean, const float* rstd,
                    const float* gamma, int M, int C, double* part, float* m1, float* m2, float* dgamma,
                    float* dbeta, float* dbias_zero, hipStream_t s, const float* conv1_x3, const float* conv1_w,
                    const float* conv1_b, const float* beta) {
    // beta given (the per-point layers): the ReLU mask comes from the recomputed pre-activation, h is not read
    if (conv1_x3 != nullptr && (C != 64 || !conv1_w || !conv1_b || !beta)) return hipErrorInvalidValue;
    if (M <= FEW_ROWS_MAX && C % 16 == 0 && h != nullptr && conv1_x3 == nullptr) {
        if (M <= 256)
            hipLaunchKernelGGL(k_few_rows_bn_bwd<16>, dim3(C / 16), dim3(256), 0, s, dh, h, y, mean, rstd, gamma, M, C,
                               dgamma, dbeta, dbias_zero);
        else
            hipLaunchKernelGGL(k_few_rows_bn_bwd<FEW_ROWS_MAX / 16>, dim3(C / 16), dim3(256), 0, s, dh, h, y, mean, rstd,
                               gamma, M, C, dgamma, dbeta, dbias_zero);
        return hipGetLastError();
    }
    const int nblk = t_colreduce_blocks(M);
    hipLaunchKernelGGL(k_colreduce, dim3(colreduce_grid_x(C), nblk), dim3(256), 0, s, (const float*)dh, h, y, mean, rstd, M,
                       C, 1, t_colreduce_rows(M), part, conv1_x3, conv1_w, conv1_b, beta != nullptr ? gamma : (const float*)nullptr,
                       beta);
    hipLaunchKernelGGL(k_bnbwd_finalize, grid1(C, FIN_COLS), dim3(256), 0, s, part, nblk, M, C, m1, m2, dgamma, dbeta,
                       dbias_zero);
    if (conv1_x3 != nullptr)
        hipLaunchKernelGGL(k_conv1_bn_bwd_dense, dim3((M + 255) / 256), dim3(256), 0, s, dh, h, conv1_x3, conv1_w, conv1_b,
                           mean, rstd, gamma, (const float*)m1, (const float*)m2, M, beta);
    else
        hipLaunchKernelGGL(k_bn_bwd_dense, grid1((size_t)M * C / 4), dim3(256), 0, s, dh, h, y, mean, rstd, gamma, m1, m2,
                           (size_t)M * C / 4, C, beta);
    return hipGetLastError();
}
hipError_t t_bn_bwd_finalize(const double* part, int nblk, int M, int C, float* m1, float* m2, float* dgamma,
                             float* dbeta, float* dbias_zero, hipStream_t s) {
    hipLaunchKernelGGL(k_bnbwd_finalize, grid1(C, FIN_COLS), dim3(256), 0, s, part, nblk, M, C, m1, m2, dgamma, dbeta,
                       dbias_zero);
    return hipGetLastError();
}
// What follows the fused backward pass of conv2 (train_bwd2.hip) in three launches: (1) the weight gradient of conv2,
// dW2[128,64] = dy2^T h1 as partial products, with the finalize step of conv1's BatchNorm backward (column sums in
// part, [nblk][2][64] doubles) riding along; (2) conv1's backward straight from dh1 - dW1 partial sums into part,
// which must hold t_red_blocks(M) * 192 doubles, and dX3 if wanted - with the reduction of dW2's partial products
// riding along; (3) the reduction of dW1's partial sums.
hipError_t t_conv2_wgrad_conv1_bwd(const float* dy2, const float* h1, float* tn_part, float* dW2, const float* dh1,
                                   const float* X3, const float* W1, const float* b1, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int M, double* part, int nblk,
                                   float* m1, float* m2, float* dgamma, float* dbeta, float* dbias_zero, float* dW1,
                                   float* dX3, hipStream_t s) {
    const int chunks = t_tn_chunks(M);
    const int n_tn = (C1 / 64) * (C2 / 64) * chunks, n_fin = 64 / FIN_COLS;
    hipLaunchKernelGGL(gemm_tn_finalize_kernel, dim3(n_fin + n_tn), dim3(256), 0, s, dy2, h1, chunks == 1 ? dW2 : tn_part, M,
                       C2, C1, TN_ROWS_PER_CHUNK, n_fin, (const double*)part, nblk, 64, m1, m2, dgamma, dbeta, dbias_zero);
    const int wblk = t_red_blocks(M);
    const size_t nw = (size_t)C2 * C1;
    const int rblk = chunks > 1 ? (int)((nw + 15) / 16) : 0;
    hipLaunchKernelGGL(k_conv1_bwd_all, dim3(wblk + rblk), dim3(256), 0, s, dh1, X3, W1, b1, mean, rstd, gamma, beta,
                       (const float*)m1, (const float*)m2, M, part, dX3, wblk, (const float*)tn_part, chunks, nw, dW2);
    hipLaunchKernelGGL(k_reduce_double, grid1(192, FIN_COLS), dim3(256), 0, s, (const double*)part, wblk, 192, dW1);
    return hipGetLastError();
}
int t_tn_chunks(int M) { return (M + TN_ROWS_PER_CHUNK - 1) / TN_ROWS_PER_CHUNK; }

// dW[N,K] = dY^T Hm (as t_gemm_tn) and dX[M,K] = dY W of one linear layer [N out, K in] as one launch (plus the
// reductions of their partial products, if any)
hipError_t t_gemm_tn_nn(const float* dY, const float* Hm, float* tn_part, float* dW, const float* W, float* dX, int M,
                        int N, int K, hipStream_t s, float* splitk_part) {
    if (N % 64 != 0 || K % 64 != 0) return hipErrorInvalidValue;
    const int chunks = t_tn_chunks(M);
    const int n_tn = (K / 64) * (N / 64) * chunks;
    const int nn_tiles = (K / 64) * ((M + 63) / 64);
    int ks = splitk_part != nullptr ? splitk_factor(nn_tiles, N / GB_K) : 1;
    if ((size_t)ks * M * K > SPLITK_PART_FLOATS) ks = 1;
    hipLaunchKernelGGL(gemm_tn_nn_kernel, dim3(n_tn + nn_tiles * ks), dim3(256), 0, s, dY, Hm, chunks == 1 ? dW : tn_part,
                       W, dX, ks > 1 ? splitk_part : (float*)nullptr, M, N, K, TN_ROWS_PER_CHUNK, n_tn, ks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (chunks > 1)
        hipLaunchKernelGGL(k_reduce_chunks, grid1((size_t)N * K, 16), dim3(256), 0, s, (const float*)tn_part, chunks,
                           (size_t)N * K, dW);
    if (ks > 1) return launch_splitk_reduce(splitk_part, ks, M, K, nullptr, 0, dX, s);
    return hipGetLastError();
}

hipError_t t_gemm_tn(const float* dY, const float* Hm, float* part, float* dW, int M, int N, int K, hipStream_t s) {
    if (N % 64 != 0 || K % 64 != 0) return hipErrorInvalidValue;
    const int chunks = t_tn_chunks(M);
    // a single chunk is the result itself
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(K / 64, N / 64, chunks), dim3(256), 0, s, dY, Hm, chunks == 1 ? dW : part, M,
                       N, K, TN_ROWS_PER_CHUNK);
    if (chunks > 1)
        hipLaunchKernelGGL(k_reduce_chunks, grid1((size_t)N * K, 16), dim3(256), 0, s, (const float*)part, chunks,
                           (size_t)N * K, dW);
    return hipGetLastError();
}
hipError_t t_conv1_bwd(const float* dy1, const float* X3, const float* W1, int M, double* part, float* dW1,
                       float* dX3, hipStream_t s) {
    const int nblk = t_red_blocks(M);
    hipLaunchKernelGGL(k_conv1_wgrad, dim3(nblk), dim3(256), 0, s, dy1, X3, M, part);
    hipLaunchKernelGGL(k_reduce_double, grid1(192, FIN_COLS), dim3(256), 0, s, (const double*)part, nblk, 192, dW1);
    if (dX3 != nullptr) {
        int blocks = (M + 63) / 64;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_conv1_dgrad, dim3(blocks), dim3(256), 0, s, dy1, W1, dX3, M);
    }
    return hipGetLastError();
}
hipError_t t_trans_grad(const float* x, const float* dX3, int B, int P, const float* dtrans_ext, float* dtrans,
                        hipStream_t s) {
    hipLaunchKernelGGL(k_trans_grad, dim3(B), dim3(256), 0, s, x, dX3, P, dtrans_ext, dtrans);
    return hipGetLastError();
}
hipError_t t_small_bwd(const float* dZ, const float* A, const float* W, float* dA, float* dW, float* db, int M, int N,
                       int K, hipStream_t s) {
    const int n_dgrad = (int)(((size_t)M * K + 255) / 256);
    hipLaunchKernelGGL(k_small_bwd, dim3(n_dgrad + (K + 15) / 16 * N), dim3(256), 0, s, dZ, A, W, dA, dW, db, M, N, K,
                       n_dgrad);
    return hipGetLastError();
}
hipError_t t_logsoftmax_bwd(const float* logp, const float* dlogp, float* dlogits, int B, int k, hipStream_t s) {
    hipLaunchKernelGGL(k_logsoftmax_bwd, grid1(B), dim3(256), 0, s, logp, dlogp, dlogits, B, k);
    return hipGetLastError();
}
hipError_t t_nll_loss(const float* logp, const long long* target, int B, int k, float* loss, float* dlogp,
                      hipStream_t s) {
    hipLaunchKernelGGL(k_nll_loss, dim3(1), dim3(256), 0, s, logp, target, B, k, loss, dlogp);
    return hipGetLastError();
}
hipError_t t_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                  int step, float grad_scale, hipStream_t s) {
    // bias corrections in double, as torch computes them in Python floats
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2_sqrt = sqrt(1.0 - pow((double)b2, (double)step));
    hipLaunchKernelGGL(k_adam, grid1(n), dim3(256), 0, s, p, g, m, v, n, (float)((double)lr / bc1), b1, b2, eps,
                       (float)bc2_sqrt, grad_scale);
    return hipGetLastError();
}

}  // namespace pngpd

// Products of a very tall matrix with a small one, exact f32 on v_mfma_f32_32x32x2_f32:
//
//     Y[M,N] = act(X[M,K] B[K,N] + bias),   (K, N) = (64,128), (128,64), (128,128),  M = candidates x points
//
// used by the training step for the per-point layers (conv1d with kernel 1 = a row-wise linear map,
// model/pointnet.py:30-31, :143-146, and their input gradients).  Such a product moves ~1 KB per row and does
// ~30 kflop per row, so it sits at the HBM roofline only if nothing else is in the way.  Hence:
//   * persistent workgroups (two per CU) that keep the whole B in LDS (<= 64 KiB, loaded once) and walk over
//     tiles of 128 rows: no per-tile prologue, no workgroup barrier inside the loop;
//   * the A operand goes straight from global memory to registers - lane (r, h) owns the contiguous half row
//     X[row r][h*K/2 .. (h+1)*K/2) - and the tile after the current one is already in flight during the MFMAs.
//     MFMA step s therefore pairs k = s (lanes h=0) with k = K/2 + s (lanes h=1); B rows are read accordingly;
//   * optionally the column statistics of Y (what train-mode BatchNorm needs next) are taken from the
//     accumulators: per workgroup and column a record (count, mean, sum of squared deviations), merged later
//     by the parallel-variance formula in double.  That saves a full pass over Y.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// merge (n, mean, M2) records: Chan et al.
__device__ __forceinline__ void merge_moments(double& n, double& mean, double& m2, double nb, double meanb,
                                              double m2b) {
    if (nb == 0.0) return;
    const double nt = n + nb, d = meanb - mean;
    mean += d * (nb / nt);
    m2 += m2b + d * d * (n * nb / nt);
    n = nt;
}

template <int K, int N>
__global__ __launch_bounds__(256, 2) void tall_gemm_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           int w_is_kn, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int M, int relu,
                                                           double* __restrict__ stat_rec) {
    // B rows k < K/2 at k*N, rows k >= K/2 shifted by 32 floats: an MFMA step reads row s for the lanes of half 0 and
    // row K/2 + s for half 1, 32 consecutive floats each; unshifted, the two rows start on the same LDS bank
    // (K/2 * N is a multiple of 64 words) and every read is a two-way conflict (1.35 M conflict cycles per launch)
    extern __shared__ __attribute__((aligned(16))) float sB[];      // [K][N] + 32
    constexpr int NT = N / 32, KH = K / 2, SHIFT = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    if (w_is_kn) {
        for (int i = tid * 4; i < K * N; i += 1024) *(float4*)(sB + i + (i >= KH * N ? SHIFT : 0)) = *(const float4*)(W + i);
    } else {                                   // W[N][K] -> B[k][n]
        for (int i = tid; i < K * N; i += 256) {
            const int n = i / K, k = i - n * K;
            sB[k * N + n + (k >= KH ? SHIFT : 0)] = W[i];
        }
    }
    __syncthreads();

    float bv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bv[t] = bias != nullptr ? bias[32 * t + r] : 0.f;

    // running column statistics of this lane: rows (i&3) + 8(i>>2) + 4h of every tile, column 32t + r
    double cnt = 0.0, s1[NT], s2[NT];
    float a0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        s1[t] = 0.0;
        s2[t] = 0.0;
        a0[t] = 0.f;
    }

    const int tiles = (M + 127) / 128;
    float4 a[KH / 4], an[KH / 4];
    auto load_rows = [&](int tile, float4* dst) {
        int row = tile * 128 + wave * 32 + r;
        row = row < M ? row : M - 1;
        const float4* src = (const float4*)(X + (size_t)row * K + h * KH);
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) dst[q] = src[q];
    };
    int tile = blockIdx.x;
    if (tile < tiles) load_rows(tile, a);
    for (; tile < tiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < tiles) load_rows(next, an);
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* brow = sB + (h * KH + 4 * q + e) * N + h * SHIFT + r;
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], brow[32 * t], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);     // keep the B reads of later steps from piling up in registers
        }
        const int row0 = tile * 128 + wave * 32 + 4 * h;
        int nvalid = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) nvalid += (row0 + (i & 3) + 8 * (i >> 2)) < M ? 1 : 0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float f1 = 0.f, f2 = 0.f;
            if (stat_rec != nullptr && cnt == 0.0 && nvalid > 0) a0[t] = acc[t][0] + bv[t];    // this lane's shift
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = row0 + (i & 3) + 8 * (i >> 2);
                if (row < M) {
                    float v = acc[t][i] + bv[t];
                    const float d = v - a0[t];
                    f1 += d;
                    f2 += d * d;
                    if (relu) v = __builtin_fmaxf(v, 0.f);
                    Y[(size_t)row * N + 32 * t + r] = v;
                }
            }
            s1[t] += (double)f1;
            s2[t] += (double)f2;
        }
        cnt += (double)nvalid;
        if (next < tiles) {
#pragma unroll
            for (int q = 0; q < KH / 4; ++q) a[q] = an[q];
        }
    }
    if (stat_rec == nullptr) return;

    // per lane -> per workgroup: 8 lane records per column (4 waves x 2 halves), merged in a fixed order
    __syncthreads();                                   // B is no longer read: reuse its LDS
    double* red = (double*)sB;                         // [8][N][3]
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        double* p = red + ((size_t)(wave * 2 + h) * N + 32 * t + r) * 3;
        const double mean = cnt > 0.0 ? (double)a0[t] + s1[t] / cnt : 0.0;
        const double m2 = cnt > 0.0 ? s2[t] - s1[t] * s1[t] / cnt : 0.0;
        p[0] = cnt;
        p[1] = mean;
        p[2] = m2 > 0.0 ? m2 : 0.0;
    }
    __syncthreads();
    if (tid < N) {
        double n = 0.0, mean = 0.0, m2 = 0.0;
        for (int k = 0; k < 8; ++k) {
            const double* p = red + ((size_t)k * N + tid) * 3;
            merge_moments(n, mean, m2, p[0], p[1], p[2]);
        }
        double* o = stat_rec + ((size_t)blockIdx.x * N + tid) * 3;
        o[0] = n;
        o[1] = mean;
        o[2] = m2;
    }
}

// Batch statistics from the per-workgroup records (count, mean, M2): N = sum n_b, mean = sum n_b mean_b / N,
// M2 = sum [M2_b + n_b (mean_b - mean)^2].  block = 4 columns x 64 record lanes; a lane sums every 64th record,
// the 64 lane sums are added in a fixed order (butterfly inside a wave, then the wave totals).  Same outputs as the
// two-pass statistics: mean, rstd (biased variance, eps 1e-5) and the running update (momentum, unbiased variance).
__global__ __launch_bounds__(256) void k_stats_from_records(const double* __restrict__ rec, int nrec, int M, int C,
                                                            float momentum, float* __restrict__ mean_out,
                                                            float* __restrict__ rstd_out,
                                                            float* __restrict__ running_mean,
                                                            float* __restrict__ running_var) {
    constexpr int COLS = 4, LANES = 64;
    __shared__ double sh[2][4][COLS];
    const int cl = threadIdx.x & (COLS - 1), rl = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + cl;
    // the 64 lane sums of a column: a butterfly over the 16 row lanes a wave holds, then the four wave totals
    auto lane_sum = [&](double v, double (*buf)[COLS]) {
#pragma unroll
        for (int off = COLS; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) < COLS) buf[threadIdx.x >> 6][cl] = v;
        __syncthreads();
        return ((buf[0][cl] + buf[1][cl]) + buf[2][cl]) + buf[3][cl];
    };
    double a = 0.0;
    if (c < C)
        for (int i = rl; i < nrec; i += LANES) {
            const double* p = rec + ((size_t)i * C + c) * 3;
            a += p[0] * p[1];
        }
    const double mean = lane_sum(a, sh[0]) / M;
    a = 0.0;
    if (c < C)
        for (int i = rl; i < nrec; i += LANES) {
            const double* p = rec + ((size_t)i * C + c) * 3;
            const double d = p[1] - mean;
            a += p[2] + p[0] * d * d;
        }
    const double m2 = lane_sum(a, sh[1]);
    if (rl != 0 || c >= C) return;
    double var = m2 / M;
    if (var < 0.0) var = 0.0;
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)(1.0 / sqrt(var + 1e-5));
    if (running_mean != nullptr) {
        const double unbiased = M > 1 ? var * ((double)M / (double)(M - 1)) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
}

template <int K, int N>
static hipError_t launch_one(const float* X, const float* W, int w_is_kn, const float* bias, float* Y, int M, int relu,
                             double* stat_rec, int blocks, hipStream_t s) {
    auto kern = tall_gemm_kernel<K, N>;
    size_t lds = ((size_t)K * N + 32) * sizeof(float);
    const size_t red = (size_t)8 * N * 3 * sizeof(double);
    if (lds < red) lds = red;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, s, X, W, w_is_kn, bias, Y, M, relu, stat_rec);
    return hipGetLastError();
}

int tall_gemm_blocks(int M) {
    const int tiles = (M + 127) / 128;
    return tiles < TALL_GEMM_MAX_BLOCKS ? tiles : TALL_GEMM_MAX_BLOCKS;
}

bool tall_gemm_supports(int M, int N, int K) {
    return M >= 1 && (N == 64 || N == 128) && (K == 64 || K == 128) && !(N == 64 && K == 64);
}

hipError_t launch_tall_gemm(const float* X, const float* W, int w_is_kn, const float* bias, float* Y, int M, int N,
                            int K, int relu, double* stat_rec, hipStream_t s) {
    if (!X || !W || !Y || !tall_gemm_supports(M, N, K)) return hipErrorInvalidValue;
    const int blocks = tall_gemm_blocks(M);
    if (K == 64 && N == 128) return launch_one<64, 128>(X, W, w_is_kn, bias, Y, M, relu, stat_rec, blocks, s);
    if (K == 128 && N == 64) return launch_one<128, 64>(X, W, w_is_kn, bias, Y, M, relu, stat_rec, blocks, s);
    return launch_one<128, 128>(X, W, w_is_kn, bias, Y, M, relu, stat_rec, blocks, s);
}

hipError_t t_stats_from_records(const double* rec, int nrec, int M, int C, float momentum, float* mean, float* rstd,
                                float* running_mean, float* running_var, hipStream_t s) {
    hipLaunchKernelGGL(k_stats_from_records, dim3((C + 3) / 4), dim3(256), 0, s, rec, nrec, M, C, momentum, mean, rstd,
                       running_mean, running_var);
    return hipGetLastError();
}

}  // namespace pngpd

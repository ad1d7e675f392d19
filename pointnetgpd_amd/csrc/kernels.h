// Internal launch interface between the C-ABI layer (api.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pngpd {

// Workgroup barrier that orders LDS only.  __syncthreads() also fences global memory, i.e. waits for every
// global load still in flight - which drains a register prefetch pipeline at each barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Waits until this wave's LDS-DMA transfers (global_load ... lds) have landed.  An LDS-DMA is a vector-memory operation:
// the workgroup release in front of a barrier orders LDS operations only, and the compiler counts a wave's
// outstanding transfers only for that wave's own reads.  A barrier that publishes DMA'd bytes to OTHER waves therefore
// needs every wave to drain its transfers first.  (s_waitcnt immediate on gfx9: vmcnt = bits 3:0 and 15:14, expcnt =
// bits 6:4, lgkmcnt = bits 11:8; 0x0F70 = vmcnt(0) with the other two counters left alone.)
__device__ __forceinline__ void lds_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }

constexpr int C1 = 64;     // conv1 output channels
constexpr int C2 = 128;    // conv2 output channels
constexpr int C3 = 1024;   // conv3 output channels

// Sizes of the packed conv-stack weight images (see pack.cpp for the element order).
constexpr size_t W2P_BF16_BYTES = (size_t)C2 * C1 * 2;      // 16 KiB
constexpr size_t W3P_BF16_BYTES = (size_t)C3 * C2 * 2;      // 256 KiB
constexpr size_t W2P_F32_BYTES = (size_t)C2 * C1 * 4;       // 32 KiB
constexpr size_t W3P_F32_BYTES = (size_t)C3 * C2 * 4;       // 512 KiB
constexpr int B2P_FLOATS = 4 * 2 * 16;                      // bias of conv2 in accumulator order

// One fused "3 -> 64 -> 128 -> 1024 -> max over points" stack.
struct ConvStackArgs {
    const float* x;        // candidate points, layout `layout`
    int layout;            // PNGPD_LAYOUT_*
    int B;                 // candidates
    int P;                 // points per candidate
    const float4* w1tab;   // [64] {w0, w1, w2, bias}: BN-folded conv1
    const float* trans;    // [B,9] per-candidate 3x3 applied to the points first, or nullptr
    const void* w2p;       // packed BN-folded conv2 weights (bf16 or f32 image)
    const float* b2p;      // [128] conv2 bias: natural order (bf16), 32x32 accumulator order [4][2][16] (f32)
    const void* w3p;       // packed BN-folded conv3 weights (bf16 or f32 image)
    const float* b3;       // [1024] BN-folded conv3 bias
    float* pooled;         // [B,1024] max over points (+bias, optional ReLU)
    int relu_out;          // 1: ReLU after the max (STN), 0: none (trunk)
    const void* w2lo = nullptr;   // bf16x3 only: second bf16 piece of the conv2 / conv3 weights, same fragment order
    const void* w3lo = nullptr;
    // Small batches (conv_stack_point_groups() > 1): the points of a candidate are cut into point_groups groups of
    // tiles_per_group tiles of 16, and `pooled` is [point_groups][B][1024] partial maxima (bias and ReLU applied to
    // each), whose maximum over the groups is the pooled row.
    int point_groups = 1;
    int tiles_per_group = 0;
};

// Into how many groups of points launch_conv_stack_bf16 / _x3 can cut a candidate of this batch (1: not at all), and
// the tiles of 16 points per group.  Only batches of at most CONV_SPLIT_MAX_B candidates are cut, into at most
// CONV_SPLIT_MAX_GROUPS groups, and only while all workgroups of the launch fit the chip at once.
constexpr int CONV_SPLIT_MAX_B = 8;
constexpr int CONV_SPLIT_MAX_GROUPS = 8;
hipError_t conv_stack_point_groups(int precision, int B, int P, int* groups, int* tiles_per_group);

hipError_t launch_conv_stack_bf16(const ConvStackArgs& a, hipStream_t s);
hipError_t launch_conv_stack_f32(const ConvStackArgs& a, hipStream_t s);
hipError_t launch_conv_stack_x3(const ConvStackArgs& a, hipStream_t s);

// launch_util.cpp: compute units of the current device; hipFuncAttributeMaxDynamicSharedMemorySize set once per
// (kernel, device) - both were three runtime calls on every launch
hipError_t device_cu_count(int* n_cu);
hipError_t set_max_lds_once(const void* kern, int bytes);

// Y[M,N] = act(X[M,K] * W[N,K]^T + b[N]); fp32 MFMA.  N % 64 == 0, K % 32 == 0.
// split-k products of the training step's FC layers (few rows): slices * tiles <= SPLITK_MAX_WORKGROUPS, so the
// partial products never exceed SPLITK_MAX_WORKGROUPS 64x64 tiles
constexpr int SPLITK_MAX_WORKGROUPS = 256;
constexpr size_t SPLITK_PART_FLOATS = (size_t)SPLITK_MAX_WORKGROUPS * 64 * 64;
int splitk_factor(int tiles, int k_tiles);
hipError_t launch_splitk_reduce(const float* part, int ks, int M, int N, const float* bias, int relu, float* out,
                                hipStream_t s);
hipError_t launch_fc_f32_splitk(const float* X, const float* W, const float* b, float* Y, int M, int N, int K, int relu,
                                float* part, hipStream_t s);
hipError_t launch_fc_f32(const float* X, const float* W, const float* b, float* Y, int M, int N,
                         int K, int relu, hipStream_t s);

// Same layer with split-bf16 operands (3 bf16 MFMAs per product term); Wp from pack_fc_split.
// N % 128 == 0, K % 32 == 0.
// groups > 1: X is [groups][M][K] and the layer's input rows are the element-wise maximum over the groups (the
// partial maxima of a conv stack cut over points, M <= 8, K <= 1024); `merged` (optional) receives those rows.
hipError_t launch_fc_bf16x3(const float* X, const void* Wp, const float* b, float* Y, int M, int N, int K, int relu,
                            hipStream_t s, int groups = 1, float* merged = nullptr);

enum { FC_TAIL_PLAIN = 0, FC_TAIL_ADD_IDENTITY = 1, FC_TAIL_LOG_SOFTMAX = 2 };
constexpr int FC_TAIL_K = 256;   // input width of the last layer of both FC chains (model/pointnet.py:24, :182)
// Y[M,N] = tail(X[M,K] * W[N,K]^T + b[N]) for tiny N (<= 16), K % 256 == 0.
hipError_t launch_fc_tail(const float* X, const float* W, const float* b, float* Y, int M, int N,
                          int K, int tail, hipStream_t s);

hipError_t launch_score(const float* logp, int B, int k, float* probs, int64_t* pred,
                        hipStream_t s);

// ---- in-gripper crop -------------------------------------------------------------------------
constexpr int CROP_PPT = 4;                  // scene points per thread
constexpr int CROP_TILE = 256 * CROP_PPT;    // scene points per workgroup
constexpr int CROP_GB = 32;                  // grasps per workgroup

struct CropLimits { double ylo, yhi, zlo, zhi, xlo, xhi; };

int crop_words_per_grasp(int S);             // 64-bit mask words per grasp (tile padded)

// uniform grid over the scene for the culled crop
constexpr int GRID_MAX_DIM = 64;
constexpr int GRID_CELLS = GRID_MAX_DIM * GRID_MAX_DIM * GRID_MAX_DIM;
struct GridParams { double origin[3]; double inv_cell; int dim[3]; int pad; };
size_t crop_grid_bytes(int S);
hipError_t launch_crop_count_grid(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                  void* grid_ws, int* cnt, hipStream_t s);
hipError_t launch_crop_emit_grid(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                 const void* grid_ws, const int* cnt, const long long* offsets, int* idx,
                                 long long capacity, double* pts_g, hipStream_t s);

hipError_t launch_crop_gather_fused(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                    void* grid_ws, int R, int P, int min_points, unsigned long long seed,
                                    int first_grasp_id, float* pts_out, unsigned char* valid, int* cnt, int* overflow,
                                    int* perm, hipStream_t s);
hipError_t launch_crop_frames(const double* grasps, double* frames, int G, hipStream_t s);
// frames [G,12] of training-data grasp records [G,stride] (centre, axis, width, angle, ...) under a 4x4 transform
hipError_t launch_dataset_frames(const double* records, int stride, int G, const double* transform16, double hand_depth,
                                 double* frames, hipStream_t s);

// PNGPD_CROP_BRUTE=1 in the environment selects the all-pairs pass with its [G, S/64] bit mask (the tests
// cross-check the culled pass against it); the mask is part of the workspace only then.
bool crop_brute_mode();
hipError_t launch_crop_count(const float* scene, int S, const double* grasps, int G, const CropLimits& L,
                             double* frames, unsigned long long* mask, void* grid_ws, int* cnt,
                             long long* offsets, hipStream_t s, bool frames_given = false);
hipError_t launch_crop_emit(const float* scene, int S, int G, const CropLimits& L, const double* frames,
                            const unsigned long long* mask, const void* grid_ws, const int* cnt,
                            const long long* offsets, int* idx, long long capacity, double* pts_g, hipStream_t s);
hipError_t launch_crop_gather(const float* scene, const double* frames, const int* idx, const long long* offsets,
                              const int* cnt, int G, int R, int P, int min_points, const int* sel,
                              unsigned long long seed, int first_grasp_id, float* pts_out, int* sel_out,
                              unsigned char* valid, hipStream_t s, int replace_when_equal = 0, int channels_first = 0);
hipError_t launch_vote(const long long* pred, const float* probs, const unsigned char* valid, int G, int R, int k,
                       int* label, float* score, hipStream_t s);

// ---- voxel down-sampling (voxel.hip) ------------------------------------------------------------
constexpr int VOXEL_MAX_N = 1600;            // n^3 voxel keys must fit 32 bits
size_t voxel_workspace_bytes(int n);
hipError_t launch_voxel_unique(const float* pts, int S, int n, int f32_edges, void* ws, float* out, long long capacity,
                               int* count, hipStream_t s);

// ---- tall-matrix products (tall_gemm.hip) ------------------------------------------------------
constexpr int TALL_GEMM_MAX_BLOCKS = 512;    // persistent workgroups: two per CU on a 256-CU part
bool tall_gemm_supports(int M, int N, int K);          // (K, N) = (64,128), (128,64) or (128,128)
int tall_gemm_blocks(int M);                           // workgroups launched = statistic records written
// Y[M,N] = act(X[M,K] B + bias); B[k][n] = W[k*N+n] (w_is_kn) or W[n*K+k]; bias optional.
// stat_rec (optional): [tall_gemm_blocks(M)][N][3] doubles (count, mean, sum of squared deviations) of the columns of
// X B + bias, for t_stats_from_records.
hipError_t launch_tall_gemm(const float* X, const float* W, int w_is_kn, const float* bias, float* Y, int M, int N,
                            int K, int relu, double* stat_rec, hipStream_t s);
hipError_t t_stats_from_records(const double* rec, int nrec, int M, int C, float momentum, float* mean, float* rstd,
                                float* running_mean, float* running_var, hipStream_t s);

// ---- training step (train.hip) ---------------------------------------------------------------
constexpr int TN_ROWS_PER_CHUNK = 512;       // rows of one weight-gradient partial (multiple of 32)

int t_red_blocks(int M);                     // row blocks of 256 rows (conv1 weight-gradient partials)
int t_colreduce_rows(int M);                 // rows per block of the column reductions over an [M, C] matrix
int t_colreduce_blocks(int M);
int t_tn_chunks(int M);                      // partials of a weight-gradient product
// Gram-matrix partials: about 256 chunks of rows (3 workgroups each: a whole number per CU on a 256-CU part),
// at least 256 and a multiple of 32 rows per chunk
inline int t_gram_rows_per_chunk(int M) {
    int rows = (M + 255) / 256;
    if (rows < 256) rows = 256;
    return (rows + 31) & ~31;
}
inline int t_gram_chunks(int M) { return (M + t_gram_rows_per_chunk(M) - 1) / t_gram_rows_per_chunk(M); }
// doubles of the Gram workspace: per chunk a 128x128 partial matrix and 128 column sums
inline size_t t_gram_ws_doubles(int M) { return (size_t)t_gram_chunks(M) * (C2 * C2 + C2); }

int t_x3_blocks(int M);
hipError_t t_conv1_bn_fwd(const float* X3, const float* W1, const float* b1, const float* gamma, const float* beta,
                          float momentum, double* part, float* mean, float* rstd, float* running_mean,
                          float* running_var, float* h1, int M, hipStream_t s);
hipError_t t_points_to_rows_moments(const float* x, const float* trans, float* X3, int B, int P, double* part,
                                    hipStream_t s);
hipError_t t_fc_bn_fwd(const float* X, const float* W, const float* b, const float* gamma, const float* beta, int M,
                       int N, int K, float momentum, double* part, float* splitk_part, float* Z, float* A, float* mean,
                       float* rstd, float* running_mean, float* running_var, hipStream_t s);
hipError_t t_bn_stats(const float* A, int M, int C, float momentum, double* part, float* mean, float* rstd,
                      float* running_mean, float* running_var, hipStream_t s);
hipError_t t_bn_act_fwd(const float* y, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        float* h, int M, int C, int relu, hipStream_t s);
hipError_t t_bn_bwd(float* dh, const float* h, const float* y, const float* mean, const float* rstd,
                    const float* gamma, int M, int C, double* part, float* m1, float* m2, float* dgamma,
                    float* dbeta, float* dbias_zero, hipStream_t s,
                    const float* conv1_x3 = nullptr, const float* conv1_w = nullptr, const float* conv1_b = nullptr,
                    const float* beta = nullptr);
// t_bn_bwd of the two per-point layers in front of conv3, cut differently (train_bwd2.hip): sums of BatchNorm 2 (by
// t_conv3_bwd's last launch, finalized by t_bn_bwd_finalize) -> one pass that turns dh2 into dy2 in place, multiplies
// it with W2 into dh1 and leaves the partial column sums of BatchNorm 1's backward in part
// ([tall_gemm_blocks(M)][2][64] doubles) -> the rest of the stack's backward (dW2, dW1, dX3) by
// t_conv2_wgrad_conv1_bwd, straight from dy2 and dh1, without storing dy1
hipError_t launch_conv2_bwd_fused(float* dh2, const float* y2, const float* mean2, const float* rstd2,
                                  const float* gamma2, const float* beta2, const float* m1, const float* m2,
                                  const float* W2, float* dh1, int M, const float* X3, const float* W1, const float* b1,
                                  const float* mean1, const float* rstd1, const float* gamma1, const float* beta1,
                                  double* part, hipStream_t s);
// Y[M,128] = X[M,128] G + ub, G [128][128] (row = input channel), products of two bfloat16 pieces per operand
hipError_t launch_dense_dgrad_x3(const float* X, const float* G, const float* ub, float* Y, int M, hipStream_t s);
// forward of conv1 / bn1 / relu / conv2 in one pass (statistics of bn1 given): h1, y2 and the column-statistic records
// of y2 ([tall_gemm_blocks(M)][128][3] doubles, for t_stats_from_records)
hipError_t launch_conv12_fwd(const float* X3, const float* W1, const float* b1, const float* mean1, const float* rstd1,
                             const float* gamma1, const float* beta1, const float* W2, const float* b2, float* h1,
                             float* y2, int M, double* stat_rec, hipStream_t s);
// conv2's weight gradient and conv1's whole backward behind launch_conv2_bwd_fused, three launches (train.hip)
hipError_t t_conv2_wgrad_conv1_bwd(const float* dy2, const float* h1, float* tn_part, float* dW2, const float* dh1,
                                   const float* X3, const float* W1, const float* b1, const float* mean,
                                   const float* rstd, const float* gamma, const float* beta, int M, double* part, int nblk,
                                   float* m1, float* m2, float* dgamma, float* dbeta, float* dbias_zero, float* dW1,
                                   float* dX3, hipStream_t s);

// conv3 + BatchNorm (batch statistics) + max over points without the [M,1024] activation (train_conv3.hip)
struct Conv3Ws {
    double* abar_d;    // [128] column mean of the input rows
    double* cov;       // [128,128] covariance of the input rows
    float* w3p;        // [1024*128] weights scaled by gamma*rstd, packed for the product/max kernel
    unsigned short* w3s;   // [3*1024*128] the same weights as three bfloat16 pieces in 32x32x16 fragment order, or (the
                           // first two thirds) as two float16 pieces of the rows scaled by 1 / wsc
    float* wsc = nullptr;  // [1024] per channel: the power of two that undoes the scaling of its float16 row
    float* sh;         // [1024] shift added after the max
    float* gmask;      // [B,1024] pooled gradient after the ReLU mask
    float* G;          // [128,128] W^T diag(q) W
    float* ub;         // [128] constant row of the dense input gradient
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;   // optional: recorded around the product/max kernel
};
// y2_pre != nullptr: h2 is an OUTPUT, relu(bn2(y2_pre)) with the given statistics and affine, written by the
// first pass over the rows; otherwise h2 is the input.
hipError_t t_conv3_fwd(float* h2, const float* W3, const float* b3, const float* gamma, const float* beta, int B,
                       int P, int relu_out, float momentum, double* gram_part, const Conv3Ws& w, float* mean,
                       float* rstd, float* running_mean, float* running_var, float* pooled, int* arg, hipStream_t s,
                       const float* y2_pre = nullptr, const float* bn2_mean = nullptr,
                       const float* bn2_rstd = nullptr, const float* bn2_gamma = nullptr,
                       const float* bn2_beta = nullptr);
hipError_t t_conv3_bwd(const float* h2, const float* W3, const float* b3, const float* gamma, const float* beta,
                       const float* dpooled,
                       const float* pooled, const int* arg, const float* mean, const float* rstd, int B, int P,
                       int relu_out, const Conv3Ws& w, float* m1, float* m2, float* dW3, float* db3_zero, float* dgamma,
                       float* dbeta, float* dh2, hipStream_t s, const float* bn2_y = nullptr,
                       const float* bn2_mean = nullptr, const float* bn2_rstd = nullptr, const float* bn2_gamma = nullptr,
                       const float* bn2_beta = nullptr, double* bn2_part = nullptr);
// the finalize step of t_bn_bwd alone: partial column sums ([nblk][2][C] doubles) -> m1, m2, dgamma, dbeta
hipError_t t_bn_bwd_finalize(const double* part, int nblk, int M, int C, float* m1, float* m2, float* dgamma,
                             float* dbeta, float* dbias_zero, hipStream_t s);
hipError_t t_gemm_tn(const float* dY, const float* Hm, float* part, float* dW, int M, int N, int K, hipStream_t s);
// both of the above for one linear layer [N out, K in] in one launch: dW[N,K] = dY^T Hm, dX[M,K] = dY W
hipError_t t_gemm_tn_nn(const float* dY, const float* Hm, float* tn_part, float* dW, const float* W, float* dX, int M,
                        int N, int K, hipStream_t s, float* splitk_part = nullptr);
hipError_t t_conv1_bwd(const float* dy1, const float* X3, const float* W1, int M, double* part, float* dW1,
                       float* dX3, hipStream_t s);
hipError_t t_trans_grad(const float* x, const float* dX3, int B, int P, const float* dtrans_ext, float* dtrans,
                        hipStream_t s);
hipError_t t_small_bwd(const float* dZ, const float* A, const float* W, float* dA, float* dW, float* db, int M, int N,
                       int K, hipStream_t s);
hipError_t t_logsoftmax_bwd(const float* logp, const float* dlogp, float* dlogits, int B, int k, hipStream_t s);
hipError_t t_nll_loss(const float* logp, const long long* target, int B, int k, float* loss, float* dlogp,
                      hipStream_t s);
hipError_t t_adam(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                  int step, float grad_scale, hipStream_t s);

}  // namespace pngpd

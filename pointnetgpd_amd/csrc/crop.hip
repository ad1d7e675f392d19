// Online in-gripper crop (apps/kinect2grasp.py:299-344 check_collision_square, :377-423
// collect_pc), subsample/pad (:746-774) and the repeat vote (:799-817), for gfx950.
//
// The reference runs, per grasp, a float64 numpy pass over the whole scene: normalise the three
// axes, d = point - centre, pg = R d, six STRICT comparisons, np.where.  To give the same index
// set bit for bit every kernel here computes in float64 with the same operation order
//     |v| = sqrt(fma(z,z,fma(y,y,x*x)))        axis = v / |v|
//     pg[r] = fma(R[r][2], d2, fma(R[r][1], d1, R[r][0]*d0))
// and contraction of separate multiplies and adds is switched off for this file.
//
// Pipeline (all on one stream; the output order is the ascending point index, as np.where gives it):
//   frames : normalised frame per grasp                          [G,12] f64 in the workspace
//   count  : points inside the region per grasp, then an exclusive scan over grasps
//   emit   : ascending scene indices (+ float64 gripper-frame points) per grasp
//            both through the spatial grid of crop_grid.hip.  The all-pairs variant below (one bit per
//            (grasp, point) in a [G, Wp] u64 mask, popcount, bit expansion) is kept as the cross-check the
//            tests run with PNGPD_CROP_BRUTE=1.
//   gather : subsample / pad to P points per (grasp, repeat), transform to the gripper frame,
//            round to float32 = the network input of main_test.py:74-76
#include <hip/hip_runtime.h>

#include "crop_math.h"
#include "crop_select.h"
#include "kernels.h"

namespace pngpd {

__global__ void crop_frames_kernel(const double* __restrict__ grasps, double* __restrict__ frames, int G) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double* in = grasps + (size_t)g * 15;       // [5,3]: row 4 centre, rows 1..3 axes
    double* f = frames + (size_t)g * 12;
    f[0] = in[12];
    f[1] = in[13];
    f[2] = in[14];
    for (int r = 0; r < 3; ++r) {
        const double x = in[3 + 3 * r], y = in[4 + 3 * r], z = in[5 + 3 * r];
        const double n = sqrt(fma(z, z, fma(y, y, x * x)));
        f[3 + 3 * r + 0] = x / n;
        f[3 + 3 * r + 1] = y / n;
        f[3 + 3 * r + 2] = z / n;
    }
}

// Gripper frame of a training-data grasp record in the frame of one view (model/dataset.py:776-835), written
// from the geometry.  A record holds the contact centre c, the closing axis a, the opening width and the angle t
// by which the approach direction is turned about the closing axis; T is the view's 4x4 transform, hd the hand
// depth.
//     y = a / |a|                                  closing axis (binormal)
//     x = (y1, -y0, 0) / |.|   ((1,0,0) when that vanishes)     a horizontal direction orthogonal to y
//     z = x cross (y / |y|)
//     approach = (cos t * x + sin t * z) / |.|     first column of [x y z] * Ry(t)
//     minor    = y cross approach
//     bottom   = c - hd * approach                 the frame's origin: the bottom centre of the open hand
// and every one of them goes through T (points with w = 1, directions with w = 0; the transformed axes are NOT
// re-normalised).  Output row: bottom, approach, binormal, minor = the [G,12] frames pngpd_crop_count_frames takes.
// Products and sums are separate roundings (contraction is off in this file) except where an fma is written: the
// norms and the two-term combination, which is where numpy's BLAS calls fuse; a 4-term row of T * v is summed as
// (p0 + p2) + (p1 + p3).  sin / cos come from the device maths library: against a host evaluation the frames agree
// to a few units in the last place, not bit for bit.
struct Transform4 { double m[16]; };

__device__ __forceinline__ double norm3(double x, double y, double z) { return sqrt(fma(z, z, fma(y, y, x * x))); }

__device__ __forceinline__ void apply_t(const Transform4& T, double v0, double v1, double v2, double w, double* out) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double p0 = T.m[4 * r + 0] * v0, p1 = T.m[4 * r + 1] * v1, p2 = T.m[4 * r + 2] * v2, p3 = T.m[4 * r + 3] * w;
        out[r] = (p0 + p2) + (p1 + p3);
    }
}

__global__ void dataset_frames_kernel(const double* __restrict__ records, int stride, int G, Transform4 T,
                                      double hand_depth, double* __restrict__ frames) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    const double* rec = records + (size_t)g * stride;
    const double cx = rec[0], cy = rec[1], cz = rec[2], angle = rec[7];
    double n = norm3(rec[3], rec[4], rec[5]);
    const double y0 = rec[3] / n, y1 = rec[4] / n, y2 = rec[5] / n;
    double x0 = y1, x1 = -y0, x2 = 0.0;
    n = norm3(x0, x1, x2);
    if (n == 0.0) {
        x0 = 1.0;
        x1 = 0.0;
        n = 1.0;
    }
    x0 /= n;
    x1 /= n;
    x2 /= n;
    n = norm3(y0, y1, y2);                               // the reference normalises the axis a second time here
    const double u0 = y0 / n, u1 = y1 / n, u2 = y2 / n;
    const double z0 = x1 * u2 - x2 * u1, z1 = x2 * u0 - x0 * u2, z2 = x0 * u1 - x1 * u0;
    const double c = cos(angle), s = sin(angle);
    // column 0 of [x u z] * Ry: the middle term u * 0 adds an exact zero
    double a0 = fma(z0, s, fma(u0, 0.0, x0 * c)), a1 = fma(z1, s, fma(u1, 0.0, x1 * c)), a2 = fma(z2, s, fma(u2, 0.0, x2 * c));
    n = norm3(a0, a1, a2);
    a0 /= n;
    a1 /= n;
    a2 /= n;
    const double m0 = y1 * a2 - y2 * a1, m1 = y2 * a0 - y0 * a2, m2 = y0 * a1 - y1 * a0;
    const double b0 = cx - hand_depth * a0, b1 = cy - hand_depth * a1, b2 = cz - hand_depth * a2;
    double* f = frames + (size_t)g * 12;
    apply_t(T, b0, b1, b2, 1.0, f + 0);
    apply_t(T, a0, a1, a2, 0.0, f + 3);
    apply_t(T, y0, y1, y2, 0.0, f + 6);
    apply_t(T, m0, m1, m2, 0.0, f + 9);
}

hipError_t launch_dataset_frames(const double* records, int stride, int G, const double* transform16, double hand_depth,
                                 double* frames, hipStream_t s) {
    if (!records || !frames || !transform16 || G < 1 || stride < 8) return hipErrorInvalidValue;
    Transform4 T;
    for (int i = 0; i < 16; ++i) T.m[i] = transform16[i];
    hipLaunchKernelGGL(dataset_frames_kernel, dim3((G + 255) / 256), dim3(256), 0, s, records, stride, G, T, hand_depth,
                       frames);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void crop_mask_kernel(const float* __restrict__ scene, int S,
                                                        const double* __restrict__ frames, int G,
                                                        CropLimits L, unsigned long long* __restrict__ mask,
                                                        int Wp) {
    __shared__ unsigned long long s_mask[CROP_GB][CROP_TILE / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, gb = blockIdx.y;
    double px[CROP_PPT], py[CROP_PPT], pz[CROP_PPT];
    bool ok[CROP_PPT];
#pragma unroll
    for (int k = 0; k < CROP_PPT; ++k) {
        const int s = tile * CROP_TILE + k * 256 + tid;
        ok[k] = s < S;
        const float* p = scene + (size_t)(ok[k] ? s : 0) * 3;
        px[k] = (double)p[0];
        py[k] = (double)p[1];
        pz[k] = (double)p[2];
    }
    const int g_end = min(CROP_GB, G - gb * CROP_GB);
    for (int j = 0; j < g_end; ++j) {
        const double* f = frames + (size_t)(gb * CROP_GB + j) * 12;   // wave-uniform: scalar loads
#pragma unroll
        for (int k = 0; k < CROP_PPT; ++k) {
            const PointD q = to_gripper(f, px[k], py[k], pz[k]);
            const unsigned long long bal = __ballot(in_box(q, L) && ok[k]);
            if (lane == 0) s_mask[j][k * 4 + wave] = bal;
        }
    }
    __syncthreads();
    constexpr int WPT = CROP_TILE / 64;    // 16 words per (grasp, tile) = one 128-byte row
    for (int i = tid; i < g_end * WPT; i += 256) {
        const int j = i / WPT, w = i % WPT;
        mask[(size_t)(gb * CROP_GB + j) * Wp + (size_t)tile * WPT + w] = s_mask[j][w];
    }
}

__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

// Exclusive prefix of v over a 256-thread block and the block total.  s_tmp: 4 ints.
__device__ __forceinline__ int block_exclusive_scan(int v, int tid, int* s_tmp, int& total) {
    const int lane = tid & 63, wave = tid >> 6;
    const int inc = wave_inclusive_scan(v, lane);
    __syncthreads();                       // previous use of s_tmp finished
    if (lane == 63) s_tmp[wave] = inc;
    __syncthreads();
    const int t0 = s_tmp[0], t1 = s_tmp[1], t2 = s_tmp[2], t3 = s_tmp[3];
    total = t0 + t1 + t2 + t3;
    const int before = wave == 0 ? 0 : wave == 1 ? t0 : wave == 2 ? t0 + t1 : t0 + t1 + t2;
    return before + inc - v;
}

__global__ __launch_bounds__(256) void crop_count_kernel(const unsigned long long* __restrict__ mask, int Wp,
                                                         int* __restrict__ cnt) {
    __shared__ int s_tmp[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    int c = 0;
    for (int w = tid; w < Wp; w += 256) c += __popcll(mask[(size_t)g * Wp + w]);
    int total;
    block_exclusive_scan(c, tid, s_tmp, total);
    if (tid == 0) cnt[g] = total;
}

__global__ __launch_bounds__(256) void crop_offsets_kernel(const int* __restrict__ cnt, int G,
                                                           long long* __restrict__ offsets) {
    __shared__ int s_tmp[4];
    const int tid = threadIdx.x;
    long long base = 0;
    for (int g0 = 0; g0 < G; g0 += 256) {
        const int g = g0 + tid;
        const int c = g < G ? cnt[g] : 0;
        int total;
        const int ex = block_exclusive_scan(c, tid, s_tmp, total);
        if (g < G) offsets[g] = base + ex;
        base += total;
    }
    if (tid == 0) offsets[G] = base;
}

__global__ __launch_bounds__(256) void crop_emit_kernel(const unsigned long long* __restrict__ mask, int Wp,
                                                        const long long* __restrict__ offsets,
                                                        const float* __restrict__ scene,
                                                        const double* __restrict__ frames,
                                                        int* __restrict__ idx, long long capacity,
                                                        double* __restrict__ pts_g) {
    __shared__ int s_tmp[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    const double* f = frames + (size_t)g * 12;
    long long base = offsets[g];
    for (int w0 = 0; w0 < Wp; w0 += 256) {
        const int w = w0 + tid;
        unsigned long long word = w < Wp ? mask[(size_t)g * Wp + w] : 0ull;
        int total;
        long long o = base + block_exclusive_scan(__popcll(word), tid, s_tmp, total);
        while (word) {
            const int b = __ffsll((long long)word) - 1;
            word &= word - 1;
            const int s = w * 64 + b;
            if (o < capacity) {
                idx[o] = s;
                if (pts_g != nullptr) {
                    const float* p = scene + (size_t)s * 3;
                    const PointD q = to_gripper(f, (double)p[0], (double)p[1], (double)p[2]);
                    pts_g[3 * o + 0] = q.x;
                    pts_g[3 * o + 1] = q.y;
                    pts_g[3 * o + 2] = q.z;
                }
            }
            ++o;
        }
        base += total;
    }
}

// One workgroup per grasp, all its repeats.  The grasp's points are transformed to the gripper frame once,
// rounded to float32 and kept in LDS; the R x P draws then read LDS.  (Drawing straight from the scene costs a
// 64-byte sector per 12-byte point, R times over; a grasp with more than GATHER_LDS_POINTS points does that.)
__global__ __launch_bounds__(256) void crop_gather_kernel(const float* __restrict__ scene,
                                                          const double* __restrict__ frames,
                                                          const int* __restrict__ idx,
                                                          const long long* __restrict__ offsets,
                                                          const int* __restrict__ cnt, int R, int P,
                                                          int min_points, const int* __restrict__ sel,
                                                          unsigned long long seed, int first_grasp_id,
                                                          float* __restrict__ pts_out,
                                                          int* __restrict__ sel_out,
                                                          unsigned char* __restrict__ valid,
                                                          int replace_when_equal, int channels_first) {
    __shared__ float s_pts[GATHER_LDS_POINTS * 3];
    const int g = blockIdx.x, tid = threadIdx.x;
    const int n = cnt[g];
    if (n < min_points) {     // apps/kinect2grasp.py:746-751: scored 0 without running the network
        float* out = pts_out + (size_t)g * R * P * 3;
        for (int i = tid; i < R * P * 3; i += 256) out[i] = 0.f;
        if (sel_out != nullptr)
            for (int i = tid; i < R * P; i += 256) sel_out[(size_t)g * R * P + i] = -1;
        if (tid == 0) valid[g] = 0;
        return;
    }
    if (tid == 0) valid[g] = 1;
    const double* f = frames + (size_t)g * 12;
    const long long base = offsets[g];
    const bool in_lds = n <= GATHER_LDS_POINTS;
    if (in_lds) {
        for (int i = tid; i < n; i += 256) {
            const float* p = scene + (size_t)idx[base + i] * 3;
            const PointD q = to_gripper(f, (double)p[0], (double)p[1], (double)p[2]);
            s_pts[3 * i + 0] = (float)q.x;
            s_pts[3 * i + 1] = (float)q.y;
            s_pts[3 * i + 2] = (float)q.z;
        }
        __syncthreads();
    }
    auto place = [&](size_t row, int i, int j) {       // draw i of candidate `row` is point j of the grasp
        float* out = pts_out + row * P * 3;
        if (sel_out != nullptr) sel_out[row * P + i] = j;
        float x, y, z;
        if (in_lds) {
            x = s_pts[3 * j + 0];
            y = s_pts[3 * j + 1];
            z = s_pts[3 * j + 2];
        } else {
            const float* p = scene + (size_t)idx[base + j] * 3;
            const PointD q = to_gripper(f, (double)p[0], (double)p[1], (double)p[2]);
            x = (float)q.x;
            y = (float)q.y;
            z = (float)q.z;
        }
        if (channels_first) {        // [3,P]: grasp_pc[...].T of the dataset (model/dataset.py:1106)
            out[i] = x;
            out[P + i] = y;
            out[2 * P + i] = z;
        } else {
            out[3 * i + 0] = x;
            out[3 * i + 1] = y;
            out[3 * i + 2] = z;
        }
    };
    if (sel != nullptr) {                               // positions given by the caller
        for (int r = 0; r < R; ++r) {
            const size_t row = (size_t)g * R + r;
            for (int i = tid; i < P; i += 256) place(row, i, sel[row * P + i]);
        }
        return;
    }
    const int lane = tid & 63, wave = tid >> 6;
    for (int r = wave; r < R; r += 4) {                 // a wave draws whole repeats (wave_draws)
        Selector pick;
        pick.init(seed, first_grasp_id + g, r, n, P, replace_when_equal);
        const size_t row = (size_t)g * R + r;
        wave_draws(pick, P, lane, [&](int i, int j) { place(row, i, j); });
    }
}

// One thread per grasp: mode of the R predictions (smallest label among the most frequent, like
// scipy.stats.mode), mean of probs[:, best] over the agreeing repeats accumulated in float32 in
// repeat order (numpy's float32 mean of fewer than 8 values), best = 2 for 3 classes else 1.
__global__ void vote_kernel(const long long* __restrict__ pred, const float* __restrict__ probs,
                            const unsigned char* __restrict__ valid, int G, int R, int k,
                            int* __restrict__ label, float* __restrict__ score) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    if (valid != nullptr && !valid[g]) {
        label[g] = 0;
        score[g] = 0.f;
        return;
    }
    int best_label = 0, best_count = -1;
    for (int c = 0; c < k; ++c) {
        int n = 0;
        for (int r = 0; r < R; ++r) n += (pred[(size_t)g * R + r] == c);
        if (n > best_count) { best_count = n; best_label = c; }
    }
    const int best = k == 3 ? 2 : 1;
    float sum = 0.f;
    for (int r = 0; r < R; ++r)
        if (pred[(size_t)g * R + r] == best_label) sum += probs[((size_t)g * R + r) * k + best];
    label[g] = best_label;
    score[g] = sum / (float)best_count;
}

int crop_words_per_grasp(int S) {
    const int tiles = (S + CROP_TILE - 1) / CROP_TILE;
    return tiles * (CROP_TILE / 64);
}

bool crop_brute_mode() {
    const char* v = getenv("PNGPD_CROP_BRUTE");
    return v && v[0] == '1';
}

hipError_t launch_crop_frames(const double* grasps, double* frames, int G, hipStream_t s) {
    if (!grasps || !frames || G < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(crop_frames_kernel, dim3((G + 255) / 256), dim3(256), 0, s, grasps, frames, G);
    return hipGetLastError();
}

hipError_t launch_crop_count(const float* scene, int S, const double* grasps, int G, const CropLimits& L,
                             double* frames, unsigned long long* mask, void* grid_ws, int* cnt,
                             long long* offsets, hipStream_t s, bool frames_given) {
    if (!scene || !grasps || !frames || !grid_ws || !cnt || !offsets || S < 1 || G < 1) return hipErrorInvalidValue;
    if (frames_given) {   // `grasps` already is [G,12]: origin + the three matrix rows, used as they are
        hipError_t e = hipMemcpyAsync(frames, grasps, (size_t)G * 12 * sizeof(double), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    } else {
        hipLaunchKernelGGL(crop_frames_kernel, dim3((G + 255) / 256), dim3(256), 0, s, grasps, frames, G);
    }
    if (crop_brute_mode()) {
        if (!mask) return hipErrorInvalidValue;
        const int Wp = crop_words_per_grasp(S);
        dim3 grid((S + CROP_TILE - 1) / CROP_TILE, (G + CROP_GB - 1) / CROP_GB);
        hipLaunchKernelGGL(crop_mask_kernel, grid, dim3(256), 0, s, scene, S, frames, G, L, mask, Wp);
        hipLaunchKernelGGL(crop_count_kernel, dim3(G), dim3(256), 0, s, mask, Wp, cnt);
    } else {
        hipError_t e = launch_crop_count_grid(scene, S, (const double*)frames, G, L, grid_ws, cnt, s);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(crop_offsets_kernel, dim3(1), dim3(256), 0, s, cnt, G, offsets);
    return hipGetLastError();
}

hipError_t launch_crop_emit(const float* scene, int S, int G, const CropLimits& L, const double* frames,
                            const unsigned long long* mask, const void* grid_ws, const int* cnt,
                            const long long* offsets, int* idx, long long capacity, double* pts_g, hipStream_t s) {
    if (!scene || !frames || !grid_ws || !cnt || !offsets || !idx || S < 1 || G < 1 || capacity < 0)
        return hipErrorInvalidValue;
    if (crop_brute_mode()) {
        if (!mask) return hipErrorInvalidValue;
        hipLaunchKernelGGL(crop_emit_kernel, dim3(G), dim3(256), 0, s, mask, crop_words_per_grasp(S), offsets, scene,
                           frames, idx, capacity, pts_g);
        return hipGetLastError();
    }
    return launch_crop_emit_grid(scene, S, frames, G, L, grid_ws, cnt, offsets, idx, capacity, pts_g, s);
}

hipError_t launch_crop_gather(const float* scene, const double* frames, const int* idx, const long long* offsets,
                              const int* cnt, int G, int R, int P, int min_points, const int* sel,
                              unsigned long long seed, int first_grasp_id, float* pts_out, int* sel_out,
                              unsigned char* valid, hipStream_t s, int replace_when_equal, int channels_first) {
    if (!scene || !frames || !idx || !offsets || !cnt || !pts_out || !valid || G < 1 || R < 1 || P < 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(crop_gather_kernel, dim3(G), dim3(256), 0, s, scene, frames, idx, offsets, cnt, R, P, min_points,
                       sel, seed, first_grasp_id, pts_out, sel_out, valid, replace_when_equal, channels_first);
    return hipGetLastError();
}

hipError_t launch_vote(const long long* pred, const float* probs, const unsigned char* valid, int G, int R, int k,
                       int* label, float* score, hipStream_t s) {
    if (!pred || !probs || !label || !score || G < 1 || R < 1 || k < 2) return hipErrorInvalidValue;
    hipLaunchKernelGGL(vote_kernel, dim3((G + 255) / 256), dim3(256), 0, s, pred, probs, valid, G, R, k, label, score);
    return hipGetLastError();
}

}  // namespace pngpd

// Spatial culling for the in-gripper crop: instead of testing every (grasp, point) pair, the scene
// is binned once per frame into a uniform grid and each grasp only visits the cells that the
// bounding box of its gripper region overlaps.  The inside test itself is unchanged (crop_math.h,
// float64, reference operation order), and the culling is conservative (box corners in float64
// plus a 1 um margin; cell coordinates are a monotone function of the coordinate), so the set of
// points that pass is exactly the set the brute-force pass finds.  Hits are recorded as bits of
// the per-grasp mask, which keeps the ascending-index order of np.where for free.
//
//   bounds  : min/max of the scene                      -> grid origin, cell size, dimensions
//   count   : cell of every point + histogram (atomics)
//   scan    : exclusive prefix over the (at most 64^3) cells
//   scatter : points regrouped by cell as float4 {x, y, z, original index}
//   cull    : one wave per grasp; x-runs of cells are contiguous in the regrouped array
#include <hip/hip_runtime.h>

#include "crop_math.h"
#include "kernels.h"

namespace pngpd {

constexpr int BOUNDS_BLOCK_PTS = 4096;

__global__ __launch_bounds__(256) void k_bounds_partial(const float* __restrict__ scene, int S,
                                                        float* __restrict__ part) {
    __shared__ float smin[3][256], smax[3][256];
    const int tid = threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const int s0 = blockIdx.x * BOUNDS_BLOCK_PTS, s1 = min(S, s0 + BOUNDS_BLOCK_PTS);
    for (int s = s0 + tid; s < s1; s += 256)
        for (int c = 0; c < 3; ++c) {
            const float v = scene[(size_t)s * 3 + c];
            mn[c] = fminf(mn[c], v);
            mx[c] = fmaxf(mx[c], v);
        }
    for (int c = 0; c < 3; ++c) { smin[c][tid] = mn[c]; smax[c][tid] = mx[c]; }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off)
            for (int c = 0; c < 3; ++c) {
                smin[c][tid] = fminf(smin[c][tid], smin[c][tid + off]);
                smax[c][tid] = fmaxf(smax[c][tid], smax[c][tid + off]);
            }
        __syncthreads();
    }
    if (tid < 3) {
        part[blockIdx.x * 6 + tid] = smin[tid][0];
        part[blockIdx.x * 6 + 3 + tid] = smax[tid][0];
    }
}

// one block: reduce the partial bounds, derive the grid, clear the histogram tail marker
__global__ __launch_bounds__(256) void k_grid_setup(const float* __restrict__ part, int nblk, double cell_req,
                                                    GridParams* __restrict__ gp) {
    __shared__ float smin[3][256], smax[3][256];
    const int tid = threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int b = tid; b < nblk; b += 256)
        for (int c = 0; c < 3; ++c) {
            mn[c] = fminf(mn[c], part[b * 6 + c]);
            mx[c] = fmaxf(mx[c], part[b * 6 + 3 + c]);
        }
    for (int c = 0; c < 3; ++c) { smin[c][tid] = mn[c]; smax[c][tid] = mx[c]; }
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off)
            for (int c = 0; c < 3; ++c) {
                smin[c][tid] = fminf(smin[c][tid], smin[c][tid + off]);
                smax[c][tid] = fmaxf(smax[c][tid], smax[c][tid + off]);
            }
        __syncthreads();
    }
    if (tid == 0) {
        double ext = 0.0;
        for (int c = 0; c < 3; ++c) ext = fmax(ext, (double)smax[c][0] - (double)smin[c][0]);
        double cell = fmax(cell_req, ext / (GRID_MAX_DIM - 1));
        if (!(cell > 1e-9)) cell = 1e-9;
        gp->inv_cell = 1.0 / cell;
        for (int c = 0; c < 3; ++c) {
            gp->origin[c] = (double)smin[c][0];
            int d = (int)floor(((double)smax[c][0] - (double)smin[c][0]) * gp->inv_cell) + 1;
            gp->dim[c] = d < 1 ? 1 : (d > GRID_MAX_DIM ? GRID_MAX_DIM : d);
        }
    }
}

__device__ __forceinline__ int cell_coord(double v, double origin, double inv_cell, int dim) {
    const double t = floor((v - origin) * inv_cell);
    return t < 0.0 ? 0 : (t > (double)(dim - 1) ? dim - 1 : (int)t);
}

__global__ void k_cell_count(const float* __restrict__ scene, int S, const GridParams* __restrict__ gp,
                             int* __restrict__ cell_of, int* __restrict__ hist) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int ix = cell_coord((double)scene[(size_t)s * 3 + 0], gp->origin[0], gp->inv_cell, gp->dim[0]);
    const int iy = cell_coord((double)scene[(size_t)s * 3 + 1], gp->origin[1], gp->inv_cell, gp->dim[1]);
    const int iz = cell_coord((double)scene[(size_t)s * 3 + 2], gp->origin[2], gp->inv_cell, gp->dim[2]);
    const int cell = ix + gp->dim[0] * (iy + gp->dim[1] * iz);
    cell_of[s] = cell;
    atomicAdd(&hist[cell], 1);
}

// exclusive scan of hist over the cells in use into start[0..ncells]; cursor = copy. One block of 1024.
__global__ __launch_bounds__(1024) void k_scan_cells(const GridParams* __restrict__ gp, const int* __restrict__ hist,
                                                     int* __restrict__ start, int* __restrict__ cursor) {
    __shared__ int s_wave[16];
    const int ncells = gp->dim[0] * gp->dim[1] * gp->dim[2];
    const int PER = (ncells + 1023) / 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = tid * PER;
    const int n_mine = max(0, min(PER, ncells - base));
    int sum = 0;
    for (int i = 0; i < n_mine; ++i) sum += hist[base + i];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    int run = before + inc - sum;
    for (int i = 0; i < n_mine; ++i) {
        start[base + i] = run;
        cursor[base + i] = run;
        run += hist[base + i];
    }
    if (n_mine > 0 && base + n_mine == ncells) start[ncells] = run;
}

__global__ void k_scatter(const float* __restrict__ scene, int S, const int* __restrict__ cell_of,
                          int* __restrict__ cursor, float4* __restrict__ sorted) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int pos = atomicAdd(&cursor[cell_of[s]], 1);
    sorted[pos] = make_float4(scene[(size_t)s * 3], scene[(size_t)s * 3 + 1], scene[(size_t)s * 3 + 2],
                              __int_as_float(s));
}

// one wave per grasp
__global__ __launch_bounds__(256) void k_cull(const double* __restrict__ frames, int G, CropLimits L,
                                              const GridParams* __restrict__ gp, const int* __restrict__ start,
                                              const float4* __restrict__ sorted,
                                              unsigned long long* __restrict__ mask, int Wp) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const double* f = frames + (size_t)g * 12;
    // bounding box of the gripper region in the world: corners c + x*a + y*b + z*m
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 8; ++k) {
        const double x = (k & 1) ? L.xhi : L.xlo, y = (k & 2) ? L.yhi : L.ylo, z = (k & 4) ? L.zhi : L.zlo;
        for (int c = 0; c < 3; ++c) {
            const double w = f[c] + x * f[3 + c] + y * f[6 + c] + z * f[9 + c];
            lo[c] = fmin(lo[c], w);
            hi[c] = fmax(hi[c], w);
        }
    }
    int c0[3], c1[3];
    for (int c = 0; c < 3; ++c) {
        const double margin = 1e-6 + 1e-9 * (fabs(lo[c]) + fabs(hi[c]));
        c0[c] = cell_coord(lo[c] - margin, gp->origin[c], gp->inv_cell, gp->dim[c]);
        c1[c] = cell_coord(hi[c] + margin, gp->origin[c], gp->inv_cell, gp->dim[c]);
    }
    const int dx = gp->dim[0], dy = gp->dim[1];
    unsigned long long* row = mask + (size_t)g * Wp;
    for (int iz = c0[2]; iz <= c1[2]; ++iz)
        for (int iy = c0[1]; iy <= c1[1]; ++iy) {
            const int cb = dx * (iy + dy * iz);
            const int p0 = start[cb + c0[0]], p1 = start[cb + c1[0] + 1];
            for (int p = p0 + lane; p < p1; p += 64) {
                const float4 v = sorted[p];
                const PointD q = to_gripper(f, (double)v.x, (double)v.y, (double)v.z);
                if (in_box(q, L)) {
                    const int s = __float_as_int(v.w);
                    atomicOr(&row[s >> 6], 1ull << (s & 63));
                }
            }
        }
}

size_t crop_grid_bytes(int S) {
    auto r = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t nblk = (size_t)(S + BOUNDS_BLOCK_PTS - 1) / BOUNDS_BLOCK_PTS;
    return r(sizeof(GridParams)) + r(nblk * 6 * sizeof(float)) + r((size_t)S * sizeof(int)) +
           3 * r(((size_t)GRID_CELLS + 1) * sizeof(int)) + r((size_t)S * sizeof(float4));
}

// Fills the per-grasp mask with the grid-culled exact test.  `grid_ws` holds crop_grid_bytes(S) bytes.
hipError_t launch_crop_mask_grid(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                 unsigned long long* mask, int Wp, void* grid_ws, hipStream_t s) {
    auto r = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const int nblk = (S + BOUNDS_BLOCK_PTS - 1) / BOUNDS_BLOCK_PTS;
    char* p = (char*)grid_ws;
    GridParams* gp = (GridParams*)p;              p += r(sizeof(GridParams));
    float* part = (float*)p;                      p += r((size_t)nblk * 6 * sizeof(float));
    int* cell_of = (int*)p;                       p += r((size_t)S * sizeof(int));
    int* hist = (int*)p;                          p += r(((size_t)GRID_CELLS + 1) * sizeof(int));
    int* start = (int*)p;                         p += r(((size_t)GRID_CELLS + 1) * sizeof(int));
    int* cursor = (int*)p;                        p += r(((size_t)GRID_CELLS + 1) * sizeof(int));
    float4* sorted = (float4*)p;
    const double ext_x = L.xhi - L.xlo, ext_y = L.yhi - L.ylo, ext_z = L.zhi - L.zlo;
    double cell_req = fmax(ext_x, fmax(ext_y, ext_z)) * 0.25;
    if (!(cell_req > 0.0)) cell_req = 1e-3;
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)G * Wp * sizeof(unsigned long long), s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(hist, 0, ((size_t)GRID_CELLS + 1) * sizeof(int), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_bounds_partial, dim3(nblk), dim3(256), 0, s, scene, S, part);
    hipLaunchKernelGGL(k_grid_setup, dim3(1), dim3(256), 0, s, (const float*)part, nblk, cell_req, gp);
    hipLaunchKernelGGL(k_cell_count, dim3((S + 255) / 256), dim3(256), 0, s, scene, S, (const GridParams*)gp, cell_of,
                       hist);
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, s, (const GridParams*)gp, (const int*)hist, start,
                       cursor);
    hipLaunchKernelGGL(k_scatter, dim3((S + 255) / 256), dim3(256), 0, s, scene, S, (const int*)cell_of, cursor, sorted);
    hipLaunchKernelGGL(k_cull, dim3((G + 3) / 4), dim3(256), 0, s, frames, G, L, (const GridParams*)gp,
                       (const int*)start, (const float4*)sorted, mask, Wp);
    return hipGetLastError();
}

}  // namespace pngpd

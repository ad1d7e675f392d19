// Spatial culling for the in-gripper crop: instead of testing every (grasp, point) pair, the scene
// is binned once per frame into a uniform grid and each grasp only visits the cells that the
// bounding box of its gripper region overlaps.  The inside test itself is unchanged (crop_math.h,
// float64, reference operation order), and the culling is conservative (box corners in float64
// plus a 1 um margin; cell coordinates are a monotone function of the coordinate; within that box a row of
// cells is narrowed to the cells whose centre maps to within half a cell diagonal of the region), so the set of
// points that pass is exactly the set the brute-force pass finds.  Nothing of size G x S is ever written:
// a first traversal counts the hits of every grasp, a second one collects them (~800 per grasp at the online
// sizes) in LDS, sorts them there - np.where order is ascending - and writes the index and point lists.
//
//   bounds  : min/max of the scene                      -> grid origin, cell size, dimensions
//   count   : cell of every point + histogram; the atomic's return value is the point's rank inside its cell
//   scan    : exclusive prefix over the (at most 64^3) cells
//   scatter : points regrouped by cell as float4 {x, y, z, original index} at start[cell] + rank (no atomics)
//   cull    : twice - count (one wave per grasp), emit (one workgroup per grasp); x-runs of cells are contiguous
//             in the regrouped array
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "crop_math.h"
#include "crop_select.h"
#include "kernels.h"

namespace pngpd {

constexpr int BOUNDS_BLOCK_PTS = 4096;

__global__ __launch_bounds__(256) void k_bounds_partial(const float* __restrict__ scene, int S,
                                                        float* __restrict__ part, int* __restrict__ hist) {
    __shared__ float smin[3][256], smax[3][256];
    const int tid = threadIdx.x;
    // the cell histogram of the pass after next starts from zero: cleared here rather than by a fill of its own
    for (int i = blockIdx.x * 256 + tid; i <= GRID_CELLS; i += gridDim.x * 256) hist[i] = 0;
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
                                                    GridParams* __restrict__ gp, int* __restrict__ clear_flag) {
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
        if (clear_flag != nullptr) *clear_flag = 0;
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

// Spatial order of the grasps for the one-pass kernel.  A grasp streams the ~4 k points of the cells around its
// box out of the cell-sorted array (16 MB at a million points: more than the 4 MiB L2 of an XCD), so which grasps
// run next to each other decides whether those lines come from the L2 or from beyond it.  perm[] lists the grasps
// by the Morton code of the coarse cell (16 per axis) that holds their box centre - a counting sort by one
// workgroup (the first workgroup of the k_cell_count launch); the order inside a cell is whatever the atomics give, which changes
// scheduling only, never results.
constexpr int ORDER_BINS = 4096;

__device__ __forceinline__ unsigned part_bits4(unsigned v) {          // abcd -> 00a00b00c00d
    v = (v | (v << 4)) & 0x0C3u;
    v = (v | (v << 2)) & 0x249u;
    return v;
}

// scale[c] = 16 coarse cells per axis over the grid's extent, prepared once per thread (binning_scale)
__device__ __forceinline__ int grasp_bin(const double* __restrict__ f, const double* mid, const double* origin,
                                         const double* scale) {
    unsigned code = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double w = f[c] + mid[0] * f[3 + c] + mid[1] * f[6 + c] + mid[2] * f[9 + c];      // box centre (unit rows assumed)
        double t = (w - origin[c]) * scale[c];
        t = t >= 0.0 ? t : 0.0;                                                     // NaN lands in bin 0
        const unsigned q = t < 15.0 ? (unsigned)t : 15u;
        code |= part_bits4(q) << c;
    }
    return (int)code;
}

constexpr int ORDER_CACHE = 16384;      // grasps whose bin is kept in LDS between the two passes

__device__ __forceinline__ void grasp_order(const double* __restrict__ frames, int G, const CropLimits& L,
                                            const GridParams* __restrict__ gp, int* __restrict__ perm, int* s_bin,
                                            int* s_wave, unsigned short* s_key) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double mid[3] = {0.5 * (L.xlo + L.xhi), 0.5 * (L.ylo + L.yhi), 0.5 * (L.zlo + L.zhi)};
    double origin[3], scale[3];
    for (int c = 0; c < 3; ++c) {
        origin[c] = gp->origin[c];
        scale[c] = gp->inv_cell * 16.0 / (double)gp->dim[c];
    }
    for (int i = tid; i < ORDER_BINS; i += 1024) s_bin[i] = 0;
    __syncthreads();
    // pass 1: bin of every grasp (kept for pass 2 while it fits), four grasps per thread and step so that their
    // frame loads overlap
    for (int g0 = tid; g0 < G; g0 += 4096) {
        int key[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 1024 * u;
            key[u] = g < G ? grasp_bin(frames + (size_t)g * 12, mid, origin, scale) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = g0 + 1024 * u;
            if (key[u] >= 0) {
                atomicAdd(&s_bin[key[u]], 1);
                if (g < ORDER_CACHE) s_key[g] = (unsigned short)key[u];
            }
        }
    }
    __syncthreads();
    const int c0 = s_bin[4 * tid], c1 = s_bin[4 * tid + 1], c2 = s_bin[4 * tid + 2], c3 = s_bin[4 * tid + 3];
    const int sum = c0 + c1 + c2 + c3;
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_wave[w];
    s_bin[4 * tid] = run;
    s_bin[4 * tid + 1] = run + c0;
    s_bin[4 * tid + 2] = run + c0 + c1;
    s_bin[4 * tid + 3] = run + c0 + c1 + c2;
    __syncthreads();
    for (int g = tid; g < G; g += 1024) {
        const int key = g < ORDER_CACHE ? (int)s_key[g] : grasp_bin(frames + (size_t)g * 12, mid, origin, scale);
        perm[atomicAdd(&s_bin[key], 1)] = g;
    }
}

// One point per thread: its cell, and its rank inside the cell from the counting atomic.  With `perm` given the
// FIRST workgroup instead makes the spatial order of the grasps (grasp_order): it needs only the grid's parameters
// and runs beside the counting workgroups, so the order costs no launch and no time of its own.
__global__ __launch_bounds__(1024) void k_cell_count(const float* __restrict__ scene, int S,
                                                     const GridParams* __restrict__ gp, int2* __restrict__ cell_rank,
                                                     int* __restrict__ hist, const double* __restrict__ frames, int G,
                                                     CropLimits L, int* __restrict__ perm) {
    int block = blockIdx.x;
    if (perm != nullptr) {
        if (block == 0) {
            __shared__ int s_wave[16];
            __shared__ int s_bin[ORDER_BINS];
            __shared__ unsigned short s_key[ORDER_CACHE];
            grasp_order(frames, G, L, gp, perm, s_bin, s_wave, s_key);
            return;
        }
        --block;
    }
    const int s = block * 1024 + threadIdx.x;
    if (s >= S) return;
    const int ix = cell_coord((double)scene[(size_t)s * 3 + 0], gp->origin[0], gp->inv_cell, gp->dim[0]);
    const int iy = cell_coord((double)scene[(size_t)s * 3 + 1], gp->origin[1], gp->inv_cell, gp->dim[1]);
    const int iz = cell_coord((double)scene[(size_t)s * 3 + 2], gp->origin[2], gp->inv_cell, gp->dim[2]);
    const int cell = ix + gp->dim[0] * (iy + gp->dim[1] * iz);
    // one device-scope atomic per point is what this pass costs (~24 G/s); taking the rank from it here spares the
    // regrouping pass a second one
    cell_rank[s] = make_int2(cell, atomicAdd(&hist[cell], 1));
}

// Position in perm[] that workgroup `b` of `n` takes: workgroups are handed to the 8 XCDs in turn, so the
// workgroups b, b + 8, b + 16, ... share an L2; they get one contiguous eighth of the spatial order.
__device__ __forceinline__ int xcd_contiguous(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// exclusive scan of hist over the cells in use into start[0..ncells]; cursor = copy.  One block of 1024 threads
// walks the cells in chunks of 4096 (one int4 per thread: coalesced) and carries the running total.
__global__ __launch_bounds__(1024) void k_scan_cells(const GridParams* __restrict__ gp, const int* __restrict__ hist,
                                                     int* __restrict__ start, int* __restrict__ cursor) {
    __shared__ int s_wave[16];
    const int ncells = gp->dim[0] * gp->dim[1] * gp->dim[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int carry = 0;
    for (int c0 = 0; c0 < ncells; c0 += 4096) {
        const int base = c0 + 4 * tid;
        int4 h = make_int4(0, 0, 0, 0);
        if (base + 3 < ncells) {
            h = *(const int4*)(hist + base);          // hist is 256-byte aligned and base a multiple of 4
        } else {
            if (base < ncells) h.x = hist[base];
            if (base + 1 < ncells) h.y = hist[base + 1];
            if (base + 2 < ncells) h.z = hist[base + 2];
        }
        const int sum = h.x + h.y + h.z + h.w;
        int inc = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
        __syncthreads();                              // s_wave of the previous chunk has been read
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        int before = carry, total = 0;
        for (int w = 0; w < 16; ++w) {
            const int v = s_wave[w];
            if (w < wave) before += v;
            total += v;
        }
        int run = before + inc - sum;
        const int v[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (base + i < ncells) {
                start[base + i] = run;
                cursor[base + i] = run;
                run += v[i];
            }
        carry += total;
    }
    if (tid == 0) start[ncells] = carry;
}

__global__ void k_scatter(const float* __restrict__ scene, int S, const int2* __restrict__ cell_rank,
                          const int* __restrict__ start, float4* __restrict__ sorted) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const int2 cr = cell_rank[s];
    const int pos = start[cr.x] + cr.y;
    sorted[pos] = make_float4(scene[(size_t)s * 3], scene[(size_t)s * 3 + 1], scene[(size_t)s * 3 + 2],
                              __int_as_float(s));
}

// ---------------------------------------------------------------------------------------------
// culled traversal: one wave per grasp
// ---------------------------------------------------------------------------------------------
// Cells the axis-aligned bounding box of the gripper region overlaps, and per frame row r the slack
// |row_r| * (half a cell diagonal): a point of a cell maps to within that slack of the cell centre's image.
struct GraspCells {
    int c0[3], c1[3];
    double slack[3];
};

__device__ __forceinline__ GraspCells grasp_cells(const double* __restrict__ f, const CropLimits& L,
                                                  const GridParams* __restrict__ gp) {
    GraspCells gc;
    // World position of a gripper-frame corner q: origin + M^-1 q, with M the three frame rows.  The rows are
    // orthonormal for the online crop (M^-1 = M^T), but frames given "as they are" (the training-data crop applies
    // a view transform without re-normalising) may carry scale or shear, so the true inverse is used:
    // M^-1 = [b x c | c x a | a x b] / det for rows a, b, c.
    const double a0 = f[3], a1 = f[4], a2 = f[5], b0 = f[6], b1 = f[7], b2 = f[8], c0 = f[9], c1 = f[10], c2 = f[11];
    const double u[3] = {b1 * c2 - b2 * c1, b2 * c0 - b0 * c2, b0 * c1 - b1 * c0};   // b x c
    const double v[3] = {c1 * a2 - c2 * a1, c2 * a0 - c0 * a2, c0 * a1 - c1 * a0};   // c x a
    const double w[3] = {a1 * b2 - a2 * b1, a2 * b0 - a0 * b2, a0 * b1 - a1 * b0};   // a x b
    const double det = a0 * u[0] + a1 * u[1] + a2 * u[2];
    const double inv_det = 1.0 / det;
    const bool singular = !(fabs(det) > 1e-290) || !(fabs(inv_det) < 1e290);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 8; ++k) {
        const double x = (k & 1) ? L.xhi : L.xlo, y = (k & 2) ? L.yhi : L.ylo, z = (k & 4) ? L.zhi : L.zlo;
        for (int c = 0; c < 3; ++c) {
            const double p = f[c] + (x * u[c] + y * v[c] + z * w[c]) * inv_det;
            lo[c] = fmin(lo[c], p);
            hi[c] = fmax(hi[c], p);
        }
    }
    for (int c = 0; c < 3; ++c) {
        const double margin = 1e-6 + 1e-9 * (fabs(lo[c]) + fabs(hi[c]));
        gc.c0[c] = cell_coord(lo[c] - margin, gp->origin[c], gp->inv_cell, gp->dim[c]);
        gc.c1[c] = cell_coord(hi[c] + margin, gp->origin[c], gp->inv_cell, gp->dim[c]);
        if (singular || !(lo[c] <= hi[c])) {            // rows that span no volume (or NaN): visit every cell
            gc.c0[c] = 0;
            gc.c1[c] = gp->dim[c] - 1;
        }
    }
    const double cell = 1.0 / gp->inv_cell;
    const double half_diag = 0.8660254037844387 * cell * (1.0 + 1e-6) + 1e-6;   // + rounding of the cell assignment
    for (int r = 0; r < 3; ++r) {
        const double a = f[3 + 3 * r], b = f[4 + 3 * r], c = f[5 + 3 * r];
        gc.slack[r] = sqrt(a * a + b * b + c * c) * half_diag;
    }
    return gc;
}

// Range [p0, p1) of the cell-sorted points of row (iy, iz) that can hold a hit: lane l looks at cell c0x + l
// (a row has at most GRID_MAX_DIM = 64 cells) and keeps it when the cell centre's image lies within the box
// widened by the slack; the run from the first to the last kept cell is contiguous in the sorted array.
// The grid of a clamped scene is exact at its border cells too: every point lies inside the grid's bounds.
__device__ __forceinline__ bool row_range(const double* __restrict__ f, const CropLimits& L, const GraspCells& gc,
                                          const GridParams* __restrict__ gp, const int* __restrict__ start, int iy,
                                          int iz, int lane, int& p0, int& p1) {
    const int ix = gc.c0[0] + lane;
    const double cell = 1.0 / gp->inv_cell;
    bool keep = false;
    if (ix <= gc.c1[0]) {
        const PointD q = to_gripper(f, gp->origin[0] + (ix + 0.5) * cell, gp->origin[1] + (iy + 0.5) * cell,
                                    gp->origin[2] + (iz + 0.5) * cell);
        keep = (q.x > L.xlo - gc.slack[0]) & (q.x < L.xhi + gc.slack[0]) & (q.y > L.ylo - gc.slack[1]) &
               (q.y < L.yhi + gc.slack[1]) & (q.z > L.zlo - gc.slack[2]) & (q.z < L.zhi + gc.slack[2]);
    }
    const unsigned long long kept = __ballot(keep);
    if (kept == 0ull) return false;
    const int first = __ffsll((long long)kept) - 1, last = 63 - __clzll((long long)kept);
    const int cb = gp->dim[0] * (iy + gp->dim[1] * iz) + gc.c0[0];
    p0 = start[cb + first];
    p1 = start[cb + last + 1];
    return p0 < p1;
}

// The same range found by ONE thread walking the row's cells (at most GRID_MAX_DIM, a handful for a gripper-sized
// box): every thread of a workgroup can then take a row of its own.
__device__ __forceinline__ bool row_range_serial(const double* __restrict__ f, const CropLimits& L, const GraspCells& gc,
                                                 const GridParams* __restrict__ gp, const int* __restrict__ start,
                                                 int iy, int iz, int& p0, int& p1) {
    const double cell = 1.0 / gp->inv_cell;
    const double cy = gp->origin[1] + (iy + 0.5) * cell, cz = gp->origin[2] + (iz + 0.5) * cell;
    int first = -1, last = -1;
    for (int ix = gc.c0[0]; ix <= gc.c1[0]; ++ix) {
        const PointD q = to_gripper(f, gp->origin[0] + (ix + 0.5) * cell, cy, cz);
        const bool keep = (q.x > L.xlo - gc.slack[0]) & (q.x < L.xhi + gc.slack[0]) & (q.y > L.ylo - gc.slack[1]) &
                          (q.y < L.yhi + gc.slack[1]) & (q.z > L.zlo - gc.slack[2]) & (q.z < L.zhi + gc.slack[2]);
        if (keep) {
            if (first < 0) first = ix;
            last = ix;
        }
    }
    if (first < 0) return false;
    const int cb = gp->dim[0] * (iy + gp->dim[1] * iz);
    p0 = start[cb + first];
    p1 = start[cb + last + 1];
    return p0 < p1;
}

// Point ranges of up to 256 rows of cells (rows row0 .. of the grasp's cell box), one row per thread, into
// row_p0 / row_p1 (empty rows get p0 == p1).  The caller places barriers around it.
__device__ __forceinline__ void fill_row_table(const double* __restrict__ f, const CropLimits& L, const GraspCells& gc,
                                               const GridParams* __restrict__ gp, const int* __restrict__ start, int ny,
                                               int nrows, int row0, int tid, int* row_p0, int* row_p1) {
    const int row = row0 + tid;
    int p0 = 0, p1 = 0;
    if (row < nrows && !row_range_serial(f, L, gc, gp, start, gc.c0[1] + row % ny, gc.c0[2] + row / ny, p0, p1))
        p0 = p1 = 0;
    row_p0[tid] = p0;
    row_p1[tid] = p1;
}

// cnt[g] = points inside the gripper region
__global__ __launch_bounds__(256) void k_cull_count(const double* __restrict__ frames, int G, CropLimits L,
                                                    const GridParams* __restrict__ gp, const int* __restrict__ start,
                                                    const float4* __restrict__ sorted, int* __restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const double* f = frames + (size_t)g * 12;
    const GraspCells gc = grasp_cells(f, L, gp);
    int n = 0;
    for (int iz = gc.c0[2]; iz <= gc.c1[2]; ++iz)
        for (int iy = gc.c0[1]; iy <= gc.c1[1]; ++iy) {
            int p0, p1;
            if (!row_range(f, L, gc, gp, start, iy, iz, lane, p0, p1)) continue;
            for (int p = p0; p < p1; p += 64) {           // wave-uniform trip count: ballots stay whole
                bool hit = false;
                if (p + lane < p1) {
                    const float4 v = sorted[p + lane];
                    hit = in_box(to_gripper(f, (double)v.x, (double)v.y, (double)v.z), L);
                }
                n += __popcll(__ballot(hit));
            }
        }
    if (lane == 0) cnt[g] = n;
}

// Ascending sort of list[0..n) by a 256-thread block: a bitonic network whose comparators all point the same way
// (the first step of every merge mirrors the partner), so the missing tail of a non-power-of-two n acts as +inf.
__device__ __forceinline__ void block_sort(int* list, int n, int tid) {
    int log_n2 = 0;
    while ((1 << log_n2) < n) ++log_n2;
    const int half = (1 << log_n2) >> 1;
    for (int lk = 1; lk <= log_n2; ++lk) {
        const int k = 1 << lk;
        for (int lj = lk - 1; lj >= 0; --lj) {
            const int j = 1 << lj;
            const bool mirror = (lj == lk - 1);
            for (int t = tid; t < half; t += 256) {
                const int i = ((t >> lj) << (lj + 1)) | (t & (j - 1));
                const int l = mirror ? ((i & ~(k - 1)) | (k - 1 - (i & (k - 1)))) : i + j;
                if (l < n) {
                    const int a = list[i], b = list[l];
                    if (a > b) {
                        list[i] = b;
                        list[l] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// Ascending sort of n scene indices held in LDS, result in `out`: the indices of one grasp are spread roughly
// evenly between their minimum and maximum, so 256 equal index ranges hold a handful each - histogram, scan,
// scatter, and every thread orders one bucket by insertion.  A bucket of more than 64 (clustered indices) falls
// back to the bitonic network.  Five barriers instead of fifty-five.
__device__ __forceinline__ const int* block_bucket_sort(int* list, int* out, int* s_small, int n, int tid) {
    // s_small: 256 counts | 257 starts | 256 cursors | 8 scratch
    int* hist = s_small;
    int* start = s_small + 256;
    int* cursor = s_small + 513;
    int* red = s_small + 769;
    const int lane = tid & 63, wave = tid >> 6;
    int mn = 0x7fffffff, mx = 0;
    for (int i = tid; i < n; i += 256) {
        const int v = list[i];
        mn = min(mn, v);
        mx = max(mx, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mn = min(mn, __shfl_xor(mn, off));
        mx = max(mx, __shfl_xor(mx, off));
    }
    if (lane == 0) {
        red[wave] = mn;
        red[4 + wave] = mx;
    }
    hist[tid] = 0;
    __syncthreads();
    mn = min(min(red[0], red[1]), min(red[2], red[3]));
    mx = max(max(red[4], red[5]), max(red[6], red[7]));
    int sh = 0;
    while (((mx - mn) >> sh) > 255) ++sh;
    for (int i = tid; i < n; i += 256) atomicAdd(&hist[(list[i] - mn) >> sh], 1);
    __syncthreads();
    if (wave == 0) {                                   // exclusive scan of 256 counts: 4 per lane
        const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
        const int sum = c0 + c1 + c2 + c3;
        int inc = sum, big = max(max(c0, c1), max(c2, c3));
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off);
            if (lane >= off) inc += t;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) big = max(big, __shfl_xor(big, off));
        int run = inc - sum;
        start[4 * lane] = run;     cursor[4 * lane] = run;     run += c0;
        start[4 * lane + 1] = run; cursor[4 * lane + 1] = run; run += c1;
        start[4 * lane + 2] = run; cursor[4 * lane + 2] = run; run += c2;
        start[4 * lane + 3] = run; cursor[4 * lane + 3] = run; run += c3;
        if (lane == 63) start[256] = run;
        if (lane == 0) red[0] = big;
    }
    __syncthreads();
    if (red[0] > 64) {
        block_sort(list, n, tid);
        return list;
    }
    for (int i = tid; i < n; i += 256) {
        const int v = list[i];
        out[atomicAdd(&cursor[(v - mn) >> sh], 1)] = v;
    }
    __syncthreads();
    {
        const int b0 = start[tid], b1 = start[tid + 1];
        for (int i = b0 + 1; i < b1; ++i) {
            const int v = out[i];
            int j = i - 1;
            while (j >= b0 && out[j] > v) {
                out[j + 1] = out[j];
                --j;
            }
            out[j + 1] = v;
        }
    }
    __syncthreads();
    return out;
}

constexpr int CULL_CAP = 4096;               // hits a workgroup sorts in LDS; also the words of its fallback bitmap

// idx[offsets[g] ..] = ascending scene indices of the points inside the gripper region, pts_g = the same points
// in the gripper frame (float64).  One workgroup per grasp; its four waves share the rows of cells.  Up to
// CULL_CAP hits are collected in LDS in the order they are found and sorted there (block_bucket_sort); a grasp
// with more goes over
// the scene's index range in slabs of 32*CULL_CAP indices, marks the hits of a slab in an LDS bitmap and expands
// it - ascending by construction.
__global__ __launch_bounds__(256) void k_cull_emit(const float* __restrict__ scene, int S,
                                                   const double* __restrict__ frames, int G, CropLimits L,
                                                   const GridParams* __restrict__ gp, const int* __restrict__ start,
                                                   const float4* __restrict__ sorted, const int* __restrict__ cnt,
                                                   const long long* __restrict__ offsets, int* __restrict__ idx,
                                                   long long capacity, double* __restrict__ pts_g) {
    __shared__ int list[CULL_CAP];
    __shared__ int s_small[800];
    __shared__ int s_fill;
    __shared__ int s_scan[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x;
    const int n = cnt[g];
    if (n == 0) return;
    const double* f = frames + (size_t)g * 12;
    const GraspCells gc = grasp_cells(f, L, gp);
    const long long base = offsets[g];
    const int ny = gc.c1[1] - gc.c0[1] + 1, nrows = ny * (gc.c1[2] - gc.c0[2] + 1);
    auto write_out = [&](long long o, int s) {
        if (o >= capacity) return;
        idx[o] = s;
        if (pts_g != nullptr) {
            const float* p = scene + (size_t)s * 3;
            const PointD q = to_gripper(f, (double)p[0], (double)p[1], (double)p[2]);
            pts_g[3 * o + 0] = q.x;
            pts_g[3 * o + 1] = q.y;
            pts_g[3 * o + 2] = q.z;
        }
    };
    const unsigned long long below = (1ull << lane) - 1ull;
    int* row_p0 = s_small;                       // [256] point range of a row of cells; free again before the sort
    int* row_p1 = s_small + 256;
    if (n <= CULL_CAP) {
        if (tid == 0) s_fill = 0;
        for (int row0 = 0; row0 < nrows; row0 += 256) {    // one thread per row of cells finds its point range
            __syncthreads();
            fill_row_table(f, L, gc, gp, start, ny, nrows, row0, tid, row_p0, row_p1);
            __syncthreads();
            const int nt = min(256, nrows - row0);
            for (int row = wave; row < nt; row += 4) {
                const int p0 = row_p0[row], p1 = row_p1[row];
                for (int p = p0; p < p1; p += 64) {       // wave-uniform trip count: ballots stay whole
                    bool hit = false;
                    int s = 0;
                    if (p + lane < p1) {
                        const float4 v = sorted[p + lane];
                        hit = in_box(to_gripper(f, (double)v.x, (double)v.y, (double)v.z), L);
                        s = __float_as_int(v.w);
                    }
                    const unsigned long long bal = __ballot(hit);
                    if (bal == 0ull) continue;
                    int at = 0;
                    if (lane == 0) at = atomicAdd(&s_fill, __popcll(bal));
                    at = __shfl(at, 0);
                    const int pos = at + (int)__popcll(bal & below);
                    if (hit && pos < CULL_CAP) list[pos] = s;  // pos < n <= CULL_CAP when cnt is this scene's count
                }
            }
        }
        __syncthreads();
        // up to half the capacity the upper half of the list is the bucket sort's output; beyond, sort in place
        const int* ordered = list;
        if (n <= CULL_CAP / 2)
            ordered = block_bucket_sort(list, list + CULL_CAP / 2, s_small, n, tid);
        else
            block_sort(list, n, tid);
        for (int i = tid; i < n; i += 256) write_out(base + i, ordered[i]);
        return;
    }
    long long o = base;
    for (int lo = 0; lo < S; lo += 32 * CULL_CAP) {
        for (int w = tid; w < CULL_CAP; w += 256) list[w] = 0;
        __syncthreads();
        for (int row0 = 0; row0 < nrows; row0 += 256) {
            fill_row_table(f, L, gc, gp, start, ny, nrows, row0, tid, row_p0, row_p1);
            __syncthreads();
            const int nt = min(256, nrows - row0);
            for (int row = wave; row < nt; row += 4) {
                const int p0 = row_p0[row], p1 = row_p1[row];
                for (int p = p0 + lane; p < p1; p += 64) {
                    const float4 v = sorted[p];
                    const int s = __float_as_int(v.w) - lo;
                    if (s >= 0 && s < 32 * CULL_CAP && in_box(to_gripper(f, (double)v.x, (double)v.y, (double)v.z), L))
                        atomicOr((unsigned int*)&list[s >> 5], 1u << (s & 31));
                }
            }
            __syncthreads();
        }
        for (int w0 = 0; w0 < CULL_CAP; w0 += 256) {
            unsigned int word = (unsigned int)list[w0 + tid];
            const int mine = __popc(word);
            int inc = mine;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(inc, off);
                if (lane >= off) inc += t;
            }
            __syncthreads();                          // s_scan of the previous chunk has been read
            if (lane == 63) s_scan[wave] = inc;
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 4; ++w) {
                if (w < wave) before += s_scan[w];
                total += s_scan[w];
            }
            long long at = o + before + inc - mine;
            while (word) {
                const int b = __ffs((int)word) - 1;
                word &= word - 1;
                write_out(at++, lo + (w0 + tid) * 32 + b);
            }
            o += total;
        }
        __syncthreads();
    }
}

#ifndef PNGPD_FUSED_CAP
#define PNGPD_FUSED_CAP 1536
#endif
constexpr int FUSED_CAP = PNGPD_FUSED_CAP;      // points of one grasp the one-pass kernel holds on chip

// Crop and subsample in one pass, for callers that want the network input and not the index lists: one
// workgroup per grasp collects the hits in LDS (as k_cull_emit does), sorts them, moves the points to the gripper
// frame (float64, rounded to float32) into LDS and writes the R x P draws.  No count pass, no offsets, no index
// list in memory, no host read of the total in between.  The draws index the ascending list, so the output is
// what k_cull_emit + crop_gather_kernel write.  A grasp with more than FUSED_CAP points cannot be held:
// it raises *overflow and leaves its rows unwritten; the caller then takes the three-call path.
__global__ __launch_bounds__(256) void k_cull_gather(const float* __restrict__ scene,
                                                     const double* __restrict__ frames, int G, CropLimits L,
                                                     const GridParams* __restrict__ gp, const int* __restrict__ start,
                                                     const float4* __restrict__ sorted, int R, int P, int min_points,
                                                     unsigned long long seed, int first_grasp_id,
                                                     float* __restrict__ pts_out, unsigned char* __restrict__ valid,
                                                     int* __restrict__ cnt, int* __restrict__ overflow,
                                                     const int* __restrict__ perm) {
    // one 18 KiB buffer: first the hit list (FUSED_CAP indices as found, then as many sorted), later the
    // float32 points - the sorted indices pass through registers in between.  LDS per workgroup decides how many
    // grasps a CU traverses at once, and the traversal is latency-bound.
    __shared__ __attribute__((aligned(16))) char s_buf[FUSED_CAP * 12];
    __shared__ int s_small[800];
    int* list = (int*)s_buf;
    float* s_pts = (float*)s_buf;
    __shared__ int s_fill;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = perm != nullptr ? perm[xcd_contiguous(blockIdx.x, G)] : blockIdx.x;
    const double* f = frames + (size_t)g * 12;
    // the cell bounds are the same for the whole workgroup: one wave works them out (a few hundred vector
    // instructions that the other three would only repeat)
    __shared__ GraspCells s_gc;
    if (wave == 0) {
        const GraspCells mine = grasp_cells(f, L, gp);
        if (lane == 0) {
            s_gc = mine;
            s_fill = 0;
        }
    }
    __syncthreads();
    const GraspCells gc = s_gc;
    const int ny = gc.c1[1] - gc.c0[1] + 1, nrows = ny * (gc.c1[2] - gc.c0[2] + 1);
    const unsigned long long below = (1ull << lane) - 1ull;
    // Rows of cells in tiles of 256: every thread finds the point range of one row (the kernel is bound by its
    // vector instructions, and a wave per row spends 64 lanes on a handful of cells), then the waves share the rows.
    int* row_p0 = s_small;
    int* row_p1 = s_small + 256;
    for (int row0 = 0; row0 < nrows; row0 += 256) {
        __syncthreads();                               // s_fill is set; the previous tile's table has been read
        fill_row_table(f, L, gc, gp, start, ny, nrows, row0, tid, row_p0, row_p1);
        __syncthreads();
        const int nt = min(256, nrows - row0);
        for (int row = wave; row < nt; row += 4) {
            const int p0 = row_p0[row], p1 = row_p1[row];
            for (int p = p0; p < p1; p += 64) {        // wave-uniform trip count: ballots stay whole
                bool hit = false;
                int s = 0;
                if (p + lane < p1) {
                    const float4 v = sorted[p + lane];
                    hit = in_box(to_gripper(f, (double)v.x, (double)v.y, (double)v.z), L);
                    s = __float_as_int(v.w);
                }
                const unsigned long long bal = __ballot(hit);
                if (bal == 0ull) continue;
                int at = 0;
                if (lane == 0) at = atomicAdd(&s_fill, __popcll(bal));
                at = __shfl(at, 0);
                const int pos = at + (int)__popcll(bal & below);
                if (hit && pos < FUSED_CAP) list[pos] = s;
            }
        }
    }
    __syncthreads();
    const int n = s_fill;                              // exact even when the list overflowed
    if (tid == 0) cnt[g] = n;
    float* out = pts_out + (size_t)g * R * P * 3;
    if (n < min_points) {     // apps/kinect2grasp.py:746-751: scored 0 without running the network
        for (int i = tid; i < R * P * 3; i += 256) out[i] = 0.f;
        if (tid == 0) valid[g] = 0;
        return;
    }
    if (tid == 0) valid[g] = 1;
    if (n > FUSED_CAP) {
        if (tid == 0) atomicOr(overflow, 1);
        return;
    }
    const int* ordered = block_bucket_sort(list, list + FUSED_CAP, s_small, n, tid);
    int mine[FUSED_CAP / 256];
#pragma unroll
    for (int k = 0; k < FUSED_CAP / 256; ++k) mine[k] = tid + 256 * k < n ? ordered[tid + 256 * k] : 0;
    __syncthreads();                                   // the index list is in registers: its LDS becomes s_pts
#pragma unroll
    for (int k = 0; k < FUSED_CAP / 256; ++k) {
        const int i = tid + 256 * k;
        if (i < n) {
            const float* p = scene + (size_t)mine[k] * 3;
            const PointD q = to_gripper(f, (double)p[0], (double)p[1], (double)p[2]);
            s_pts[3 * i + 0] = (float)q.x;
            s_pts[3 * i + 1] = (float)q.y;
            s_pts[3 * i + 2] = (float)q.z;
        }
    }
    __syncthreads();
    for (int r = wave; r < R; r += 4) {                 // a wave draws whole repeats (wave_draws)
        Selector pick;
        pick.init(seed, first_grasp_id + g, r, n, P, 0);
        float* o = out + (size_t)r * P * 3;
        wave_draws(pick, P, lane, [&](int i, int j) {
            o[3 * i + 0] = s_pts[3 * j + 0];
            o[3 * i + 1] = s_pts[3 * j + 1];
            o[3 * i + 2] = s_pts[3 * j + 2];
        });
    }
}

size_t crop_grid_bytes(int S) {
    auto r = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t nblk = (size_t)(S + BOUNDS_BLOCK_PTS - 1) / BOUNDS_BLOCK_PTS;
    return r(sizeof(GridParams)) + r(nblk * 6 * sizeof(float)) + r((size_t)S * sizeof(int2)) +
           3 * r(((size_t)GRID_CELLS + 1) * sizeof(int)) + r((size_t)S * sizeof(float4));
}

namespace {
struct GridWs {
    GridParams* gp;
    float* part;
    int2* cell_rank;
    int *hist, *start, *cursor;
    float4* sorted;
    GridWs(void* grid_ws, int S) {
        auto r = [](size_t n) { return (n + 255) & ~(size_t)255; };
        const int nblk = (S + BOUNDS_BLOCK_PTS - 1) / BOUNDS_BLOCK_PTS;
        char* p = (char*)grid_ws;
        gp = (GridParams*)p;          p += r(sizeof(GridParams));
        part = (float*)p;             p += r((size_t)nblk * 6 * sizeof(float));
        cell_rank = (int2*)p;         p += r((size_t)S * sizeof(int2));
        hist = (int*)p;               p += r(((size_t)GRID_CELLS + 1) * sizeof(int));
        start = (int*)p;              p += r(((size_t)GRID_CELLS + 1) * sizeof(int));
        cursor = (int*)p;             p += r(((size_t)GRID_CELLS + 1) * sizeof(int));
        sorted = (float4*)p;
    }
};
}  // namespace

// Bins the scene: bounds, cell of every point, prefix over the cells, points regrouped by cell.
// With `perm`, the spatial order of the G grasps (frames) is produced on the way.
// `clear_flag` (optional) is set to 0 on the way.
static hipError_t build_grid(const float* scene, int S, const CropLimits& L, const GridWs& w, hipStream_t s,
                             const double* frames = nullptr, int G = 0, int* perm = nullptr, int* clear_flag = nullptr) {
    const int nblk = (S + BOUNDS_BLOCK_PTS - 1) / BOUNDS_BLOCK_PTS;
    const double ext_x = L.xhi - L.xlo, ext_y = L.yhi - L.ylo, ext_z = L.zhi - L.zlo;
    // cells of a quarter of the region's longest side (the scene's extent / 63 if that is larger): smaller cells
    // test fewer points but visit more rows, and a row costs a dependent pair of loads (measured at S = 1M,
    // G = 10k: count + emit + grid build 0.72 / 0.51 / 0.46 / 0.50 ms for 1/8, 1/5, 1/4, 1/3)
    double cell_req = fmax(ext_x, fmax(ext_y, ext_z)) * 0.25;
    if (!(cell_req > 0.0)) cell_req = 1e-3;
    hipLaunchKernelGGL(k_bounds_partial, dim3(nblk), dim3(256), 0, s, scene, S, w.part, w.hist);
    hipLaunchKernelGGL(k_grid_setup, dim3(1), dim3(256), 0, s, (const float*)w.part, nblk, cell_req, w.gp, clear_flag);
    hipLaunchKernelGGL(k_cell_count, dim3((S + 1023) / 1024 + (perm != nullptr ? 1 : 0)), dim3(1024), 0, s, scene, S,
                       (const GridParams*)w.gp, w.cell_rank, w.hist, frames, G, L, perm);
    hipLaunchKernelGGL(k_scan_cells, dim3(1), dim3(1024), 0, s, (const GridParams*)w.gp, (const int*)w.hist, w.start,
                       w.cursor);
    hipLaunchKernelGGL(k_scatter, dim3((S + 255) / 256), dim3(256), 0, s, scene, S, (const int2*)w.cell_rank,
                       (const int*)w.start, w.sorted);
    return hipGetLastError();
}

// Bins the scene and counts, per grasp, the points inside the gripper region.  `grid_ws` holds
// crop_grid_bytes(S) bytes and keeps the binned scene for launch_crop_emit_grid.
hipError_t launch_crop_count_grid(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                  void* grid_ws, int* cnt, hipStream_t s) {
    GridWs w(grid_ws, S);
    hipError_t e = build_grid(scene, S, L, w, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cull_count, dim3((G + 3) / 4), dim3(256), 0, s, frames, G, L, (const GridParams*)w.gp,
                       (const int*)w.start, (const float4*)w.sorted, cnt);
    return hipGetLastError();
}

// Bins the scene, then crops and subsamples every grasp in one kernel (k_cull_gather).
hipError_t launch_crop_gather_fused(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                    void* grid_ws, int R, int P, int min_points, unsigned long long seed,
                                    int first_grasp_id, float* pts_out, unsigned char* valid, int* cnt, int* overflow,
                                    int* perm, hipStream_t s) {
    if (!scene || !frames || !grid_ws || !pts_out || !valid || !cnt || !overflow || S < 1 || G < 1 || R < 1 || P < 1)
        return hipErrorInvalidValue;
    static const bool ordered = [] { const char* v = getenv("PNGPD_CROP_ORDER"); return !(v && v[0] == '0'); }();
    if (!ordered || G < 64) perm = nullptr;            // a handful of grasps: nothing to keep apart
    GridWs w(grid_ws, S);
    hipError_t e = build_grid(scene, S, L, w, s, frames, G, perm, overflow);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_cull_gather, dim3(G), dim3(256), 0, s, scene, frames, G, L, (const GridParams*)w.gp,
                       (const int*)w.start, (const float4*)w.sorted, R, P, min_points, seed, first_grasp_id, pts_out,
                       valid, cnt, overflow, (const int*)perm);
    return hipGetLastError();
}

hipError_t launch_crop_emit_grid(const float* scene, int S, const double* frames, int G, const CropLimits& L,
                                 const void* grid_ws, const int* cnt, const long long* offsets, int* idx,
                                 long long capacity, double* pts_g, hipStream_t s) {
    GridWs w((void*)grid_ws, S);
    hipLaunchKernelGGL(k_cull_emit, dim3(G), dim3(256), 0, s, scene, S, frames, G, L,
                       (const GridParams*)w.gp, (const int*)w.start, (const float4*)w.sorted, cnt, offsets, idx,
                       capacity, pts_g);
    return hipGetLastError();
}

}  // namespace pngpd

// Voxel down-sampling of a scene to the unique centres of its occupied voxels:
//   get_voxel_fun(points, n)          apps/kinect2grasp.py:145-150
//   VoxelGrid(points, n, n, n).compute()   apps/voxelgrid.py:115-160
//
// The reference builds n+1 bin edges per axis with np.linspace over the bounding box made cubic,
// finds each point's bin with np.searchsorted(edges, x) - 1, looks up the float32 bin centres and
// keeps np.unique(rows): the unique centre triples in lexicographic order.
//
// Here: a float32 min/max reduction, one small setup block that redoes the reference's scalar
// arithmetic operation by operation (float32 box, edges = i*step + start with two roundings,
// float32 centres), then one pass over the points that sets a bit per occupied voxel,
// and a popcount / scan / expand pass over the bitmap.  Ascending bit order is the lexicographic
// order of the centres, so nothing is sorted.  Two voxels whose float32 centres coincide on an
// axis are the same row to np.unique; each axis index is therefore first mapped to the lowest
// index with the same float32 centre.
// The type of the edges is numpy's choice, not the reference's: np.linspace on two float32 scalars works in
// float64 under numpy 1.x (value-based promotion with its Python-float factor) and in float32 under numpy >= 2.
// Both are implemented (f32_edges); they differ in the step and in the two roundings of every edge.
// HBM-bound: S*12 bytes of points, n^3/8 bytes of bitmap written once and read twice.
#include <hip/hip_runtime.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace pngpd {

namespace {

constexpr int VX_THREADS = 256;
constexpr int VX_RED_BLOCKS = 512;
constexpr int VX_WORDS_PER_BLOCK = 1024;     // bitmap words per block of the count / emit passes

struct VoxelParams {
    double start[3], stop[3], step[3];
    double inv_step[3];   // 1 / step (0 where step is 0): only for the first guess of a bin, which is then made exact
    int n;
    int f32_edges;   // 1: edges formed in float32 (numpy >= 2); start / stop / step then hold float32 values
};

struct VoxelWs {
    VoxelParams* prm;
    float* partial;          // [VX_RED_BLOCKS][6]
    float* mid;              // [3][n] float32 bin centres
    int* canon;              // [3][n] lowest index with the same float32 centre
    unsigned int* blocksum;  // [nblocks + 1]
    unsigned int* bitmap;    // [words]
};

__host__ __device__ inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

inline size_t vx_words(int n) { return ((size_t)n * n * n + 31) / 32; }
inline size_t vx_blocks(int n) { return (vx_words(n) + VX_WORDS_PER_BLOCK - 1) / VX_WORDS_PER_BLOCK; }

VoxelWs carve(void* ws, int n) {
    char* p = (char*)ws;
    VoxelWs w;
    w.prm = (VoxelParams*)p;
    p += align256(sizeof(VoxelParams));
    w.partial = (float*)p;
    p += align256((size_t)VX_RED_BLOCKS * 6 * 4);
    w.mid = (float*)p;
    p += align256((size_t)3 * n * 4);
    w.canon = (int*)p;
    p += align256((size_t)3 * n * 4);
    w.blocksum = (unsigned int*)p;
    p += align256((vx_blocks(n) + 1) * 4);
    w.bitmap = (unsigned int*)p;
    return w;
}

// min / max of the points; on the way the occupancy bitmap is cleared (16-byte stores spread over the same
// workgroups: a fill of its own is one more launch in front of a chain of dependent launches)
__global__ __launch_bounds__(VX_THREADS) void k_vox_minmax(const float* pts, int S, float* partial, uint4* bitmap4,
                                                           size_t quads) {
    for (size_t i = blockIdx.x * (size_t)VX_THREADS + threadIdx.x; i < quads; i += (size_t)gridDim.x * VX_THREADS)
        bitmap4[i] = make_uint4(0u, 0u, 0u, 0u);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long long i = blockIdx.x * (long long)VX_THREADS + threadIdx.x; i < S; i += (long long)gridDim.x * VX_THREADS) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = pts[i * 3 + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
    __shared__ float s[VX_THREADS / 64][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
        if (lane == 0) {
            s[wave][a] = lo[a];
            s[wave][3 + a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s[0][threadIdx.x];
        for (int w = 1; w < VX_THREADS / 64; ++w) v = threadIdx.x < 3 ? fminf(v, s[w][threadIdx.x]) : fmaxf(v, s[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = v;
    }
}

// Bin edge i of np.linspace(start, stop, n + 1): arange * step, then + start, last one = stop.
__device__ __forceinline__ double vox_edge(const VoxelParams& p, int a, int i) {
    if (i >= p.n) return p.stop[a];
    if (p.step[a] == 0.0) return p.start[a];
    if (p.f32_edges) {
        const float t = (float)i * (float)p.step[a];
        return (double)(t + (float)p.start[a]);
    }
    const double t = (double)i * p.step[a];
    return t + p.start[a];
}

// The grid of a call from the min / max records: float32 cubic box, edges, float32 bin centres `mid[3][n]` and, per axis,
// the lowest index with the same float32 centre `canon[3][n]`.  Run by EVERY workgroup of the marking pass into its own
// LDS (a launch of one workgroup for it cost 6.7 us in front of a chain of dependent launches; here it is ~1 us of a
// pass that waits for its atomics anyway); workgroup 0 also stores the centres for the emit pass.
__device__ __forceinline__ void vox_setup_block(const float* partial, int nparts, int n, int f32_edges, VoxelParams& s_p,
                                                float (&s_red)[VX_THREADS / 64][6], float (&s_mm)[6], float* s_mid,
                                                int* s_canon) {
    {
        // all threads share the partial records (a serial walk by six threads is 512 dependent loads)
        float v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = k < 3 ? INFINITY : -INFINITY;
        for (int b = threadIdx.x; b < nparts; b += VX_THREADS)
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float x = partial[b * 6 + k];
                v[k] = k < 3 ? fminf(v[k], x) : fmaxf(v[k], x);
            }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float o = __shfl_xor(v[k], off);
                v[k] = k < 3 ? fminf(v[k], o) : fmaxf(v[k], o);
            }
            if (lane == 0) s_red[wave][k] = v[k];
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            float r = s_red[0][threadIdx.x];
            for (int w = 1; w < VX_THREADS / 64; ++w)
                r = threadIdx.x < 3 ? fminf(r, s_red[w][threadIdx.x]) : fmaxf(r, s_red[w][threadIdx.x]);
            s_mm[threadIdx.x] = r;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // float32, as numpy does on float32 points: the box is grown to a cube around itself
        float d[3], m = -INFINITY;
        for (int a = 0; a < 3; ++a) {
            d[a] = s_mm[3 + a] - s_mm[a];
            m = fmaxf(m, d[a]);
        }
        for (int a = 0; a < 3; ++a) {
            const float margin = m - d[a];
            const float half = margin / 2.0f;
            const float lo = s_mm[a] - half, hi = s_mm[3 + a] + half;
            s_p.start[a] = (double)lo;
            s_p.stop[a] = (double)hi;
            if (f32_edges) {
                // numpy >= 2: np.linspace on two float32 scalars stays in float32 (delta, step and every edge)
                const float delta = hi - lo;
                s_p.step[a] = (double)(delta / (float)n);
            } else {
                // numpy 1.x: float64 from here on (two float32 scalars and a Python float)
                const double delta = (double)hi - (double)lo;
                s_p.step[a] = delta / (double)n;
            }
        }
        for (int a = 0; a < 3; ++a) s_p.inv_step[a] = s_p.step[a] != 0.0 ? 1.0 / s_p.step[a] : 0.0;
        s_p.n = n;
        s_p.f32_edges = f32_edges;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * n; t += VX_THREADS) {
        const int a = t / n, i = t - a * n;
        const double e0 = vox_edge(s_p, a, i), e1 = vox_edge(s_p, a, i + 1);
        s_mid[t] = (float)((e1 + e0) / 2.0);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 3 * n; t += VX_THREADS) {
        const int a = t / n;
        int i = t - a * n;
        const float v = s_mid[t];
        while (i > 0 && s_mid[a * n + i - 1] == v) --i;      // centres are non-decreasing along an axis
        s_canon[t] = i;
    }
    __syncthreads();
}

// np.clip(np.searchsorted(edges, x) - 1, 0, n): searchsorted (side 'left') counts the edges < x.
__device__ __forceinline__ int vox_bin(const VoxelParams& p, int a, float xf) {
    const double x = (double)xf;
    const int n = p.n;
    if (p.step[a] == 0.0) return 0;
    // first guess of the count (a multiplication: the two loops below make it exact whatever the guess), then exact
    const double g = (x - p.start[a]) * p.inv_step[a];
    int c = g < 0.0 ? 0 : (g > (double)n ? n : (int)g);
    while (c <= n && vox_edge(p, a, c) < x) ++c;
    while (c > 0 && !(vox_edge(p, a, c - 1) < x)) --c;
    int idx = c - 1;
    idx = idx < 0 ? 0 : idx;
    return idx > n - 1 ? n - 1 : idx;   // x <= stop = the last edge, so the reference never exceeds n-1 either
}

__global__ __launch_bounds__(VX_THREADS) void k_vox_mark(const float* pts, int S, const float* partial, int nparts, int n,
                                                         int f32_edges, float* mid, unsigned int* bitmap) {
    extern __shared__ __attribute__((aligned(16))) char vx_smem[];       // [3n] float centres, [3n] int canon
    __shared__ VoxelParams s_p;
    __shared__ float s_red[VX_THREADS / 64][6];
    __shared__ float s_mm[6];
    float* s_mid = (float*)vx_smem;
    int* s_canon = (int*)(vx_smem + (size_t)3 * n * 4);
    vox_setup_block(partial, nparts, n, f32_edges, s_p, s_red, s_mm, s_mid, s_canon);
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < 3 * n; t += VX_THREADS) mid[t] = s_mid[t];
    for (long long i = blockIdx.x * (long long)VX_THREADS + threadIdx.x; i < S; i += (long long)gridDim.x * VX_THREADS) {
        const int ix = s_canon[vox_bin(s_p, 0, pts[i * 3 + 0])];
        const int iy = s_canon[n + vox_bin(s_p, 1, pts[i * 3 + 1])];
        const int iz = s_canon[2 * n + vox_bin(s_p, 2, pts[i * 3 + 2])];
        const unsigned long long key = ((unsigned long long)ix * n + iy) * n + iz;
        atomicOr(&bitmap[key >> 5], 1u << (key & 31));
    }
}

__global__ __launch_bounds__(VX_THREADS) void k_vox_count(const unsigned int* bitmap, size_t words, unsigned int* blocksum) {
    const size_t base = (size_t)blockIdx.x * VX_WORDS_PER_BLOCK;
    unsigned int c = 0;
    for (int j = threadIdx.x; j < VX_WORDS_PER_BLOCK; j += VX_THREADS)
        if (base + j < words) c += __popc(bitmap[base + j]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    __shared__ unsigned int s[VX_THREADS / 64];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int w = 0; w < VX_THREADS / 64; ++w) t += s[w];
        blocksum[blockIdx.x] = t;
    }
}

struct __attribute__((packed, aligned(4))) VoxRow {
    float x, y, z;
};

// Rows of the set bits in ascending bit order.  A workgroup's first output row is the number of set bits in the
// workgroups before it, which it adds up from their counts itself (blocksum, 3,815 entries at n = 500: on average
// half of them, eight loads per thread - a scan launch between the count and this pass took 5 us and a launch gap; the
// scan folded into the count kernel, the block that finishes last scanning, was slower still: thousands of blocks
// bumping ONE device-scope counter serialise at the memory side).  The last workgroup also stores the total.
__global__ __launch_bounds__(VX_THREADS) void k_vox_emit(const unsigned int* bitmap, size_t words,
                                                         const unsigned int* blocksum, const float* mid, int n,
                                                         float* out, long long capacity, int* count) {
    // each thread owns 4 consecutive words: ascending bit order is preserved
    constexpr int WPT = VX_WORDS_PER_BLOCK / VX_THREADS;
    const size_t base = (size_t)blockIdx.x * VX_WORDS_PER_BLOCK + (size_t)threadIdx.x * WPT;
    unsigned int w[WPT], c = 0;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        w[j] = base + j < words ? bitmap[base + j] : 0u;
        c += __popc(w[j]);
    }
    unsigned int pre = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += VX_THREADS) pre += blocksum[i];
    // exclusive prefix of c over the block: shuffles inside a wave, one barrier for the wave totals (and for the sums
    // of the earlier workgroups' counts)
    __shared__ unsigned int s_wave[VX_THREADS / 64], s_pre[VX_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pre += __shfl_xor(pre, off);
    if (lane == 63) s_wave[wave] = inc;
    if (lane == 0) s_pre[wave] = pre;
    __syncthreads();
    unsigned int before = 0, first = 0, total = 0;
    for (int k = 0; k < VX_THREADS / 64; ++k) {
        if (k < wave) before += s_wave[k];
        first += s_pre[k];
        total += s_wave[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *count = (int)(first + total);
    long long pos = (long long)first + before + inc - c;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
        unsigned int bits = w[j];
        if (bits == 0u) continue;
        // voxel of the word's first bit, by 32-bit divisions once per word (n <= VOXEL_MAX_N keeps every key below
        // 2^32); the voxel of a set bit follows by stepping along z with carries.  (One 64-bit division pair per set
        // bit made this kernel 24 us at a million points: the divisions, not the memory, were its cost.)
        const unsigned int key0 = (unsigned int)(base + j) * 32u, un = (unsigned int)n;
        const unsigned int r0 = key0 / un;
        int iz0 = (int)(key0 - r0 * un), iy0 = (int)(r0 % un), ix0 = (int)(r0 / un);
        int prev = 0;
        while (bits) {
            const int b = __ffs(bits) - 1;
            bits &= bits - 1;
            iz0 += b - prev;
            prev = b;
            while (iz0 >= n) {
                iz0 -= n;
                if (++iy0 == n) {
                    iy0 = 0;
                    ++ix0;
                }
            }
            // a row is one 12-byte store (working the centres out here instead of looking them up is slower: 24 us
            // against 20 at a million points)
            if (pos < capacity) *(VoxRow*)(out + pos * 3) = VoxRow{mid[ix0], mid[n + iy0], mid[2 * n + iz0]};
            ++pos;
        }
    }
}

}  // namespace

size_t voxel_workspace_bytes(int n) {
    return align256(sizeof(VoxelParams)) + align256((size_t)VX_RED_BLOCKS * 6 * 4) + 2 * align256((size_t)3 * n * 4) +
           align256((vx_blocks(n) + 1) * 4) + align256(vx_words(n) * 4);      // the bitmap: a whole number of 16-byte quads
}

hipError_t launch_voxel_unique(const float* pts, int S, int n, int f32_edges, void* ws, float* out, long long capacity,
                               int* count, hipStream_t s) {
    const VoxelWs w = carve(ws, n);
    const size_t words = vx_words(n);
    const int nb = (int)vx_blocks(n);
    // the workspace is caller memory of at least voxel_workspace_bytes(n), whose bitmap region is rounded up to 256 bytes
    const size_t quads = (words * 4 + 15) / 16;
    int red_blocks = VX_RED_BLOCKS;          // also clears the bitmap: always the full grid
    hipLaunchKernelGGL(k_vox_minmax, dim3(red_blocks), dim3(VX_THREADS), 0, s, pts, S, w.partial, (uint4*)w.bitmap, quads);
    // the marking pass derives the grid itself (vox_setup_block); fewer, longer workgroups than one per 256 points keep
    // that prologue a small share of the pass
    int mark_blocks = (S + VX_THREADS - 1) / VX_THREADS;
    mark_blocks = mark_blocks < 1024 ? mark_blocks : 1024;
    hipLaunchKernelGGL(k_vox_mark, dim3(mark_blocks), dim3(VX_THREADS), (size_t)3 * n * 8, s, pts, S, w.partial, red_blocks, n,
                       f32_edges, w.mid, w.bitmap);
    hipLaunchKernelGGL(k_vox_count, dim3(nb), dim3(VX_THREADS), 0, s, w.bitmap, words, w.blocksum);
    hipLaunchKernelGGL(k_vox_emit, dim3(nb), dim3(VX_THREADS), 0, s, w.bitmap, words, w.blocksum, w.mid, n, out, capacity,
                       count);
    return hipGetLastError();
}

}  // namespace pngpd

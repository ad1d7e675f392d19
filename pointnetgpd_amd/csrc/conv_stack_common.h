// Device helpers shared by the conv-stack kernels (conv_stack.hip, conv_stack_x3.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "pngpd.h"

namespace pngpd {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

// One wave-instruction of LDS-DMA: 64 lanes x 16 B, LDS destination = lds_piece + lane*16.
__device__ __forceinline__ void dma_piece(const char* gsrc_piece, char* lds_piece, int lane) {
    __builtin_amdgcn_global_load_lds((const void*)(gsrc_piece + lane * 16), (lds_void*)lds_piece,
                                     16, 0, 0);
}

__device__ __forceinline__ float max3(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}

typedef f32x4 f32x4v;

// Running maximum over one accumulator, as two v_max3.  The empty asm makes the result opaque: in an unrolled batch
// the compiler otherwise re-associates the maxima of all tiles into one tree at the end of the batch and keeps
// every accumulator alive (in scratch memory) until then.
__device__ __forceinline__ float pinned_max(float m, const f32x4v& v) {
    m = max3(m, max3(v[0], v[1], v[2]), v[3]);
    asm volatile("" : "+v"(m));
    return m;
}

// Orders this wave's own LDS accesses (a wave-private table is written, later read, by one wave).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// relu + bf16 pack of 8 accumulator values.  Rounding keeps the sign, so the clamp can follow the
// rounding, where it is a packed int16 max against 0; the conversions are written two at a time
// so that each is one v_cvt_pk_bf16_f32 (8 values: 4 conversions + 4 packed max).
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned int relu_pack2(float a, float b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    s16x2 v = __builtin_bit_cast(s16x2, h);
    v = __builtin_elementwise_max(v, (s16x2)(0));
    return __builtin_bit_cast(unsigned int, v);
}

__device__ __forceinline__ bf16x8 relu_pack(const f32x4v& lo, const f32x4v& hi) {
    u32x4 r;
    r[0] = relu_pack2(lo[0], lo[1]);
    r[1] = relu_pack2(lo[2], lo[3]);
    r[2] = relu_pack2(hi[0], hi[1]);
    r[3] = relu_pack2(hi[2], hi[3]);
    return __builtin_bit_cast(bf16x8, r);
}

__device__ __forceinline__ unsigned int f2u(float v) { return __builtin_bit_cast(unsigned int, v); }
__device__ __forceinline__ float u2f(unsigned int v) { return __builtin_bit_cast(float, v); }

// bf16 bit patterns (upper halves) of the three pieces of v
__device__ __forceinline__ void split3(float v, unsigned int& h, unsigned int& m, unsigned int& l) {
    h = f2u(v) & 0xffff0000u;
    const float r1 = v - u2f(h);
    m = f2u(r1) & 0xffff0000u;
    l = f2u(r1 - u2f(m));                      // at most 8 significant bits are left: exact as bf16
}

constexpr int W1F_BYTES = 4 * 64 * 16;            // [tile 2s+hl][q][row] x 16 B: the layer-1 A fragments of one candidate

// Coordinate q of point c of the two tiles a wave produces next (lanes with q = 3 hold 0).
struct TilePoints {
    float x[2];
};

__device__ __forceinline__ void load_tile_points(const ConvStackArgs& a, int b, int tile0, int c, int q, TilePoints& p) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int pt = (tile0 + u) * 16 + c;
        pt = pt < a.P ? pt : a.P - 1;
        const size_t at = a.layout == PNGPD_LAYOUT_C3P ? ((size_t)b * 3 + q) * a.P + pt : ((size_t)b * a.P + pt) * 3 + q;
        p.x[u] = q < 3 ? a.x[at] : 0.f;
    }
}

// The layer-1 A fragments of candidate b (model/pointnet.py:138-142: the 3x3 input transform is applied to the
// points before conv1; here it is applied to conv1's weights).  One wave; lane = channel.
__device__ __forceinline__ void fold_w1_frags(const ConvStackArgs& a, int b, const float4* s_w1base, char* tab, int lane) {
    float4 w = s_w1base[lane];
    if (a.trans != nullptr) {
        const float* T = a.trans + (size_t)b * 9;
        float4 f;
        f.x = w.x * T[0] + w.y * T[1] + w.z * T[2];
        f.y = w.x * T[3] + w.y * T[4] + w.z * T[5];
        f.z = w.x * T[6] + w.y * T[7] + w.z * T[8];
        f.w = w.w;
        w = f;
    }
    // channel 32s + 8q' + 4hl + i is row 4q' + i of tile 2s + hl
    const int t = 2 * (lane >> 5) + ((lane >> 2) & 1), r = 4 * ((lane & 31) >> 3) + (lane & 3);
    unsigned int bh, bm, bl;
    split3(w.w, bh, bm, bl);
    const float comp[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        unsigned int h, m, l;
        split3(comp[d], h, m, l);
        u32x4 v;
        v[0] = (h >> 16) | m;                  // slots 0, 1: wh, wm
        v[1] = (h >> 16) | l;                  //       2, 3: wh, wl
        v[2] = (m >> 16) | h;                  //       4, 5: wm, wh
        v[3] = d == 0 ? ((bh >> 16) | bm) : d == 1 ? (bl >> 16) : 0u;
        *(u32x4*)(tab + ((t * 4 + d) * 16 + r) * 16) = v;
    }
}

}  // namespace pngpd

// float64 frame transform and box test shared by the crop kernels.  The operation order is the
// one numpy/OpenBLAS execute for check_collision_square (apps/kinect2grasp.py:313-317, 332-337):
// d = p - c;  pg[r] = fma(R[r][2], d2, fma(R[r][1], d1, R[r][0]*d0));  six strict comparisons.
// Contraction of separate multiplies and adds is off for every file that includes this header.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

#pragma clang fp contract(off)

namespace pngpd {

struct PointD { double x, y, z; };

// f: centre[3], approach[3], binormal[3], minor[3] (unit axes)
__device__ __forceinline__ PointD to_gripper(const double* __restrict__ f, double px, double py, double pz) {
    const double d0 = px - f[0], d1 = py - f[1], d2 = pz - f[2];
    PointD q;
    q.x = fma(f[5], d2, fma(f[4], d1, f[3] * d0));
    q.y = fma(f[8], d2, fma(f[7], d1, f[6] * d0));
    q.z = fma(f[11], d2, fma(f[10], d1, f[9] * d0));
    return q;
}

__device__ __forceinline__ bool in_box(const PointD& q, const CropLimits& L) {
    return (L.ylo < q.y) & (L.yhi > q.y) & (L.zhi > q.z) & (L.zlo < q.z) & (L.xhi > q.x) & (L.xlo < q.x);
}

}  // namespace pngpd

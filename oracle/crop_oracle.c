/*
 * TEST INFRASTRUCTURE ONLY - plain-C restatement of the in-gripper crop of the reference
 * (apps/kinect2grasp.py:299-344, check_collision_square, way="p_open") with the floating-point
 * operation order written out.  Compiled with -ffp-contract=off so that only the fma() calls
 * below fuse.  Only tests/, smoke() and bench.py's cpu_baseline leg use it.
 *
 * Order pinned (verified bit-for-bit against numpy 2.2 / OpenBLAS 0.3.29 in the build container,
 * tests/test_crop_oracle.py):
 *   |v|   = sqrt(fma(vz, vz, fma(vy, vy, vx*vx)))              np.linalg.norm   (:303,305,307)
 *   axis  = v / |v|                                                              (:303,305,307)
 *   d     = (double)point - centre                                               (:313)
 *   pg[r] = fma(R[r][2], d[2], fma(R[r][1], d[1], R[r][0]*d[0]))   np.dot        (:316)
 *   inside: six STRICT comparisons                                               (:332-337)
 */
#include <math.h>
#include <stdint.h>

void oracle_fma(const double* a, const double* b, const double* c, double* out, long n) {
    for (long i = 0; i < n; ++i) out[i] = fma(a[i], b[i], c[i]);
}

/* frame: centre[3], approach[3], binormal[3], minor[3] (axes not yet normalised).
 * limits: ylo, yhi, zlo, zhi, xlo, xhi.  Writes ascending indices and their gripper-frame
 * coordinates; returns the count. */
long oracle_crop_one(const float* scene, long S, const double* frame, const double* limits,
                     int64_t* idx_out, double* pts_out) {
    double R[3][3];
    for (int r = 0; r < 3; ++r) {
        const double* v = frame + 3 + 3 * r;
        const double n = sqrt(fma(v[2], v[2], fma(v[1], v[1], v[0] * v[0])));
        for (int c = 0; c < 3; ++c) R[r][c] = v[c] / n;
    }
    const double ylo = limits[0], yhi = limits[1], zlo = limits[2], zhi = limits[3], xlo = limits[4],
                 xhi = limits[5];
    long n = 0;
    for (long s = 0; s < S; ++s) {
        const double d0 = (double)scene[3 * s + 0] - frame[0];
        const double d1 = (double)scene[3 * s + 1] - frame[1];
        const double d2 = (double)scene[3 * s + 2] - frame[2];
        double pg[3];
        for (int r = 0; r < 3; ++r) pg[r] = fma(R[r][2], d2, fma(R[r][1], d1, R[r][0] * d0));
        if (ylo < pg[1] && yhi > pg[1] && zhi > pg[2] && zlo < pg[2] && xhi > pg[0] && xlo < pg[0]) {
            idx_out[n] = s;
            pts_out[3 * n + 0] = pg[0];
            pts_out[3 * n + 1] = pg[1];
            pts_out[3 * n + 2] = pg[2];
            ++n;
        }
    }
    return n;
}

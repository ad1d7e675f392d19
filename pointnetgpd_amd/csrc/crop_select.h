// Counter-based selection of the points a candidate is made of, shared by the gather kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace pngpd {

constexpr int GATHER_LDS_POINTS = 2048;      // points of one grasp held in LDS by the gather kernels

// ---- repo-defined counter-based selection (the reference draws from numpy's global, time-seeded
// generator, apps/kinect2grasp.py:768-774, which cannot be reproduced) ---------------------------
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Draws for (grasp g, repeat r) out of n cropped points, P draws in total.
// n >= P: i -> a keyed 4-round Feistel permutation of [0, 4^hb) cycle-walked into [0, n): the P
//         draws are distinct (sampling without replacement);  n < P: independent uniform draws.
// replace_when_equal: n == P is drawn WITH replacement (the dataset's `len > N` rule,
// model/dataset.py:1104-1111) instead of without (the online `len >= N` rule, apps/kinect2grasp.py:756).
// The per-(g, r) part (key, four 32-bit round keys) is set up once; a draw then costs 32-bit arithmetic only
// (the round function is a 2-multiply 32-bit mixer): the gather kernel is bound by this arithmetic.
__host__ __device__ __forceinline__ unsigned int mix32(unsigned int x) {
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    x *= 0x297a2d39u;
    x ^= x >> 15;
    return x;
}

struct Selector {
    unsigned long long key;
    unsigned int rk[4], m;
    int n, hb;
    bool with_replacement;

    __host__ __device__ __forceinline__ void init(unsigned long long seed, int g, int r, int n_, int P,
                                                  int replace_when_equal) {
        key = mix64(seed ^ mix64((unsigned long long)g * 0x9E3779B97F4A7C15ull + (unsigned long long)r));
        n = n_;
        with_replacement = n < P + replace_when_equal;
        hb = 1;
        while ((1ll << (2 * hb)) < (long long)n) ++hb;
        m = (1u << hb) - 1u;
        for (int t = 0; t < 4; ++t) rk[t] = (unsigned int)(mix64(key + ((unsigned long long)(t + 1) << 32)) >> 32);
    }

    __host__ __device__ __forceinline__ int draw(int i) const {
        if (with_replacement) {
            const unsigned long long u = mix64(key + (unsigned long long)i) >> 32;
            return (int)((u * (unsigned long long)n) >> 32);
        }
        unsigned int x = (unsigned int)i;
        do {
            unsigned int Lh = x >> hb, Rh = x & m;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned int F = mix32(Rh ^ rk[t]) & m;
                const unsigned int nl = Rh;
                Rh = Lh ^ F;
                Lh = nl;
            }
            x = (Lh << hb) | Rh;
        } while (x >= (unsigned int)n);
        return (int)x;
    }
};

}  // namespace pngpd

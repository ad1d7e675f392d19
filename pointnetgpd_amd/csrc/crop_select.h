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
// n >= P: sampling without replacement.  pi is a keyed 4-round unbalanced Feistel permutation of [0, 2^b), b =
//         ceil(log2 n) (at least 2): the draws are the first P numbers below n in the sequence pi(0), pi(1), ... -
//         distinct by construction.  A wave finds them by scanning 64 indices at a time and compacting the ones below
//         n with a ballot (wave_draws): every lane evaluates pi once per index scanned, about P * 2^b / n < 2P
//         evaluations per repeat.  (Cycle-walking each draw separately, x = pi(x) until x < n, makes all 64 lanes of
//         a wave wait for the unluckiest one: about four evaluations per draw.)
// n < P:  independent uniform draws, one 32-bit hash each.
// replace_when_equal: n == P is drawn WITH replacement (the dataset's `len > N` rule,
// model/dataset.py:1104-1111) instead of without (the online `len >= N` rule, apps/kinect2grasp.py:756).
// The per-(g, r) part (four 32-bit round keys) is set up once; an evaluation is 32-bit arithmetic only.
__host__ __device__ __forceinline__ unsigned int mix32(unsigned int x) {
    x ^= x >> 15;
    x *= 0x2c1b3c6du;
    x ^= x >> 12;
    x *= 0x297a2d39u;
    x ^= x >> 15;
    return x;
}

// Feistel round function: the upper half of a 32-bit multiplicative hash of (half ^ round key).
__host__ __device__ __forceinline__ unsigned int feistel_f(unsigned int half, unsigned int key) {
    return ((half ^ key) * 0x9E3779B1u) >> 16;
}

struct Selector {
    unsigned int rk[4];
    unsigned int n, domain, a_mask, c_mask;
    int a_bits;
    bool with_replacement;

    __host__ __device__ __forceinline__ void init(unsigned long long seed, int g, int r, int n_, int P,
                                                  int replace_when_equal) {
        const unsigned long long key =
            mix64(seed ^ mix64((unsigned long long)g * 0x9E3779B97F4A7C15ull + (unsigned long long)r));
        n = (unsigned int)n_;
        with_replacement = n_ < P + replace_when_equal;
        int b = 2;
        while ((1ll << b) < (long long)n_) ++b;
        a_bits = b >> 1;
        a_mask = (1u << a_bits) - 1u;
        c_mask = (1u << (b - a_bits)) - 1u;
        domain = 1u << b;
        for (int t = 0; t < 4; ++t) rk[t] = (unsigned int)(mix64(key + ((unsigned long long)(t + 1) << 32)) >> 32);
    }

    // bijection of [0, domain): each round XORs one half with a function of the other
    __host__ __device__ __forceinline__ unsigned int perm(unsigned int t) const {
        unsigned int hi = t >> a_bits, lo = t & a_mask;
        hi ^= feistel_f(lo, rk[0]) & c_mask;
        lo ^= feistel_f(hi, rk[1]) & a_mask;
        hi ^= feistel_f(lo, rk[2]) & c_mask;
        lo ^= feistel_f(hi, rk[3]) & a_mask;
        return (hi << a_bits) | lo;
    }

    // i-th draw when sampling with replacement
    __host__ __device__ __forceinline__ int draw_replace(int i) const {
        const unsigned int u = mix32((unsigned int)i * 0x9E3779B9u + rk[0]);
        return (int)(((unsigned long long)u * (unsigned long long)n) >> 32);
    }
};

#ifdef __HIPCC__
// All P draws of one (grasp, repeat) by ONE wave: emit(i, j) is called once for every draw i in [0, P) with its
// position j in [0, n), by the lane that found it.  Every lane of the wave must call this (it uses ballots).
template <typename Emit>
__device__ __forceinline__ void wave_draws(const Selector& pick, int P, int lane, Emit emit) {
    if (pick.with_replacement) {
        for (int i = lane; i < P; i += 64) emit(i, pick.draw_replace(i));
        return;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int found = 0;
    for (unsigned int t0 = 0; found < P && t0 < pick.domain; t0 += 64) {      // n >= P numbers below n exist in [0, domain)
        const unsigned int t = t0 + (unsigned int)lane;
        const unsigned int x = pick.perm(t);
        const bool ok = t < pick.domain && x < pick.n;
        const unsigned long long bal = __ballot(ok);
        const int i = found + (int)__popcll(bal & below);
        if (ok && i < P) emit(i, (int)x);
        found += (int)__popcll(bal);
    }
}
#endif

}  // namespace pngpd

#include "pack.h"

#include <math.h>
#include <string.h>

#include "kernels.h"

namespace pngpd {

void fold_bn(const float* W, const float* b, const float* gamma, const float* beta,
             const float* mean, const float* var, int n_out, int n_in, std::vector<float>& Wf,
             std::vector<float>& bf) {
    Wf.resize((size_t)n_out * n_in);
    bf.resize(n_out);
    for (int o = 0; o < n_out; ++o) {
        double s = 1.0, shift = 0.0, mu = 0.0;
        if (gamma != nullptr) {
            s = (double)gamma[o] / sqrt((double)var[o] + 1e-5);
            shift = (double)beta[o];
            mu = (double)mean[o];
        }
        for (int i = 0; i < n_in; ++i) Wf[(size_t)o * n_in + i] = (float)((double)W[(size_t)o * n_in + i] * s);
        bf[o] = (float)(((double)b[o] - mu) * s + shift);
    }
}

uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // keep NaN a NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static float bf16_to_f32(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void pack_fc_split(const std::vector<float>& Wf, int N, int K, std::vector<uint16_t>& out) {
    const int ksteps = K / 16;
    out.resize((size_t)N * K * 2);
    for (int nt = 0; nt < N / 32; ++nt)
        for (int ks = 0; ks < ksteps; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int n = 32 * nt + (l & 31), k = 16 * ks + 8 * (l >> 5) + j;
                    const float w = Wf[(size_t)n * K + k];
                    const uint16_t hi = f32_to_bf16_rne(w);
                    const uint16_t lo = f32_to_bf16_rne(w - bf16_to_f32(hi));
                    const size_t base = (((size_t)nt * ksteps + ks) * 2) * 512 + (size_t)l * 8 + j;
                    out[base] = hi;
                    out[base + 512] = lo;
                }
}

void pack_stack(const std::vector<float>& W1f, const std::vector<float>& b1f,
                const std::vector<float>& W2f, const std::vector<float>& b2f,
                const std::vector<float>& W3f, const std::vector<float>& b3f, PackedStack& out) {
    out.w1tab.resize(C1 * 4);
    for (int o = 0; o < C1; ++o) {
        out.w1tab[4 * o + 0] = W1f[3 * o + 0];
        out.w1tab[4 * o + 1] = W1f[3 * o + 1];
        out.w1tab[4 * o + 2] = W1f[3 * o + 2];
        out.w1tab[4 * o + 3] = b1f[o];
    }
    out.b2p.resize(B2P_FLOATS);
    for (int t = 0; t < 4; ++t)
        for (int h = 0; h < 2; ++h)
            for (int i = 0; i < 16; ++i) out.b2p[(t * 2 + h) * 16 + i] = b2f[32 * t + acc_row(i, h)];
    out.b3 = b3f;

    // bf16 images for the 16x16x32 shape: lane l holds row/col l%16 and k = 8(l/16) + j.
    // conv2 is computed transposed (A = weights): row tile T covers channels 16T..16T+15.
    out.b2 = b2f;
    out.w2_bf16.resize((size_t)C2 * C1);
    out.w2_lo.resize((size_t)C2 * C1);
    for (int T = 0; T < 8; ++T)
        for (int s = 0; s < 2; ++s)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int ch2 = 16 * T + (l & 15), k = 32 * s + 8 * (l >> 4) + j;
                    const float w = W2f[(size_t)ch2 * C1 + k];
                    const uint16_t hi = f32_to_bf16_rne(w);
                    out.w2_bf16[(((size_t)T * 2 + s) * 64 + l) * 8 + j] = hi;
                    out.w2_lo[(((size_t)T * 2 + s) * 64 + l) * 8 + j] = f32_to_bf16_rne(w - bf16_to_f32(hi));
                }
    // conv3's k slot (quarter q, elem j) of step ks is register j&3 of conv2 row tile 2ks + (j>>2)
    out.w3_bf16.resize((size_t)C3 * C2);
    out.w3_lo.resize((size_t)C3 * C2);
    for (int NT = 0; NT < 64; ++NT)
        for (int ks = 0; ks < 4; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 8; ++j) {
                    const int ch3 = 16 * NT + (l & 15);
                    const int ch2 = 16 * (2 * ks + (j >> 2)) + 4 * (l >> 4) + (j & 3);
                    const float w = W3f[(size_t)ch3 * C2 + ch2];
                    const uint16_t hi = f32_to_bf16_rne(w);
                    out.w3_bf16[(((size_t)NT * 4 + ks) * 64 + l) * 8 + j] = hi;
                    out.w3_lo[(((size_t)NT * 4 + ks) * 64 + l) * 8 + j] = f32_to_bf16_rne(w - bf16_to_f32(hi));
                }

    // f32 images: 32x32x2 fragments, one k per lane half; 4 consecutive steps per 16-byte read
    out.w2_f32.resize((size_t)C2 * C1);
    for (int t = 0; t < 4; ++t)
        for (int q = 0; q < 8; ++q)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const int ch2 = 32 * t + (l & 31), k = 2 * (4 * q + e) + (l >> 5);
                    out.w2_f32[(((size_t)t * 8 + q) * 64 + l) * 4 + e] = W2f[(size_t)ch2 * C1 + k];
                }
    out.w3_f32.resize((size_t)C3 * C2);
    for (int nt = 0; nt < 32; ++nt)
        for (int q = 0; q < 16; ++q)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 4; ++e) {
                    const int ch3 = 32 * nt + (l & 31);
                    const int ch2 = 32 * (q >> 2) + acc_row(4 * (q & 3) + e, l >> 5);
                    out.w3_f32[(((size_t)nt * 16 + q) * 64 + l) * 4 + e] = W3f[(size_t)ch3 * C2 + ch2];
                }
}

}  // namespace pngpd

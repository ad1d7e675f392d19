// Host-side weight preparation: eval-mode BatchNorm folding (float64) and permutation of the
// conv-stack weights into the register-fragment order of the gfx950 MFMA instructions.
#pragma once
#include <stdint.h>

#include <vector>

namespace pngpd {

// y = bn(W x + b) with running statistics  ==  (s*W) x + ((b - mean)*s + beta),  s = gamma/sqrt(var+eps)
// (torch.nn.BatchNorm1d eval semantics, eps = 1e-5; model/pointnet.py:21-25,129-131,182-183).
// W is [n_out, n_in] row-major.  gamma == nullptr means "no BatchNorm after this layer".
void fold_bn(const float* W, const float* b, const float* gamma, const float* beta,
             const float* mean, const float* var, int n_out, int n_in, std::vector<float>& Wf,
             std::vector<float>& bf);

uint16_t f32_to_bf16_rne(float f);

// Row of a 32x32 MFMA accumulator held in register `reg` of lane half `half`.
inline int acc_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct PackedStack {
    std::vector<float> w1tab;      // [64][4]  {w0,w1,w2,bias}
    std::vector<float> b2p;        // [4][2][16]  conv2 bias in 32x32 accumulator order (f32 path)
    std::vector<float> b2;         // [128]       conv2 bias, natural order (bf16 path)
    std::vector<float> b3;         // [1024]
    std::vector<uint16_t> w2_bf16; // [T8][s2][lane64][j8]    16x16x32 fragments
    std::vector<uint16_t> w3_bf16; // [NT64][ks4][lane64][j8]
    std::vector<uint16_t> w2_lo;   // bf16(w - w2_bf16), same order: the second piece of the split-bf16 mode
    std::vector<uint16_t> w3_lo;   // bf16(w - w3_bf16), same order
    std::vector<float> w2_f32;     // [t4][q8][lane64][e4]
    std::vector<float> w3_f32;     // [nt32][q16][lane64][e4]
};

// Split-bf16 image of an FC weight matrix Wf [N,K] (N % 32 == 0, K % 16 == 0):
// out[nt][ks][plane][lane][j], plane 0 = bf16(w), plane 1 = bf16(w - plane0), as the B operand
// fragments of v_mfma_f32_32x32x16_bf16 (n = 32nt + lane%32, k = 16ks + 8(lane/32) + j).
void pack_fc_split(const std::vector<float>& Wf, int N, int K, std::vector<uint16_t>& out);

// W1f [64,3], W2f [128,64], W3f [1024,128]: BN-folded, row-major [out,in].
void pack_stack(const std::vector<float>& W1f, const std::vector<float>& b1f,
                const std::vector<float>& W2f, const std::vector<float>& b2f,
                const std::vector<float>& W3f, const std::vector<float>& b3f, PackedStack& out);

}  // namespace pngpd

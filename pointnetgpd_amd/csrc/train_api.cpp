// C-ABI of the training step: sequencing of the train.hip kernels over a caller-owned workspace.
#include <hip/hip_runtime.h>

#include "api_common.h"
#include "kernels.h"
#include "pngpd.h"

namespace {

using pngpd::fail;

constexpr int C1 = pngpd::C1, C2 = pngpd::C2, C3 = pngpd::C3;

// bump allocator over the workspace; every block is 256-byte aligned
struct Bump {
    char* base;
    size_t off = 0;
    explicit Bump(void* b) : base((char*)b) {}
    template <typename T>
    T* take(size_t n) {
        T* p = (T*)(base + off);
        off += (n * sizeof(T) + 255) & ~(size_t)255;
        return p;
    }
};

struct StackWs {
    float *X3, *h1, *y2, *h2, *pooled, *dh2, *dh1, *dX3;      // conv1's pre-activation is recomputed, not stored
    int* arg;
    float *mean[3], *rstd[3];
    pngpd::Conv3Ws c3;
};

struct FcWs {
    float *z1, *a1, *z2, *a2, *da2, *da1, *din;
    float *mean[2], *rstd[2];
};

struct TrainWs {
    StackWs stn, trunk;
    FcWs stn_fc, head_fc;
    double* part;
    double* gram_part;
    float *tn_part, *fc_part, *m1, *m2, *dlogits, *dtrans, *trans;
    size_t bytes;

    TrainWs(void* base, int B, int P, int k) {
        Bump a(base);
        const size_t M = (size_t)B * P;
        for (StackWs* s : {&stn, &trunk}) {
            s->X3 = a.take<float>(M * 3);
            s->h1 = a.take<float>(M * C1);
            s->y2 = a.take<float>(M * C2);
            s->h2 = a.take<float>(M * C2);
            s->pooled = a.take<float>((size_t)B * C3);
            s->arg = a.take<int>((size_t)B * C3);
            s->dh2 = a.take<float>(M * C2);
            s->dh1 = a.take<float>(M * C1);
            s->dX3 = a.take<float>(M * 3);
            const int ch[3] = {C1, C2, C3};
            for (int i = 0; i < 3; ++i) {
                s->mean[i] = a.take<float>(ch[i]);
                s->rstd[i] = a.take<float>(ch[i]);
            }
            s->c3.abar_d = a.take<double>(C2);
            s->c3.cov = a.take<double>((size_t)C2 * C2);
            s->c3.w3p = a.take<float>((size_t)C3 * C2);
            s->c3.w3s = a.take<unsigned short>((size_t)3 * C3 * C2);
            s->c3.sh = a.take<float>(C3);
            s->c3.wsc = a.take<float>(C3);
            s->c3.gmask = a.take<float>((size_t)B * C3);
            s->c3.G = a.take<float>((size_t)C2 * C2);
            s->c3.ub = a.take<float>(C2);
        }
        for (FcWs* f : {&stn_fc, &head_fc}) {
            f->z1 = a.take<float>((size_t)B * 512);
            f->a1 = a.take<float>((size_t)B * 512);
            f->z2 = a.take<float>((size_t)B * 256);
            f->a2 = a.take<float>((size_t)B * 256);
            f->da2 = a.take<float>((size_t)B * 256);
            f->da1 = a.take<float>((size_t)B * 512);
            f->din = a.take<float>((size_t)B * 1024);
            f->mean[0] = a.take<float>(512);
            f->rstd[0] = a.take<float>(512);
            f->mean[1] = a.take<float>(256);
            f->rstd[1] = a.take<float>(256);
        }
        // column-reduction partials [blocks][2][C]: the per-point layers (C <= 128) and the FC layers (B rows, C <= 512);
        // the conv1 weight gradient ([t_red_blocks][192]) and conv2's statistics records use the same buffer
        size_t part_n = (size_t)pngpd::t_colreduce_blocks((int)M) * 2 * C2;
        const size_t part_fc = (size_t)pngpd::t_colreduce_blocks(B) * 2 * 512;
        const size_t part_w1 = (size_t)pngpd::t_red_blocks((int)M) * 192;
        const size_t part_rec = (size_t)pngpd::tall_gemm_blocks((int)M) * C2 * 3;      // conv2 statistics records
        if (part_n < part_fc) part_n = part_fc;
        if (part_n < part_w1) part_n = part_w1;
        if (part_n < part_rec) part_n = part_rec;
        const size_t part_x3 = (size_t)pngpd::t_x3_blocks((int)M) * 9;                   // moments of the points
        if (part_n < part_x3) part_n = part_x3;
        part = a.take<double>(part_n);
        const size_t chunks = (size_t)pngpd::t_tn_chunks((int)M);
        gram_part = a.take<double>(pngpd::t_gram_ws_doubles((int)M));
        size_t tn = chunks * (size_t)C2 * C1;       // conv2 weight-gradient partials
        if (tn < (size_t)512 * 1024) tn = (size_t)512 * 1024;
        tn_part = a.take<float>(tn);
        fc_part = a.take<float>(pngpd::SPLITK_PART_FLOATS);      // split-k partial products of the FC layers
        m1 = a.take<float>(C3);
        m2 = a.take<float>(C3);
        dlogits = a.take<float>((size_t)B * 16);
        dtrans = a.take<float>((size_t)B * 9);
        trans = a.take<float>((size_t)B * 9);
        (void)k;
        bytes = a.off;
    }
};

// slot helpers
inline float* slot(float* const* t, int unit, int field) { return t[unit * PNGPD_FIELDS + field]; }
inline const float* cslot(const float* const* t, int unit, int field) { return t[unit * PNGPD_FIELDS + field]; }

// units a model with these parts owns: STN 0..5, trunk 6..8, head 9..11
inline bool unit_in_parts(int u, int parts) {
    if (u <= PNGPD_U_STN_FC3) return parts & PNGPD_PART_STN;
    if (u <= PNGPD_U_FEAT_CONV3) return parts & PNGPD_PART_TRUNK;
    return parts & PNGPD_PART_HEAD;
}

int check_tables(float* const* params, bool need_running, int parts) {
    for (int u = 0; u < PNGPD_UNITS; ++u) {
        if (!unit_in_parts(u, parts)) continue;
        const bool has_bn = (u != PNGPD_U_STN_FC3 && u != PNGPD_U_FC3);
        if (!slot(params, u, PNGPD_F_WEIGHT) || !slot(params, u, PNGPD_F_BIAS))
            return fail(PNGPD_ERR_INVALID, "training table: unit %d has no weight/bias", u);
        if (has_bn && (!slot(params, u, PNGPD_F_BN_WEIGHT) || !slot(params, u, PNGPD_F_BN_BIAS)))
            return fail(PNGPD_ERR_INVALID, "training table: unit %d lacks BatchNorm affine tensors", u);
        if (has_bn && need_running && (!slot(params, u, PNGPD_F_BN_MEAN) || !slot(params, u, PNGPD_F_BN_VAR)))
            return fail(PNGPD_ERR_INVALID, "training table: unit %d lacks running statistics", u);
    }
    return PNGPD_OK;
}

int check_shape(int B, int P, int k, const void* ws, size_t ws_bytes) {
    if (B < 2) return fail(PNGPD_ERR_SHAPE, "train-mode BatchNorm needs a batch of at least 2, got %d", B);
    if (P < 1 || k < 2 || k > 16) return fail(PNGPD_ERR_SHAPE, "training: P=%d k=%d", P, k);
    if ((size_t)B * P > (size_t)1 << 30) return fail(PNGPD_ERR_SHAPE, "training: B*P too large");
    if (!ws || ws_bytes < pngpd_train_workspace_bytes(B, P, k))
        return fail(PNGPD_ERR_WORKSPACE, "training workspace of %zu bytes is smaller than the %zu needed", ws_bytes,
                    pngpd_train_workspace_bytes(B, P, k));
    return PNGPD_OK;
}

// conv1 -> bn -> relu -> conv2 -> bn -> relu -> conv3 -> bn (-> relu) -> max over points
int fwd_stack(StackWs& S, float* const* p, int u1, const float* x, const float* trans, int B, int P, int relu_out,
              float momentum, double* part, double* gram_part, hipStream_t s) {
    const int M = B * P;
    HIP_TRY(pngpd::t_points_to_rows_moments(x, trans, S.X3, B, P, part, s));
    // bn1's statistics from the moments of the points; then conv1 + bn1 + relu + conv2 in one pass that writes h1 and
    // y2 and takes the column statistics of y2 from the accumulators (PNGPD_TRAIN_FWD12=split: h1 by a pass of its own,
    // conv2 as a tall product reading it).  relu(bn(conv2)) is applied by the first pass of the conv3 block (t_conv3_fwd)
    const char* fwd12_env = getenv("PNGPD_TRAIN_FWD12");
    const bool fwd12_split = fwd12_env && fwd12_env[0] == 's';
    HIP_TRY(pngpd::t_conv1_bn_fwd(S.X3, slot(p, u1, PNGPD_F_WEIGHT), slot(p, u1, PNGPD_F_BIAS),
                                  slot(p, u1, PNGPD_F_BN_WEIGHT), slot(p, u1, PNGPD_F_BN_BIAS), momentum, part, S.mean[0],
                                  S.rstd[0], slot(p, u1, PNGPD_F_BN_MEAN), slot(p, u1, PNGPD_F_BN_VAR),
                                  fwd12_split ? S.h1 : nullptr, M, s));
    if (fwd12_split)
        HIP_TRY(pngpd::launch_tall_gemm(S.h1, slot(p, u1 + 1, PNGPD_F_WEIGHT), 0, slot(p, u1 + 1, PNGPD_F_BIAS), S.y2, M,
                                        C2, C1, 0, part, s));
    else
        HIP_TRY(pngpd::launch_conv12_fwd(S.X3, slot(p, u1, PNGPD_F_WEIGHT), slot(p, u1, PNGPD_F_BIAS), S.mean[0], S.rstd[0],
                                         slot(p, u1, PNGPD_F_BN_WEIGHT), slot(p, u1, PNGPD_F_BN_BIAS),
                                         slot(p, u1 + 1, PNGPD_F_WEIGHT), slot(p, u1 + 1, PNGPD_F_BIAS), S.h1, S.y2, M,
                                         part, s));
    HIP_TRY(pngpd::t_stats_from_records(part, pngpd::tall_gemm_blocks(M), M, C2, momentum, S.mean[1], S.rstd[1],
                                        slot(p, u1 + 1, PNGPD_F_BN_MEAN), slot(p, u1 + 1, PNGPD_F_BN_VAR), s));
    const int u3 = u1 + 2;
    HIP_TRY(pngpd::t_conv3_fwd(S.h2, slot(p, u3, PNGPD_F_WEIGHT), slot(p, u3, PNGPD_F_BIAS), slot(p, u3, PNGPD_F_BN_WEIGHT),
                               slot(p, u3, PNGPD_F_BN_BIAS), B, P, relu_out, momentum, gram_part, S.c3, S.mean[2],
                               S.rstd[2], slot(p, u3, PNGPD_F_BN_MEAN), slot(p, u3, PNGPD_F_BN_VAR), S.pooled, S.arg, s,
                               S.y2, S.mean[1], S.rstd[1], slot(p, u1 + 1, PNGPD_F_BN_WEIGHT),
                               slot(p, u1 + 1, PNGPD_F_BN_BIAS)));
    return PNGPD_OK;
}

// linear -> bn -> relu -> linear -> bn -> relu -> linear (+ identity | log_softmax)
int fwd_fc(FcWs& F, float* const* p, int u1, const float* in, float* out, int B, int n_last, int tail, float momentum,
           double* part, float* fc_part, hipStream_t s) {
    const int n[2] = {512, 256}, kk[2] = {1024, 512};
    float* zs[2] = {F.z1, F.z2};
    float* as[2] = {F.a1, F.a2};
    const float* cur = in;
    for (int l = 0; l < 2; ++l) {
        const int u = u1 + l;
        HIP_TRY(pngpd::t_fc_bn_fwd(cur, slot(p, u, PNGPD_F_WEIGHT), slot(p, u, PNGPD_F_BIAS), slot(p, u, PNGPD_F_BN_WEIGHT),
                                   slot(p, u, PNGPD_F_BN_BIAS), B, n[l], kk[l], momentum, part, fc_part, zs[l], as[l],
                                   F.mean[l], F.rstd[l], slot(p, u, PNGPD_F_BN_MEAN), slot(p, u, PNGPD_F_BN_VAR), s));
        cur = as[l];
    }
    HIP_TRY(pngpd::launch_fc_tail(F.a2, slot(p, u1 + 2, PNGPD_F_WEIGHT), slot(p, u1 + 2, PNGPD_F_BIAS), out, B, n_last,
                                  256, tail, s));
    return PNGPD_OK;
}

int bwd_fc(FcWs& F, const float* const* p, float* const* g, int u1, const float* in, const float* dOut, int B,
           int n_last, TrainWs& W, hipStream_t s) {
    // last layer (no BatchNorm): real bias gradient
    HIP_TRY(pngpd::t_small_bwd(dOut, F.a2, cslot(p, u1 + 2, PNGPD_F_WEIGHT), F.da2, slot(g, u1 + 2, PNGPD_F_WEIGHT),
                               slot(g, u1 + 2, PNGPD_F_BIAS), B, n_last, 256, s));
    // layer 2
    HIP_TRY(pngpd::t_bn_bwd(F.da2, F.a2, F.z2, F.mean[1], F.rstd[1], cslot(p, u1 + 1, PNGPD_F_BN_WEIGHT), B, 256,
                            W.part, W.m1, W.m2, slot(g, u1 + 1, PNGPD_F_BN_WEIGHT), slot(g, u1 + 1, PNGPD_F_BN_BIAS),
                            slot(g, u1 + 1, PNGPD_F_BIAS), s));
    HIP_TRY(pngpd::t_gemm_tn_nn(F.da2, F.a1, W.tn_part, slot(g, u1 + 1, PNGPD_F_WEIGHT), cslot(p, u1 + 1, PNGPD_F_WEIGHT),
                                F.da1, B, 256, 512, s, W.fc_part));
    // layer 1
    HIP_TRY(pngpd::t_bn_bwd(F.da1, F.a1, F.z1, F.mean[0], F.rstd[0], cslot(p, u1, PNGPD_F_BN_WEIGHT), B, 512, W.part,
                            W.m1, W.m2, slot(g, u1, PNGPD_F_BN_WEIGHT), slot(g, u1, PNGPD_F_BN_BIAS),
                            slot(g, u1, PNGPD_F_BIAS), s));
    HIP_TRY(pngpd::t_gemm_tn_nn(F.da1, in, W.tn_part, slot(g, u1, PNGPD_F_WEIGHT), cslot(p, u1, PNGPD_F_WEIGHT), F.din, B,
                                512, 1024, s, W.fc_part));
    return PNGPD_OK;
}

int bwd_stack(StackWs& S, const float* const* p, float* const* g, int u1, const float* dpooled, int B, int P,
              int relu_out, bool need_dX3, TrainWs& W, hipStream_t s) {
    const int M = B * P;
    // PNGPD_TRAIN_BWD2=split: one kernel per step of the backward of the layers in front of conv3, as before
    const char* bwd2_env = getenv("PNGPD_TRAIN_BWD2");
    const bool fused = !(bwd2_env && bwd2_env[0] == 's');
    // conv3 / bn3 / max-pool; its last launch (conv3's weight gradient) also sums the columns for bn2's backward
    HIP_TRY(pngpd::t_conv3_bwd(S.h2, cslot(p, u1 + 2, PNGPD_F_WEIGHT), cslot(p, u1 + 2, PNGPD_F_BIAS),
                               cslot(p, u1 + 2, PNGPD_F_BN_WEIGHT), cslot(p, u1 + 2, PNGPD_F_BN_BIAS), dpooled, S.pooled, S.arg, S.mean[2], S.rstd[2], B, P,
                               relu_out, S.c3, W.m1, W.m2, slot(g, u1 + 2, PNGPD_F_WEIGHT), slot(g, u1 + 2, PNGPD_F_BIAS),
                               slot(g, u1 + 2, PNGPD_F_BN_WEIGHT), slot(g, u1 + 2, PNGPD_F_BN_BIAS), S.dh2, s,
                               fused ? S.y2 : nullptr, S.mean[1], S.rstd[1], cslot(p, u1 + 1, PNGPD_F_BN_WEIGHT),
                               cslot(p, u1 + 1, PNGPD_F_BN_BIAS), fused ? W.part : nullptr));
    // conv2 / bn2 / conv1's bn in one pass over dh2
    if (fused) {
        HIP_TRY(pngpd::t_bn_bwd_finalize(W.part, pngpd::t_colreduce_blocks(M), M, C2, W.m1, W.m2,
                                         slot(g, u1 + 1, PNGPD_F_BN_WEIGHT), slot(g, u1 + 1, PNGPD_F_BN_BIAS),
                                         slot(g, u1 + 1, PNGPD_F_BIAS), s));
        HIP_TRY(pngpd::launch_conv2_bwd_fused(S.dh2, S.y2, S.mean[1], S.rstd[1], cslot(p, u1 + 1, PNGPD_F_BN_WEIGHT),
                                              cslot(p, u1 + 1, PNGPD_F_BN_BIAS), W.m1, W.m2,
                                              cslot(p, u1 + 1, PNGPD_F_WEIGHT), S.dh1, M, S.X3,
                                              cslot(p, u1, PNGPD_F_WEIGHT), cslot(p, u1, PNGPD_F_BIAS), S.mean[0],
                                              S.rstd[0], cslot(p, u1, PNGPD_F_BN_WEIGHT), cslot(p, u1, PNGPD_F_BN_BIAS),
                                              W.part, s));
        HIP_TRY(pngpd::t_conv2_wgrad_conv1_bwd(S.dh2, S.h1, W.tn_part, slot(g, u1 + 1, PNGPD_F_WEIGHT), S.dh1, S.X3,
                                               cslot(p, u1, PNGPD_F_WEIGHT), cslot(p, u1, PNGPD_F_BIAS), S.mean[0],
                                               S.rstd[0], cslot(p, u1, PNGPD_F_BN_WEIGHT), cslot(p, u1, PNGPD_F_BN_BIAS),
                                               M, W.part, pngpd::tall_gemm_blocks(M), W.m1, W.m2,
                                               slot(g, u1, PNGPD_F_BN_WEIGHT), slot(g, u1, PNGPD_F_BN_BIAS),
                                               slot(g, u1, PNGPD_F_BIAS), slot(g, u1, PNGPD_F_WEIGHT),
                                               need_dX3 ? S.dX3 : nullptr, s));
        return PNGPD_OK;
    }
    HIP_TRY(pngpd::t_bn_bwd(S.dh2, S.h2, S.y2, S.mean[1], S.rstd[1], cslot(p, u1 + 1, PNGPD_F_BN_WEIGHT), M, C2, W.part,
                            W.m1, W.m2, slot(g, u1 + 1, PNGPD_F_BN_WEIGHT), slot(g, u1 + 1, PNGPD_F_BN_BIAS),
                            slot(g, u1 + 1, PNGPD_F_BIAS), s, nullptr, nullptr, nullptr,
                            cslot(p, u1 + 1, PNGPD_F_BN_BIAS)));
    HIP_TRY(pngpd::t_gemm_tn(S.dh2, S.h1, W.tn_part, slot(g, u1 + 1, PNGPD_F_WEIGHT), M, C2, C1, s));
    HIP_TRY(pngpd::launch_tall_gemm(S.dh2, cslot(p, u1 + 1, PNGPD_F_WEIGHT), 1, nullptr, S.dh1, M, C1, C2, 0, nullptr, s));
    // conv1 / bn1
    HIP_TRY(pngpd::t_bn_bwd(S.dh1, S.h1, nullptr, S.mean[0], S.rstd[0], cslot(p, u1, PNGPD_F_BN_WEIGHT), M, C1, W.part,
                            W.m1, W.m2, slot(g, u1, PNGPD_F_BN_WEIGHT), slot(g, u1, PNGPD_F_BN_BIAS),
                            slot(g, u1, PNGPD_F_BIAS), s, S.X3, cslot(p, u1, PNGPD_F_WEIGHT), cslot(p, u1, PNGPD_F_BIAS),
                            cslot(p, u1, PNGPD_F_BN_BIAS)));
    HIP_TRY(pngpd::t_conv1_bwd(S.dh1, S.X3, cslot(p, u1, PNGPD_F_WEIGHT), M, W.part, slot(g, u1, PNGPD_F_WEIGHT),
                               need_dX3 ? S.dX3 : nullptr, s));
    return PNGPD_OK;
}

}  // namespace

extern "C" {

size_t pngpd_train_workspace_bytes(int B, int P, int k) {
    if (B < 1 || P < 1) return 0;
    TrainWs w(nullptr, B, P, k);
    return w.bytes;
}

// ev: nullptr, or 4 events recorded around the conv3 product/max kernel of the STN (0,1) and the trunk (2,3)
static int valid_model_parts(int parts) {
    const int all = PNGPD_PART_STN | PNGPD_PART_TRUNK | PNGPD_PART_HEAD;
    return parts == PNGPD_PART_STN || parts == (PNGPD_PART_STN | PNGPD_PART_TRUNK) || parts == all;
}

static int train_forward_impl(int B, int P, int k, const float* x_dev, float* const* params_dev, int n_slots,
                              float momentum, float* logp_dev, float* trans_dev, float* feat_dev, void* workspace_dev,
                              size_t workspace_bytes, void* stream, hipEvent_t* ev,
                              int parts = PNGPD_PART_STN | PNGPD_PART_TRUNK | PNGPD_PART_HEAD) {
    if (!x_dev || !params_dev) return fail(PNGPD_ERR_INVALID, "pngpd_train_forward: NULL argument");
    if (!valid_model_parts(parts))
        return fail(PNGPD_ERR_INVALID, "parts must be STN, STN|TRUNK or STN|TRUNK|HEAD, got %d", parts);
    if ((parts & PNGPD_PART_HEAD) && !logp_dev) return fail(PNGPD_ERR_INVALID, "pngpd_train_forward: NULL logp_dev");
    if (n_slots != PNGPD_SLOTS) return fail(PNGPD_ERR_INVALID, "training table must have %d slots", PNGPD_SLOTS);
    int rc = check_shape(B, P, k, workspace_dev, workspace_bytes);
    if (rc) return rc;
    if ((rc = check_tables(params_dev, true, parts))) return rc;
    hipStream_t s = (hipStream_t)stream;
    TrainWs W(workspace_dev, B, P, k);
    if (ev) {
        W.stn.c3.ev_begin = ev[0];
        W.stn.c3.ev_end = ev[1];
        W.trunk.c3.ev_begin = ev[2];
        W.trunk.c3.ev_end = ev[3];
    }
    float* trans = trans_dev ? trans_dev : W.trans;
    if ((rc = fwd_stack(W.stn, params_dev, PNGPD_U_STN_CONV1, x_dev, nullptr, B, P, 1, momentum, W.part, W.gram_part, s)))
        return rc;
    if ((rc = fwd_fc(W.stn_fc, params_dev, PNGPD_U_STN_FC1, W.stn.pooled, trans, B, 9, pngpd::FC_TAIL_ADD_IDENTITY,
                     momentum, W.part, W.fc_part, s)))
        return rc;
    if (!(parts & PNGPD_PART_TRUNK)) return PNGPD_OK;
    if ((rc = fwd_stack(W.trunk, params_dev, PNGPD_U_FEAT_CONV1, x_dev, trans, B, P, 0, momentum, W.part, W.gram_part,
                       s)))
        return rc;
    if (!(parts & PNGPD_PART_HEAD)) {
        if (feat_dev)
            HIP_TRY(hipMemcpyAsync(feat_dev, W.trunk.pooled, (size_t)B * C3 * sizeof(float), hipMemcpyDeviceToDevice, s));
        return PNGPD_OK;
    }
    if ((rc = fwd_fc(W.head_fc, params_dev, PNGPD_U_FC1, W.trunk.pooled, logp_dev, B, k, pngpd::FC_TAIL_LOG_SOFTMAX,
                     momentum, W.part, W.fc_part, s)))
        return rc;
    if (feat_dev)
        HIP_TRY(hipMemcpyAsync(feat_dev, W.trunk.pooled, (size_t)B * C3 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return PNGPD_OK;
}

int pngpd_train_forward(int B, int P, int k, const float* x_dev, float* const* params_dev, int n_slots,
                        float momentum, float* logp_dev, float* trans_dev, float* feat_dev, void* workspace_dev,
                        size_t workspace_bytes, void* stream) {
    return train_forward_impl(B, P, k, x_dev, params_dev, n_slots, momentum, logp_dev, trans_dev, feat_dev,
                              workspace_dev, workspace_bytes, stream, nullptr);
}

int pngpd_train_forward_parts(int parts, int B, int P, int k, const float* x_dev, float* const* params_dev, int n_slots,
                              float momentum, float* logp_dev, float* trans_dev, float* feat_dev, void* workspace_dev,
                              size_t workspace_bytes, void* stream) {
    return train_forward_impl(B, P, k, x_dev, params_dev, n_slots, momentum, logp_dev, trans_dev, feat_dev,
                              workspace_dev, workspace_bytes, stream, nullptr, parts);
}

int pngpd_train_forward_timed(int B, int P, int k, const float* x_dev, float* const* params_dev, int n_slots,
                              float momentum, float* logp_dev, float* trans_dev, float* feat_dev, void* workspace_dev,
                              size_t workspace_bytes, float* conv3_ms_out, void* stream) {
    if (!conv3_ms_out) return fail(PNGPD_ERR_INVALID, "pngpd_train_forward_timed: NULL conv3_ms_out");
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipEventCreate(&ev[i]);
        if (e != hipSuccess) {
            while (i-- > 0) (void)hipEventDestroy(ev[i]);
            return fail(PNGPD_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    int rc = train_forward_impl(B, P, k, x_dev, params_dev, n_slots, momentum, logp_dev, trans_dev, feat_dev,
                                workspace_dev, workspace_bytes, stream, ev);
    if (rc == PNGPD_OK) {
        hipError_t e = hipEventSynchronize(ev[3]);
        if (e == hipSuccess) e = hipEventElapsedTime(&conv3_ms_out[0], ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&conv3_ms_out[1], ev[2], ev[3]);
        if (e != hipSuccess) rc = fail(PNGPD_ERR_HIP, "event timing failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

int pngpd_train_argmax(int B, int P, int k, const void* workspace_dev, size_t workspace_bytes, int* stn_arg_dev,
                       int* trunk_arg_dev, void* stream) {
    int rc = check_shape(B, P, k, workspace_dev, workspace_bytes);
    if (rc) return rc;
    TrainWs W((void*)workspace_dev, B, P, k);
    const size_t bytes = (size_t)B * C3 * sizeof(int);
    hipStream_t s = (hipStream_t)stream;
    if (stn_arg_dev) HIP_TRY(hipMemcpyAsync(stn_arg_dev, W.stn.arg, bytes, hipMemcpyDeviceToDevice, s));
    if (trunk_arg_dev) HIP_TRY(hipMemcpyAsync(trunk_arg_dev, W.trunk.arg, bytes, hipMemcpyDeviceToDevice, s));
    return PNGPD_OK;
}

int pngpd_train_saved(int B, int P, int k, const void* workspace_dev, size_t workspace_bytes, int which,
                      const float** data_dev_out, size_t* count_out) {
    if (!data_dev_out || !count_out) return fail(PNGPD_ERR_INVALID, "pngpd_train_saved: NULL output");
    int rc = check_shape(B, P, k, workspace_dev, workspace_bytes);
    if (rc) return rc;
    TrainWs W((void*)workspace_dev, B, P, k);
    const size_t M = (size_t)B * P;
    switch (which) {
        case PNGPD_SAVED_STN_H1: *data_dev_out = W.stn.h1; *count_out = M * C1; break;
        case PNGPD_SAVED_STN_H2: *data_dev_out = W.stn.h2; *count_out = M * C2; break;
        case PNGPD_SAVED_STN_POOLED: *data_dev_out = W.stn.pooled; *count_out = (size_t)B * C3; break;
        case PNGPD_SAVED_STN_A1: *data_dev_out = W.stn_fc.a1; *count_out = (size_t)B * 512; break;
        case PNGPD_SAVED_STN_A2: *data_dev_out = W.stn_fc.a2; *count_out = (size_t)B * 256; break;
        case PNGPD_SAVED_TRUNK_H1: *data_dev_out = W.trunk.h1; *count_out = M * C1; break;
        case PNGPD_SAVED_TRUNK_H2: *data_dev_out = W.trunk.h2; *count_out = M * C2; break;
        case PNGPD_SAVED_TRUNK_POOLED: *data_dev_out = W.trunk.pooled; *count_out = (size_t)B * C3; break;
        case PNGPD_SAVED_HEAD_A1: *data_dev_out = W.head_fc.a1; *count_out = (size_t)B * 512; break;
        case PNGPD_SAVED_HEAD_A2: *data_dev_out = W.head_fc.a2; *count_out = (size_t)B * 256; break;
        default: return fail(PNGPD_ERR_INVALID, "pngpd_train_saved: unknown activation %d", which);
    }
    return PNGPD_OK;
}

// Backward in the order the gradients become final: head, trunk, transform regressor (STN).
static int train_backward_impl(int B, int P, int k, const float* x_dev, const float* const* params_dev,
                               float* const* grads_dev, int n_slots, const float* logp_dev, const float* dlogp_dev,
                               const float* dtrans_dev, void* workspace_dev, size_t workspace_bytes, int parts,
                               void* stream, const float* dfeat_dev = nullptr) {
    if (!x_dev || !params_dev || !grads_dev) return fail(PNGPD_ERR_INVALID, "pngpd_train_backward: NULL argument");
    if ((parts & PNGPD_PART_HEAD) && (!logp_dev || !dlogp_dev))
        return fail(PNGPD_ERR_INVALID, "pngpd_train_backward: the head part needs logp_dev and dlogp_dev");
    if ((parts & PNGPD_PART_HEAD) && dfeat_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_train_backward: dfeat_dev is the input gradient of a model without a head");
    if (n_slots != PNGPD_SLOTS) return fail(PNGPD_ERR_INVALID, "training table must have %d slots", PNGPD_SLOTS);
    const int all = PNGPD_PART_STN | PNGPD_PART_TRUNK | PNGPD_PART_HEAD;
    if (parts == 0 || (parts & ~all)) return fail(PNGPD_ERR_INVALID, "pngpd_train_backward_part: bad part mask %d", parts);
    int rc = check_shape(B, P, k, workspace_dev, workspace_bytes);
    if (rc) return rc;
    if ((rc = check_tables((float* const*)params_dev, false, parts))) return rc;
    if ((rc = check_tables(grads_dev, false, parts))) return rc;
    hipStream_t s = (hipStream_t)stream;
    TrainWs W(workspace_dev, B, P, k);
    if (parts & PNGPD_PART_HEAD) {
        HIP_TRY(pngpd::t_logsoftmax_bwd(logp_dev, dlogp_dev, W.dlogits, B, k, s));
        if ((rc = bwd_fc(W.head_fc, params_dev, grads_dev, PNGPD_U_FC1, W.trunk.pooled, W.dlogits, B, k, W, s))) return rc;
    }
    if (parts & PNGPD_PART_TRUNK) {
        // the gradient of the pooled feature: the caller's (a model without a head), else what the head's backward
        // pass - this call's or an earlier one's - left in the workspace
        const float* dpooled = dfeat_dev ? dfeat_dev : W.head_fc.din;
        if ((rc = bwd_stack(W.trunk, params_dev, grads_dev, PNGPD_U_FEAT_CONV1, dpooled, B, P, 0, true, W, s)))
            return rc;
        HIP_TRY(pngpd::t_trans_grad(x_dev, W.trunk.dX3, B, P, dtrans_dev, W.dtrans, s));
    } else if (parts == PNGPD_PART_STN && dtrans_dev) {
        // transform regressor alone: the gradient of its output is the caller's
        HIP_TRY(hipMemcpyAsync(W.dtrans, dtrans_dev, (size_t)B * 9 * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    if (parts & PNGPD_PART_STN) {
        if ((rc = bwd_fc(W.stn_fc, params_dev, grads_dev, PNGPD_U_STN_FC1, W.stn.pooled, W.dtrans, B, 9, W, s))) return rc;
        if ((rc = bwd_stack(W.stn, params_dev, grads_dev, PNGPD_U_STN_CONV1, W.stn_fc.din, B, P, 1, false, W, s)))
            return rc;
    }
    return PNGPD_OK;
}

int pngpd_train_backward(int B, int P, int k, const float* x_dev, const float* const* params_dev,
                         float* const* grads_dev, int n_slots, const float* logp_dev, const float* dlogp_dev,
                         const float* dtrans_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    return train_backward_impl(B, P, k, x_dev, params_dev, grads_dev, n_slots, logp_dev, dlogp_dev, dtrans_dev,
                               workspace_dev, workspace_bytes, PNGPD_PART_STN | PNGPD_PART_TRUNK | PNGPD_PART_HEAD,
                               stream);
}

int pngpd_train_backward_part(int B, int P, int k, const float* x_dev, const float* const* params_dev,
                              float* const* grads_dev, int n_slots, const float* logp_dev, const float* dlogp_dev,
                              const float* dtrans_dev, const float* dfeat_dev, void* workspace_dev,
                              size_t workspace_bytes, int parts, void* stream) {
    return train_backward_impl(B, P, k, x_dev, params_dev, grads_dev, n_slots, logp_dev, dlogp_dev, dtrans_dev,
                               workspace_dev, workspace_bytes, parts, stream, dfeat_dev);
}

int pngpd_nll_loss(const float* logp_dev, const int64_t* target_dev, int B, int k, float* loss_dev, float* dlogp_dev,
                   void* stream) {
    if (!logp_dev || !target_dev || !loss_dev || !dlogp_dev) return fail(PNGPD_ERR_INVALID, "pngpd_nll_loss: NULL argument");
    if (B < 1 || k < 2) return fail(PNGPD_ERR_SHAPE, "pngpd_nll_loss: B=%d k=%d", B, k);
    HIP_TRY(pngpd::t_nll_loss(logp_dev, (const long long*)target_dev, B, k, loss_dev, dlogp_dev, (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_adam_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                    float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* stream) {
    if (!params_dev || !grads_dev || !exp_avg_dev || !exp_avg_sq_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_adam_step: NULL argument");
    if (n < 1 || step < 1) return fail(PNGPD_ERR_SHAPE, "pngpd_adam_step: n=%lld step=%d", (long long)n, step);
    HIP_TRY(pngpd::t_adam(params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, (size_t)n, lr, beta1, beta2, eps, step,
                          grad_scale, (hipStream_t)stream));
    return PNGPD_OK;
}

}  // extern "C"

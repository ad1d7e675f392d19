// C-ABI layer: argument checking, weight upload, kernel sequencing.  No kernel code here.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <atomic>
#include <new>
#include <string>
#include <vector>

#include "api_common.h"
#include "kernels.h"
#include "pack.h"
#include "pack_device.h"
#include "pngpd.h"

namespace {
thread_local std::string g_last_error;
}

namespace pngpd {
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
}  // namespace pngpd

namespace {

using pngpd::fail;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    // a re-upload of the same size (weights after a training step) keeps the allocation: no hipFree, which
    // would wait for the whole device, and the copy is ordered on the stream behind earlier forwards
    hipError_t upload(const void* src, size_t n, hipStream_t s) {
        if (p && n != bytes) {
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, n);
            if (e != hipSuccess) return e;
            bytes = n;
        }
        return hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, s);
    }
    // the allocation alone (a kernel fills it)
    hipError_t reserve(size_t n) {
        if (p && n != bytes) {
            (void)hipFree(p);
            p = nullptr;
            bytes = 0;
        }
        if (!p) {
            hipError_t e = hipMalloc(&p, n);
            if (e != hipSuccess) return e;
            bytes = n;
        }
        return hipSuccess;
    }
};

struct StackDev {
    DevBuf w1tab, b2p, b2, b3, w2_bf16, w3_bf16, w2_lo, w3_lo, w2_f32, w3_f32;
};

struct FcDev {
    DevBuf w[3], b[3], wsplit[2];
    int n[3] = {0, 0, 0}, k[3] = {0, 0, 0};
};

}  // namespace

struct pngpd_model {
    int device = 0;
    int num_points = 0;
    int k = 0;
    int parts = 0;
    bool loaded = false;
    StackDev stn, trunk;
    FcDev stn_fc, head_fc;
};

namespace {

using pngpd::PackedStack;

// folded conv/linear layer of unit `u`; n_out x n_in
int fold_unit(const float* const* t, int u, int n_out, int n_in, bool has_bn, std::vector<float>& W,
              std::vector<float>& b) {
    const float* const* f = t + u * PNGPD_FIELDS;
    if (!f[PNGPD_F_WEIGHT] || !f[PNGPD_F_BIAS])
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d has no weight/bias", u);
    if (has_bn && (!f[PNGPD_F_BN_WEIGHT] || !f[PNGPD_F_BN_BIAS] || !f[PNGPD_F_BN_MEAN] || !f[PNGPD_F_BN_VAR]))
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d lacks BatchNorm tensors", u);
    pngpd::fold_bn(f[PNGPD_F_WEIGHT], f[PNGPD_F_BIAS], has_bn ? f[PNGPD_F_BN_WEIGHT] : nullptr,
                   f[PNGPD_F_BN_BIAS], f[PNGPD_F_BN_MEAN], f[PNGPD_F_BN_VAR], n_out, n_in, W, b);
    return PNGPD_OK;
}

int load_stack(const float* const* t, int u_conv1, StackDev& d, std::vector<PackedStack>& keep,
               hipStream_t s) {
    std::vector<float> W1, b1, W2, b2, W3, b3;
    int rc;
    if ((rc = fold_unit(t, u_conv1 + 0, pngpd::C1, 3, true, W1, b1))) return rc;
    if ((rc = fold_unit(t, u_conv1 + 1, pngpd::C2, pngpd::C1, true, W2, b2))) return rc;
    if ((rc = fold_unit(t, u_conv1 + 2, pngpd::C3, pngpd::C2, true, W3, b3))) return rc;
    keep.emplace_back();
    PackedStack& p = keep.back();
    pngpd::pack_stack(W1, b1, W2, b2, W3, b3, p);
    HIP_TRY(d.w1tab.upload(p.w1tab.data(), p.w1tab.size() * 4, s));
    HIP_TRY(d.b2p.upload(p.b2p.data(), p.b2p.size() * 4, s));
    HIP_TRY(d.b3.upload(p.b3.data(), p.b3.size() * 4, s));
    HIP_TRY(d.w2_bf16.upload(p.w2_bf16.data(), p.w2_bf16.size() * 2, s));
    HIP_TRY(d.w3_bf16.upload(p.w3_bf16.data(), p.w3_bf16.size() * 2, s));
    HIP_TRY(d.w2_lo.upload(p.w2_lo.data(), p.w2_lo.size() * 2, s));
    HIP_TRY(d.w3_lo.upload(p.w3_lo.data(), p.w3_lo.size() * 2, s));
    HIP_TRY(d.b2.upload(p.b2.data(), p.b2.size() * 4, s));
    HIP_TRY(d.w2_f32.upload(p.w2_f32.data(), p.w2_f32.size() * 4, s));
    HIP_TRY(d.w3_f32.upload(p.w3_f32.data(), p.w3_f32.size() * 4, s));
    return PNGPD_OK;
}

int load_fc(const float* const* t, int u_fc1, int n_last, FcDev& d,
            std::vector<std::vector<float>>& keep, std::vector<std::vector<uint16_t>>& keep16, hipStream_t s) {
    const int n[3] = {512, 256, n_last}, k[3] = {1024, 512, 256};
    for (int i = 0; i < 3; ++i) {
        keep.emplace_back();
        std::vector<float>& W = keep.back();
        keep.emplace_back();
        std::vector<float>& b = keep.back();
        int rc = fold_unit(t, u_fc1 + i, n[i], k[i], i < 2, W, b);
        if (rc) return rc;
        HIP_TRY(d.w[i].upload(W.data(), W.size() * 4, s));
        HIP_TRY(d.b[i].upload(b.data(), b.size() * 4, s));
        if (i < 2) {
            keep16.emplace_back();
            pngpd::pack_fc_split(W, n[i], k[i], keep16.back());
            HIP_TRY(d.wsplit[i].upload(keep16.back().data(), keep16.back().size() * 2, s));
        }
        d.n[i] = n[i];
        d.k[i] = k[i];
    }
    return PNGPD_OK;
}

size_t round4(size_t n) { return (n + 3) & ~(size_t)3; }

// Floats of the area that holds the partial maxima of a conv stack cut over points (small batches only); part of
// every workspace so that its size grows monotonically with B.
constexpr size_t workspace_part_floats(int B) {
    return (size_t)pngpd::CONV_SPLIT_MAX_GROUPS * 1024 * (size_t)(B < pngpd::CONV_SPLIT_MAX_B ? B : pngpd::CONV_SPLIT_MAX_B);
}

struct Workspace {
    float *pooled, *h512, *h256, *trans, *feat, *part;
    static size_t floats(int B) {
        return (size_t)B * (1024 + 512 + 256 + 1024) + round4((size_t)B * 9) + workspace_part_floats(B);
    }
    Workspace(void* base, int B) {
        float* p = (float*)base;
        pooled = p; p += (size_t)B * 1024;
        h512 = p;   p += (size_t)B * 512;
        h256 = p;   p += (size_t)B * 256;
        feat = p;   p += (size_t)B * 1024;
        trans = p;  p += round4((size_t)B * 9);
        part = p;
    }
};

// groups > 1: `in` is [groups][B][1024] partial maxima; the first layer merges them (and stores the merged rows to
// `merged` when given).
int run_fc_chain(const FcDev& fc, const float* in, Workspace& ws, float* out, int B, int tail, int precision,
                 hipStream_t s, int groups = 1, float* merged = nullptr) {
    if (precision != PNGPD_FP32) {
        // split-bf16 products: f32-grade results on the bf16 matrix pipe
        HIP_TRY(pngpd::launch_fc_bf16x3(in, fc.wsplit[0].p, (const float*)fc.b[0].p, ws.h512, B, fc.n[0], fc.k[0], 1, s,
                                        groups, merged));
        HIP_TRY(pngpd::launch_fc_bf16x3(ws.h512, fc.wsplit[1].p, (const float*)fc.b[1].p, ws.h256, B, fc.n[1], fc.k[1],
                                        1, s));
    } else {
        HIP_TRY(pngpd::launch_fc_f32(in, (const float*)fc.w[0].p, (const float*)fc.b[0].p, ws.h512, B,
                                     fc.n[0], fc.k[0], 1, s));
        HIP_TRY(pngpd::launch_fc_f32(ws.h512, (const float*)fc.w[1].p, (const float*)fc.b[1].p, ws.h256,
                                     B, fc.n[1], fc.k[1], 1, s));
    }
    HIP_TRY(pngpd::launch_fc_tail(ws.h256, (const float*)fc.w[2].p, (const float*)fc.b[2].p, out, B,
                                  fc.n[2], fc.k[2], tail, s));
    return PNGPD_OK;
}

pngpd::ConvStackArgs stack_args(const StackDev& d, int precision, const float* x, int layout, int B,
                                int P, const float* trans, float* pooled, int relu_out) {
    pngpd::ConvStackArgs a;
    a.x = x;
    a.layout = layout;
    a.B = B;
    a.P = P;
    a.w1tab = (const float4*)d.w1tab.p;
    a.trans = trans;
    const bool f32 = precision == PNGPD_FP32;
    a.w2p = f32 ? d.w2_f32.p : d.w2_bf16.p;
    a.b2p = (const float*)(f32 ? d.b2p.p : d.b2.p);
    a.w3p = f32 ? d.w3_f32.p : d.w3_bf16.p;
    if (precision == PNGPD_BF16X3) {
        a.w2lo = d.w2_lo.p;
        a.w3lo = d.w3_lo.p;
    }
    a.b3 = (const float*)d.b3.p;
    a.pooled = pooled;
    a.relu_out = relu_out;
    return a;
}

}  // namespace

// ---- workspace of pngpd_forward_infer_noalloc -------------------------------------------------------------
// The per-batch intermediates of a forward (pooled maxima, the two hidden FC layers, the global feature, the
// transform) for up to PNGPD_NOALLOC_MAX_B candidates live in a device array of the library's code object
// instead of a caller workspace.  Every GPU the library runs on holds its own copy at its own address; in host
// code the array's name is only a handle for hipGetSymbolAddress, never a device address.
static_assert(PNGPD_NOALLOC_MAX_B >= 1, "PNGPD_NOALLOC_MAX_B");
static constexpr size_t SCRATCH_FLOATS =
    (size_t)PNGPD_NOALLOC_MAX_B * (1024 + 512 + 256 + 1024) + (((size_t)PNGPD_NOALLOC_MAX_B * 9 + 3) & ~(size_t)3) +
    workspace_part_floats(PNGPD_NOALLOC_MAX_B);
static __device__ __align__(256) float scratch[SCRATCH_FLOATS];

namespace {
constexpr int MAX_DEVICES = 64;
std::atomic<float*> g_scratch_addr[MAX_DEVICES];

// One array per GPU serves every model and stream, so two forwards that overlap on a GPU would overwrite each other's
// intermediates.  The library cannot order them, but it can tell: per device, a flag says that a thread is inside the
// call, and an event recorded behind the last forward says whether the GPU is still working on it.  A call on a
// different stream (or thread) before that event has completed is refused with PNGPD_ERR_STATE instead of racing.
// The event is created at set-up (pngpd_model_create); the call itself only records and queries it.
struct ScratchUse {
    std::atomic<int> in_call{0};
    std::atomic<hipEvent_t> done{nullptr};
    hipStream_t last_stream = nullptr;
    bool has_last = false;
};
ScratchUse g_scratch_use[MAX_DEVICES];

// Device address of `scratch` on device `dev`, which must be the current device.  Looked up once per device.
int scratch_on_current_device(int dev, float** out) {
    if (dev < 0 || dev >= MAX_DEVICES) return fail(PNGPD_ERR_INVALID, "device ordinal %d out of range", dev);
    float* p = g_scratch_addr[dev].load(std::memory_order_acquire);
    if (!p) {
        void* q = nullptr;
        hipError_t e = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !q)
            return fail(PNGPD_ERR_HIP, "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        p = (float*)q;
        g_scratch_addr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return PNGPD_OK;
}

// set-up only (allocates): the event behind which a device's scratch array is free again
int scratch_event_on_current_device(int dev) {
    if (dev < 0 || dev >= MAX_DEVICES) return fail(PNGPD_ERR_INVALID, "device ordinal %d out of range", dev);
    if (g_scratch_use[dev].done.load(std::memory_order_acquire)) return PNGPD_OK;
    hipEvent_t ev = nullptr;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return fail(PNGPD_ERR_HIP, "hipEventCreate on device %d: %s", dev, hipGetErrorString(e));
    hipEvent_t expected = nullptr;
    if (!g_scratch_use[dev].done.compare_exchange_strong(expected, ev, std::memory_order_acq_rel))
        (void)hipEventDestroy(ev);            // another thread was first
    return PNGPD_OK;
}
}  // namespace

extern "C" {

const char* pngpd_last_error(void) { return g_last_error.c_str(); }

int pngpd_abi_version(void) { return 2; }

int pngpd_model_create(pngpd_model** out, int num_points, int k, int parts) {
    if (!out) return fail(PNGPD_ERR_INVALID, "pngpd_model_create: out is NULL");
    *out = nullptr;
    if (num_points < 1) return fail(PNGPD_ERR_SHAPE, "num_points must be >= 1, got %d", num_points);
    if (k < 1 || k > 16) return fail(PNGPD_ERR_SHAPE, "k must be in [1,16], got %d", k);
    const int all = PNGPD_PART_STN | PNGPD_PART_TRUNK | PNGPD_PART_HEAD;
    if (parts != PNGPD_PART_STN && parts != (PNGPD_PART_STN | PNGPD_PART_TRUNK) && parts != all)
        return fail(PNGPD_ERR_INVALID, "parts must be STN, STN|TRUNK or STN|TRUNK|HEAD, got %d", parts);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    float* scratch_addr = nullptr;   // resolved here, at set-up, so that the first no-allocation forward only reads it
    int rc = scratch_on_current_device(dev, &scratch_addr);
    if (rc) return rc;
    if ((rc = scratch_event_on_current_device(dev))) return rc;
    pngpd_model* m = new (std::nothrow) pngpd_model();
    if (!m) return fail(PNGPD_ERR_INVALID, "out of host memory");
    m->device = dev;
    m->num_points = num_points;
    m->k = k;
    m->parts = parts;
    *out = m;
    return PNGPD_OK;
}

int pngpd_model_destroy(pngpd_model* m) {
    delete m;
    return PNGPD_OK;
}

int pngpd_model_load(pngpd_model* m, const float* const* host_tensors, int n_slots, void* stream) {
    if (!m || !host_tensors) return fail(PNGPD_ERR_INVALID, "pngpd_model_load: NULL argument");
    if (n_slots != PNGPD_SLOTS)
        return fail(PNGPD_ERR_INVALID, "weight table must have %d slots, got %d", PNGPD_SLOTS, n_slots);
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != m->device)
        return fail(PNGPD_ERR_STATE, "model lives on device %d but device %d is current", m->device, dev);
    // validate the whole table before the first upload: a table that fails half-way must not leave the model with
    // one part new and one part old
    for (int u = 0; u < PNGPD_UNITS; ++u) {
        const bool owned = u <= PNGPD_U_STN_FC3 ? (m->parts & PNGPD_PART_STN)
                           : u <= PNGPD_U_FEAT_CONV3 ? (m->parts & PNGPD_PART_TRUNK) : (m->parts & PNGPD_PART_HEAD);
        if (!owned) continue;
        const float* const* f = host_tensors + u * PNGPD_FIELDS;
        const bool has_bn = (u != PNGPD_U_STN_FC3 && u != PNGPD_U_FC3);
        if (!f[PNGPD_F_WEIGHT] || !f[PNGPD_F_BIAS])
            return fail(PNGPD_ERR_INVALID, "weight table: unit %d has no weight/bias", u);
        if (has_bn && (!f[PNGPD_F_BN_WEIGHT] || !f[PNGPD_F_BN_BIAS] || !f[PNGPD_F_BN_MEAN] || !f[PNGPD_F_BN_VAR]))
            return fail(PNGPD_ERR_INVALID, "weight table: unit %d lacks BatchNorm tensors", u);
    }
    m->loaded = false;
    std::vector<PackedStack> keep_stacks;
    keep_stacks.reserve(2);
    std::vector<std::vector<float>> keep_fc;
    keep_fc.reserve(12);
    std::vector<std::vector<uint16_t>> keep16;
    keep16.reserve(4);
    // The uploads below are asynchronous copies out of the staging vectors above.  Declared after them, this guard
    // is destroyed first: whichever way the function is left, the stream has drained before a vector is freed.
    struct DrainOnExit {
        hipStream_t s;
        ~DrainOnExit() { (void)hipStreamSynchronize(s); }
    } drain{s};
    int rc;
    if (m->parts & PNGPD_PART_STN) {
        if ((rc = load_stack(host_tensors, PNGPD_U_STN_CONV1, m->stn, keep_stacks, s))) return rc;
        if ((rc = load_fc(host_tensors, PNGPD_U_STN_FC1, 9, m->stn_fc, keep_fc, keep16, s))) return rc;
    }
    if (m->parts & PNGPD_PART_TRUNK)
        if ((rc = load_stack(host_tensors, PNGPD_U_FEAT_CONV1, m->trunk, keep_stacks, s))) return rc;
    if (m->parts & PNGPD_PART_HEAD)
        if ((rc = load_fc(host_tensors, PNGPD_U_FC1, m->k, m->head_fc, keep_fc, keep16, s))) return rc;
    HIP_TRY(hipStreamSynchronize(s));   // staging vectors die at return
    m->loaded = true;
    return PNGPD_OK;
}

namespace {

// `n` floats readable at device pointer `p` on the current device?  A host pointer handed to a kernel would fault
// the GPU, so every tensor of a device table is looked up before anything is launched.
int check_device_tensor(const float* p, size_t n, int dev, int unit, int field) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d field %d is not a device pointer", unit, field);
    }
    if (at.type != hipMemoryTypeDevice || at.device != dev)
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d field %d does not live on device %d", unit, field, dev);
    hipDeviceptr_t base = nullptr;
    size_t extent = 0;
    if (hipMemGetAddressRange(&base, &extent, (hipDeviceptr_t)p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d field %d has no allocation behind it", unit, field);
    }
    const size_t off = (size_t)((const char*)p - (const char*)base);
    if (off + n * sizeof(float) > extent)
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d field %d: %zu floats do not fit its allocation", unit, field,
                    n);
    if (((uintptr_t)p & 3u) != 0)
        return fail(PNGPD_ERR_INVALID, "weight table: unit %d field %d is not aligned to 4 bytes", unit, field);
    return PNGPD_OK;
}

pngpd::UnitDev unit_dev(const float* const* t, int u, bool has_bn) {
    const float* const* f = t + u * PNGPD_FIELDS;
    pngpd::UnitDev d;
    d.W = f[PNGPD_F_WEIGHT];
    d.b = f[PNGPD_F_BIAS];
    d.gamma = has_bn ? f[PNGPD_F_BN_WEIGHT] : nullptr;
    d.beta = has_bn ? f[PNGPD_F_BN_BIAS] : nullptr;
    d.mean = has_bn ? f[PNGPD_F_BN_MEAN] : nullptr;
    d.var = has_bn ? f[PNGPD_F_BN_VAR] : nullptr;
    return d;
}

int load_stack_device(const float* const* t, int u_conv1, StackDev& d, hipStream_t s) {
    using namespace pngpd;
    HIP_TRY(d.w1tab.reserve((size_t)C1 * 4 * 4));
    HIP_TRY(d.b2p.reserve((size_t)B2P_FLOATS * 4));
    HIP_TRY(d.b2.reserve((size_t)C2 * 4));
    HIP_TRY(d.b3.reserve((size_t)C3 * 4));
    HIP_TRY(d.w2_bf16.reserve((size_t)C2 * C1 * 2));
    HIP_TRY(d.w3_bf16.reserve((size_t)C3 * C2 * 2));
    HIP_TRY(d.w2_lo.reserve((size_t)C2 * C1 * 2));
    HIP_TRY(d.w3_lo.reserve((size_t)C3 * C2 * 2));
    HIP_TRY(d.w2_f32.reserve((size_t)C2 * C1 * 4));
    HIP_TRY(d.w3_f32.reserve((size_t)C3 * C2 * 4));
    StackImages o;
    o.w1tab = (float*)d.w1tab.p;
    o.b2p = (float*)d.b2p.p;
    o.b2 = (float*)d.b2.p;
    o.b3 = (float*)d.b3.p;
    o.w2_bf16 = (uint16_t*)d.w2_bf16.p;
    o.w3_bf16 = (uint16_t*)d.w3_bf16.p;
    o.w2_lo = (uint16_t*)d.w2_lo.p;
    o.w3_lo = (uint16_t*)d.w3_lo.p;
    o.w2_f32 = (float*)d.w2_f32.p;
    o.w3_f32 = (float*)d.w3_f32.p;
    HIP_TRY(launch_pack_stack_device(unit_dev(t, u_conv1, true), unit_dev(t, u_conv1 + 1, true),
                                     unit_dev(t, u_conv1 + 2, true), o, s));
    return PNGPD_OK;
}

int load_fc_device(const float* const* t, int u_fc1, int n_last, FcDev& d, hipStream_t s) {
    const int n[3] = {512, 256, n_last}, k[3] = {1024, 512, 256};
    for (int i = 0; i < 3; ++i) {
        HIP_TRY(d.w[i].reserve((size_t)n[i] * k[i] * 4));
        HIP_TRY(d.b[i].reserve((size_t)n[i] * 4));
        if (i < 2) HIP_TRY(d.wsplit[i].reserve((size_t)n[i] * k[i] * 2 * 2));
        HIP_TRY(pngpd::launch_pack_fc_device(unit_dev(t, u_fc1 + i, i < 2), n[i], k[i], (float*)d.w[i].p, (float*)d.b[i].p,
                                             i < 2 ? (uint16_t*)d.wsplit[i].p : nullptr, s));
        d.n[i] = n[i];
        d.k[i] = k[i];
    }
    return PNGPD_OK;
}

// floats of field `field` of unit `u` (k = classes of the head)
size_t unit_field_floats(int u, int field, int k) {
    static const int n_out[PNGPD_UNITS] = {64, 128, 1024, 512, 256, 9, 64, 128, 1024, 512, 256, 0};
    static const int n_in[PNGPD_UNITS] = {3, 64, 128, 1024, 512, 256, 3, 64, 128, 1024, 512, 256};
    const size_t o = u == PNGPD_U_FC3 ? (size_t)k : (size_t)n_out[u];
    return field == PNGPD_F_WEIGHT ? o * n_in[u] : o;
}

}  // namespace

int pngpd_model_load_device(pngpd_model* m, const float* const* device_tensors, int n_slots, void* stream) {
    if (!m || !device_tensors) return fail(PNGPD_ERR_INVALID, "pngpd_model_load_device: NULL argument");
    if (n_slots != PNGPD_SLOTS)
        return fail(PNGPD_ERR_INVALID, "weight table must have %d slots, got %d", PNGPD_SLOTS, n_slots);
    hipStream_t s = (hipStream_t)stream;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != m->device)
        return fail(PNGPD_ERR_STATE, "model lives on device %d but device %d is current", m->device, dev);
    // the whole table is validated before the first launch
    for (int u = 0; u < PNGPD_UNITS; ++u) {
        const bool owned = u <= PNGPD_U_STN_FC3 ? (m->parts & PNGPD_PART_STN)
                           : u <= PNGPD_U_FEAT_CONV3 ? (m->parts & PNGPD_PART_TRUNK) : (m->parts & PNGPD_PART_HEAD);
        if (!owned) continue;
        const float* const* f = device_tensors + u * PNGPD_FIELDS;
        const bool has_bn = (u != PNGPD_U_STN_FC3 && u != PNGPD_U_FC3);
        if (!f[PNGPD_F_WEIGHT] || !f[PNGPD_F_BIAS])
            return fail(PNGPD_ERR_INVALID, "weight table: unit %d has no weight/bias", u);
        if (has_bn && (!f[PNGPD_F_BN_WEIGHT] || !f[PNGPD_F_BN_BIAS] || !f[PNGPD_F_BN_MEAN] || !f[PNGPD_F_BN_VAR]))
            return fail(PNGPD_ERR_INVALID, "weight table: unit %d lacks BatchNorm tensors", u);
        for (int field = 0; field < PNGPD_FIELDS; ++field) {
            if (!f[field] || (!has_bn && field >= PNGPD_F_BN_WEIGHT)) continue;
            const int rc = check_device_tensor(f[field], unit_field_floats(u, field, m->k), dev, u, field);
            if (rc) return rc;
        }
    }
    m->loaded = false;
    int rc;
    if (m->parts & PNGPD_PART_STN) {
        if ((rc = load_stack_device(device_tensors, PNGPD_U_STN_CONV1, m->stn, s))) return rc;
        if ((rc = load_fc_device(device_tensors, PNGPD_U_STN_FC1, 9, m->stn_fc, s))) return rc;
    }
    if (m->parts & PNGPD_PART_TRUNK)
        if ((rc = load_stack_device(device_tensors, PNGPD_U_FEAT_CONV1, m->trunk, s))) return rc;
    if (m->parts & PNGPD_PART_HEAD)
        if ((rc = load_fc_device(device_tensors, PNGPD_U_FC1, m->k, m->head_fc, s))) return rc;
    m->loaded = true;
    return PNGPD_OK;
}

size_t pngpd_forward_workspace_bytes(const pngpd_model* m, int B) {
    if (!m || B < 1) return 0;
    return Workspace::floats(B) * sizeof(float);
}

// ev: nullptr, or 4 events recorded on the stream around the STN (0,1) and trunk (2,3) conv-stack launches
static int forward_impl(const pngpd_model* m, const float* x_dev, int layout, int B, int P, int precision,
                        float* logp_dev, float* trans_dev, float* feat_dev, void* workspace_dev,
                        size_t workspace_bytes, void* stream, hipEvent_t* ev) {
    if (!m || !x_dev) return fail(PNGPD_ERR_INVALID, "pngpd_forward_infer: NULL model or input");
    if (!m->loaded) return fail(PNGPD_ERR_STATE, "pngpd_forward_infer: weights not loaded");
    if (B < 1) return fail(PNGPD_ERR_SHAPE, "batch must be >= 1, got %d", B);
    if (P != m->num_points)
        return fail(PNGPD_ERR_SHAPE, "input has %d points per candidate but the model was built for num_points=%d",
                    P, m->num_points);
    if (layout != PNGPD_LAYOUT_C3P && layout != PNGPD_LAYOUT_P3)
        return fail(PNGPD_ERR_INVALID, "unknown layout %d", layout);
    if (precision != PNGPD_FP32 && precision != PNGPD_BF16 && precision != PNGPD_BF16X3)
        return fail(PNGPD_ERR_INVALID, "unknown precision %d", precision);
    if (!workspace_dev || workspace_bytes < pngpd_forward_workspace_bytes(m, B))
        return fail(PNGPD_ERR_WORKSPACE, "workspace of %zu bytes is smaller than the %zu needed for B=%d",
                    workspace_bytes, pngpd_forward_workspace_bytes(m, B), B);
    if (logp_dev && !(m->parts & PNGPD_PART_HEAD))
        return fail(PNGPD_ERR_STATE, "log-probabilities requested from a model without a head");
    if (feat_dev && !(m->parts & PNGPD_PART_TRUNK))
        return fail(PNGPD_ERR_STATE, "global feature requested from a model without a trunk");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != m->device)
        return fail(PNGPD_ERR_STATE, "model lives on device %d but device %d is current", m->device, dev);
    hipStream_t s = (hipStream_t)stream;
    Workspace ws(workspace_dev, B);
    auto launch_stack = precision == PNGPD_BF16 ? pngpd::launch_conv_stack_bf16
                        : precision == PNGPD_BF16X3 ? pngpd::launch_conv_stack_x3 : pngpd::launch_conv_stack_f32;

    // STN3d: conv stack (ReLU after the max), then the 1024-512-256-9 regressor + identity
    float* trans = trans_dev ? trans_dev : ws.trans;
    // A batch of a few candidates (the reference scores ONE per call, main_test.py:73-83) cannot fill the chip with
    // one workgroup per candidate and channel piece: the conv stacks then also cut the points of a candidate into
    // groups, and the FC layer behind each stack takes the maximum over the groups' partial maxima.
    int groups = 1, tiles_per_group = 0;
    HIP_TRY(pngpd::conv_stack_point_groups(precision, B, P, &groups, &tiles_per_group));
    auto split_args = [&](pngpd::ConvStackArgs a) {
        a.point_groups = groups;
        a.tiles_per_group = tiles_per_group;
        a.pooled = ws.part;
        return a;
    };
    if (ev) HIP_TRY(hipEventRecord(ev[0], s));
    if (groups > 1)
        HIP_TRY(launch_stack(split_args(stack_args(m->stn, precision, x_dev, layout, B, P, nullptr, ws.pooled, 1)), s));
    else
        HIP_TRY(launch_stack(stack_args(m->stn, precision, x_dev, layout, B, P, nullptr, ws.pooled, 1), s));
    if (ev) HIP_TRY(hipEventRecord(ev[1], s));
    int rc = run_fc_chain(m->stn_fc, groups > 1 ? ws.part : ws.pooled, ws, trans, B, pngpd::FC_TAIL_ADD_IDENTITY, precision,
                          s, groups);
    if (rc) return rc;
    if (!(m->parts & PNGPD_PART_TRUNK)) return PNGPD_OK;

    // PointNetfeat: points times trans, conv stack without the last ReLU
    float* feat = feat_dev ? feat_dev : ws.feat;
    const bool head = (m->parts & PNGPD_PART_HEAD) && logp_dev;
    if (!head) groups = 1;                               // no FC layer behind the trunk to merge the groups
    if (ev) HIP_TRY(hipEventRecord(ev[2], s));
    if (groups > 1)
        HIP_TRY(launch_stack(split_args(stack_args(m->trunk, precision, x_dev, layout, B, P, trans, feat, 0)), s));
    else
        HIP_TRY(launch_stack(stack_args(m->trunk, precision, x_dev, layout, B, P, trans, feat, 0), s));
    if (ev) HIP_TRY(hipEventRecord(ev[3], s));
    if (!head) return PNGPD_OK;

    // PointNetCls head + log_softmax (the merged global feature goes to `feat` on the way)
    return run_fc_chain(m->head_fc, groups > 1 ? ws.part : feat, ws, logp_dev, B, pngpd::FC_TAIL_LOG_SOFTMAX, precision, s,
                        groups, groups > 1 ? feat : nullptr);
}

int pngpd_forward_infer(const pngpd_model* m, const float* x_dev, int layout, int B, int P,
                        int precision, float* logp_dev, float* trans_dev, float* feat_dev,
                        void* workspace_dev, size_t workspace_bytes, void* stream) {
    return forward_impl(m, x_dev, layout, B, P, precision, logp_dev, trans_dev, feat_dev, workspace_dev,
                        workspace_bytes, stream, nullptr);
}

int pngpd_forward_infer_noalloc(const pngpd_model* m, const float* x_dev, int layout, int B, int P, int precision,
                                float* logp_dev, float* trans_dev, float* feat_dev, void* stream) {
    if (!m) return fail(PNGPD_ERR_INVALID, "pngpd_forward_infer_noalloc: NULL model");
    if (B > PNGPD_NOALLOC_MAX_B)
        return fail(PNGPD_ERR_SHAPE, "pngpd_forward_infer_noalloc takes at most %d candidates per call, got %d "
                    "(use pngpd_forward_infer with a caller workspace)", PNGPD_NOALLOC_MAX_B, B);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != m->device)
        return fail(PNGPD_ERR_STATE, "model lives on device %d but device %d is current", m->device, dev);
    float* ws = nullptr;
    int rc = scratch_on_current_device(dev, &ws);
    if (rc) return rc;
    ScratchUse& use = g_scratch_use[dev];
    hipEvent_t done = use.done.load(std::memory_order_acquire);
    if (!done) return fail(PNGPD_ERR_STATE, "pngpd_forward_infer_noalloc: no model was created on device %d", dev);
    if (use.in_call.exchange(1, std::memory_order_acquire) != 0)
        return fail(PNGPD_ERR_STATE, "pngpd_forward_infer_noalloc: another thread is inside the call on device %d "
                    "(all calls on a GPU share one scratch array)", dev);
    struct Leave {
        std::atomic<int>& f;
        ~Leave() { f.store(0, std::memory_order_release); }
    } leave{use.in_call};
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &capturing);       // an event recorded under capture cannot be queried: skip the check
    if (capturing == hipStreamCaptureStatusNone && use.has_last && use.last_stream != s &&
        hipEventQuery(done) == hipErrorNotReady)
        return fail(PNGPD_ERR_STATE, "pngpd_forward_infer_noalloc: the scratch array of device %d is still in use by a "
                    "forward on another stream (synchronise the two streams, or use pngpd_forward_infer with a "
                    "workspace per stream)", dev);
    // every float of the workspace that a forward reads is written earlier in the same forward: nothing to clear
    rc = forward_impl(m, x_dev, layout, B, P, precision, logp_dev, trans_dev, feat_dev, ws,
                      SCRATCH_FLOATS * sizeof(float), stream, nullptr);
    if (rc == PNGPD_OK && capturing == hipStreamCaptureStatusNone) {
        HIP_TRY(hipEventRecord(done, s));
        use.last_stream = s;
        use.has_last = true;
    }
    return rc;
}

int pngpd_forward_infer_timed(const pngpd_model* m, const float* x_dev, int layout, int B, int P,
                              int precision, float* logp_dev, float* trans_dev, float* feat_dev,
                              void* workspace_dev, size_t workspace_bytes, float* conv_ms_out, void* stream) {
    if (!conv_ms_out) return fail(PNGPD_ERR_INVALID, "pngpd_forward_infer_timed: NULL conv_ms_out");
    if (m && !(m->parts & PNGPD_PART_TRUNK)) return fail(PNGPD_ERR_STATE, "pngpd_forward_infer_timed: model has no trunk");
    hipEvent_t ev[4];
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipEventCreate(&ev[i]);
        if (e != hipSuccess) {
            while (i-- > 0) (void)hipEventDestroy(ev[i]);
            return fail(PNGPD_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    int rc = forward_impl(m, x_dev, layout, B, P, precision, logp_dev, trans_dev, feat_dev, workspace_dev,
                          workspace_bytes, stream, ev);
    if (rc == PNGPD_OK) {
        hipError_t e = hipEventSynchronize(ev[3]);
        if (e == hipSuccess) e = hipEventElapsedTime(&conv_ms_out[0], ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&conv_ms_out[1], ev[2], ev[3]);
        if (e != hipSuccess) rc = fail(PNGPD_ERR_HIP, "event timing failed: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

// crop workspace: [G,12] f64 frames | [G] i32 spatial order of the grasps | the scene's spatial grid | only in the
// all-pairs test mode a [G, Wp] u64 mask.  Every region is rounded up to 256 B.
static size_t crop_frames_bytes(int G) { return ((size_t)G * 12 * sizeof(double) + 255) & ~(size_t)255; }
static size_t crop_perm_bytes(int G) { return ((size_t)G * sizeof(int) + 255) & ~(size_t)255; }

static size_t crop_mask_bytes(int S, int G) {
    if (!pngpd::crop_brute_mode()) return 0;
    return ((size_t)G * pngpd::crop_words_per_grasp(S) * sizeof(unsigned long long) + 255) & ~(size_t)255;
}

namespace {
struct CropWs {
    double* frames;
    int* perm;
    void* grid;
    unsigned long long* mask;
    CropWs(const void* base, int S, int G) {
        char* p = (char*)base;
        frames = (double*)p;              p += crop_frames_bytes(G);
        perm = (int*)p;                   p += crop_perm_bytes(G);
        grid = p;                         p += pngpd::crop_grid_bytes(S);
        mask = pngpd::crop_brute_mode() ? (unsigned long long*)p : nullptr;
    }
};
}  // namespace

size_t pngpd_crop_workspace_bytes(int S, int G) {
    if (S < 1 || G < 1) return 0;
    return crop_frames_bytes(G) + crop_perm_bytes(G) + pngpd::crop_grid_bytes(S) + crop_mask_bytes(S, G);
}

static int crop_ws_check(int S, int G, const void* ws, size_t bytes) {
    if (S < 1 || G < 1) return fail(PNGPD_ERR_SHAPE, "crop: S=%d G=%d", S, G);
    if (!ws || bytes < pngpd_crop_workspace_bytes(S, G))
        return fail(PNGPD_ERR_WORKSPACE, "crop workspace of %zu bytes is smaller than the %zu needed for S=%d G=%d",
                    bytes, pngpd_crop_workspace_bytes(S, G), S, G);
    return PNGPD_OK;
}

int pngpd_crop_count(const float* scene_dev, int S, const double* grasps_dev, int G, const double* limits,
                     void* workspace_dev, size_t workspace_bytes, int32_t* cnt_dev, int64_t* offsets_dev,
                     void* stream) {
    if (!scene_dev || !grasps_dev || !limits || !cnt_dev || !offsets_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_crop_count: NULL argument");
    int rc = crop_ws_check(S, G, workspace_dev, workspace_bytes);
    if (rc) return rc;
    pngpd::CropLimits L{limits[0], limits[1], limits[2], limits[3], limits[4], limits[5]};
    CropWs w(workspace_dev, S, G);
    HIP_TRY(pngpd::launch_crop_count(scene_dev, S, grasps_dev, G, L, w.frames, w.mask, w.grid, cnt_dev,
                                     (long long*)offsets_dev, (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_crop_count_frames(const float* scene_dev, int S, const double* frames_dev, int G, const double* limits,
                            void* workspace_dev, size_t workspace_bytes, int32_t* cnt_dev, int64_t* offsets_dev,
                            void* stream) {
    if (!scene_dev || !frames_dev || !limits || !cnt_dev || !offsets_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_crop_count_frames: NULL argument");
    int rc = crop_ws_check(S, G, workspace_dev, workspace_bytes);
    if (rc) return rc;
    pngpd::CropLimits L{limits[0], limits[1], limits[2], limits[3], limits[4], limits[5]};
    CropWs w(workspace_dev, S, G);
    HIP_TRY(pngpd::launch_crop_count(scene_dev, S, frames_dev, G, L, w.frames, w.mask, w.grid, cnt_dev,
                                     (long long*)offsets_dev, (hipStream_t)stream, true));
    return PNGPD_OK;
}

int pngpd_dataset_frames(const double* records_dev, int G, int record_stride, const double* transform,
                         double hand_depth, double* frames_dev, void* stream) {
    if (!records_dev || !transform || !frames_dev) return fail(PNGPD_ERR_INVALID, "pngpd_dataset_frames: NULL argument");
    if (G < 1 || record_stride < 8)
        return fail(PNGPD_ERR_SHAPE, "pngpd_dataset_frames: G=%d record_stride=%d (a record has at least 8 fields)", G,
                    record_stride);
    HIP_TRY(pngpd::launch_dataset_frames(records_dev, record_stride, G, transform, hand_depth, frames_dev,
                                         (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_crop_emit(const float* scene_dev, int S, int G, const double* limits, const void* workspace_dev,
                    size_t workspace_bytes, const int32_t* cnt_dev, const int64_t* offsets_dev, int32_t* idx_dev,
                    int64_t idx_capacity, double* pts_g_dev, void* stream) {
    if (!scene_dev || !limits || !cnt_dev || !offsets_dev || !idx_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_crop_emit: NULL argument");
    if (idx_capacity < 0) return fail(PNGPD_ERR_SHAPE, "pngpd_crop_emit: negative capacity");
    int rc = crop_ws_check(S, G, workspace_dev, workspace_bytes);
    if (rc) return rc;
    pngpd::CropLimits L{limits[0], limits[1], limits[2], limits[3], limits[4], limits[5]};
    CropWs w(workspace_dev, S, G);
    HIP_TRY(pngpd::launch_crop_emit(scene_dev, S, G, L, w.frames, w.mask, w.grid, cnt_dev, (const long long*)offsets_dev,
                                    idx_dev, idx_capacity, pts_g_dev, (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_crop_gather(const float* scene_dev, int S, int G, const void* workspace_dev, size_t workspace_bytes,
                      const int32_t* idx_dev, const int64_t* offsets_dev, const int32_t* cnt_dev, int repeat,
                      int P, int min_points, const int32_t* sel_dev, uint64_t seed, int first_grasp_id,
                      float* pts_out_dev, int32_t* sel_out_dev, uint8_t* valid_dev, void* stream) {
    if (!scene_dev || !idx_dev || !offsets_dev || !cnt_dev || !pts_out_dev || !valid_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_crop_gather: NULL argument");
    if (repeat < 1 || P < 1 || min_points < 1)
        return fail(PNGPD_ERR_SHAPE, "pngpd_crop_gather: repeat=%d P=%d min_points=%d", repeat, P, min_points);
    int rc = crop_ws_check(S, G, workspace_dev, workspace_bytes);
    if (rc) return rc;
    HIP_TRY(pngpd::launch_crop_gather(scene_dev, (const double*)workspace_dev, idx_dev, (const long long*)offsets_dev,
                                      cnt_dev, G, repeat, P, min_points, sel_dev, seed, first_grasp_id, pts_out_dev, sel_out_dev,
                                      valid_dev, (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_crop_subsample(const float* scene_dev, int S, const double* grasps_dev, int G, const double* limits,
                         void* workspace_dev, size_t workspace_bytes, int repeat, int P, int min_points, uint64_t seed,
                         int first_grasp_id, float* pts_out_dev, uint8_t* valid_dev, int32_t* cnt_dev,
                         int32_t* overflow_dev, void* stream) {
    if (!scene_dev || !grasps_dev || !limits || !pts_out_dev || !valid_dev || !cnt_dev || !overflow_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_crop_subsample: NULL argument");
    if (repeat < 1 || P < 1 || min_points < 1)
        return fail(PNGPD_ERR_SHAPE, "pngpd_crop_subsample: repeat=%d P=%d min_points=%d", repeat, P, min_points);
    int rc = crop_ws_check(S, G, workspace_dev, workspace_bytes);
    if (rc) return rc;
    pngpd::CropLimits L{limits[0], limits[1], limits[2], limits[3], limits[4], limits[5]};
    CropWs w(workspace_dev, S, G);
    HIP_TRY(pngpd::launch_crop_frames(grasps_dev, w.frames, G, (hipStream_t)stream));
    HIP_TRY(pngpd::launch_crop_gather_fused(scene_dev, S, w.frames, G, L, w.grid, repeat, P, min_points, seed,
                                            first_grasp_id, pts_out_dev, valid_dev, cnt_dev, overflow_dev,
                                            w.perm, (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_crop_gather_dataset(const float* scene_dev, int S, int G, const void* workspace_dev, size_t workspace_bytes,
                              const int32_t* idx_dev, const int64_t* offsets_dev, const int32_t* cnt_dev, int P,
                              int min_points, const int32_t* sel_dev, uint64_t seed, int first_grasp_id,
                              float* pts_out_dev, int32_t* sel_out_dev, uint8_t* valid_dev, void* stream) {
    if (!scene_dev || !idx_dev || !offsets_dev || !cnt_dev || !pts_out_dev || !valid_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_crop_gather_dataset: NULL argument");
    if (P < 1 || min_points < 1) return fail(PNGPD_ERR_SHAPE, "pngpd_crop_gather_dataset: P=%d min_points=%d", P, min_points);
    int rc = crop_ws_check(S, G, workspace_dev, workspace_bytes);
    if (rc) return rc;
    HIP_TRY(pngpd::launch_crop_gather(scene_dev, (const double*)workspace_dev, idx_dev, (const long long*)offsets_dev,
                                      cnt_dev, G, 1, P, min_points, sel_dev, seed, first_grasp_id, pts_out_dev, sel_out_dev,
                                      valid_dev, (hipStream_t)stream, 1, 1));
    return PNGPD_OK;
}

int pngpd_vote(const int64_t* pred_dev, const float* probs_dev, const uint8_t* valid_dev, int G, int repeat, int k,
               int32_t* label_dev, float* score_dev, void* stream) {
    if (!pred_dev || !probs_dev || !label_dev || !score_dev) return fail(PNGPD_ERR_INVALID, "pngpd_vote: NULL argument");
    if (G < 1 || repeat < 1 || k < 2 || k > 16) return fail(PNGPD_ERR_SHAPE, "pngpd_vote: G=%d repeat=%d k=%d", G, repeat, k);
    HIP_TRY(pngpd::launch_vote((const long long*)pred_dev, probs_dev, valid_dev, G, repeat, k, label_dev, score_dev,
                               (hipStream_t)stream));
    return PNGPD_OK;
}

size_t pngpd_voxel_workspace_bytes(int n) {
    if (n < 1 || n > pngpd::VOXEL_MAX_N) return 0;
    return pngpd::voxel_workspace_bytes(n);
}

int pngpd_voxel_unique(const float* points_dev, int S, int n, int edges, void* workspace_dev, size_t workspace_bytes,
                       float* out_dev, int64_t capacity, int32_t* count_dev, void* stream) {
    if (!points_dev || !workspace_dev || !out_dev || !count_dev)
        return fail(PNGPD_ERR_INVALID, "pngpd_voxel_unique: NULL argument");
    if (S < 1 || n < 1 || n > pngpd::VOXEL_MAX_N || capacity < 0)
        return fail(PNGPD_ERR_SHAPE, "pngpd_voxel_unique: S=%d n=%d (1..%d) capacity=%lld", S, n, pngpd::VOXEL_MAX_N,
                    (long long)capacity);
    if (edges != PNGPD_EDGES_F64 && edges != PNGPD_EDGES_F32)
        return fail(PNGPD_ERR_INVALID, "pngpd_voxel_unique: unknown edge type %d", edges);
    if (workspace_bytes < pngpd::voxel_workspace_bytes(n))
        return fail(PNGPD_ERR_WORKSPACE, "pngpd_voxel_unique: workspace %zu < %zu bytes", workspace_bytes,
                    pngpd::voxel_workspace_bytes(n));
    HIP_TRY(pngpd::launch_voxel_unique(points_dev, S, n, edges == PNGPD_EDGES_F32, workspace_dev, out_dev, (long long)capacity, count_dev,
                                       (hipStream_t)stream));
    return PNGPD_OK;
}

int pngpd_score(const float* logp_dev, int B, int k, float* probs_dev, int64_t* pred_dev, void* stream) {
    if (!logp_dev || !probs_dev || !pred_dev) return fail(PNGPD_ERR_INVALID, "pngpd_score: NULL argument");
    if (B < 1 || k < 1) return fail(PNGPD_ERR_SHAPE, "pngpd_score: B=%d k=%d", B, k);
    HIP_TRY(pngpd::launch_score(logp_dev, B, k, probs_dev, pred_dev, (hipStream_t)stream));
    return PNGPD_OK;
}

}  // extern "C"

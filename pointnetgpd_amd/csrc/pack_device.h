// Device-side weight preparation (pack_device.hip): the images of pack.h built by kernels from parameter tensors in
// device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pngpd {

// one layer unit of the weight table, device pointers; gamma == nullptr: no BatchNorm behind the layer
struct UnitDev {
    const float *W, *b, *gamma, *beta, *mean, *var;
};

// destinations of a conv stack's images (device), sizes as in PackedStack
struct StackImages {
    float *w1tab, *b2p, *b2, *b3;
    uint16_t *w2_bf16, *w3_bf16, *w2_lo, *w3_lo;
    float *w2_f32, *w3_f32;
};

hipError_t launch_pack_stack_device(const UnitDev& u1, const UnitDev& u2, const UnitDev& u3, const StackImages& o,
                                    hipStream_t s);
// w [N,K] and b [N]: folded float32 layer; wsplit (optional, N % 32 == 0 and K % 16 == 0): its split-bf16 image
hipError_t launch_pack_fc_device(const UnitDev& u, int N, int K, float* w, float* b, uint16_t* wsplit, hipStream_t s);

}  // namespace pngpd

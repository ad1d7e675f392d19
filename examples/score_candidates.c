/* A C host of the library: no Python, no torch.  Loads a PointNetCls from a flat weight file, scores
 * a batch of candidates and writes the log-probabilities, i.e. model.eval(); output, _ = model(data)
 * of main_test.py:73-83 / main_1v.py:318-319 for a whole batch.
 *
 *   score_candidates <in.bin> <out.bin>
 *
 * in.bin : int32 P, k, B, precision; then PNGPD_SLOTS records { int64 n; float32 data[n] } (n = 0 for the
 *          BatchNorm fields the two fc3 layers do not have), in the slot order of pngpd.h; then float32
 *          x[B,3,P].   out.bin: float32 logp[B,k], float32 trans[B,9].
 * Build (build.py does it): gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude ... -lpngpd -lamdhip64
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pngpd.h"

#define CHECK_HIP(e)                                                                   \
    do {                                                                               \
        hipError_t e_ = (e);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_));                    \
            return 2;                                                                  \
        }                                                                              \
    } while (0)
#define CHECK_PNGPD(e)                                                                 \
    do {                                                                               \
        if ((e) != PNGPD_OK) {                                                         \
            fprintf(stderr, "%s: %s\n", #e, pngpd_last_error());                       \
            return 3;                                                                  \
        }                                                                              \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[4];
    if (fread(hdr, sizeof(int32_t), 4, f) != 4) return 1;
    const int P = hdr[0], k = hdr[1], B = hdr[2], precision = hdr[3];
    float* tensors[PNGPD_SLOTS];
    for (int s = 0; s < PNGPD_SLOTS; ++s) {
        int64_t n;
        if (fread(&n, sizeof n, 1, f) != 1) return 1;
        tensors[s] = NULL;
        if (n > 0) {
            tensors[s] = (float*)malloc((size_t)n * sizeof(float));
            if (!tensors[s] || fread(tensors[s], sizeof(float), (size_t)n, f) != (size_t)n) return 1;
        }
    }
    const size_t nx = (size_t)B * 3 * P;
    float* x = (float*)malloc(nx * sizeof(float));
    if (!x || fread(x, sizeof(float), nx, f) != nx) return 1;
    fclose(f);

    CHECK_HIP(hipSetDevice(0));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    pngpd_model* m = NULL;
    CHECK_PNGPD(pngpd_model_create(&m, P, k, PNGPD_PART_STN | PNGPD_PART_TRUNK | PNGPD_PART_HEAD));
    CHECK_PNGPD(pngpd_model_load(m, (const float* const*)tensors, PNGPD_SLOTS, stream));

    float *x_dev, *logp_dev, *trans_dev;
    void* ws;
    const size_t ws_bytes = pngpd_forward_workspace_bytes(m, B);
    CHECK_HIP(hipMalloc((void**)&x_dev, nx * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&logp_dev, (size_t)B * k * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&trans_dev, (size_t)B * 9 * sizeof(float)));
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_HIP(hipMemcpyAsync(x_dev, x, nx * sizeof(float), hipMemcpyHostToDevice, stream));
    CHECK_PNGPD(pngpd_forward_infer(m, x_dev, PNGPD_LAYOUT_C3P, B, P, precision, logp_dev, trans_dev, NULL, ws, ws_bytes,
                                    stream));
    float* out = (float*)malloc((size_t)B * (k + 9) * sizeof(float));
    CHECK_HIP(hipMemcpyAsync(out, logp_dev, (size_t)B * k * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(out + (size_t)B * k, trans_dev, (size_t)B * 9 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    f = fopen(argv[2], "wb");
    if (!f || fwrite(out, sizeof(float), (size_t)B * (k + 9), f) != (size_t)B * (k + 9)) return 1;
    fclose(f);
    CHECK_PNGPD(pngpd_model_destroy(m));
    printf("scored %d candidates of %d points (%s)\n", B, P, precision == PNGPD_BF16 ? "bf16" : precision == PNGPD_BF16X3 ? "bf16x3" : "fp32");
    return 0;
}

/* A consumer of include/mdx_hip.h written in plain C -- no Python, no torch: the drop-in boundary is a C ABI, and this is the
 * smallest program that exercises it the way any host language would (extern "C", caller-owned device buffers,
 * status returns, a stream argument).
 *
 *   abi_consumer <out.bin>
 *
 * 1. S1: builds the schedule tables of BASELINE configs[2] (T = 1000, linear, sigma 1e-4 .. 0.2, corrector_step_epsilon 2.5e-8,
 *    C = 2 classes: noise_schedulers/noise_scheduler.py:112-267) with mdx_noise_schedule_build and writes them to <out.bin>
 *    (9 vectors [T], then 3 tensors [T, C, C], float32) -- tests/test_kernels_gpu.py compares the file with the reference-made
 *    fixture tests/golden/schedules.npz bit for bit.
 * 2. F1: mdx_noise_relative_coordinates on 6 numbers with sigma = 0 (wrap only: utils/basis_transformations.py:95-119),
 *    appended to the file.
 * Compiled by the test with gcc -std=c11 against include/mdx_hip.h and libmdx_hip.so (+ the HIP runtime's C host API for
 * hipMalloc / hipMemcpy). */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "mdx_hip.h"

#define CHECK_HIP(call)                                                               \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define CHECK_MDX(call)                                                               \
    do {                                                                              \
        int s_ = (call);                                                              \
        if (s_ != MDX_OK) { fprintf(stderr, "%s: %s\n", #call, mdx_status_string(s_)); return 3; } \
    } while (0)

int main(int argc, char** argv)
{
    if (argc != 2) { fprintf(stderr, "usage: %s out.bin\n", argv[0]); return 1; }
    if (mdx_abi_version() != MDX_ABI_VERSION) { fprintf(stderr, "ABI mismatch: header %d, library %d\n", MDX_ABI_VERSION, mdx_abi_version()); return 1; }
    const int T = 1000, C = 2;
    const size_t vec = (size_t)T, mat = (size_t)T * C * C, total = 9 * vec + 3 * mat;
    float* d = NULL;
    CHECK_HIP(hipMalloc((void**)&d, total * sizeof(float)));
    float* p[12];
    for (int k = 0; k < 9; ++k) p[k] = d + k * vec;
    for (int k = 0; k < 3; ++k) p[9 + k] = d + 9 * vec + k * mat;
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_MDX(mdx_noise_schedule_build(T, 1, 1e-5, 1e-4, 0.2, 2.5e-8, C, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9],
                                       p[10], p[11], (mdx_stream_t)stream));
    /* F1 with sigma = 0: the periodic wrap of 6 coordinates */
    const float x_host[6] = {-1e-8f, 1.0f, 2.0f, -0.25f, 1.75f, 0.5f};
    float *x = NULL, *z = NULL, *out = NULL;
    CHECK_HIP(hipMalloc((void**)&x, sizeof(x_host)));
    CHECK_HIP(hipMalloc((void**)&z, sizeof(x_host)));
    CHECK_HIP(hipMalloc((void**)&out, sizeof(x_host)));
    CHECK_HIP(hipMemcpyAsync(x, x_host, sizeof(x_host), hipMemcpyHostToDevice, stream));
    CHECK_HIP(hipMemsetAsync(z, 0, sizeof(x_host), stream));
    CHECK_MDX(mdx_noise_relative_coordinates(x, z, 0.0f, 6, out, (mdx_stream_t)stream));
    /* an invalid argument comes back as a status, not as an abort */
    if (mdx_noise_schedule_build(0, 1, 1e-5, 1e-4, 0.2, 2.5e-8, C, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10],
                                 p[11], (mdx_stream_t)stream) != MDX_ERR_INVALID_ARG) { fprintf(stderr, "T = 0 was accepted\n"); return 4; }
    float* host = (float*)malloc((total + 6) * sizeof(float));
    CHECK_HIP(hipMemcpyAsync(host, d, total * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(host + total, out, sizeof(x_host), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    FILE* f = fopen(argv[1], "wb");
    if (!f || fwrite(host, sizeof(float), total + 6, f) != total + 6) { fprintf(stderr, "cannot write %s\n", argv[1]); return 5; }
    fclose(f);
    printf("abi_consumer: ABI %d, %zu floats written\n", mdx_abi_version(), total + 6);
    free(host);
    CHECK_HIP(hipFree(out)); CHECK_HIP(hipFree(z)); CHECK_HIP(hipFree(x)); CHECK_HIP(hipFree(d));
    CHECK_HIP(hipStreamDestroy(stream));
    return 0;
}

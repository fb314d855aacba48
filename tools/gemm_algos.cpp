// Times every hipBLASLt heuristic candidate (and optionally all algos) for the EGNN's edge GEMM shape:
//   out[M,N] = SiLU(x[M,K] w[N,K]^T + b), fp32, M ~ 8e5, N = K = 256.   Build: see tools/gemm_algos.sh
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do { auto s_ = (x); if (s_ != 0) { printf("fail %s -> %d line %d\n", #x, (int)s_, __LINE__); exit(1);} } while (0)

int main(int argc, char** argv)
{
    const int64_t M = argc > 1 ? atoll(argv[1]) : 819200;
    const int K = argc > 2 ? atoi(argv[2]) : 256, N = argc > 3 ? atoi(argv[3]) : 256;
    const int all = argc > 4 ? atoi(argv[4]) : 0;
    float *x, *w, *b, *out; void* ws; const uint64_t wsb = 64ull << 20;
    CK(hipMalloc(&x, M * K * 4)); CK(hipMalloc(&w, (size_t)N * K * 4)); CK(hipMalloc(&b, N * 4)); CK(hipMalloc(&out, M * N * 4));
    CK(hipMalloc(&ws, wsb));
    std::vector<float> hx((size_t)M * K), hw((size_t)N * K), hb(N);
    for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) * 6e-5f;
    for (auto& v : hb) v = 0.01f;
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    hipblasLtHandle_t lt; CK(hipblasLtCreate(&lt));
    hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const int32_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
    uint32_t epi = HIPBLASLT_EPILOGUE_SWISH_BIAS_EXT;
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi));
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &b, sizeof(b));
    const float one = 1.0f;
    hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE_ACT_ARG0_EXT, &one, sizeof(one));
    hipblasLtMatrixLayout_t la, lb, ld;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, K, N, K)); CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, K, M, K));
    CK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_32F, N, M, N));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(64);
    int found = 0;
    if (!all) {
        CK(hipblasLtMatmulAlgoGetHeuristic(lt, desc, la, lb, ld, ld, pref, 64, res.data(), &found));
        res.resize(found);
    } else {
        res.clear();
        CK(hipblaslt_ext::getAllAlgos(lt, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, HIPBLAS_OP_T, HIPBLAS_OP_N, HIP_R_32F, HIP_R_32F,
                                      HIP_R_32F, HIP_R_32F, HIPBLAS_COMPUTE_32F, res));
    }
    printf("M=%lld K=%d N=%d candidates=%zu\n", (long long)M, K, N, res.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const float alpha = 1.0f, beta = 0.0f;
    const double flop = 2.0 * M * K * N;
    std::vector<std::pair<float, int>> times;
    for (size_t i = 0; i < res.size(); ++i) {
        size_t need = 0;
        if (hipblaslt_ext::matmulIsAlgoSupported(lt, desc, &alpha, la, lb, &beta, ld, ld, res[i].algo, need) != HIPBLAS_STATUS_SUCCESS || need > wsb)
            continue;
        if (hipblasLtMatmul(lt, desc, &alpha, w, la, x, lb, &beta, out, ld, out, ld, &res[i].algo, ws, wsb, 0) != HIPBLAS_STATUS_SUCCESS) continue;
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 5; ++r) hipblasLtMatmul(lt, desc, &alpha, w, la, x, lb, &beta, out, ld, out, ld, &res[i].algo, ws, wsb, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        times.push_back({ms, (int)i});
        if (!all) printf("  heuristic #%zu idx %d: %.3f ms  %.1f TFLOP/s  %s\n", i, hipblaslt_ext::getIndexFromAlgo(res[i].algo), ms, flop / ms * 1e-9,
               hipblaslt_ext::getKernelNameFromAlgo(lt, res[i].algo).substr(0, 90).c_str());
    }
    std::sort(times.begin(), times.end());
    for (size_t k = 0; k < times.size() && k < 8; ++k) {
        auto& a = res[times[k].second].algo;
        printf("best %zu: idx %d  %.3f ms  %.1f TFLOP/s  %s\n", k, hipblaslt_ext::getIndexFromAlgo(a), times[k].first, flop / times[k].first * 1e-9,
               hipblaslt_ext::getKernelNameFromAlgo(lt, a).substr(0, 110).c_str());
    }
    return 0;
}

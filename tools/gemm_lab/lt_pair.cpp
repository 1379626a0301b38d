// lt_pair.cpp -- which hipBLASLt solution should two CONCURRENT sessions use?
//
// For D[M,N] = relu(X[M,K] W[N,K]^T + b) in bf16 (the heads' hidden layers) this lists the
// heuristic's top solutions with (a) the time of one GEMM alone and (b) the time per GEMM when two
// streams each run that GEMM back to back -- what a session sees while the other session's kernels
// share the chip.  A solution with big tiles can be slower alone (it leaves CUs idle) and still
// win (b) because it costs less CU-time.   Build/run: tools/gemm_lab/run_lt_pair.sh
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#define CK(x) do { auto _e = (x); if (_e != 0) { printf("error %d at %s:%d\n", (int)_e, __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 2048, N = argc > 2 ? atoll(argv[2]) : 1344, K = argc > 3 ? atoll(argv[3]) : 1344;
  const int want = argc > 4 ? atoi(argv[4]) : 48;
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  const size_t ws_bytes = 64 << 20;
  void *x[2], *w, *b, *d[2], *ws[2];
  for (int i = 0; i < 2; i++) { CK(hipMalloc(&x[i], M * K * 2)); CK(hipMalloc(&d[i], M * N * 2)); CK(hipMalloc(&ws[i], ws_bytes)); CK(hipMemset(x[i], 0x3c, M * K * 2)); }
  CK(hipMalloc(&w, N * K * 2)); CK(hipMalloc(&b, N * 2)); CK(hipMemset(w, 0x3b, N * K * 2)); CK(hipMemset(b, 0, N * 2));
  // column-major view: D^T[N,M] = W[N,K] (stored [K,N], ld K, op T) * X^T[K,M] (ld K, op N)
  hipblasLtMatmulDesc_t desc; CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof ta));
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof tb));
  hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_RELU_BIAS;
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof ep));
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &b, sizeof b));
  int32_t bt = HIP_R_16BF; CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt));
  hipblasLtMatrixLayout_t la, lb, ld;
  CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, N, K));
  CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, K, M, K));
  CK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_16BF, N, M, N));
  hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
  CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof ws_bytes));
  std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
  int got = 0;
  CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, want, res.data(), &got));
  printf("M=%ld N=%ld K=%ld: %d heuristic solutions\n", (long)M, (long)N, (long)K, got);
  hipStream_t st[2]; CK(hipStreamCreate(&st[0])); CK(hipStreamCreate(&st[1]));
  hipEvent_t e0, e1, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f1));
  const float alpha = 1.f, beta = 0.f;
  const int reps = 100;
  auto launch = [&](int i, hipblasLtMatmulAlgo_t* a) { return hipblasLtMatmul(h, desc, &alpha, w, la, x[i], lb, &beta, d[i], ld, d[i], ld, a, ws[i], ws_bytes, st[i]); };
  for (int i = 0; i < got; i++) {
    if (res[i].state != HIPBLAS_STATUS_SUCCESS) continue;
    hipblasLtMatmulAlgo_t a = res[i].algo;
    if (launch(0, &a) != HIPBLAS_STATUS_SUCCESS) { printf("%2d: launch failed\n", i); continue; }
    for (int r = 0; r < 10; r++) { launch(0, &a); launch(1, &a); }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st[0]));
    for (int r = 0; r < reps; r++) launch(0, &a);
    CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
    float solo; CK(hipEventElapsedTime(&solo, e0, e1));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st[0])); CK(hipStreamWaitEvent(st[1], e0, 0));
    for (int r = 0; r < reps; r++) { launch(0, &a); launch(1, &a); }
    CK(hipEventRecord(f1, st[1])); CK(hipStreamWaitEvent(st[0], f1, 0));
    CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
    float pair; CK(hipEventElapsedTime(&pair, e0, e1));
    std::string nm = hipblaslt_ext::getKernelNameFromAlgo(h, a);
    size_t p = nm.find("_MT"); std::string mt = p == std::string::npos ? nm.substr(0, 40) : nm.substr(p + 1, 14);
    size_t q = nm.find("_SK"); std::string sk = q == std::string::npos ? "" : nm.substr(q + 1, 4);
    size_t g = nm.find("_GSU"); std::string gsu = g == std::string::npos ? "" : nm.substr(g + 1, 5);
    printf("%2d idx %7d  %-14s %-5s %-5s ws %6zu KB  alone %6.2f us   two streams %6.2f us per GEMM pair = %6.2f us each\n", i, hipblaslt_ext::getIndexFromAlgo(a),
           mt.c_str(), sk.c_str(), gsu.c_str(), res[i].workspaceSize >> 10, solo * 1e3 / reps, pair * 1e3 / reps, pair * 1e3 / reps / 2);
  }
  return 0;
}

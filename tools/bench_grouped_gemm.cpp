// Micro-benchmark: weight-gradient GEMMs dW = dY^T X of L same-shaped layers whose operands are SEPARATE allocations
// (what a deferred weight-gradient phase holds): hipBLASLt grouped GEMM (pointer per problem) vs one strided-batched launch over
// a contiguous arena vs L separate launches.   hipcc -O2 tools/bench_grouped_gemm.cpp -lhipblaslt -o /tmp/bgg
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <vector>

#define CK(x) do { auto _s = (x); if (_s != 0) { printf("FAIL %s -> %d (line %d)\n", #x, (int)_s, __LINE__); return 1; } } while (0)

int run(int L, int64_t M, int64_t K, int64_t N) {   // tokens M, in-features K, out-features N
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  std::vector<void*> X(L), DY(L), DW(L);
  void *arenaX, *arenaDY, *arenaDW;
  CK(hipMalloc(&arenaX, (size_t)L * M * K * 2));
  CK(hipMalloc(&arenaDY, (size_t)L * M * N * 2));
  CK(hipMalloc(&arenaDW, (size_t)L * N * K * 2));
  CK(hipMemset(arenaX, 0, (size_t)L * M * K * 2));
  CK(hipMemset(arenaDY, 0, (size_t)L * M * N * 2));
  for (int i = 0; i < L; ++i) {   // separate allocations with odd gaps between them
    CK(hipMalloc(&X[i], (size_t)M * K * 2 + 4096 * (i % 3)));
    CK(hipMalloc(&DY[i], (size_t)M * N * 2 + 4096 * (i % 5)));
    CK(hipMalloc(&DW[i], (size_t)N * K * 2));
    CK(hipMemset(X[i], 0, (size_t)M * K * 2));
    CK(hipMemset(DY[i], 0, (size_t)M * N * 2));
  }
  void* ws;
  const size_t wsz = 256u << 20;
  CK(hipMalloc(&ws, wsz));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const float alpha = 1.f, beta = 0.f;
  const double fl = 2.0 * L * M * N * K;
  // row-major dW [N,K] = dY^T X  ==  column-major D (K x N) = X_cm (K x M) * dY_cm^T (M x N):  blas m = K, n = N, k = M, opA = N, opB = T
  // ---- grouped
  {
    hipblaslt_ext::GroupedGemm gg(h, HIPBLAS_OP_N, HIPBLAS_OP_T, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF, HIP_R_16BF, HIPBLAS_COMPUTE_32F);
    std::vector<int64_t> m(L, K), n(L, N), k(L, M), bc(L, 1);
    std::vector<hipblaslt_ext::GemmEpilogue> epi(L);
    std::vector<hipblaslt_ext::GemmInputs> in(L);
    for (int i = 0; i < L; ++i) {
      in[i].setA(X[i]); in[i].setB(DY[i]); in[i].setC(DW[i]); in[i].setD(DW[i]); in[i].setAlpha(&alpha); in[i].setBeta(&beta);
    }
    CK(gg.setProblem(m, n, k, bc, epi, in));
    hipblaslt_ext::GemmPreference pref;
    pref.setMaxWorkspaceBytes(wsz);
    std::vector<hipblasLtMatmulHeuristicResult_t> res;
    CK(gg.algoGetHeuristic(4, pref, res));
    printf("  grouped: %zu algos\n", res.size());
    for (size_t a = 0; a < res.size(); ++a) {
      size_t need = 0;
      if (gg.isAlgoSupported(res[a].algo, need) != HIPBLAS_STATUS_SUCCESS || need > wsz) continue;
      CK(gg.initialize(res[a].algo, ws, false, st));
      for (int r = 0; r < 3; ++r) CK(gg.run(st));
      CK(hipEventRecord(e0, st));
      for (int r = 0; r < 10; ++r) CK(gg.run(st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  grouped algo %zu: %8.1f us  %6.0f TF/s (ws %zu)\n", a, ms * 100, fl / (ms / 10 * 1e-3) / 1e12, need);
    }
  }
  // ---- strided batched over the arena, and L separate launches
  for (int mode = 0; mode < 2; ++mode) {
    hipblasLtMatmulDesc_t desc;
    CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    const hipblasOperation_t opA = HIPBLAS_OP_N, opB = HIPBLAS_OP_T;
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opA, sizeof(int32_t)));
    CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opB, sizeof(int32_t)));
    hipblasLtMatrixLayout_t la, lb, ld;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, K, M, K));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, N, M, N));
    CK(hipblasLtMatrixLayoutCreate(&ld, HIP_R_16BF, K, N, K));
    if (mode == 0) {
      int32_t bcnt = L;
      int64_t sa = M * K, sb = M * N, sd = N * K;
      for (auto pr : {std::make_pair(la, sa), std::make_pair(lb, sb), std::make_pair(ld, sd)}) {
        CK(hipblasLtMatrixLayoutSetAttribute(pr.first, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &bcnt, sizeof(bcnt)));
        CK(hipblasLtMatrixLayoutSetAttribute(pr.first, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &pr.second, sizeof(int64_t)));
      }
    }
    hipblasLtMatmulPreference_t pref;
    CK(hipblasLtMatmulPreferenceCreate(&pref));
    uint64_t w = wsz;
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &w, sizeof(w)));
    hipblasLtMatmulHeuristicResult_t res[1];
    int found = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, ld, ld, pref, 1, res, &found));
    if (!found) { printf("  no algo (mode %d)\n", mode); continue; }
    auto once = [&]() {
      if (mode == 0) return (int)hipblasLtMatmul(h, desc, &alpha, arenaX, la, arenaDY, lb, &beta, arenaDW, ld, arenaDW, ld, &res[0].algo, ws, wsz, st);
      int rc = 0;
      for (int i = 0; i < L; ++i) rc |= (int)hipblasLtMatmul(h, desc, &alpha, X[i], la, DY[i], lb, &beta, DW[i], ld, DW[i], ld, &res[0].algo, ws, wsz, st);
      return rc;
    };
    for (int r = 0; r < 3; ++r) CK(once());
    CK(hipEventRecord(e0, st));
    for (int r = 0; r < 10; ++r) CK(once());
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %s: %8.1f us  %6.0f TF/s\n", mode == 0 ? "strided-batched arena" : "separate launches    ", ms * 100, fl / (ms / 10 * 1e-3) / 1e12);
  }
  for (int i = 0; i < L; ++i) { hipFree(X[i]); hipFree(DY[i]); hipFree(DW[i]); }
  hipFree(arenaX); hipFree(arenaDY); hipFree(arenaDW); hipFree(ws);
  return 0;
}

int main() {
  struct { const char* name; int L; int64_t M, K, N; } cases[] = {
      {"cnx s2 pw1", 27, 8192, 512, 2048}, {"cnx s2 pw2", 27, 8192, 2048, 512}, {"pvt s3 fc1", 6, 8192, 320, 1280}, {"pvt s3 q", 6, 8192, 320, 320},
      {"cnx s0 pw1", 3, 131072, 128, 512}, {"pvt s1 fc1", 3, 131072, 64, 512}};
  for (auto& c : cases) {
    printf("%s  L=%d tokens=%lld in=%lld out=%lld\n", c.name, c.L, (long long)c.M, (long long)c.K, (long long)c.N);
    if (run(c.L, c.M, c.K, c.N)) return 1;
    fflush(stdout);
  }
  return 0;
}

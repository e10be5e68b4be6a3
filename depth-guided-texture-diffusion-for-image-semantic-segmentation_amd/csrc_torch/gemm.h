// gemm.h — bf16 / fp16 GEMMs of the Linear layers through hipBLASLt with CACHED plans.
// at::mm / at::addmm re-create the matmul descriptor and re-run hipblasLtMatmulAlgoGetHeuristic on every call (18.5 us of host
// time per GEMM, tools/host_call_cost.py; 543 GEMMs per training step).  Here a plan (descriptor, layouts, heuristic algorithm)
// is built once per (shape, transposes, epilogue, batch) and per thread (forward runs on the Python thread, backward on autograd's
// device thread), so a call is one hipblasLtMatmul.  Library GEMM, not a hand-written kernel: plain GEMMs belong to hipBLASLt.
#pragma once
#include <ATen/ATen.h>
#include <hip/hip_runtime_api.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>

#include <vector>

#include <cstdlib>
#include <unordered_map>

namespace dgemm {

struct Key {
  int64_t m, n, k, sa, sb, sd;
  int ta, tb, bias, batch, dtype;
  bool operator==(const Key& o) const {
    return m == o.m && n == o.n && k == o.k && sa == o.sa && sb == o.sb && sd == o.sd && ta == o.ta && tb == o.tb && bias == o.bias &&
           batch == o.batch && dtype == o.dtype;
  }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    size_t h = 1469598103934665603ull;
    for (int64_t v : {k.m, k.n, k.k, k.sa, k.sb, k.sd, (int64_t)k.ta, (int64_t)k.tb, (int64_t)k.bias, (int64_t)k.batch, (int64_t)k.dtype}) {
      h ^= (size_t)v + 0x9e3779b97f4a7c15ull + (h << 6) + (h >> 2);
    }
    return h;
  }
};
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
  hipblasLtMatmulAlgo_t algo;
  bool ok = false;
};
struct Ctx {
  hipblasLtHandle_t handle = nullptr;
  hipblasLtMatmulPreference_t pref = nullptr;
  std::unordered_map<Key, Plan, KeyHash> plans;
  at::Tensor ws;
  bool dead = false;
};
constexpr size_t kWorkspace = 64u << 20;

inline bool enabled() {
  static const bool on = [] { const char* e = std::getenv("DGTD_DIRECT_GEMM"); return !(e && e[0] == '0'); }();
  return on;
}

// Pick among the heuristic's first candidates by measurement when a plan is created (once per shape and thread).  With the step
// replayed as a hipGraph (GPU-bound) this is worth 0.7 ms of 33.4 per step at config 2 (245 vs 240 images/s); in the host-bound eager
// step of round 1 it showed nothing.  DGTD_GEMM_TUNE=0 takes the heuristic's first answer.
inline bool tune() {
  static const bool on = [] { const char* e = std::getenv("DGTD_GEMM_TUNE"); return !(e && e[0] == '0'); }();
  return on;
}

// DGTD_GEMM_EXHAUSTIVE=1: time EVERY library solution that supports the problem instead of the heuristic's first candidates
// (hipblaslt_ext::getAllAlgos; hundreds per plan, seconds per shape - an offline-tuning switch, off by default)
inline bool exhaustive() {
  static const bool on = [] { const char* e = std::getenv("DGTD_GEMM_EXHAUSTIVE"); return e && e[0] == '1'; }();
  return on;
}

// One context (handle, plan cache, 64 MB workspace) per thread AND stream: the deferred weight-gradient phase can run on a side stream
// beside the backward pass (bindings.cpp flush_deferred_async); two GEMMs in flight on different streams must not share a workspace.
inline Ctx& ctx(hipStream_t st) {   // never destroyed: a thread_local destructor could run after the HIP runtime has shut down
  thread_local std::unordered_map<hipStream_t, Ctx*>* m = new std::unordered_map<hipStream_t, Ctx*>();
  auto it = m->find(st);
  if (it == m->end()) it = m->emplace(st, new Ctx()).first;
  return *it->second;
}

inline bool ok(hipblasStatus_t s) { return s == HIPBLAS_STATUS_SUCCESS; }

inline void set_batch(hipblasLtMatrixLayout_t l, int32_t batch, int64_t stride) {
  hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch));
  hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &stride, sizeof(stride));
}

// D[batch][M][N] (row-major) = op(A) op(B) (+ bias[N]);  A is [M,K] (transA: [K,M]), B is [K,N] (transB: [N,K]), all of the 16-bit type `half_type` (at::kBFloat16 or at::kHalf), dense.
// Returns false when hipBLASLt offers no algorithm for the problem: the caller then uses the ATen GEMM.
inline bool matmul_16(at::ScalarType half_type, const void* A, const void* B, void* D, const void* bias, int64_t M, int64_t N, int64_t K,
                      bool transA, bool transB, int batch, int64_t sA, int64_t sB, int64_t sD, const at::TensorOptions& dev_opts, hipStream_t st) {
  if (!enabled()) return false;
  const hipDataType DT = half_type == at::kHalf ? HIP_R_16F : HIP_R_16BF;
  Ctx& c = ctx(st);
  if (c.dead) return false;
  if (!c.handle) {
    uint64_t wsz = kWorkspace;
    if (!ok(hipblasLtCreate(&c.handle)) || !ok(hipblasLtMatmulPreferenceCreate(&c.pref)) ||
        !ok(hipblasLtMatmulPreferenceSetAttribute(c.pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)))) {
      c.dead = true;
      return false;
    }
    c.ws = at::empty({(int64_t)kWorkspace}, dev_opts.dtype(at::kByte));
  }
  const Key key{M, N, K, sA, sB, sD, (int)transA, (int)transB, bias ? 1 : 0, batch, (int)DT};
  auto it = c.plans.find(key);
  if (it == c.plans.end()) {
    Plan p;
    // column-major view of the row-major problem: D^T (N x M) = op(B)^T-as-stored ... i.e. blas A := B memory, blas B := A memory
    const hipblasOperation_t opA = transB ? HIPBLAS_OP_T : HIPBLAS_OP_N, opB = transA ? HIPBLAS_OP_T : HIPBLAS_OP_N;
    bool good = ok(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    good = good && ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opA, sizeof(int32_t)));
    good = good && ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opB, sizeof(int32_t)));
    if (good && bias) {
      const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
      const int32_t bt = (int32_t)DT;
      good = ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi))) &&
             ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt))) &&
             ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(void*)));
    }
    good = good && ok(transB ? hipblasLtMatrixLayoutCreate(&p.a, DT, K, N, K) : hipblasLtMatrixLayoutCreate(&p.a, DT, N, K, N));
    good = good && ok(transA ? hipblasLtMatrixLayoutCreate(&p.b, DT, M, K, M) : hipblasLtMatrixLayoutCreate(&p.b, DT, K, M, K));
    good = good && ok(hipblasLtMatrixLayoutCreate(&p.d, DT, N, M, N));
    if (good && batch > 1) {
      set_batch(p.a, batch, sB);
      set_batch(p.b, batch, sA);
      set_batch(p.d, batch, sD);
    }
    if (good) {
      constexpr int kMaxTry = 32;
      static const int kTry = [] { const char* e = std::getenv("DGTD_GEMM_CANDIDATES"); const int v = e ? std::atoi(e) : 32; return v < 1 ? 1 : (v > kMaxTry ? kMaxTry : v); }();
      hipblasLtMatmulHeuristicResult_t res[kMaxTry];
      int found = 0;
      // a plan first needed while the stream is being captured cannot be timed (event synchronisation is illegal there): first answer
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const bool timing = tune() && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
      good = ok(hipblasLtMatmulAlgoGetHeuristic(c.handle, p.desc, p.a, p.b, p.d, p.d, c.pref, timing ? kTry : 1, res, &found)) && found > 0;
      std::vector<hipblasLtMatmulHeuristicResult_t> all;
      if (good && timing && exhaustive()) {
        const float one = 1.f, zero = 0.f;
        std::vector<hipblasLtMatmulHeuristicResult_t> cand;
        if (ok(hipblaslt_ext::getAllAlgos(c.handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opA, opB, DT, DT, DT, DT, HIPBLAS_COMPUTE_32F, cand))) {
          for (auto& r : cand) {
            size_t need = 0;
            if (ok(hipblaslt_ext::matmulIsAlgoSupported(c.handle, p.desc, &one, p.a, p.b, &zero, p.d, p.d, r.algo, need)) && need <= kWorkspace) {
              r.workspaceSize = need;
              r.state = HIPBLAS_STATUS_SUCCESS;
              all.push_back(r);
            }
          }
        }
      }
      hipblasLtMatmulHeuristicResult_t* list = res;
      if (!all.empty()) {
        for (int i = 0; i < found; ++i) all.push_back(res[i]);
        list = all.data();
        found = (int)all.size();
      }
      int best = -1;
      if (good && found > 1) {
        // one-time selection among the heuristic's candidates by measurement (the top-1 pick is not always the fastest for the
        // skinny token-major shapes of this model): 1 warm-up + 3 timed runs each, on the caller's stream.  (7 timed runs pick
        // differently - solutions that win back to back on L2-hot operands - and the step is 0.4 ms slower: measured, not adopted.)
        const float alpha = 1.f, beta = 0.f;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        float best_ms = 1e30f;
        for (int i = 0; i < found; ++i) {
          if (list[i].state != HIPBLAS_STATUS_SUCCESS || list[i].workspaceSize > kWorkspace) continue;
          bool run_ok = true;
          for (int r = 0; r < 4 && run_ok; ++r) {       // 1 warm-up + 3 timed
            if (r == 1) (void)hipEventRecord(e0, st);
            run_ok = ok(hipblasLtMatmul(c.handle, p.desc, &alpha, B, p.a, A, p.b, &beta, D, p.d, D, p.d, &list[i].algo, c.ws.data_ptr(),
                                        kWorkspace, st));
          }
          (void)hipEventRecord(e1, st);
          (void)hipEventSynchronize(e1);
          float ms = 0.f;
          if (run_ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best_ms) { best_ms = ms; best = i; }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
      } else if (good && list[0].workspaceSize <= kWorkspace) {
        best = 0;
      }
      good = best >= 0;
      if (good) p.algo = list[best].algo;
    }
    p.ok = good;
    it = c.plans.emplace(key, p).first;
  }
  Plan& p = it->second;
  if (!p.ok) return false;
  if (bias && !ok(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(void*)))) return false;
  const float alpha = 1.f, beta = 0.f;
  return ok(hipblasLtMatmul(c.handle, p.desc, &alpha, B, p.a, A, p.b, &beta, D, p.d, D, p.d, &p.algo, c.ws.data_ptr(), kWorkspace, st));
}

}  // namespace dgemm

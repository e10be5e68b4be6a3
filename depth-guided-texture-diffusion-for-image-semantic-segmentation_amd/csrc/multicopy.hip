// multicopy.hip — many small tensors <-> one flat buffer in ONE launch per <= 128 tensors, with dtype conversion.
// Replaces torch.cat / torch.stack on the training path: the gradient reducer's bucket gather (leaf gradients -> flat fp32 bucket,
// dist/reducer.py), the per-step stacks of the 16 prompt-decoder weights (ops/conv3x3.py) and of the five low-resolution logit maps
// (ops/loss.py).  Two reasons: (i) one pass (16-bit gradient -> fp32 bucket) instead of concat + cast; (ii) on ROCm torch.cat passes
// its tensor table through a PINNED HOST buffer + an H2D copy; captured in a hipGraph, every replay re-reads that host buffer, which
// the host allocator has meanwhile handed to some later torch.cat - replays then gather from stale pointers (the NaNs of the
// round-1 graph attempt, tools/debug_graph_nan.py).  Here the table travels BY VALUE in the kernel arguments (< 4 KB), which a graph
// node owns.  HBM-bound: (src + dst element size) * n bytes.
#include "common.h"

namespace {

constexpr int MC_MAX = 128;        // table entries per launch: 128 * (8 + 8 + 8 + 4) B = 3.5 KB of kernel arguments
constexpr int MC_CHUNK = 2048;     // elements per workgroup-iteration (256 lanes x 8)

struct McTable {
  const void* src[MC_MAX];
  int64_t dst_off[MC_MAX];         // element offset into dst
  int64_t n[MC_MAX];               // elements
  int first_chunk[MC_MAX + 1];     // prefix sum of ceil(n / MC_CHUNK)
  int count;
};

template <typename S, typename D>
__global__ __launch_bounds__(256) void multi_copy_kernel(McTable t, D* __restrict__ dst, int to_tensors) {
  // to_tensors = 0: tensors (t.src) -> flat (dst);  1: flat (dst, read) -> tensors (t.src, written)
  const int chunk = blockIdx.x;
  int lo = 0, hi = t.count;                       // entry e with first_chunk[e] <= chunk < first_chunk[e + 1]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t.first_chunk[mid] <= chunk) lo = mid; else hi = mid; }
  const int e = lo;
  const int64_t base = (int64_t)(chunk - t.first_chunk[e]) * MC_CHUNK;
  const int64_t n = t.n[e];
  S* s = (S*)t.src[e];
  D* d = dst + t.dst_off[e];
  // 8 elements per lane as one vector access each side when the tensor's and the slot's addresses allow it (16-byte multiples:
  // the reducer aligns every slot to 8 elements), scalars otherwise and on the ragged tail
  typedef S s8 __attribute__((ext_vector_type(8)));
  typedef D d8 __attribute__((ext_vector_type(8)));
  const int64_t i0 = base + (int64_t)threadIdx.x * 8;
  const bool vec = (((uintptr_t)s | (uintptr_t)d) & 15) == 0 && i0 + 8 <= n;
  if (vec) {
    if (to_tensors) {
      const d8 v = *reinterpret_cast<const d8*>(d + i0);
      s8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (S)(float)v[j];
      *reinterpret_cast<s8*>(s + i0) = o;
    } else {
      const s8 v = *reinterpret_cast<const s8*>(s + i0);
      d8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (D)(float)v[j];
      *reinterpret_cast<d8*>(d + i0) = o;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int64_t i = i0 + j;
    if (i < n) {
      if (to_tensors) s[i] = (S)(float)d[i];
      else d[i] = (D)(float)s[i];
    }
  }
}

template <typename S, typename D>
int launch_tables(const void* const* tensors, const int64_t* offs, const int64_t* ns, int count, void* flat, int to_tensors, hipStream_t st) {
  for (int b0 = 0; b0 < count; b0 += MC_MAX) {
    McTable t;
    const int m = std::min(MC_MAX, count - b0);
    t.count = m;
    int chunks = 0;
    for (int i = 0; i < m; ++i) {
      t.src[i] = tensors[b0 + i];
      t.dst_off[i] = offs[b0 + i];
      t.n[i] = ns[b0 + i];
      t.first_chunk[i] = chunks;
      chunks += (int)cdiv(ns[b0 + i], MC_CHUNK);
    }
    for (int i = m; i <= MC_MAX; ++i) t.first_chunk[i] = chunks;
    if (chunks == 0) continue;
    hipLaunchKernelGGL((multi_copy_kernel<S, D>), dim3(chunks), dim3(256), 0, st, t, (D*)flat, to_tensors);
    DGTD_CHECK_LAUNCH("multi_copy");
  }
  return 0;
}

template <typename S>
int by_dst(const void* const* tensors, const int64_t* offs, const int64_t* ns, int count, void* flat, dgtd_dtype ft, int to_tensors, hipStream_t st) {
  if (ft == DGTD_F32) return launch_tables<S, float>(tensors, offs, ns, count, flat, to_tensors, st);
  if (ft == DGTD_BF16) return launch_tables<S, bf16_t>(tensors, offs, ns, count, flat, to_tensors, st);
  if (ft == DGTD_F16) return launch_tables<S, f16_t>(tensors, offs, ns, count, flat, to_tensors, st);
  DGTD_FAIL(2, "multi_copy: bad flat dtype %d", (int)ft);
}

}  // namespace

extern "C" int dgtd_multi_copy(const void* const* tensors, const int64_t* offsets, const int64_t* counts, int n_tensors, dgtd_dtype tensor_dt,
                               void* flat, dgtd_dtype flat_dt, int to_tensors, dgtd_stream s) {
  DGTD_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (tensors && offsets && counts && flat)), "multi_copy: bad arguments");
  if (n_tensors == 0) return 0;
  double total = 0;
  for (int i = 0; i < n_tensors; ++i) {
    DGTD_REQUIRE((tensors[i] || counts[i] == 0) && counts[i] >= 0 && offsets[i] >= 0, "multi_copy: bad entry %d", i);
    total += (double)counts[i];
  }
  DGTD_PROF(s, DGTD_HBM, total * (dgtd_esize(tensor_dt) + dgtd_esize(flat_dt)), "dgtd_multi_copy[n=%d,elems=%.0f]", n_tensors, total);
  hipStream_t st = (hipStream_t)s;
  if (tensor_dt == DGTD_F32) return by_dst<float>(tensors, offsets, counts, n_tensors, flat, flat_dt, to_tensors, st);
  if (tensor_dt == DGTD_BF16) return by_dst<bf16_t>(tensors, offsets, counts, n_tensors, flat, flat_dt, to_tensors, st);
  if (tensor_dt == DGTD_F16) return by_dst<f16_t>(tensors, offsets, counts, n_tensors, flat, flat_dt, to_tensors, st);
  DGTD_FAIL(2, "multi_copy: bad tensor dtype %d", (int)tensor_dt);
}

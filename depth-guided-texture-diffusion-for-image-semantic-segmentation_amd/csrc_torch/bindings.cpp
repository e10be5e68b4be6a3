// bindings.cpp — torch C++ autograd bindings over the C ABI of libdgtd.so (include/dgtd.h).
// Same role as the reference's twig/ops/src/vision.cpp + functions/ms_deform_attn_func.py: tensors in, the C ABI called with raw
// pointers on the current HIP stream, one autograd node per op.  Host plumbing only (no device code here): it exists because the
// training step is host-bound in eager Python (~1600 Python autograd.Function calls per step); these nodes cost a few microseconds.
#include <ATen/ATen.h>
#include <atomic>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <set>
#include <tuple>
#include <mutex>
#include <vector>
#include <c10/hip/HIPStream.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPCachingAllocator.h>
#include <torch/autograd.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/library.h>

#include "../../include/dgtd.h"
#include "gemm.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

inline void* stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
inline dgtd_dtype code(const Tensor& t) {
  if (t.scalar_type() == at::kFloat) return DGTD_F32;
  if (t.scalar_type() == at::kBFloat16) return DGTD_BF16;
  if (t.scalar_type() == at::kHalf) return DGTD_F16;
  TORCH_CHECK(false, "dgtd kernels take float32, bfloat16 or float16 tensors, got ", t.scalar_type());
}
inline void check(int rc, const char* name) { TORCH_CHECK(rc == 0, name, " failed (code ", rc, "): ", dgtd_last_error()); }
inline void on_device(const Tensor& t) { TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "dgtd ops need contiguous tensors on the HIP device"); }
inline Tensor f32(const Tensor& t) { return t.scalar_type() == at::kFloat ? t.contiguous() : t.to(at::kFloat).contiguous(); }
inline Tensor undefined() { return Tensor(); }
inline bool is16(at::ScalarType t) { return t == at::kBFloat16 || t == at::kHalf; }
inline at::ScalarType from_code(int64_t dt_code) { return dt_code == DGTD_BF16 ? at::kBFloat16 : (dt_code == DGTD_F16 ? at::kHalf : at::kFloat); }
inline int64_t st_id(const Tensor& t) { return (int64_t)t.scalar_type(); }                  // a dtype remembered across forward -> backward
inline at::ScalarType st_of(const c10::IValue& v) { return (at::ScalarType)v.toInt(); }

// ------------------------------------------------------------------------------------------------ deferred second-stage reductions
// Every backward node that also produces a per-column sum (LayerNorm dgamma/dbeta, Linear bias gradients, layer-scale gradients) needs
// a second pass over its per-workgroup partial rows: ~290 launches of ~6 us per training step.  Between `set_deferred(true)` and the
// next `flush_deferred()` (the gradient reducer brackets backward() with them, dist/reducer.py) the nodes run only their first stage,
// hand out the still unwritten gradient tensors and park (partial buffer, destination) here; ONE dgtd_multi_reduce per 56 entries then
// fills them all.  Nothing may read those gradients before the flush: the reducer's gather does so only after it, and parameters
// shared between call sites (autograd adds their gradients on the fly) never take this path (only LayerNorm, Linear bias and gamma
// gradients of the two trunks do; their parameters are used once per step).  Destinations are remembered as raw pointers: holding the
// tensors would stop AccumulateGrad from stealing them (it would clone the unwritten memory instead).
//
// SOUNDNESS RULE: an unwritten tensor may be handed to autograd only if autograd will STEAL it as the leaf's .grad and no
// other gradient of that leaf arrives before the flush.  Every node that defers a parameter gradient therefore asks may_defer(leaf)
// in its backward: the leaf must be a true leaf seen by exactly ONE deferring forward call in this window (a bias shared by two call
// sites, a module called twice: immediate path for all its calls, autograd's accumulation adds then only see written tensors), and
// its .grad must still be undefined - if an earlier backward of the window already handed one over (gradient accumulation, two
// backward() calls, retain_graph), the pending work is flushed first, so AccumulateGrad's `grad += new` reads written memory, and the
// call takes the immediate path.
struct PendingReduce { dgtd_reduce_entry e; Tensor ws; };
static std::mutex g_pending_mu;
static std::map<const void*, int> g_uses;                 // leaf data pointer -> deferring forward calls in this window
static std::vector<PendingReduce> g_pending;
static std::atomic<bool> g_defer{false};

inline bool deferring() { return g_defer.load(std::memory_order_relaxed); }
void flush_deferred();
static void refresh_transposed();
inline bool own_gemm_on();
inline bool is_leaf(const Tensor& t) { return t.defined() && t.requires_grad() && !t.grad_fn(); }
// forward side: remember the leaf (kept in the node, so its .grad can be inspected in the backward) and count the call
inline void note_leaf(AutogradContext* ctx, const char* key, const Tensor& t) {
  if (!deferring() || !is_leaf(t)) return;
  ctx->saved_data[key] = t;
  std::lock_guard<std::mutex> lk(g_pending_mu);
  ++g_uses[t.data_ptr()];
}
inline bool may_defer(AutogradContext* ctx, const char* key) {
  if (!deferring()) return false;
  auto it = ctx->saved_data.find(key);
  if (it == ctx->saved_data.end() || !it->second.isTensor()) return false;
  const Tensor& leaf = it->second.toTensor();
  if (leaf.grad().defined()) { flush_deferred(); return false; }
  std::lock_guard<std::mutex> lk(g_pending_mu);
  auto u = g_uses.find(leaf.data_ptr());
  return u != g_uses.end() && u->second == 1;
}
// SAFETY NET: whoever parks work during a backward pass also queues ONE engine callback that flushes at the end of that pass.  Under the
// reducer this is where the single flush of a step happens anyway (its own flush calls then find nothing); without it - a step that
// raised between zero_grad() and finish() leaves deferral switched on, and a later plain backward() would otherwise return with
// unwritten gradients.
static std::atomic<bool> g_cb_queued{false};
inline void queue_final_flush() {
  if (g_cb_queued.exchange(true)) return;
  try {
    torch::autograd::Engine::get_default_engine().queue_callback([] { g_cb_queued.store(false); flush_deferred(); });
  } catch (...) {            // not inside a backward pass (a node run by hand): the caller flushes
    g_cb_queued.store(false);
  }
}
inline void park(const Tensor& ws, int nblocks, int ncols, float* outA, int nA, void* outB, dgtd_dtype dtB, int tr_rows = 0, int tr_cols = 0,
                 void* outC = nullptr) {
  queue_final_flush();
  std::lock_guard<std::mutex> lk(g_pending_mu);
  g_pending.push_back(PendingReduce{dgtd_reduce_entry{ws.data_ptr<float>(), nblocks, ncols, outA, nA, outB, (int32_t)dtB, tr_rows, tr_cols, outC}, ws});
}
// Deferred WEIGHT GRADIENTS of the depthwise convolutions: the node only remembers its (input, output-gradient) pair; at the flush
// every parked layer of one shape is served by ONE dgtd_dwconv_bwd_weight_batched launch (27 ConvNeXt stage-3 layers at config 2:
// 453 MB streamed once, instead of 27 launches of 16 MB that are over before the chip is busy and write 4x their input in partials).
struct PendingDw { Tensor x, du; void* dw; void* db; int32_t dw_dt; bool has_bias; int B, H, W, C, K; };
static std::vector<PendingDw> g_pending_dw;

inline void park_dw(const Tensor& x, const Tensor& du, Tensor& dw, Tensor& db, bool has_bias, int C, int K) {
  queue_final_flush();
  std::lock_guard<std::mutex> lk(g_pending_mu);
  g_pending_dw.push_back(PendingDw{x, du, dw.data_ptr(), has_bias ? db.data_ptr() : nullptr, (int32_t)code(dw), has_bias, (int)x.size(0), (int)x.size(1),
                                   (int)x.size(2), C, K});
}
// Deferred WEIGHT GRADIENTS of the Linear layers of a run of identical blocks (ConvNeXt stage: 27 x pwconv1 / pwconv2 at config 2).
// One layer's dW = dY^T X is a GEMM with a few hundred output tiles and an 8192-long reduction: the library needs split-K plus a
// separate sum to fill the chip (36 us per layer).  The same GEMM batched over all layers of the stage fills it by itself
// (tools/bench_grouped_gemm.cpp: 27 layers 344 us instead of 27 x 36).  hipBLASLt batches by STRIDE, so the operands of layer i must
// sit at base + i * stride: while a block runs under an "arena hint" (group id, layer index, layer count; set by the module), the
// producers of X (LayerNorm output, GELU output) and of dY (GELU' / layer-scale backward) write straight into slot i of a
// per-(group, role) arena [count, rows, cols] instead of a fresh allocation - no copies - and the weight gradient itself is slot i
// of a dW arena handed to autograd as the gradient tensor and filled at the flush (same contract as the parked reductions).
// Arena roles of a block: X[k] = input of its k-th deferred Linear, DY[k] = gradient w.r.t. that Linear's output, DW[k] = its weight
// gradient.  The module says which role the NEXT arena-capable node writes (arena_roles: forward output -> X[out]; the gradients its
// backward produces -> DY[ga], DY[gb]); a Linear node finds its own k by looking its input / output-gradient pointers up in the arenas.
constexpr int ROLE_X = 0, ROLE_DY = 8, ROLE_DW = 16, NROLE = 8;
struct Hint { int64_t group = -1; int idx = 0, count = 0; };
struct Roles { int out = -1, ga = -1, gb = -1; };
static thread_local Hint t_hint;
static thread_local Roles t_roles;
void arena_roles(int64_t out, int64_t ga, int64_t gb) { t_roles = Roles{(int)out, (int)ga, (int)gb}; }
inline int take_out_role() { const int r = t_roles.out; t_roles.out = -1; return r < NROLE ? r : -1; }
inline std::pair<int, int> take_grad_roles() {
  const std::pair<int, int> r{t_roles.ga < NROLE ? t_roles.ga : -1, t_roles.gb < NROLE ? t_roles.gb : -1};
  t_roles.ga = t_roles.gb = -1;
  return r;
}
static std::map<std::pair<int64_t, int>, Tensor> g_arenas;
// slots handed out in this window: a slot requested twice before the flush (two forwards of a module before one backward, a second
// backward through a retained graph) would overwrite what autograd saved from the first request - the second request gets a plain
// allocation instead (and, not being in an arena, takes the per-layer weight-gradient path)
static std::set<std::tuple<int64_t, int, int>> g_slot_used;
// a captured hipGraph holds raw arena addresses: while pinned, re-allocating an arena is an error (runner/graph.py)
static std::map<int64_t, int> g_arena_pins;              // group -> number of live graphs referencing it
void arena_pin(int64_t group, bool on) {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  if (on) ++g_arena_pins[group];
  else if (g_arena_pins.count(group) && --g_arena_pins[group] <= 0) g_arena_pins.erase(group);
}

void arena_hint(int64_t group, int64_t idx, int64_t count) { t_hint = Hint{count > 1 ? group : -1, (int)idx, (int)count}; t_roles = Roles{}; }
void arena_release(int64_t group) {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  g_arena_pins.erase(group);
  for (auto it = g_arenas.begin(); it != g_arenas.end();) it = it->first.first == group ? g_arenas.erase(it) : std::next(it);
}
int64_t arena_bytes() {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  int64_t n = 0;
  for (auto& kv : g_arenas) n += kv.second.numel() * kv.second.element_size();
  return n;
}
inline bool hinted(const Hint& h, at::ScalarType dt) { return h.group >= 0 && is16(dt) && deferring(); }
inline Tensor arena_slot(const Hint& h, int role, at::IntArrayRef shape, const at::TensorOptions& opt) {
  std::vector<int64_t> full{h.count};
  full.insert(full.end(), shape.begin(), shape.end());
  std::lock_guard<std::mutex> lk(g_pending_mu);
  if (!g_slot_used.insert({h.group, role, h.idx}).second) return at::empty(shape, opt);
  Tensor& a = g_arenas[{h.group, role}];
  if (!a.defined() || a.sizes() != at::IntArrayRef(full) || a.scalar_type() != c10::typeMetaToScalarType(opt.dtype()) || a.device() != opt.device()) {
    TORCH_CHECK(!g_arena_pins.count(h.group), "dgtd: a captured hipGraph references the per-stage arenas; a grad-enabled step with another shape would "
                "re-allocate them under it (GraphedTrainStep.release() first)");
    a = at::empty(full, opt);
  }
  // NOT a view: views share the arena's version counter, and every write into another slot would invalidate the tensors autograd
  // saved from this one.  A blob over the slot's memory with its own storage object (the deleter keeps the arena alive).
  Tensor keep = a;
  int64_t n = 1;
  for (auto d : shape) n *= d;
  return at::from_blob((char*)a.data_ptr() + (int64_t)h.idx * n * a.element_size(), shape, [keep](void*) mutable { keep.reset(); }, opt);
}
inline bool in_arena(const Hint& h, int role, const Tensor& t) {       // is t exactly slot idx of this arena?
  std::lock_guard<std::mutex> lk(g_pending_mu);
  auto it = g_arenas.find({h.group, role});
  return it != g_arenas.end() && it->second.defined() && it->second.size(0) == h.count && t.is_contiguous() &&
         it->second.numel() / h.count == t.numel() && t.data_ptr() == (char*)it->second.data_ptr() + (int64_t)h.idx * t.numel() * t.element_size();
}
inline int find_role(const Hint& h, int base, const Tensor& t) {      // k with t == slot idx of arena base + k, or -1
  if (h.group < 0) return -1;
  for (int k = 0; k < NROLE; ++k)
    if (in_arena(h, base + k, t)) return k;
  return -1;
}
// what a node keeps for its backward: the hint + up to two role indices
inline c10::IValue hint_iv(const Hint& h, int r0 = -1, int r1 = -1) { return std::vector<int64_t>{h.group, h.idx, h.count, r0, r1}; }
inline Hint hint_of(const c10::IValue& v) { auto m = v.toIntVector(); return Hint{m[0], (int)m[1], (int)m[2]}; }
inline int role_of(const c10::IValue& v, int i) { return (int)v.toIntVector()[3 + i]; }

struct PendingGemm { int64_t group; int which, idx; Tensor x, dy; void* dw; int64_t M, N, K; };   // dW [N,K] = dy[M,N]^T x[M,K]
static std::vector<PendingGemm> g_pending_gemm;
inline void park_gemm(const Hint& h, int which, const Tensor& x2, const Tensor& dy2, const Tensor& dw) {
  queue_final_flush();
  std::lock_guard<std::mutex> lk(g_pending_mu);
  g_pending_gemm.push_back(PendingGemm{h.group, which, h.idx, x2, dy2, dw.data_ptr(), dy2.size(0), dy2.size(1), x2.size(1)});
}
Tensor gemm_dw(const Tensor& dy2, const Tensor& x2);
namespace dgemm_fwd { bool batched_dw(at::ScalarType dt, const void* dy, const void* x, void* dw, int64_t N, int64_t K, int64_t M, int batch, int64_t s_dy,
                                      int64_t s_x, int64_t s_dw, const at::TensorOptions& o); }
static void flush_gemms(std::vector<PendingGemm>& gs) {
  std::sort(gs.begin(), gs.end(), [](const PendingGemm& a, const PendingGemm& b) {
    return std::tie(a.group, a.which, a.idx) < std::tie(b.group, b.which, b.idx); });
  for (size_t i = 0; i < gs.size();) {
    // maximal run of consecutive layers of one (group, linear) whose operands are equally spaced: one strided-batched GEMM
    const PendingGemm& a = gs[i];
    const int64_t es = a.x.element_size();
    size_t j = i + 1;
    int64_t sx = 0, sdy = 0, sdw = 0;
    while (j < gs.size() && gs[j].group == a.group && gs[j].which == a.which && gs[j].idx == gs[j - 1].idx + 1 && gs[j].M == a.M && gs[j].N == a.N &&
           gs[j].K == a.K) {
      const int64_t dx = ((char*)gs[j].x.data_ptr() - (char*)gs[j - 1].x.data_ptr()) / es, ddy = ((char*)gs[j].dy.data_ptr() - (char*)gs[j - 1].dy.data_ptr()) / es,
                    ddw = ((char*)gs[j].dw - (char*)gs[j - 1].dw) / es;
      if (j == i + 1) { sx = dx; sdy = ddy; sdw = ddw; }
      else if (dx != sx || ddy != sdy || ddw != sdw) break;
      ++j;
    }
    const int n = (int)(j - i);
    // few layers with a long token dimension (3 blocks x 131072 tokens at stage 0): the batch alone does not fill the chip, so each
    // layer is additionally split along the tokens - the slots are dense, so (layer, split) is still ONE uniform stride - and the
    // bf16 partials are summed per layer (fp32 accumulation inside the reduction), as the per-layer path does
    static const int64_t max_split = [] { const char* e = std::getenv("DGTD_WGRAD_SPLIT"); return e ? (int64_t)std::atol(e) : (int64_t)32; }();
    int64_t S = n >= 16 ? 1 : std::min<int64_t>(max_split, a.M / 1024);
    while (S > 1 && a.M % S) --S;
    const bool dense = n == 1 || (sx == a.M * a.K && sdy == a.M * a.N && sdw == a.N * a.K);
    bool done = false;
    if (S >= 2 && dense) {
      Tensor part = at::empty({n * S, a.N * a.K}, a.x.options());
      if (dgemm_fwd::batched_dw(a.x.scalar_type(), a.dy.data_ptr(), a.x.data_ptr(), part.data_ptr(), a.N, a.K, a.M / S, (int)(n * S), (a.M / S) * a.N,
                                (a.M / S) * a.K, a.N * a.K, a.x.options())) {
        Tensor out = at::from_blob(a.dw, {n, a.N * a.K}, a.x.options());
        at::sum_out(out, part.view({n, S, a.N * a.K}), {1});
        done = true;
      }
    }
    if (!done && !dgemm_fwd::batched_dw(a.x.scalar_type(), a.dy.data_ptr(), a.x.data_ptr(), a.dw, a.N, a.K, a.M, n, sdy, sx, sdw, a.x.options())) {
      for (size_t k = i; k < j; ++k) {     // no library plan: the per-layer path, copied into the promised gradient slots
        Tensor dw = gemm_dw(gs[k].dy, gs[k].x);
        Tensor dst = at::from_blob(gs[k].dw, {gs[k].N, gs[k].K}, dw.options());
        dst.copy_(dw);
      }
    }
    i = j;
  }
}

// Deferred WEIGHT GRADIENTS of the single dense 3x3 convolutions (Hitnet CAB bodies, conv4: small maps, 20-40 us launches that
// write several times their input in partial sums).  Parked per call; at the flush all calls of one geometry go through ONE
// dgtd_conv3x3_wgrad_batched launch.  A module called m times per step (the CABs of the iterative decoder) has ONE gradient tensor:
// the first parked call hands it to autograd, the later calls return nothing, and the reduce kernel sums the partials of all m calls
// into it (no per-call gradients, no autograd adds).  That is only sound when EVERY call of the weight in this step is deferred into
// the same launch: the forward registers (weight, geometry) pairs, and a weight seen with two geometries, or through a non-leaf
// tensor, keeps the immediate path for the whole step.
struct ConvKey { int B, H, W, Ci, Co; int dt; bool operator==(const ConvKey& o) const { return B == o.B && H == o.H && W == o.W && Ci == o.Ci && Co == o.Co && dt == o.dt; } };
struct ConvSeen { ConvKey k; bool mixed; };
struct ConvDest { void* dw; void* db; };
struct PendingConv { Tensor x, dy, mask; const void* w; ConvKey k; };
static std::map<const void*, ConvSeen> g_conv_seen;      // forward registry of this step
static std::map<const void*, ConvDest> g_conv_dest;      // backward: weight -> the one gradient tensor handed to autograd
static std::vector<PendingConv> g_pending_conv;
// the one PReLU slope shared by every activation of the decoder (cod.py:686): all backward calls of a step add into ONE fp32 accumulator
// (first call: zeroed here and handed to autograd through a parked 1x1 "reduction" that converts it; later calls return nothing)
static std::map<const void*, Tensor> g_prelu_acc;
static std::map<const void*, Tensor> g_conv_flip;        // weight -> its flipped copy, valid for one step (cleared at every flush)
// the two 1x1 weights of a CAB's channel-attention MLP (cod.py:420-425) get one gradient per call of the module (4 per step): every call
// writes its { dw1 | dw2 } row into a per-module buffer, the first call hands autograd the (unwritten) sums, the flush adds the rows up
struct CaAcc { Tensor buf; int count, rc; void* out1; void* out2; int rows = 1; };   // rows: buffer rows written per call
static std::map<const void*, CaAcc> g_ca_acc;            // key: the first weight's data pointer
constexpr int CA_MAX_CALLS = 8;

// Deferrals that hand ONE gradient tensor to autograd for several calls are only sound when nothing is flushed between the first and
// the last of those calls: the reducer switches them off when it gathers buckets from inside the backward pass (eager overlap mode).
static std::atomic<bool> g_shared_ok{true};
void set_shared_deferral(bool on) { g_shared_ok.store(on); }
inline bool conv_defer_on() {
  static const bool on = [] { const char* e = std::getenv("DGTD_DEFER_CONV3X3"); return !e || std::atoi(e) != 0; }();
  return on && g_shared_ok.load(std::memory_order_relaxed);
}
// forward: note that this weight runs with this geometry; returns whether the call may be deferred as far as the forward can tell
inline bool conv_register(const Tensor& w_arg, const Tensor& w, const ConvKey& k) {
  if (!deferring() || !conv_defer_on()) return false;
  const bool leaf = w_arg.requires_grad() && !w_arg.grad_fn() && w_arg.data_ptr() == w.data_ptr();
  std::lock_guard<std::mutex> lk(g_pending_mu);
  auto it = g_conv_seen.find(w.data_ptr());
  if (it == g_conv_seen.end()) g_conv_seen.emplace(w.data_ptr(), ConvSeen{k, !leaf});
  else if (!(it->second.k == k) || !leaf) it->second.mixed = true;
  return leaf;
}
// has an earlier backward of this window already handed autograd a gradient for the leaf kept under `key`?  (then: flush, so that it
// is written, and take the immediate path - the SOUNDNESS RULE above)
inline bool leaf_has_grad(AutogradContext* ctx, const char* key) {
  auto it = ctx->saved_data.find(key);
  if (it == ctx->saved_data.end() || !it->second.isTensor()) return true;      // unknown leaf: never defer
  if (!it->second.toTensor().grad().defined()) return false;
  flush_deferred();
  return true;
}
inline bool conv_dest_exists(const Tensor& w) {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  return g_conv_dest.count(w.data_ptr()) > 0;
}
// backward: park this call; dw / db are set only for the first parked call of the weight.  false: take the immediate path.
inline bool conv_park(const Tensor& x, const Tensor& dy, const Tensor& mask, const Tensor& w, const ConvKey& k, bool has_b, Tensor& dw, Tensor& db) {
  queue_final_flush();
  std::lock_guard<std::mutex> lk(g_pending_mu);
  auto seen = g_conv_seen.find(w.data_ptr());
  if (seen == g_conv_seen.end() || seen->second.mixed) return false;
  auto it = g_conv_dest.find(w.data_ptr());
  if (it == g_conv_dest.end()) {
    dw = at::empty_like(w);
    if (has_b) db = at::empty({k.Co}, w.options());
    g_conv_dest.emplace(w.data_ptr(), ConvDest{dw.data_ptr(), has_b ? db.data_ptr() : nullptr});
  }
  g_pending_conv.push_back(PendingConv{x, dy, mask, w.data_ptr(), k});
  return true;
}
static void flush_convs(std::vector<PendingConv>& cs, const std::map<const void*, ConvDest>& dest) {
  std::vector<bool> done(cs.size(), false);
  for (size_t i = 0; i < cs.size(); ++i) {
    if (done[i]) continue;
    // all calls of this geometry, grouped by weight
    std::vector<const void*> weights;
    std::map<const void*, std::vector<size_t>> by_w;
    for (size_t j = i; j < cs.size(); ++j)
      if (!done[j] && cs[j].k == cs[i].k && cs[j].x.device() == cs[i].x.device()) {
        done[j] = true;
        if (!by_w.count(cs[j].w)) weights.push_back(cs[j].w);
        by_w[cs[j].w].push_back(j);
      }
    // one launch per call count (a launch needs equal shares), at most 32 convolutions each
    std::map<size_t, std::vector<const void*>> by_count;
    for (auto w : weights) by_count[by_w[w].size()].push_back(w);
    const ConvKey& k = cs[i].k;
    for (auto& bc : by_count) {
      const size_t per = bc.first;
      TORCH_CHECK(per <= 32, "dgtd conv3x3: one weight used ", per, " times per step (deferred weight gradient takes at most 32)");
      const size_t wmax = 32 / per;
      for (size_t w0 = 0; w0 < bc.second.size(); w0 += wmax) {
        const size_t nw = std::min(wmax, bc.second.size() - w0);
        std::vector<const void*> xs, dys, ms;
        std::vector<int> slot;
        std::vector<void*> dws, dbs;
        bool any_b = false;
        for (size_t a = 0; a < nw; ++a) {
          const void* w = bc.second[w0 + a];
          const ConvDest& d = dest.at(w);
          dws.push_back(d.dw); dbs.push_back(d.db); any_b = any_b || d.db;
          for (size_t j : by_w[w]) {
            xs.push_back(cs[j].x.data_ptr()); dys.push_back(cs[j].dy.data_ptr());
            ms.push_back(cs[j].mask.defined() ? cs[j].mask.data_ptr() : nullptr);
            slot.push_back((int)a);
          }
        }
        const int n = (int)xs.size();
        Tensor ws = at::empty({dgtd_conv3x3_wgrad_batched_workspace(n, k.B, k.H, k.W, k.Ci, k.Co)}, cs[i].x.options().dtype(at::kByte));
        check(dgtd_conv3x3_wgrad_batched(xs.data(), dys.data(), ms.data(), slot.data(), n, dws.data(), any_b ? dbs.data() : nullptr, (int)nw, ws.data_ptr(),
                                         k.B, k.H, k.W, k.Ci, k.Co, (dgtd_dtype)k.dt, stream()), "dgtd_conv3x3_wgrad_batched");
      }
    }
  }
}

static std::atomic<int64_t> g_flushed{0};
// ASYNCHRONOUS FLUSH: the parked weight-gradient work is a handful of LARGE launches (batched GEMMs over 27 layers, batched depthwise
// and 3x3 weight gradients: ~4.4 ms per step at config 2) that nothing in the backward pass waits for, while the backward pass itself
// is a chain of small latency-bound launches that leave most of the chip idle.  flush_deferred_async() runs everything parked so far
// on a SIDE stream forked from the caller's stream (event edges both ways, so it is captured into a hipGraph as a parallel branch);
// the next synchronous flush - at the latest the end-of-backward callback - joins it.  The caller picks a point where no shared
// deferral is half-way (dgtd.nn: when the gradient of the texture-diffuser embedding arrives, i.e. after the Hitnet decoder, every PVT
// block and the prompt decoders have run backward, with the whole ConvNeXt trunk still to go).
// Tensors the side stream reads are recorded on it (the caching allocator must not hand their memory to main-stream allocations
// while the side stream is still running); the library-GEMM context is per stream (gemm.h).
static c10::optional<c10::hip::HIPStream> g_side_stream;
static hipEvent_t g_fork_event = nullptr, g_join_event = nullptr;     // plain HIP events (record / stream-wait are capturable)
static bool g_async_outstanding = false;
static thread_local const c10::hip::HIPStream* t_record_on = nullptr;
inline void keep_for_side(const Tensor& t) {
  if (t_record_on && t.defined() && t.has_storage() && t.is_cuda()) c10::hip::HIPCachingAllocator::recordStream(t.storage().data_ptr(), *t_record_on);
}
static void join_async() {
  if (!g_async_outstanding) return;
  TORCH_CHECK(hipStreamWaitEvent(c10::hip::getCurrentHIPStream().stream(), g_join_event, 0) == hipSuccess, "dgtd: joining the side stream failed");
  g_async_outstanding = false;
}
static void flush_deferred_impl();
void flush_deferred() {
  join_async();
  flush_deferred_impl();
}
void flush_deferred_async() {
  // measured at config 2 (profiles/r03_async_flush.txt): 31.13 ms/step with the side branch, 30.42 without - the chip-filling weight-gradient
  // launches slow the latency-bound main chain down by more than they hide.  Kept behind DGTD_ASYNC_FLUSH=1 for other shapes.
  static const bool on = [] { const char* e = std::getenv("DGTD_ASYNC_FLUSH"); return e && std::atoi(e) != 0; }();
  if (!on) return;
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    if (g_pending.empty() && g_pending_dw.empty() && g_pending_gemm.empty() && g_pending_conv.empty() && g_ca_acc.empty()) return;
  }
  join_async();                                     // one side branch at a time
  const auto cur = c10::hip::getCurrentHIPStream();
  if (!g_side_stream || g_side_stream->device_index() != cur.device_index()) g_side_stream = c10::hip::getStreamFromPool(false, cur.device_index());
  if (!g_fork_event) {
    TORCH_CHECK(hipEventCreateWithFlags(&g_fork_event, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&g_join_event, hipEventDisableTiming) == hipSuccess, "dgtd: hipEventCreate failed");
  }
  // the side stream sees everything enqueued so far (partials, saved activations)
  TORCH_CHECK(hipEventRecord(g_fork_event, cur.stream()) == hipSuccess && hipStreamWaitEvent(g_side_stream->stream(), g_fork_event, 0) == hipSuccess,
              "dgtd: forking the side stream failed");
  {
    c10::hip::HIPStreamGuard guard(*g_side_stream);
    t_record_on = &*g_side_stream;
    try { flush_deferred_impl(); } catch (...) { t_record_on = nullptr; throw; }
    t_record_on = nullptr;
  }
  TORCH_CHECK(hipEventRecord(g_join_event, g_side_stream->stream()) == hipSuccess, "dgtd: recording the join event failed");
  g_async_outstanding = true;
}
static void flush_deferred_impl() {
  {
    std::vector<PendingConv> convs;
    std::map<const void*, ConvDest> dest;
    {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      convs.swap(g_pending_conv);
      dest.swap(g_conv_dest);
      g_prelu_acc.clear();
      g_conv_flip.clear();
    }
    g_flushed += (int64_t)convs.size();
    for (auto& c : convs) { keep_for_side(c.x); keep_for_side(c.dy); keep_for_side(c.mask); }
    if (!convs.empty()) flush_convs(convs, dest);
  }
  std::vector<PendingReduce> todo;
  std::vector<PendingDw> dws;
  std::vector<PendingGemm> gemms;
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    gemms.swap(g_pending_gemm);
  }
  for (auto& gm : gemms) { keep_for_side(gm.x); keep_for_side(gm.dy); }
  if (!gemms.empty()) flush_gemms(gemms);
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    todo.swap(g_pending);
    dws.swap(g_pending_dw);
  }
  std::map<const void*, CaAcc> cas;
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    cas.swap(g_ca_acc);
  }
  for (auto& kv : cas) {
    CaAcc& a = kv.second;
    keep_for_side(a.buf);
    todo.push_back(PendingReduce{dgtd_reduce_entry{a.buf.data_ptr<float>(), a.count * a.rows, 2 * a.rc, (float*)a.out1, a.rc, a.out2, (int32_t)DGTD_F32, 0, 0, nullptr}, a.buf});
  }
  g_flushed += (int64_t)(gemms.size() + todo.size() + dws.size());
  for (auto& p : todo) keep_for_side(p.ws);
  for (auto& d : dws) { keep_for_side(d.x); keep_for_side(d.du); }
  if (todo.empty() && dws.empty()) return;
  std::vector<dgtd_reduce_entry> es;
  es.reserve(todo.size() + dws.size());
  for (auto& p : todo) es.push_back(p.e);
  std::vector<Tensor> keep;
  std::vector<bool> done(dws.size(), false);
  for (size_t i = 0; i < dws.size(); ++i) {
    if (done[i]) continue;
    const PendingDw& a = dws[i];
    std::vector<size_t> grp;
    for (size_t j = i; j < dws.size(); ++j) {
      const PendingDw& b = dws[j];
      if (!done[j] && b.B == a.B && b.H == a.H && b.W == a.W && b.C == a.C && b.K == a.K && b.has_bias == a.has_bias &&
          b.x.scalar_type() == a.x.scalar_type() && b.x.device() == a.x.device()) { grp.push_back(j); done[j] = true; }
    }
    const int n = (int)grp.size(), KK = a.K * a.K;
    const int blocks = dgtd_dwconv_bwd_weight_batched_blocks(n, a.B, a.H, a.W, a.C, a.K);
    const int64_t per = (int64_t)blocks * (KK + 1) * a.C;
    Tensor ws = at::empty({(int64_t)n * per}, a.x.options().dtype(at::kFloat));
    std::vector<const void*> xs(n), dus(n);
    for (int k = 0; k < n; ++k) { xs[k] = dws[grp[k]].x.data_ptr(); dus[k] = dws[grp[k]].du.data_ptr(); }
    check(dgtd_dwconv_bwd_weight_batched(xs.data(), dus.data(), n, a.has_bias ? 1 : 0, ws.data_ptr(), a.B, a.H, a.W, a.C, a.K, code(a.x), stream()),
          "dgtd_dwconv_bwd_weight_batched");
    for (int k = 0; k < n; ++k) {
      const PendingDw& d = dws[grp[k]];
      es.push_back(dgtd_reduce_entry{ws.data_ptr<float>() + (int64_t)k * per, blocks, (int32_t)((KK + 1) * a.C), nullptr, 0, d.dw, d.dw_dt, (int32_t)KK,
                                     (int32_t)a.C, d.db});
    }
    keep.push_back(ws);
  }
  check(dgtd_multi_reduce(es.data(), (int)es.size(), stream()), "dgtd_multi_reduce");
}
// PREPARED WEIGHTS: the packed fp32 form { w_t | w_t flipped | bias } every depthwise convolution computes from.  Weights change
// only between steps, so under the reducer a layer packs itself once (first step) into a PERSISTENT buffer and registers it; from
// then on set_deferred(true) - the start of every step - re-packs all registered layers in one launch (52 launches -> 1 per step).
// Entries die with the weight's storage (weak reference); outside zero_grad() .. finish() the nodes pack on the fly as before.
struct PreparedDw { c10::weak_intrusive_ptr<c10::StorageImpl> wstore; const void* w; const void* b; Tensor packed; int C, K; at::ScalarType dt; };
static std::map<const void*, PreparedDw> g_prepared_dw;
inline bool prepare_on() {
  static const bool on = [] { const char* e = std::getenv("DGTD_PREPARE_WEIGHTS"); return !e || std::atoi(e) != 0; }();
  return on;
}
static void refresh_prepared() {
  std::map<int, std::vector<PreparedDw*>> by_dt;
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    for (auto it = g_prepared_dw.begin(); it != g_prepared_dw.end();) {
      if (it->second.wstore.expired()) it = g_prepared_dw.erase(it);
      else { by_dt[(int)it->second.dt].push_back(&it->second); ++it; }
    }
  }
  for (auto& kv : by_dt) {
    std::vector<const void*> ws, bs;
    std::vector<float*> outs;
    std::vector<int> Cs, Ks;
    for (auto* p : kv.second) { ws.push_back(p->w); bs.push_back(p->b); outs.push_back(p->packed.data_ptr<float>()); Cs.push_back(p->C); Ks.push_back(p->K); }
    const dgtd_dtype dt = kv.first == (int)at::kFloat ? DGTD_F32 : (kv.first == (int)at::kBFloat16 ? DGTD_BF16 : DGTD_F16);
    check(dgtd_dwconv_pack_batched(ws.data(), bs.data(), outs.data(), Cs.data(), Ks.data(), (int)ws.size(), dt, stream()), "dgtd_dwconv_pack_batched");
  }
}
// the packed weights of (weight, bias) for this step: the registered buffer (already refreshed), or a fresh pack that is registered
inline Tensor packed_dw(const Tensor& weight, const Tensor& bias, bool has_bias, int64_t C, int64_t K) {
  const int64_t KK = K * K;
  const bool use = deferring() && prepare_on() && weight.has_storage();
  if (use) {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    auto it = g_prepared_dw.find(weight.data_ptr());
    if (it != g_prepared_dw.end()) {
      PreparedDw& p = it->second;
      auto st = p.wstore.lock();
      if (st && st.get() == weight.storage().unsafeGetStorageImpl() && p.C == C && p.K == K && p.dt == weight.scalar_type() &&
          p.b == (has_bias ? bias.data_ptr() : nullptr) && p.packed.device() == weight.device())
        return p.packed;
      g_prepared_dw.erase(it);
    }
  }
  Tensor packed = at::empty({(2 * KK + 1) * C}, weight.options().dtype(at::kFloat));   // { w_t | w_t flipped | bias }
  check(dgtd_dwconv_pack(weight.data_ptr(), has_bias ? bias.data_ptr() : nullptr, packed.data_ptr<float>(), (int)C, (int)K, code(weight), stream()),
        "dgtd_dwconv_pack");
  if (use) {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    g_prepared_dw.emplace(weight.data_ptr(),
                          PreparedDw{c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(weight.storage().unsafeGetStorageImpl())),
                                     weight.data_ptr(), has_bias ? bias.data_ptr() : nullptr, packed, (int)C, (int)K, weight.scalar_type()});
  }
  return packed;
}

void set_deferred(bool on) {
  if (on) {   // entries left over from a backward that was never flushed point at dead memory: drop them
    std::lock_guard<std::mutex> lk(g_pending_mu);
    g_pending.clear();
    g_pending_dw.clear();
    g_pending_gemm.clear();
    g_pending_conv.clear();
    g_conv_dest.clear();
    g_conv_seen.clear();
    g_prelu_acc.clear();
    g_conv_flip.clear();
    g_ca_acc.clear();
    g_uses.clear();
    g_slot_used.clear();
  } else {
    flush_deferred();
  }
  g_defer.store(on);
  if (on && prepare_on()) refresh_prepared();
  if (on && own_gemm_on()) refresh_transposed();
}
int64_t flushed_reductions() { return g_flushed.load(); }    // cumulative number of parked entries that flush_deferred() has served
int64_t pending_reductions() {
  std::lock_guard<std::mutex> lk(g_pending_mu);
  return (int64_t)(g_pending.size() + g_pending_dw.size() + g_pending_gemm.size() + g_pending_conv.size() + g_ca_acc.size());
}

// ------------------------------------------------------------------------------------------------ LayerNorm
struct LayerNormFn : public torch::autograd::Function<LayerNormFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& w, const Tensor& b, double eps) {
    Tensor x = x_.contiguous();
    on_device(x);
    const int64_t C = x.size(-1), rows = x.numel() / C;
    Tensor w32 = f32(w), b32 = f32(b);
    const int out_role = take_out_role();
    Tensor y = (out_role >= 0 && hinted(t_hint, x.scalar_type())) ? arena_slot(t_hint, ROLE_X + out_role, x.sizes(), x.options()) : at::empty_like(x);
    Tensor stats = at::empty({2, rows}, x.options().dtype(at::kFloat));
    check(dgtd_layernorm_fwd(x.data_ptr(), w32.data_ptr<float>(), b32.data_ptr<float>(), y.data_ptr(), stats.data_ptr<float>(),
                             stats.data_ptr<float>() + rows, rows, (int)C, (float)eps, code(x), stream()), "dgtd_layernorm_fwd");
    ctx->save_for_backward({x, w32, stats});
    note_leaf(ctx, "leaf_w", w);
    note_leaf(ctx, "leaf_b", b);
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w32 = saved[1], &stats = saved[2];
    const int64_t C = x.size(-1), rows = x.numel() / C;
    Tensor dy = g[0].contiguous();
    if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
    Tensor dx = at::empty_like(x);
    Tensor dgb = at::empty({2, C}, x.options().dtype(at::kFloat));
    Tensor ws = at::empty({dgtd_layernorm_bwd_workspace((int)C) / 4}, x.options().dtype(at::kFloat));
    if (may_defer(ctx, "leaf_w") && may_defer(ctx, "leaf_b")) {
      int nb = 0;
      check(dgtd_layernorm_bwd_partial(dy.data_ptr(), x.data_ptr(), w32.data_ptr<float>(), stats.data_ptr<float>(), stats.data_ptr<float>() + rows,
                                       nullptr, dx.data_ptr(), ws.data_ptr(), rows, (int)C, code(x), &nb, stream()), "dgtd_layernorm_bwd_partial");
      park(ws, nb, 2 * (int)C, dgb.data_ptr<float>(), 2 * (int)C, nullptr, DGTD_F32);
    } else {
      check(dgtd_layernorm_bwd(dy.data_ptr(), x.data_ptr(), w32.data_ptr<float>(), stats.data_ptr<float>(), stats.data_ptr<float>() + rows,
                               dx.data_ptr(), dgb.data_ptr<float>(), dgb.data_ptr<float>() + C, ws.data_ptr(), rows, (int)C, code(x), stream()),
            "dgtd_layernorm_bwd");
    }
    return {dx, dgb[0], dgb[1], undefined()};
  }
};

// Pre-norm fork: returns (LayerNorm(x), x).  The second output is x itself, to be used for the skip connection: autograd then hands
// BOTH gradients to this node and the backward adds them inside the LayerNorm backward kernel (dgtd_layernorm_bwd_add) instead
// of a separate elementwise add per block.
struct LayerNormForkFn : public torch::autograd::Function<LayerNormForkFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x_, const Tensor& w, const Tensor& b, double eps) {
    Tensor x = x_.contiguous();
    on_device(x);
    const int64_t C = x.size(-1), rows = x.numel() / C;
    Tensor w32 = f32(w), b32 = f32(b);
    const int out_role = take_out_role();
    Tensor y = (out_role >= 0 && hinted(t_hint, x.scalar_type())) ? arena_slot(t_hint, ROLE_X + out_role, x.sizes(), x.options()) : at::empty_like(x);
    Tensor stats = at::empty({2, rows}, x.options().dtype(at::kFloat));
    check(dgtd_layernorm_fwd(x.data_ptr(), w32.data_ptr<float>(), b32.data_ptr<float>(), y.data_ptr(), stats.data_ptr<float>(),
                             stats.data_ptr<float>() + rows, rows, (int)C, (float)eps, code(x), stream()), "dgtd_layernorm_fwd");
    ctx->save_for_backward({x, w32, stats});
    note_leaf(ctx, "leaf_w", w);
    note_leaf(ctx, "leaf_b", b);
    return {y, x_};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w32 = saved[1], &stats = saved[2];
    const int64_t C = x.size(-1), rows = x.numel() / C;
    Tensor dres;
    if (g[1].defined()) dres = (g[1].scalar_type() == x.scalar_type() ? g[1] : g[1].to(x.scalar_type())).contiguous();
    if (!g[0].defined()) return {dres, undefined(), undefined(), undefined()};
    Tensor dy = g[0].contiguous();
    if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
    Tensor dx = at::empty_like(x);
    Tensor dgb = at::empty({2, C}, x.options().dtype(at::kFloat));
    Tensor ws = at::empty({dgtd_layernorm_bwd_workspace((int)C) / 4}, x.options().dtype(at::kFloat));
    if (may_defer(ctx, "leaf_w") && may_defer(ctx, "leaf_b")) {
      int nb = 0;
      check(dgtd_layernorm_bwd_partial(dy.data_ptr(), x.data_ptr(), w32.data_ptr<float>(), stats.data_ptr<float>(), stats.data_ptr<float>() + rows,
                                       dres.defined() ? dres.data_ptr() : nullptr, dx.data_ptr(), ws.data_ptr(), rows, (int)C, code(x), &nb, stream()),
            "dgtd_layernorm_bwd_partial");
      park(ws, nb, 2 * (int)C, dgb.data_ptr<float>(), 2 * (int)C, nullptr, DGTD_F32);
    } else {
      check(dgtd_layernorm_bwd_add(dy.data_ptr(), x.data_ptr(), w32.data_ptr<float>(), stats.data_ptr<float>(), stats.data_ptr<float>() + rows,
                                   dres.defined() ? dres.data_ptr() : nullptr, dx.data_ptr(), dgb.data_ptr<float>(), dgb.data_ptr<float>() + C,
                                   ws.data_ptr(), rows, (int)C, code(x), stream()), "dgtd_layernorm_bwd_add");
    }
    return {dx, dgb[0], dgb[1], undefined()};
  }
};

// ------------------------------------------------------------------------------------------------ SRA attention
struct SraAttnFn : public torch::autograd::Function<SraAttnFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& q_, const Tensor& kv_, int64_t heads, double scale) {
    Tensor q = q_.contiguous(), kv = kv_.contiguous();
    on_device(q); on_device(kv);
    const int64_t B = q.size(0), N = q.size(1), C = q.size(2), Nkv = kv.size(1);
    TORCH_CHECK(C == heads * 64 && kv.size(2) == 2 * C && kv.scalar_type() == q.scalar_type(), "sra_attention: bad shapes");
    const Hint hint = t_hint;
    const int out_role = take_out_role();
    const auto groles = take_grad_roles();
    const bool ar = hinted(hint, q.scalar_type());
    Tensor out = (ar && out_role >= 0) ? arena_slot(hint, ROLE_X + out_role, q.sizes(), q.options()) : at::empty_like(q);
    ctx->saved_data["hint"] = hint_iv(ar ? hint : Hint{}, groles.first, groles.second);
    Tensor lse = at::empty({B, heads, N}, q.options().dtype(at::kFloat));
    check(dgtd_sra_attn_fwd(q.data_ptr(), kv.data_ptr(), out.data_ptr(), lse.data_ptr<float>(), (int)B, (int)N, (int)Nkv, (int)heads,
                            (float)scale, code(q), stream()), "dgtd_sra_attn_fwd");
    ctx->save_for_backward({q, kv, out, lse});
    ctx->saved_data["heads"] = heads;
    ctx->saved_data["scale"] = scale;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &q = saved[0], &kv = saved[1], &out = saved[2], &lse = saved[3];
    const int64_t heads = ctx->saved_data["heads"].toInt();
    const double scale = ctx->saved_data["scale"].toDouble();
    const int64_t B = q.size(0), N = q.size(1), Nkv = kv.size(1);
    Tensor dout = g[0].contiguous();
    if (dout.scalar_type() != q.scalar_type()) dout = dout.to(q.scalar_type());
    const Hint hint = hint_of(ctx->saved_data["hint"]);
    const int ra = role_of(ctx->saved_data["hint"], 0), rb = role_of(ctx->saved_data["hint"], 1);
    const bool ar = hint.group >= 0 && deferring();
    Tensor dq = (ar && ra >= 0) ? arena_slot(hint, ROLE_DY + ra, q.sizes(), q.options()) : at::empty_like(q);
    // 16-bit paths overwrite every element (plain stores or the slab reduce); the fp32 kernel accumulates with atomics into zeros
    Tensor dkv = is16(q.scalar_type()) ? at::empty(kv.sizes(), kv.options().dtype(at::kFloat)) : at::zeros(kv.sizes(), kv.options().dtype(at::kFloat));
    Tensor ws = at::empty({dgtd_sra_attn_bwd_workspace((int)B, (int)N, (int)heads)}, q.options().dtype(at::kByte));
    check(dgtd_sra_attn_bwd(q.data_ptr(), kv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr<float>(), dq.data_ptr(),
                            dkv.data_ptr<float>(), ws.data_ptr(), (int)B, (int)N, (int)Nkv, (int)heads, (float)scale, code(q), stream()),
          "dgtd_sra_attn_bwd");
    if (ar && rb >= 0 && kv.scalar_type() != at::kFloat) {
      Tensor dkv16 = arena_slot(hint, ROLE_DY + rb, kv.sizes(), kv.options());
      dkv16.copy_(dkv);
      return {dq, dkv16, undefined(), undefined()};
    }
    return {dq, dkv.to(kv.scalar_type()), undefined(), undefined()};
  }
};

// ------------------------------------------------------------------------------------------------ depthwise conv (NHWC)
// weight / bias gradient of a depthwise convolution: first stage into per-workgroup partial rows { dw_t | db }, then ONE multi-reduce
// entry that also transposes into the Conv2d layout and converts to the parameters' dtype (was: reduce + unpack, two launches) - parked
// with the other column reductions of the backward pass when the reducer has switched deferral on.
inline void dwconv_weight_grads(const Tensor& x, const Tensor& du, bool has_bias, int64_t C, int64_t K, Tensor& dw, Tensor& db, bool defer) {
  const int64_t KK = K * K;
  static const bool batch_dw = [] { const char* e = std::getenv("DGTD_DEFER_DWCONV"); return !e || std::atoi(e) != 0; }();
  if (defer && batch_dw && C % 128 == 0) { park_dw(x, du, dw, db, has_bias, (int)C, (int)K); return; }   // the batched kernel walks 128-channel groups
  Tensor ws = at::empty({dgtd_dwconv_bwd_weight_workspace((int)x.size(0), (int)x.size(1), (int)x.size(2), (int)C, (int)K) / 4}, x.options().dtype(at::kFloat));
  int nb = 0;
  check(dgtd_dwconv_bwd_weight_partial(x.data_ptr(), du.data_ptr(), has_bias ? 1 : 0, ws.data_ptr(), (int)x.size(0), (int)x.size(1), (int)x.size(2),
                                       (int)C, (int)K, code(x), &nb, stream()), "dgtd_dwconv_bwd_weight_partial");
  if (defer) {
    park(ws, nb, (int)((KK + 1) * C), nullptr, 0, dw.data_ptr(), code(dw), (int)KK, (int)C, has_bias ? db.data_ptr() : nullptr);
  } else {
    const dgtd_reduce_entry e{ws.data_ptr<float>(), nb, (int32_t)((KK + 1) * C), nullptr, 0, dw.data_ptr(), (int32_t)code(dw), (int32_t)KK, (int32_t)C,
                              has_bias ? db.data_ptr() : nullptr};
    check(dgtd_multi_reduce(&e, 1, stream()), "dgtd_multi_reduce");
  }
}

struct DwConvFn : public torch::autograd::Function<DwConvFn> {
  static Tensor launch(const Tensor& x, const float* wt, const float* bias, const Tensor* aux, int mode, int K, Tensor y = Tensor()) {
    if (!y.defined()) y = at::empty_like(x);
    check(dgtd_dwconv_fwd(x.data_ptr(), wt, bias, aux ? aux->data_ptr() : nullptr, y.data_ptr(), (int)x.size(0), (int)x.size(1),
                          (int)x.size(2), (int)x.size(3), K, mode, code(x), stream()), "dgtd_dwconv_fwd");
    return y;
  }
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& weight_, const c10::optional<Tensor>& bias_, bool gelu) {
    Tensor x = x_.contiguous(), weight = weight_.contiguous();
    on_device(x); on_device(weight);
    const int64_t C = weight.size(0), K = weight.size(3), KK = K * K;
    const bool has_bias = bias_.has_value() && bias_->defined();
    TORCH_CHECK(x.size(3) == C && weight.size(1) == 1, "dwconv_nhwc: bad shapes");
    Tensor bias = has_bias ? bias_->contiguous() : Tensor();
    if (has_bias) TORCH_CHECK(bias.scalar_type() == weight.scalar_type(), "dwconv_nhwc: weight/bias dtype mismatch");
    Tensor packed = packed_dw(weight, bias, has_bias, C, K);                        // { w_t | w_t flipped | bias }
    const float* base = packed.data_ptr<float>();
    const Hint hint = t_hint;
    const int out_role = take_out_role();
    const auto groles = take_grad_roles();
    const bool ar = hinted(hint, x.scalar_type());
    Tensor y = launch(x, base, has_bias ? base + 2 * KK * C : nullptr, nullptr, gelu ? 1 : 0, (int)K,
                      (ar && out_role >= 0) ? arena_slot(hint, ROLE_X + out_role, x.sizes(), x.options()) : Tensor());
    ctx->saved_data["hint"] = hint_iv(ar ? hint : Hint{}, groles.first);
    note_leaf(ctx, "leaf_w", weight_);
    if (has_bias) note_leaf(ctx, "leaf_b", *bias_);
    ctx->save_for_backward({x, packed});
    ctx->saved_data["gelu"] = gelu;
    ctx->saved_data["K"] = K;
    ctx->saved_data["has_bias"] = has_bias;
    ctx->saved_data["w_dt"] = st_id(weight);
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &packed = saved[1];
    const bool gelu = ctx->saved_data["gelu"].toBool(), has_bias = ctx->saved_data["has_bias"].toBool();
    const int64_t K = ctx->saved_data["K"].toInt(), KK = K * K, C = x.size(3);
    const auto wdtype = st_of(ctx->saved_data["w_dt"]);
    Tensor dy = g[0].contiguous();
    if (dy.scalar_type() != x.scalar_type()) dy = dy.to(x.scalar_type());
    const float* base = packed.data_ptr<float>();
    const float* bias = has_bias ? base + 2 * KK * C : nullptr;
    Tensor du = gelu ? launch(x, base, bias, &dy, 2, (int)K) : dy;          // through the GELU: recompute the pre-activation
    const Hint hint = hint_of(ctx->saved_data["hint"]);
    const int ra = role_of(ctx->saved_data["hint"], 0);
    Tensor dx = launch(du, base + KK * C, nullptr, nullptr, 0, (int)K,      // bwd-data = same kernel, flipped filter
                       (hint.group >= 0 && ra >= 0 && deferring()) ? arena_slot(hint, ROLE_DY + ra, x.sizes(), x.options()) : Tensor());
    Tensor dw = at::empty({C, 1, K, K}, x.options().dtype(wdtype));
    Tensor db = has_bias ? at::empty({C}, x.options().dtype(wdtype)) : Tensor();
    dwconv_weight_grads(x, du, has_bias, C, K, dw, db, may_defer(ctx, "leaf_w") && (!has_bias || may_defer(ctx, "leaf_b")));
    return {dx, dw, db, undefined()};
  }
};

// Residual fork: returns (dwconv(x), x); x is used for the skip connection, both gradients arrive here and the skip gradient is
// added inside the input-gradient convolution (dgtd_dwconv_fwd mode 3) instead of a separate elementwise add per block.
struct DwConvForkFn : public torch::autograd::Function<DwConvForkFn> {
  static variable_list forward(AutogradContext* ctx, const Tensor& x_, const Tensor& weight_, const c10::optional<Tensor>& bias_) {
    Tensor x = x_.contiguous(), weight = weight_.contiguous();
    on_device(x); on_device(weight);
    const int64_t C = weight.size(0), K = weight.size(3), KK = K * K;
    const bool has_bias = bias_.has_value() && bias_->defined();
    TORCH_CHECK(x.size(3) == C && weight.size(1) == 1, "dwconv_nhwc: bad shapes");
    Tensor bias = has_bias ? bias_->contiguous() : Tensor();
    Tensor packed = packed_dw(weight, bias, has_bias, C, K);
    const float* base = packed.data_ptr<float>();
    Tensor y = DwConvFn::launch(x, base, has_bias ? base + 2 * KK * C : nullptr, nullptr, 0, (int)K);
    note_leaf(ctx, "leaf_w", weight_);
    if (has_bias) note_leaf(ctx, "leaf_b", *bias_);
    ctx->save_for_backward({x, packed});
    ctx->saved_data["K"] = K;
    ctx->saved_data["has_bias"] = has_bias;
    ctx->saved_data["w_dt"] = st_id(weight);
    return {y, x_};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &packed = saved[1];
    const bool has_bias = ctx->saved_data["has_bias"].toBool();
    const int64_t K = ctx->saved_data["K"].toInt(), KK = K * K, C = x.size(3);
    const auto wdtype = st_of(ctx->saved_data["w_dt"]);
    Tensor skip;
    if (g[1].defined()) skip = (g[1].scalar_type() == x.scalar_type() ? g[1] : g[1].to(x.scalar_type())).contiguous();
    if (!g[0].defined()) return {skip, undefined(), undefined()};
    Tensor du = g[0].contiguous();
    if (du.scalar_type() != x.scalar_type()) du = du.to(x.scalar_type());
    const float* base = packed.data_ptr<float>();
    Tensor dx = skip.defined() ? DwConvFn::launch(du, base + KK * C, nullptr, &skip, 3, (int)K) : DwConvFn::launch(du, base + KK * C, nullptr, nullptr, 0, (int)K);
    Tensor dw = at::empty({C, 1, K, K}, x.options().dtype(wdtype));
    Tensor db = has_bias ? at::empty({C}, x.options().dtype(wdtype)) : Tensor();
    dwconv_weight_grads(x, du, has_bias, C, K, dw, db, may_defer(ctx, "leaf_w") && (!has_bias || may_defer(ctx, "leaf_b")));
    return {dx, dw, db};
  }
};

// ------------------------------------------------------------------------------------------------ x + s[b]*gamma[c]*y
struct ScaleResidualFn : public torch::autograd::Function<ScaleResidualFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& y_, const c10::optional<Tensor>& s_, const c10::optional<Tensor>& gamma_) {
    Tensor x = x_.contiguous(), y = y_.scalar_type() == x_.scalar_type() ? y_.contiguous() : y_.to(x_.scalar_type()).contiguous();
    on_device(x);
    const bool has_s = s_.has_value() && s_->defined(), has_g = gamma_.has_value() && gamma_->defined();
    const int64_t B = x.size(0), C = x.size(-1), rows = x.numel() / C;
    Tensor s = has_s ? f32(*s_) : Tensor(), g32 = has_g ? f32(*gamma_) : Tensor();
    Tensor out = at::empty_like(x);
    check(dgtd_scale_residual_fwd(x.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr, has_g ? g32.data_ptr<float>() : nullptr,
                                  out.data_ptr(), rows, (int)C, rows / B, code(x), stream()), "dgtd_scale_residual_fwd");
    ctx->save_for_backward({y, s, g32});
    ctx->saved_data["has_s"] = has_s;
    ctx->saved_data["has_g"] = has_g;
    ctx->saved_data["g_dt"] = has_g ? st_id(*gamma_) : (int64_t)at::kFloat;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &y = saved[0], &s = saved[1], &g32 = saved[2];
    const bool has_s = ctx->saved_data["has_s"].toBool(), has_g = ctx->saved_data["has_g"].toBool();
    Tensor g = gr[0].contiguous();
    if (g.scalar_type() != y.scalar_type()) g = g.to(y.scalar_type());
    if (!has_s && !has_g) return {g, g, undefined(), undefined()};
    const int64_t B = y.size(0), C = y.size(-1), rows = y.numel() / C;
    Tensor dy = at::empty_like(y);
    Tensor dgamma = has_g ? at::empty({C}, y.options().dtype(at::kFloat)) : Tensor();
    Tensor ws = has_g ? at::empty({dgtd_colsum_workspace((int)C)}, y.options().dtype(at::kByte)) : Tensor();
    check(dgtd_scale_residual_bwd(g.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr, has_g ? g32.data_ptr<float>() : nullptr,
                                  dy.data_ptr(), has_g ? dgamma.data_ptr<float>() : nullptr, has_g ? ws.data_ptr() : nullptr, rows, (int)C,
                                  rows / B, code(y), stream()), "dgtd_scale_residual_bwd");
    if (has_g && st_of(ctx->saved_data["g_dt"]) != at::kFloat) dgamma = dgamma.to(st_of(ctx->saved_data["g_dt"]));
    return {g, dy, undefined(), dgamma};
  }
};

// ------------------------------------------------------------------------------------------------ Linear (library GEMMs + colsum bias grad)
Tensor colsum(const Tensor& x2, at::ScalarType out_dt, bool defer) {
  const int64_t rows = x2.size(0), C = x2.size(1);
  if (C % (16 / (int64_t)x2.element_size())) return at::sum(x2, {0}, false, at::kFloat).to(out_dt);   // narrower than a 16-byte chunk per lane
  Tensor out = at::empty({C}, x2.options().dtype(out_dt));
  Tensor ws = at::empty({dgtd_colsum_workspace((int)C) / 4}, x2.options().dtype(at::kFloat));
  if (defer) {
    int nb = 0;
    check(dgtd_colsum_partial(x2.data_ptr(), ws.data_ptr(), rows, (int)C, code(x2), &nb, stream()), "dgtd_colsum_partial");
    const bool f = out_dt == at::kFloat;
    park(ws, nb, (int)C, f ? out.data_ptr<float>() : nullptr, f ? (int)C : 0, f ? nullptr : out.data_ptr(), code(out));
  } else {
    check(dgtd_colsum(x2.data_ptr(), out.data_ptr(), code(out), ws.data_ptr(), rows, (int)C, code(x2), stream()), "dgtd_colsum");
  }
  return out;
}

// ------------------------------------------------------------------------------------------------ the package's own MFMA GEMM (csrc/gemm.hip)
// DGTD_OWN_GEMM=0 keeps every Linear on the library GEMM (A/B switch).  Shapes outside the kernel (M % 128, N % 64, K % 64) always do.
inline bool own_gemm_on() {
  static const bool on = [] { const char* e = std::getenv("DGTD_OWN_GEMM"); return !e || std::atoi(e) != 0; }();
  return on;
}
inline bool aligned16(const Tensor& t) { return ((uintptr_t)t.data_ptr() & 15) == 0; }
inline bool own_ok(at::ScalarType dt, int64_t M, int64_t N, int64_t K) {
  return own_gemm_on() && is16(dt) && M < (1ll << 31) && N < (1ll << 31) && K < (1ll << 31) &&
         dgtd_gemm_supported((int)M, (int)N, (int)K, dt == at::kHalf ? DGTD_F16 : DGTD_BF16);
}
// TRANSPOSED WEIGHT COPIES: the input-gradient GEMM dX = dY W runs through the same K-contiguous kernel on W^T.  Weights change only
// between steps, so under the reducer a weight transposes itself once (first backward) into a PERSISTENT buffer and registers it;
// from then on set_deferred(true) - the start of every step - refreshes all registered copies in one launch per 64 weights.  Entries
// die with the weight's storage.  Outside zero_grad() .. finish() the copy is made on the fly.
struct PreparedWt { c10::weak_intrusive_ptr<c10::StorageImpl> wstore; const void* w; Tensor wt; int rows, cols; at::ScalarType dt; };
static std::map<const void*, PreparedWt> g_prepared_wt;
static void refresh_transposed() {
  std::map<int, std::vector<PreparedWt*>> by_dt;
  {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    for (auto it = g_prepared_wt.begin(); it != g_prepared_wt.end();) {
      if (it->second.wstore.expired()) it = g_prepared_wt.erase(it);
      else { by_dt[(int)it->second.dt].push_back(&it->second); ++it; }
    }
  }
  for (auto& kv : by_dt) {
    std::vector<const void*> src;
    std::vector<void*> dst;
    std::vector<int> rows, cols;
    for (auto* p : kv.second) { src.push_back(p->w); dst.push_back(p->wt.data_ptr()); rows.push_back(p->rows); cols.push_back(p->cols); }
    check(dgtd_transpose_batched(src.data(), dst.data(), rows.data(), cols.data(), (int)src.size(), kv.first == (int)at::kHalf ? DGTD_F16 : DGTD_BF16,
                                 stream()), "dgtd_transpose_batched");
  }
}
inline Tensor transposed_weight(const Tensor& wc) {                  // [N,K] contiguous 16-bit -> [K,N]
  const bool use = deferring() && wc.has_storage();
  if (use) {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    auto it = g_prepared_wt.find(wc.data_ptr());
    if (it != g_prepared_wt.end()) {
      PreparedWt& p = it->second;
      auto st = p.wstore.lock();
      if (st && st.get() == wc.storage().unsafeGetStorageImpl() && p.rows == wc.size(0) && p.cols == wc.size(1) && p.dt == wc.scalar_type() &&
          p.wt.device() == wc.device())
        return p.wt;
      g_prepared_wt.erase(it);
    }
  }
  Tensor wt = at::empty({wc.size(1), wc.size(0)}, wc.options());
  const void* src = wc.data_ptr();
  void* dst = wt.data_ptr();
  const int rows = (int)wc.size(0), cols = (int)wc.size(1);
  check(dgtd_transpose_batched(&src, &dst, &rows, &cols, 1, code(wc), stream()), "dgtd_transpose_batched");
  if (use) {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    g_prepared_wt.emplace(wc.data_ptr(),
                          PreparedWt{c10::weak_intrusive_ptr<c10::StorageImpl>(c10::intrusive_ptr<c10::StorageImpl>::reclaim_copy(wc.storage().unsafeGetStorageImpl())),
                                     wc.data_ptr(), wt, rows, cols, wc.scalar_type()});
  }
  return wt;
}

// GEMM helpers shared by the Linear nodes: bf16 / fp16 go to hipBLASLt with cached plans (gemm.h), anything else to ATen.
inline Tensor gemm_fwd(const Tensor& x2, const Tensor& wc, const Tensor& bc, Tensor o2 = Tensor()) {   // [M,K] x [N,K]^T (+ bias[N]) -> [M,N]
  const int64_t M = x2.size(0), K = x2.size(1), N = wc.size(0);
  if (!o2.defined()) o2 = at::empty({M, N}, x2.options());
  if (own_ok(x2.scalar_type(), M, N, K) && x2.is_contiguous() && wc.is_contiguous() && o2.is_contiguous() && aligned16(x2) && aligned16(wc) &&
      aligned16(o2) && (!bc.defined() || aligned16(bc))) {
    check(dgtd_gemm_bias(x2.data_ptr(), wc.data_ptr(), bc.defined() ? bc.data_ptr() : nullptr, o2.data_ptr(), (int)M, (int)N, (int)K, code(x2), stream()),
          "dgtd_gemm_bias");
    return o2;
  }
  const bool direct = is16(x2.scalar_type()) && M > 0 && x2.is_contiguous() && wc.is_contiguous() &&
                      dgemm::matmul_16(x2.scalar_type(), x2.data_ptr(), wc.data_ptr(), o2.data_ptr(), bc.defined() ? bc.data_ptr() : nullptr, M, N, K, false,
                                         true, 1, 0, 0, 0, x2.options(), (hipStream_t)stream());
  if (!direct) {
    if (bc.defined()) at::addmm_out(o2, bc, x2, wc.t());
    else at::mm_out(o2, x2, wc.t());
  }
  return o2;
}
inline Tensor gemm_dx(const Tensor& dy2, const Tensor& wc) {                              // [M,N] x [N,K] -> [M,K]
  const int64_t M = dy2.size(0), N = wc.size(0), K = wc.size(1);
  Tensor dx2 = at::empty({M, K}, dy2.options());
  if (own_ok(dy2.scalar_type(), M, K, N) && dy2.is_contiguous() && wc.is_contiguous() && aligned16(dy2) && aligned16(wc)) {
    Tensor wt = transposed_weight(wc);                      // [K,N]: contiguous in the reduction dim
    check(dgtd_gemm_bias(dy2.data_ptr(), wt.data_ptr(), nullptr, dx2.data_ptr(), (int)M, (int)K, (int)N, code(dy2), stream()), "dgtd_gemm_bias (input gradient)");
    return dx2;
  }
  const bool bf = is16(dy2.scalar_type()) && wc.is_contiguous() && M > 0;
  if (!(bf && dgemm::matmul_16(dy2.scalar_type(), dy2.data_ptr(), wc.data_ptr(), dx2.data_ptr(), nullptr, M, K, N, false, false, 1, 0, 0, 0, dy2.options(),
                                 (hipStream_t)stream())))
    at::mm_out(dx2, dy2, wc);
  return dx2;
}
// dW = dY^T X; long token dimensions are split into S batches (library batched GEMM) and summed in fp32: the plain GEMM has only a
// few hundred output tiles, each reducing over all tokens (latency-bound, tools/bench_gemm.py)
Tensor gemm_dw(const Tensor& dy2, const Tensor& x2) {                              // [M,N]^T x [M,K] -> [N,K]
  const int64_t M = dy2.size(0), N = dy2.size(1), K = x2.size(1);
  const bool bf = is16(dy2.scalar_type()) && x2.is_contiguous() && M > 0;
  hipStream_t st = (hipStream_t)stream();
  static const int64_t max_split = [] { const char* e = std::getenv("DGTD_WGRAD_SPLIT"); return e ? (int64_t)std::atol(e) : (int64_t)32; }();
  const int64_t S = std::min<int64_t>(max_split, M / 1024);
  if (S >= 4 && M % S == 0 && is16(dy2.scalar_type())) {
    Tensor part = at::empty({S, N, K}, dy2.options());
    if (!(bf && dgemm::matmul_16(dy2.scalar_type(), dy2.data_ptr(), x2.data_ptr(), part.data_ptr(), nullptr, N, K, M / S, true, false, (int)S, (M / S) * N,
                                   (M / S) * K, N * K, dy2.options(), st)))
      at::bmm_out(part, dy2.view({S, M / S, -1}).transpose(1, 2), x2.view({S, M / S, -1}));
    return at::sum(part, {0});   // bf16 in, fp32 accumulation inside the reduction, bf16 out: one launch
  }
  Tensor dw = at::empty({N, K}, dy2.options());
  if (!(bf && dgemm::matmul_16(dy2.scalar_type(), dy2.data_ptr(), x2.data_ptr(), dw.data_ptr(), nullptr, N, K, M, true, false, 1, 0, 0, 0, dy2.options(), st)))
    at::mm_out(dw, dy2.t(), x2);
  return dw;
}
bool dgemm_fwd::batched_dw(at::ScalarType dt, const void* dy, const void* x, void* dw, int64_t N, int64_t K, int64_t M, int batch, int64_t s_dy, int64_t s_x,
                           int64_t s_dw, const at::TensorOptions& o) {
  return dgemm::matmul_16(dt, dy, x, dw, nullptr, N, K, M, true, false, batch, batch > 1 ? s_dy : 0, batch > 1 ? s_x : 0, batch > 1 ? s_dw : 0, o,
                          (hipStream_t)stream());
}
inline Tensor as_rows(const Tensor& t, at::ScalarType dt) {          // [..., C] -> contiguous [rows, C] in the compute dtype
  Tensor t2 = t.reshape({-1, t.size(-1)});
  if (t2.scalar_type() != dt) t2 = t2.to(dt);
  return t2.contiguous();
}
inline std::vector<int64_t> with_last(const Tensor& x, int64_t n) {
  std::vector<int64_t> sh(x.sizes().begin(), x.sizes().end());
  sh.back() = n;
  return sh;
}

struct LinearFn : public torch::autograd::Function<LinearFn> {
  // compute dtype `dt` is decided by the caller (autocast policy lives in Python)
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b_, int64_t dt_code) {
    const auto dt = from_code(dt_code);
    const bool has_b = b_.has_value() && b_->defined();
    Tensor x2 = as_rows(x, dt);
    Tensor wc = w.scalar_type() == dt ? w : w.to(dt);
    Tensor bc;
    if (has_b) bc = b_->scalar_type() == dt ? b_->contiguous() : b_->to(dt);
    // the output is allocated in its final shape and the GEMM writes into a 2-D alias of it: what autograd sees is a base tensor
    // (a view created inside a Function may not be modified in place, e.g. by nn.ReLU(inplace=True))
    Tensor out = at::empty(with_last(x, w.size(0)), x2.options());
    gemm_fwd(x2, wc, bc, out.view({-1, w.size(0)}));
    ctx->save_for_backward({x2, wc});
    // a Linear of a run of identical blocks whose input was written into an arena (PVT q / kv / fc1): its weight gradient is deferred
    // when the output gradient arrives in an arena slot as well (attention / depthwise backward outputs)
    const Hint hint = t_hint;
    const int xr = (hinted(hint, dt) && w.scalar_type() == dt) ? find_role(hint, ROLE_X, x2) : -1;
    ctx->saved_data["hint"] = hint_iv(xr >= 0 ? hint : Hint{}, xr);
    note_leaf(ctx, "leaf_w", w);
    if (has_b) note_leaf(ctx, "leaf_b", *b_);
    ctx->saved_data["xshape"] = x.sizes().vec();
    ctx->saved_data["w_dt"] = st_id(w);
    ctx->saved_data["b_dt"] = has_b ? st_id(*b_) : (int64_t)-1;
    ctx->saved_data["need_dx"] = x.requires_grad();
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x2 = saved[0], &wc = saved[1];
    Tensor dy2 = as_rows(g[0], x2.scalar_type());
    Tensor dx;
    if (ctx->saved_data["need_dx"].toBool()) dx = gemm_dx(dy2, wc).view(ctx->saved_data["xshape"].toIntVector());
    const Hint hint = hint_of(ctx->saved_data["hint"]);
    const int yr = (hint.group >= 0 && deferring()) ? find_role(hint, ROLE_DY, dy2) : -1;
    Tensor dw;
    if (yr >= 0 && may_defer(ctx, "leaf_w")) {
      dw = arena_slot(hint, ROLE_DW + yr, wc.sizes(), wc.options());
      park_gemm(hint, yr, x2, dy2, dw);
    } else {
      dw = gemm_dw(dy2, x2);
      const auto w_dt = st_of(ctx->saved_data["w_dt"]);
      if (dw.scalar_type() != w_dt) dw = dw.to(w_dt);
    }
    Tensor db;
    const int64_t bk = ctx->saved_data["b_dt"].toInt();
    if (bk >= 0) db = colsum(dy2, (at::ScalarType)bk, may_defer(ctx, "leaf_b"));
    return {dx, dw, db, undefined()};
  }
};

// h = gelu(x W^T + b) as ONE node (convnext_Block pwconv1 + act, cod.py:1097-1098): the backward computes dpre = dh * gelu'(pre)
// and the bias gradient sum(dpre) in one pass (dgtd_gelu_bias_bwd) instead of GELU-backward + column-sum + reduce.
struct LinearGeluFn : public torch::autograd::Function<LinearGeluFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x, const Tensor& w, const Tensor& b, int64_t dt_code) {
    const auto dt = from_code(dt_code);
    Tensor x2 = as_rows(x, dt);
    Tensor wc = w.scalar_type() == dt ? w : w.to(dt);
    Tensor bc = b.scalar_type() == dt ? b.contiguous() : b.to(dt);
    const int64_t Mg = x2.size(0), Kg = x2.size(1), Ng = wc.size(0);
    const bool fused = own_ok(dt, Mg, Ng, Kg) && x2.is_contiguous() && wc.is_contiguous() && aligned16(x2) && aligned16(wc) && aligned16(bc);
    Tensor pre = fused ? at::empty({Mg, Ng}, x2.options()) : gemm_fwd(x2, wc, bc);
    // under an arena hint (a block of a run of identical blocks): the GELU output = pwconv2's input goes to slot idx of the stage arena,
    // and the backward defers dW when this node's own input already sits in its arena (written there by the hinted LayerNorm)
    const Hint hint = t_hint;
    const int out_role = take_out_role();
    const bool ar = hinted(hint, dt) && w.scalar_type() == dt;
    Tensor h = (ar && out_role >= 0) ? arena_slot(hint, ROLE_X + out_role, with_last(x, w.size(0)), pre.options())
                                     : at::empty(with_last(x, w.size(0)), pre.options());
    Tensor h2 = h.view({-1, w.size(0)});
    if (fused)       // ONE launch: GEMM, bias, pre-activation and GELU(pre) written from the same accumulators (csrc/gemm.hip)
      check(dgtd_gemm_bias_gelu(x2.data_ptr(), wc.data_ptr(), bc.data_ptr(), pre.data_ptr(), h2.data_ptr(), (int)Mg, (int)Ng, (int)Kg, code(x2), stream()),
            "dgtd_gemm_bias_gelu");
    else
      at::gelu_out(h2, pre);
    const int xr = ar ? find_role(hint, ROLE_X, x2) : -1;
    ctx->saved_data["hint"] = hint_iv(xr >= 0 ? hint : Hint{}, xr);
    note_leaf(ctx, "leaf_w", w);
    note_leaf(ctx, "leaf_b", b);
    ctx->save_for_backward({x2, wc, pre});
    ctx->saved_data["xshape"] = x.sizes().vec();
    ctx->saved_data["w_dt"] = st_id(w);
    ctx->saved_data["b_dt"] = st_id(b);
    ctx->saved_data["need_dx"] = x.requires_grad();
    return h;
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x2 = saved[0], &wc = saved[1], &pre = saved[2];
    Tensor dh = as_rows(g[0], pre.scalar_type());
    const int64_t rows = pre.size(0), C = pre.size(1);
    const Hint hint = hint_of(ctx->saved_data["hint"]);
    const int xr = role_of(ctx->saved_data["hint"], 0);
    const bool ar = hint.group >= 0 && deferring();
    Tensor dpre = ar ? arena_slot(hint, ROLE_DY + xr, pre.sizes(), pre.options()) : at::empty_like(pre);
    Tensor db = at::empty({C}, pre.options().dtype(st_of(ctx->saved_data["b_dt"])));
    Tensor ws = at::empty({dgtd_colsum2_workspace((int)C) / 4}, pre.options().dtype(at::kFloat));
    if (may_defer(ctx, "leaf_b")) {
      int nb = 0;
      check(dgtd_gelu_bias_bwd_partial(dh.data_ptr(), pre.data_ptr(), dpre.data_ptr(), ws.data_ptr(), rows, (int)C, code(pre), &nb, stream()),
            "dgtd_gelu_bias_bwd_partial");
      park(ws, nb, 2 * (int)C, nullptr, (int)C, db.data_ptr(), code(db));
    } else {
      check(dgtd_gelu_bias_bwd(dh.data_ptr(), pre.data_ptr(), dpre.data_ptr(), db.data_ptr(), code(db), ws.data_ptr(), rows, (int)C, code(pre),
                               stream()), "dgtd_gelu_bias_bwd");
    }
    Tensor dx;
    if (ctx->saved_data["need_dx"].toBool()) dx = gemm_dx(dpre, wc).view(ctx->saved_data["xshape"].toIntVector());
    Tensor dw;
    if (ar && may_defer(ctx, "leaf_w")) {
      dw = arena_slot(hint, ROLE_DW + xr, wc.sizes(), wc.options());
      park_gemm(hint, xr, x2, dpre, dw);
    } else {
      dw = gemm_dw(dpre, x2);
      const auto w_dt = st_of(ctx->saved_data["w_dt"]);
      if (dw.scalar_type() != w_dt) dw = dw.to(w_dt);
    }
    return {dx, dw, db, undefined()};
  }
};

// out = x + s[b] * gamma[c] * (h W^T + b) as ONE node (pwconv2 + layer scale + DropPath + residual, cod.py:1099-1116; attn.proj /
// Mlp.fc2 + DropPath + residual, cod.py:958-959): the backward computes dy = s*gamma*g, dgamma and the bias gradient sum(dy) in one
// pass (dgtd_scale_residual_bias_bwd) instead of scale_residual_bwd + column-sum (4 launches -> 2).
struct LinearResidualFn : public torch::autograd::Function<LinearResidualFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& h, const Tensor& w, const Tensor& b, const Tensor& x_, const c10::optional<Tensor>& s_,
                        const c10::optional<Tensor>& gamma_, int64_t dt_code) {
    const auto dt = from_code(dt_code);
    const bool has_s = s_.has_value() && s_->defined(), has_g = gamma_.has_value() && gamma_->defined();
    Tensor h2 = as_rows(h, dt);
    Tensor wc = w.scalar_type() == dt ? w : w.to(dt);
    Tensor bc = b.scalar_type() == dt ? b.contiguous() : b.to(dt);
    Tensor x = (x_.scalar_type() == dt ? x_ : x_.to(dt)).contiguous();
    const int64_t B = x.size(0), C = x.size(-1), rows = x.numel() / C;
    Tensor s = has_s ? f32(*s_) : Tensor(), g32 = has_g ? f32(*gamma_) : Tensor();
    Tensor out = at::empty_like(x);
    const int64_t Kg = h2.size(1);
    const bool fused = own_ok(dt, rows, C, Kg) && h2.size(0) == rows && wc.size(0) == C && h2.is_contiguous() && wc.is_contiguous() && aligned16(h2) &&
                       aligned16(wc) && aligned16(bc) && aligned16(x);
    Tensor y;
    if (fused) {     // ONE launch: GEMM + bias + x + s*gamma*y; y itself is kept only where the backward reads it (the layer-scale gradient)
      y = has_g ? at::empty({rows, C}, x.options()) : at::empty({0}, x.options());
      check(dgtd_gemm_bias_residual(h2.data_ptr(), wc.data_ptr(), bc.data_ptr(), x.data_ptr(), has_s ? s.data_ptr<float>() : nullptr,
                                    has_g ? g32.data_ptr<float>() : nullptr, has_g ? y.data_ptr() : nullptr, out.data_ptr(), (int)rows, (int)C, (int)Kg,
                                    rows / B, code(x), stream()), "dgtd_gemm_bias_residual");
    } else {
      y = gemm_fwd(h2, wc, bc);
      check(dgtd_scale_residual_fwd(x.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr, has_g ? g32.data_ptr<float>() : nullptr,
                                    out.data_ptr(), rows, (int)C, rows / B, code(x), stream()), "dgtd_scale_residual_fwd");
    }
    ctx->save_for_backward({h2, wc, y, s, g32});
    const Hint hint = t_hint;
    const int xr = (hinted(hint, dt) && w.scalar_type() == dt) ? find_role(hint, ROLE_X, h2) : -1;
    ctx->saved_data["hint"] = hint_iv(xr >= 0 ? hint : Hint{}, xr);
    note_leaf(ctx, "leaf_w", w);
    note_leaf(ctx, "leaf_b", b);
    if (has_g) note_leaf(ctx, "leaf_g", *gamma_);
    ctx->saved_data["hshape"] = h.sizes().vec();
    ctx->saved_data["meta"] = std::vector<int64_t>{has_s, has_g, st_id(w), st_id(b), has_g ? st_id(*gamma_) : (int64_t)at::kFloat,
                                                   h.requires_grad(), B};
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &h2 = saved[0], &wc = saved[1], &y = saved[2], &s = saved[3], &g32 = saved[4];
    const auto m = ctx->saved_data["meta"].toIntVector();
    const bool has_s = m[0], has_g = m[1], need_dh = m[5];
    const auto w_dt = (at::ScalarType)m[2], b_dt = (at::ScalarType)m[3], g_dt = (at::ScalarType)m[4];
    Tensor g = gr[0].contiguous();
    if (g.scalar_type() != h2.scalar_type()) g = g.to(h2.scalar_type());
    // y (the Linear's own output) may be an empty placeholder: the fused forward keeps it only for the layer-scale gradient
    const int64_t rows = h2.size(0), C = wc.size(0), B = m[6];
    const Hint hint = hint_of(ctx->saved_data["hint"]);
    const int xr = role_of(ctx->saved_data["hint"], 0);
    const bool ar = hint.group >= 0 && deferring();
    Tensor dy = ar ? arena_slot(hint, ROLE_DY + xr, {rows, C}, h2.options()) : at::empty({rows, C}, h2.options());
    Tensor db = at::empty({C}, h2.options().dtype(b_dt));
    Tensor dgamma = has_g ? at::empty({C}, h2.options().dtype(at::kFloat)) : Tensor();
    Tensor ws = at::empty({dgtd_colsum2_workspace((int)C) / 4}, h2.options().dtype(at::kFloat));
    // the layer-scale gradient is deferred only when it needs no dtype conversion afterwards (gamma is an fp32 parameter on this path)
    if ((!has_g || (g_dt == at::kFloat && may_defer(ctx, "leaf_g"))) && may_defer(ctx, "leaf_b")) {
      int nb = 0;
      check(dgtd_scale_residual_bias_bwd_partial(g.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr,
                                                 has_g ? g32.data_ptr<float>() : nullptr, dy.data_ptr(), ws.data_ptr(), rows, (int)C, rows / B,
                                                 code(h2), &nb, stream()), "dgtd_scale_residual_bias_bwd_partial");
      park(ws, nb, 2 * (int)C, has_g ? dgamma.data_ptr<float>() : nullptr, (int)C, db.data_ptr(), code(db));
    } else {
      check(dgtd_scale_residual_bias_bwd(g.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr, has_g ? g32.data_ptr<float>() : nullptr,
                                         dy.data_ptr(), has_g ? dgamma.data_ptr<float>() : nullptr, db.data_ptr(), code(db), ws.data_ptr(), rows,
                                         (int)C, rows / B, code(h2), stream()), "dgtd_scale_residual_bias_bwd");
    }
    Tensor dh;
    if (need_dh) dh = gemm_dx(dy, wc).view(ctx->saved_data["hshape"].toIntVector());
    Tensor dw;
    if (ar && may_defer(ctx, "leaf_w")) {
      dw = arena_slot(hint, ROLE_DW + xr, wc.sizes(), wc.options());
      park_gemm(hint, xr, h2, dy, dw);
    } else {
      dw = gemm_dw(dy, h2);
      if (dw.scalar_type() != w_dt) dw = dw.to(w_dt);
    }
    if (has_g && g_dt != at::kFloat) dgamma = dgamma.to(g_dt);
    return {dh, dw, db, g, undefined(), dgamma, undefined()};
  }
};


// out = x + s[b] * gamma[c] * (gelu(v W1^T + b1) W2^T + b2) as ONE node: the whole pointwise half of a convnext_Block (cod.py:1097-1116)
// on the package's own GEMM.  Forward = two launches (GEMM + bias + GELU writing pre and h; GEMM + bias + layer scale + DropPath +
// residual).  Backward = scale-residual backward (dy2, dgamma, db2 partials), then the input gradient of pwconv2 taken THROUGH the GELU
// inside its GEMM's epilogue (dpre = (dy2 W2) * gelu'(pre), bias-gradient partials of pwconv1 from the same tile), then the input
// gradient of pwconv1: no pass over the [tokens, 4C] hidden tensor outside a GEMM.  Weight gradients: deferred batched GEMMs (arenas).
struct MlpResidualFn : public torch::autograd::Function<MlpResidualFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& v, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, const Tensor& x_,
                        const c10::optional<Tensor>& s_, const c10::optional<Tensor>& gamma_, int64_t dt_code) {
    const auto dt = from_code(dt_code);
    const bool has_s = s_.has_value() && s_->defined(), has_g = gamma_.has_value() && gamma_->defined();
    Tensor v2 = as_rows(v, dt);
    Tensor wc1 = (w1.scalar_type() == dt ? w1 : w1.to(dt)).contiguous(), wc2 = (w2.scalar_type() == dt ? w2 : w2.to(dt)).contiguous();
    Tensor bc1 = b1.scalar_type() == dt ? b1.contiguous() : b1.to(dt), bc2 = b2.scalar_type() == dt ? b2.contiguous() : b2.to(dt);
    Tensor x = (x_.scalar_type() == dt ? x_ : x_.to(dt)).contiguous();
    const int64_t M = v2.size(0), C = v2.size(1), H4 = wc1.size(0), B = x.size(0);
    TORCH_CHECK(own_ok(dt, M, H4, C) && own_ok(dt, M, C, H4) && wc2.size(0) == C && wc2.size(1) == H4 && x.numel() == M * C && aligned16(v2) &&
                aligned16(wc1) && aligned16(wc2) && aligned16(bc1) && aligned16(bc2) && aligned16(x),
                "dgtd mlp_residual: shapes outside the package's GEMM (check dgtd.gemm_ok first)");
    const Hint hint = t_hint;
    const int out_role = take_out_role();
    const bool ar = hinted(hint, dt) && w1.scalar_type() == dt && w2.scalar_type() == dt;
    Tensor pre = at::empty({M, H4}, v2.options());
    Tensor h = (ar && out_role >= 0) ? arena_slot(hint, ROLE_X + out_role, {M, H4}, v2.options()) : at::empty({M, H4}, v2.options());
    check(dgtd_gemm_bias_gelu(v2.data_ptr(), wc1.data_ptr(), bc1.data_ptr(), pre.data_ptr(), h.data_ptr(), (int)M, (int)H4, (int)C, code(v2), stream()),
          "dgtd_gemm_bias_gelu");
    Tensor s = has_s ? f32(*s_) : Tensor(), g32 = has_g ? f32(*gamma_) : Tensor();
    Tensor y = has_g ? at::empty({M, C}, v2.options()) : at::empty({0}, v2.options());
    Tensor out = at::empty_like(x);
    check(dgtd_gemm_bias_residual(h.data_ptr(), wc2.data_ptr(), bc2.data_ptr(), x.data_ptr(), has_s ? s.data_ptr<float>() : nullptr,
                                  has_g ? g32.data_ptr<float>() : nullptr, has_g ? y.data_ptr() : nullptr, out.data_ptr(), (int)M, (int)C, (int)H4, M / B,
                                  code(v2), stream()), "dgtd_gemm_bias_residual");
    const int r1 = ar ? find_role(hint, ROLE_X, v2) : -1, r2 = ar ? find_role(hint, ROLE_X, h) : -1;
    ctx->saved_data["hint"] = hint_iv((r1 >= 0 && r2 >= 0) ? hint : Hint{}, r1, r2);
    note_leaf(ctx, "leaf_w1", w1); note_leaf(ctx, "leaf_b1", b1); note_leaf(ctx, "leaf_w2", w2); note_leaf(ctx, "leaf_b2", b2);
    if (has_g) note_leaf(ctx, "leaf_g", *gamma_);
    ctx->save_for_backward({v2, wc1, wc2, pre, h, y, s, g32});
    ctx->saved_data["vshape"] = v.sizes().vec();
    ctx->saved_data["meta"] = std::vector<int64_t>{has_s, has_g, st_id(w1), st_id(b1), st_id(w2), st_id(b2), has_g ? st_id(*gamma_) : (int64_t)at::kFloat,
                                                   v.requires_grad(), B};
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &v2 = saved[0], &wc1 = saved[1], &wc2 = saved[2], &pre = saved[3], &h = saved[4], &y = saved[5], &s = saved[6], &g32 = saved[7];
    const auto m = ctx->saved_data["meta"].toIntVector();
    const bool has_s = m[0], has_g = m[1], need_dv = m[7];
    const auto w1_dt = (at::ScalarType)m[2], b1_dt = (at::ScalarType)m[3], w2_dt = (at::ScalarType)m[4], b2_dt = (at::ScalarType)m[5], g_dt = (at::ScalarType)m[6];
    const int64_t M = v2.size(0), C = v2.size(1), H4 = wc1.size(0), B = m[8];
    Tensor g = gr[0].contiguous();
    if (g.scalar_type() != v2.scalar_type()) g = g.to(v2.scalar_type());
    const Hint hint = hint_of(ctx->saved_data["hint"]);
    const int r1 = role_of(ctx->saved_data["hint"], 0), r2 = role_of(ctx->saved_data["hint"], 1);
    const bool ar = hint.group >= 0 && deferring();
    // (1) through the residual epilogue: dy2 = s * gamma * g, dgamma = sum g * s * y, db2 = sum dy2
    Tensor dy2 = ar ? arena_slot(hint, ROLE_DY + r2, {M, C}, v2.options()) : at::empty({M, C}, v2.options());
    Tensor db2 = at::empty({C}, v2.options().dtype(b2_dt));
    Tensor dgamma = has_g ? at::empty({C}, v2.options().dtype(at::kFloat)) : Tensor();
    Tensor ws2 = at::empty({dgtd_colsum2_workspace((int)C) / 4}, v2.options().dtype(at::kFloat));
    if ((!has_g || (g_dt == at::kFloat && may_defer(ctx, "leaf_g"))) && may_defer(ctx, "leaf_b2")) {
      int nb = 0;
      check(dgtd_scale_residual_bias_bwd_partial(g.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr, has_g ? g32.data_ptr<float>() : nullptr,
                                                 dy2.data_ptr(), ws2.data_ptr(), M, (int)C, M / B, code(v2), &nb, stream()),
            "dgtd_scale_residual_bias_bwd_partial");
      park(ws2, nb, 2 * (int)C, has_g ? dgamma.data_ptr<float>() : nullptr, (int)C, db2.data_ptr(), code(db2));
    } else {
      check(dgtd_scale_residual_bias_bwd(g.data_ptr(), y.data_ptr(), has_s ? s.data_ptr<float>() : nullptr, has_g ? g32.data_ptr<float>() : nullptr,
                                         dy2.data_ptr(), has_g ? dgamma.data_ptr<float>() : nullptr, db2.data_ptr(), code(db2), ws2.data_ptr(), M, (int)C,
                                         M / B, code(v2), stream()), "dgtd_scale_residual_bias_bwd");
    }
    // (2) input gradient of pwconv2 through the GELU, bias-gradient partials of pwconv1 from the same tiles
    Tensor dpre = ar ? arena_slot(hint, ROLE_DY + r1, {M, H4}, v2.options()) : at::empty({M, H4}, v2.options());
    Tensor db1 = at::empty({H4}, v2.options().dtype(b1_dt));
    Tensor ws1 = at::empty({dgtd_gemm_gelu_bwd_workspace((int)M, (int)H4) / 4}, v2.options().dtype(at::kFloat));
    int nb1 = 0;
    check(dgtd_gemm_gelu_bwd(dy2.data_ptr(), transposed_weight(wc2).data_ptr(), pre.data_ptr(), dpre.data_ptr(), ws1.data_ptr(), &nb1, (int)M, (int)H4,
                             (int)C, code(v2), stream()), "dgtd_gemm_gelu_bwd");
    if (may_defer(ctx, "leaf_b1")) {
      park(ws1, nb1, (int)H4, nullptr, 0, db1.data_ptr(), code(db1));
    } else {
      const dgtd_reduce_entry e{ws1.data_ptr<float>(), nb1, (int32_t)H4, nullptr, 0, db1.data_ptr(), (int32_t)code(db1), 0, 0, nullptr};
      check(dgtd_multi_reduce(&e, 1, stream()), "dgtd_multi_reduce");
    }
    // (3) input gradient of pwconv1
    Tensor dv;
    if (need_dv) {
      dv = at::empty({M, C}, v2.options());
      check(dgtd_gemm_bias(dpre.data_ptr(), transposed_weight(wc1).data_ptr(), nullptr, dv.data_ptr(), (int)M, (int)C, (int)H4, code(v2), stream()),
            "dgtd_gemm_bias (input gradient)");
      dv = dv.view(ctx->saved_data["vshape"].toIntVector());
    }
    // (4) weight gradients: parked for the batched GEMMs of the deferred phase, or per layer
    Tensor dw1, dw2;
    if (ar && may_defer(ctx, "leaf_w2")) { dw2 = arena_slot(hint, ROLE_DW + r2, wc2.sizes(), wc2.options()); park_gemm(hint, r2, h, dy2, dw2); }
    else { dw2 = gemm_dw(dy2, h); if (dw2.scalar_type() != w2_dt) dw2 = dw2.to(w2_dt); }
    if (ar && may_defer(ctx, "leaf_w1")) { dw1 = arena_slot(hint, ROLE_DW + r1, wc1.sizes(), wc1.options()); park_gemm(hint, r1, v2, dpre, dw1); }
    else { dw1 = gemm_dw(dpre, v2); if (dw1.scalar_type() != w1_dt) dw1 = dw1.to(w1_dt); }
    if (has_g && g_dt != at::kFloat) dgamma = dgamma.to(g_dt);
    return {dv, dw1, db1, dw2, db2, g, undefined(), dgamma, undefined()};
  }
};

// ------------------------------------------------------------------------------------------------ dense conv3x3 (NHWC bf16 MFMA)
// raw form: x [Z|1,B,H,W,Ci], w [Z,Co,3,3,Ci], b [Z,Co]? -> y [Z,B,H,W,Co]
struct ConvGeom { int Z, B, H, W, Ci, Co; bool shared; };
inline void conv3x3_backward_raw(const Tensor& x, const Tensor& w, const Tensor& y_mask, const Tensor& dy, const ConvGeom& g, bool need_dx,
                                 bool has_b, Tensor& dx, Tensor& dw, Tensor& db, bool skip_wgrad = false) {
  const void* mask = y_mask.defined() ? y_mask.data_ptr() : nullptr;
  if (need_dx) {
    // the transposed + flipped kernel of the input-gradient convolution: a module called several times per step (CABs of the decoder
    // iterations) flips the same weight every time - under deferral (one step = one set of weight values) the first flip is kept
    Tensor wt;
    const bool cache = deferring() && g.Z == 1;
    if (cache) {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      auto it = g_conv_flip.find(w.data_ptr());
      if (it != g_conv_flip.end()) wt = it->second;
    }
    if (!wt.defined()) {
      wt = at::empty({g.Z, g.Ci, 3, 3, g.Co}, w.options());
      check(dgtd_conv3x3_flip(w.data_ptr(), wt.data_ptr(), g.Z, g.Co, g.Ci, stream()), "dgtd_conv3x3_flip");
      if (cache) {
        std::lock_guard<std::mutex> lk(g_pending_mu);
        g_conv_flip.emplace(w.data_ptr(), wt);
      }
    }
    check(dgtd_conv3x3_fwd(dy.data_ptr(), mask, wt.data_ptr(), nullptr, dx.data_ptr(), g.Z, g.B, g.H, g.W, g.Co, g.Ci, 0, 0, code(dy), stream()),
          "dgtd_conv3x3_fwd (input gradient)");
  }
  if (skip_wgrad) return;                          // weight gradient parked for the deferred phase
  Tensor ws = at::empty({dgtd_conv3x3_wgrad_workspace(g.Z, g.B, g.H, g.W, g.Ci, g.Co)}, x.options().dtype(at::kByte));
  check(dgtd_conv3x3_wgrad(x.data_ptr(), dy.data_ptr(), mask, dw.data_ptr(), has_b ? db.data_ptr() : nullptr, ws.data_ptr(), g.Z, g.B, g.H,
                           g.W, g.Ci, g.Co, g.shared ? 1 : 0, code(x), stream()), "dgtd_conv3x3_wgrad");
}

struct Conv3x3Fn : public torch::autograd::Function<Conv3x3Fn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& w_, const c10::optional<Tensor>& b_, bool relu) {
    Tensor x = x_.contiguous(), w = w_.contiguous();
    on_device(x);
    TORCH_CHECK(x.dim() == 5 && w.dim() == 5 && is16(x.scalar_type()) && w.scalar_type() == x.scalar_type(),
                "dgtd conv3x3 takes bf16 or fp16 x [Z|1,B,H,W,Ci] and w [Z,Co,3,3,Ci] of one dtype");
    const bool has_b = b_.has_value() && b_->defined();
    ConvGeom g{(int)w.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)w.size(4), (int)w.size(1), x.size(0) == 1 && w.size(0) > 1};
    Tensor b = has_b ? b_->contiguous() : Tensor();
    Tensor y = at::empty({g.Z, g.B, g.H, g.W, g.Co}, x.options());
    check(dgtd_conv3x3_fwd(x.data_ptr(), nullptr, w.data_ptr(), has_b ? b.data_ptr() : nullptr, y.data_ptr(), g.Z, g.B, g.H, g.W, g.Ci, g.Co,
                           relu ? 1 : 0, g.shared ? 1 : 0, code(x), stream()), "dgtd_conv3x3_fwd");
    ctx->save_for_backward({x, w, relu ? y : Tensor()});
    ctx->saved_data["has_b"] = has_b;
    ctx->saved_data["need_dx"] = x_.requires_grad();
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w = saved[1], &ym = saved[2];
    const bool has_b = ctx->saved_data["has_b"].toBool(), need_dx = ctx->saved_data["need_dx"].toBool();
    ConvGeom g{(int)w.size(0), (int)x.size(1), (int)x.size(2), (int)x.size(3), (int)w.size(4), (int)w.size(1), x.size(0) == 1 && w.size(0) > 1};
    Tensor dy = (gr[0].scalar_type() == x.scalar_type() ? gr[0] : gr[0].to(x.scalar_type())).contiguous();
    Tensor dx = need_dx ? at::empty({g.Z, g.B, g.H, g.W, g.Ci}, x.options()) : Tensor();
    Tensor dw = at::empty_like(w), db = has_b ? at::empty({g.Z, g.Co}, w.options()) : Tensor();
    conv3x3_backward_raw(x, w, ym, dy, g, need_dx, has_b, dx, dw, db);
    if (need_dx && g.shared) dx = at::sum(dx, {0}, true);
    return {dx, dw, db, undefined()};
  }
};

// logical form: x [B,Ci,H,W] and w [Co,Ci,3,3] in channels_last memory (= NHWC / OHWI); every tensor handed to autograd is a base
// tensor in channels_last memory format, so no permute / view nodes surround the kernel
struct Conv3x3ClFn : public torch::autograd::Function<Conv3x3ClFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& w_, const c10::optional<Tensor>& b_, bool relu) {
    Tensor x = x_.contiguous(at::MemoryFormat::ChannelsLast), w = w_.contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(x.is_cuda() && is16(x.scalar_type()) && w.scalar_type() == x.scalar_type(), "dgtd conv3x3 takes bf16 or fp16 tensors (one dtype) on the HIP device");
    const bool has_b = b_.has_value() && b_->defined();
    ConvGeom g{1, (int)x.size(0), (int)x.size(2), (int)x.size(3), (int)w.size(1), (int)w.size(0), false};
    Tensor b = has_b ? b_->contiguous() : Tensor();
    Tensor y = at::empty({g.B, g.Co, g.H, g.W}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    check(dgtd_conv3x3_fwd(x.data_ptr(), nullptr, w.data_ptr(), has_b ? b.data_ptr() : nullptr, y.data_ptr(), 1, g.B, g.H, g.W, g.Ci, g.Co,
                           relu ? 1 : 0, 0, code(x), stream()), "dgtd_conv3x3_fwd");
    ctx->save_for_backward({x, w, relu ? y : Tensor()});
    ctx->saved_data["has_b"] = has_b;
    ctx->saved_data["need_dx"] = x_.requires_grad();
    ctx->saved_data["defer"] = conv_register(w_, w, ConvKey{g.B, g.H, g.W, g.Ci, g.Co, (int)code(x)}) &&
                               (!has_b || (b_->requires_grad() && !b_->grad_fn() && b_->data_ptr() == b.data_ptr()));
    note_leaf(ctx, "leaf_w", w_);
    if (has_b) note_leaf(ctx, "leaf_b", *b_);
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w = saved[1], &ym = saved[2];
    const bool has_b = ctx->saved_data["has_b"].toBool(), need_dx = ctx->saved_data["need_dx"].toBool();
    ConvGeom g{1, (int)x.size(0), (int)x.size(2), (int)x.size(3), (int)w.size(1), (int)w.size(0), false};
    Tensor dy = (gr[0].scalar_type() == x.scalar_type() ? gr[0] : gr[0].to(x.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    Tensor dx = need_dx ? at::empty_like(x) : Tensor();
    Tensor dw, db;
    // first call of this weight in the backward pass: its leaf (and bias) must not own a gradient yet; later calls share the destination
    const bool parked = ctx->saved_data["defer"].toBool() && deferring() &&
                        (conv_dest_exists(w) || !(leaf_has_grad(ctx, "leaf_w") || (has_b && leaf_has_grad(ctx, "leaf_b")))) &&
                        conv_park(x, dy, ym, w, ConvKey{g.B, g.H, g.W, g.Ci, g.Co, (int)code(x)}, has_b, dw, db);
    if (!parked) { dw = at::empty_like(w); db = has_b ? at::empty({g.Co}, w.options()) : Tensor(); }
    conv3x3_backward_raw(x, w, ym, dy, g, need_dx, has_b, dx, dw, db, parked);
    return {dx, dw, db, undefined()};
  }
};

// ------------------------------------------------------------------------------------------------ Hitnet glue: PReLU, CA gate, bilinear
inline Tensor dense(const Tensor& x) { return x.is_non_overlapping_and_dense() ? x : x.contiguous(); }

struct PReLUFn : public torch::autograd::Function<PReLUFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& a) {
    Tensor x = dense(x_);
    TORCH_CHECK(x.is_cuda() && a.numel() == 1, "dgtd prelu: single-slope PReLU on the HIP device");
    Tensor a32 = f32(a);
    Tensor y = at::empty_like(x);
    check(dgtd_prelu_fwd(x.data_ptr(), a32.data_ptr<float>(), y.data_ptr(), x.numel(), code(x), stream()), "dgtd_prelu_fwd");
    ctx->save_for_backward({x, a32});
    ctx->saved_data["a_dt"] = st_id(a);
    ctx->saved_data["a_key"] = (a.requires_grad() && !a.grad_fn()) ? (int64_t)(intptr_t)a.data_ptr() : (int64_t)0;
    note_leaf(ctx, "leaf_a", a);
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &a32 = saved[1];
    Tensor g = gr[0];
    if (g.scalar_type() != x.scalar_type() || g.strides() != x.strides()) g = at::empty_like(x).copy_(g);
    Tensor dx = at::empty_like(x);
    const void* key = (const void*)(intptr_t)ctx->saved_data["a_key"].toInt();
    bool share = key && deferring() && g_shared_ok.load(std::memory_order_relaxed);
    if (share) {
      bool have;
      queue_final_flush();
      { std::lock_guard<std::mutex> lk(g_pending_mu); have = g_prelu_acc.count(key) > 0; }
      if (!have && leaf_has_grad(ctx, "leaf_a")) share = false;       // accumulation: the slope already owns a gradient
    }
    if (share) {
      Tensor acc, ret;
      {
        std::lock_guard<std::mutex> lk(g_pending_mu);
        auto it = g_prelu_acc.find(key);
        if (it == g_prelu_acc.end()) {
          acc = at::zeros({1}, x.options().dtype(at::kFloat));
          ret = at::empty({1}, x.options().dtype(st_of(ctx->saved_data["a_dt"])));
          g_prelu_acc.emplace(key, acc);
          g_pending.push_back(PendingReduce{dgtd_reduce_entry{acc.data_ptr<float>(), 1, 1, nullptr, 0, ret.data_ptr(), (int32_t)code(ret), 0, 0, nullptr}, acc});
        } else {
          acc = it->second;
        }
      }
      check(dgtd_prelu_bwd(x.data_ptr(), g.data_ptr(), a32.data_ptr<float>(), dx.data_ptr(), acc.data_ptr<float>(), x.numel(), code(x), stream()),
            "dgtd_prelu_bwd");
      return {dx, ret};
    }
    Tensor da = at::zeros({1}, x.options().dtype(at::kFloat));
    check(dgtd_prelu_bwd(x.data_ptr(), g.data_ptr(), a32.data_ptr<float>(), dx.data_ptr(), da.data_ptr<float>(), x.numel(), code(x), stream()),
          "dgtd_prelu_bwd");
    if (st_of(ctx->saved_data["a_dt"]) != at::kFloat) da = da.to(st_of(ctx->saved_data["a_dt"]));
    return {dx, da};
  }
};

struct CAGateFn : public torch::autograd::Function<CAGateFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& res_, const Tensor& x_, const Tensor& w1, const Tensor& w2) {
    Tensor res = res_.contiguous(at::MemoryFormat::ChannelsLast);
    Tensor x = (x_.scalar_type() == res.scalar_type() ? x_ : x_.to(res.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(res.is_cuda(), "dgtd ca_gate runs on the HIP device");
    const int64_t B = res.size(0), C = res.size(1), HW = res.size(2) * res.size(3), R = w1.size(0);
    Tensor w1f = f32(w1.reshape({R, C})), w2f = f32(w2.reshape({C, R}));
    Tensor stats = at::empty({2 * B * C + B * R + 64 * B * C}, res.options().dtype(at::kFloat));
    Tensor out = at::empty_like(res);
    check(dgtd_ca_gate_fwd(res.data_ptr(), x.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), out.data_ptr(), stats.data_ptr<float>(),
                           (int)B, (int)HW, (int)C, (int)R, code(res), stream()), "dgtd_ca_gate_fwd");
    ctx->save_for_backward({res, w1f, w2f, stats});
    ctx->saved_data["w_dt"] = st_id(w1);
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &res = saved[0], &w1f = saved[1], &w2f = saved[2], &stats = saved[3];
    const int64_t B = res.size(0), C = res.size(1), HW = res.size(2) * res.size(3), R = w1f.size(0);
    Tensor g = (gr[0].scalar_type() == res.scalar_type() ? gr[0] : gr[0].to(res.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    Tensor dres = at::empty_like(res);
    Tensor small = at::empty({2 * R * C + B * C + 64 * B * C + B * 2 * R * C}, res.options().dtype(at::kFloat));
    float* sp = small.data_ptr<float>();
    check(dgtd_ca_gate_bwd(g.data_ptr(), res.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), stats.data_ptr<float>(), dres.data_ptr(),
                           sp, sp + R * C, sp + 2 * R * C, (int)B, (int)HW, (int)C, (int)R, code(res), stream()), "dgtd_ca_gate_bwd");
    Tensor dw1 = small.narrow(0, 0, R * C).view({R, C, 1, 1}), dw2 = small.narrow(0, R * C, R * C).view({C, R, 1, 1});
    if (st_of(ctx->saved_data["w_dt"]) != at::kFloat) { dw1 = dw1.to(st_of(ctx->saved_data["w_dt"])); dw2 = dw2.to(st_of(ctx->saved_data["w_dt"])); }
    return {dres, g, dw1, dw2};
  }
};

// SAM (cod.py:454-506) as one node: 2 launches forward, 2 backward (csrc/sam.hip)
struct SamFn : public torch::autograd::Function<SamFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& xh_, const Tensor& xl_, const Tensor& w1, const Tensor& w2, const Tensor& v1,
                        const Tensor& v2) {
    Tensor xh = xh_.contiguous(at::MemoryFormat::ChannelsLast);
    Tensor xl = (xl_.scalar_type() == xh.scalar_type() ? xl_ : xl_.to(xh.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(xh.is_cuda() && xh.dim() == 4 && xl.sizes() == xh.sizes(), "dgtd sam: two 4-D maps of one shape on the HIP device");
    const int64_t B = xh.size(0), C = xh.size(1), HW = xh.size(2) * xh.size(3), R = w1.size(0);
    TORCH_CHECK(w1.numel() == R * C && w2.numel() == C * R && v1.numel() == R * C && v2.numel() == R, "dgtd sam: weight shapes");
    Tensor w1f = f32(w1.reshape({R, C})), w2f = f32(w2.reshape({C, R})), v1f = f32(v1.reshape({R, C})), v2f = f32(v2.reshape({R}));
    Tensor stats = at::empty({dgtd_sam_stats_floats((int)B, (int)C, (int)R)}, xh.options().dtype(at::kFloat));
    Tensor out = at::empty_like(xh);
    check(dgtd_sam_fwd(xh.data_ptr(), xl.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), v1f.data_ptr<float>(), v2f.data_ptr<float>(),
                       out.data_ptr(), stats.data_ptr<float>(), (int)B, (int)HW, (int)C, (int)R, code(xh), stream()), "dgtd_sam_fwd");
    ctx->save_for_backward({xh, xl, w1f, w2f, v1f, v2f, stats});
    ctx->saved_data["w_dt"] = std::vector<int64_t>{st_id(w1), st_id(w2), st_id(v1), st_id(v2)};
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &xh = saved[0], &xl = saved[1], &w1f = saved[2], &w2f = saved[3], &v1f = saved[4], &v2f = saved[5], &stats = saved[6];
    const int64_t B = xh.size(0), C = xh.size(1), HW = xh.size(2) * xh.size(3), R = w1f.size(0);
    Tensor g = (gr[0].scalar_type() == xh.scalar_type() ? gr[0] : gr[0].to(xh.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    Tensor dxh = at::empty_like(xh), dxl = at::empty_like(xh);
    Tensor dw = at::empty({3 * R * C + R}, xh.options().dtype(at::kFloat));
    Tensor scratch = at::empty({dgtd_sam_scratch_floats((int)B, (int)C)}, xh.options().dtype(at::kFloat));
    check(dgtd_sam_bwd(g.data_ptr(), xh.data_ptr(), xl.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), v1f.data_ptr<float>(),
                       v2f.data_ptr<float>(), stats.data_ptr<float>(), dxh.data_ptr(), dxl.data_ptr(), dw.data_ptr<float>(), scratch.data_ptr<float>(),
                       (int)B, (int)HW, (int)C, (int)R, code(xh), stream()), "dgtd_sam_bwd");
    const auto dts = ctx->saved_data["w_dt"].toIntVector();
    auto as = [&](Tensor t, int i) { return st_of(dts[i]) == at::kFloat ? t : t.to(st_of(dts[i])); };
    return {dxh, dxl, as(dw.narrow(0, 0, R * C).view({R, C}), 0), as(dw.narrow(0, R * C, R * C).view({C, R}), 1),
            as(dw.narrow(0, 2 * R * C, R * C).view({R, C}), 2), as(dw.narrow(0, 3 * R * C, R).view({1, R}), 3)};
  }
};

// nn.BatchNorm2d of BasicConv2d (cod.py:359, :366) as one node: 2 launches each way (csrc/batchnorm.hip), the running statistics and
// num_batches_tracked moved inside the second forward launch.  A module applied several times per step (compress_out, compress_out2,
// conv4 inside the decoder loop, cod.py:757-789) writes one { dgamma | dbeta } row per call; the flush adds the rows up (as for the CAB's
// channel-attention weights) so autograd sees one gradient per parameter per step.
struct BatchNormFn : public torch::autograd::Function<BatchNormFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& gamma, const Tensor& beta, Tensor running_mean, Tensor running_var,
                        Tensor num_batches, bool training, double momentum, double eps) {
    Tensor x = x_.contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(x.is_cuda() && x.dim() == 4, "dgtd batch_norm takes a 4-D map on the HIP device");
    const int64_t C = x.size(1), N = x.numel() / C;
    TORCH_CHECK(gamma.scalar_type() == at::kFloat && beta.scalar_type() == at::kFloat && running_mean.scalar_type() == at::kFloat &&
                running_var.scalar_type() == at::kFloat && num_batches.scalar_type() == at::kLong && gamma.is_contiguous() && beta.is_contiguous() &&
                running_mean.is_contiguous() && running_var.is_contiguous(), "dgtd batch_norm: fp32 parameters / statistics and an int64 counter");
    Tensor y = at::empty_like(x);
    Tensor save = at::empty({training ? 2 * C : 0}, x.options().dtype(at::kFloat));
    Tensor scratch = at::empty({training ? dgtd_batchnorm_scratch((int)C) : 0}, x.options().dtype(at::kFloat));
    check(dgtd_batchnorm_fwd(x.data_ptr(), gamma.data_ptr<float>(), beta.data_ptr<float>(), running_mean.data_ptr<float>(), running_var.data_ptr<float>(),
                             (long long*)num_batches.data_ptr<int64_t>(), y.data_ptr(), training ? save.data_ptr<float>() : nullptr,
                             training ? scratch.data_ptr<float>() : nullptr, N, (int)C, (float)eps, (float)momentum, training ? 1 : 0, code(x), stream()),
          "dgtd_batchnorm_fwd");
    ctx->saved_data["training"] = training;
    ctx->save_for_backward({x, gamma, save});
    note_leaf(ctx, "leaf_g", gamma); note_leaf(ctx, "leaf_b", beta);
    ctx->saved_data["key"] = (is_leaf(gamma) && is_leaf(beta)) ? (int64_t)(intptr_t)gamma.data_ptr() : (int64_t)0;
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &gamma = saved[1], &save = saved[2];
    TORCH_CHECK(ctx->saved_data["training"].toBool(), "dgtd batch_norm: the backward exists for training statistics only");
    const int64_t C = x.size(1), N = x.numel() / C;
    Tensor g = (gr[0].scalar_type() == x.scalar_type() ? gr[0] : gr[0].to(x.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    Tensor dx = at::empty_like(x);
    Tensor scratch = at::empty({dgtd_batchnorm_scratch((int)C)}, x.options().dtype(at::kFloat));
    Tensor dg, db, own;
    float* row = nullptr;
    const void* key = (const void*)(intptr_t)ctx->saved_data["key"].toInt();
    bool shared = key && deferring() && g_shared_ok.load(std::memory_order_relaxed);
    if (shared) {
      queue_final_flush();
      bool have;
      { std::lock_guard<std::mutex> lk(g_pending_mu); have = g_ca_acc.count(key) > 0; }
      if (!have && (leaf_has_grad(ctx, "leaf_g") || leaf_has_grad(ctx, "leaf_b"))) shared = false;
    }
    if (shared) {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      auto it = g_ca_acc.find(key);
      if (it == g_ca_acc.end()) {
        dg = at::empty({C}, x.options().dtype(at::kFloat));
        db = at::empty({C}, x.options().dtype(at::kFloat));
        CaAcc acc{at::empty({CA_MAX_CALLS, 2 * C}, x.options().dtype(at::kFloat)), 0, (int)C, dg.data_ptr(), db.data_ptr()};
        it = g_ca_acc.emplace(key, acc).first;
      }
      if (it->second.count < CA_MAX_CALLS) row = it->second.buf.data_ptr<float>() + (int64_t)(it->second.count++) * 2 * C;
      else shared = false;
    }
    if (!shared) {
      own = at::empty({2 * C}, x.options().dtype(at::kFloat));
      row = own.data_ptr<float>();
      dg = own.narrow(0, 0, C);
      db = own.narrow(0, C, C);
    }
    check(dgtd_batchnorm_bwd(g.data_ptr(), x.data_ptr(), gamma.data_ptr<float>(), save.data_ptr<float>(), dx.data_ptr(), row, row + C,
                             scratch.data_ptr<float>(), N, (int)C, code(x), stream()), "dgtd_batchnorm_bwd");
    return {dx, dg, db, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

struct BilinearFn : public torch::autograd::Function<BilinearFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, int64_t Ho, int64_t Wo, bool align) {
    Tensor x = x_.contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(x.is_cuda() && x.dim() == 4, "dgtd bilinear_resize takes a 4-D tensor on the HIP device");
    const int64_t B = x.size(0), C = x.size(1), Hi = x.size(2), Wi = x.size(3);
    Tensor y = at::empty({B, C, Ho, Wo}, x.options().memory_format(at::MemoryFormat::ChannelsLast));
    check(dgtd_bilinear_fwd(x.data_ptr(), y.data_ptr(), (int)B, (int)Hi, (int)Wi, (int)Ho, (int)Wo, (int)C, align ? 1 : 0, code(x), stream()),
          "dgtd_bilinear_fwd");
    ctx->saved_data["geom"] = std::vector<int64_t>{B, C, Hi, Wi, Ho, Wo, align ? 1 : 0, st_id(x)};
    return y;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    const auto q = ctx->saved_data["geom"].toIntVector();
    const auto dt = (at::ScalarType)q[7];
    Tensor g = (gr[0].scalar_type() == dt ? gr[0] : gr[0].to(dt)).contiguous(at::MemoryFormat::ChannelsLast);
    Tensor dx = at::empty({q[0], q[1], q[2], q[3]}, g.options().memory_format(at::MemoryFormat::ChannelsLast));
    check(dgtd_bilinear_bwd(g.data_ptr(), dx.data_ptr(), (int)q[0], (int)q[2], (int)q[3], (int)q[4], (int)q[5], (int)q[1], (int)q[6], code(g), stream()),
          "dgtd_bilinear_bwd");
    return {dx, undefined(), undefined(), undefined()};
  }
};

// ------------------------------------------------------------------------------------------------ Hitnet CAB as ONE node
// out = x + gate(res) * res, res = conv3x3(PReLU(conv3x3(x, w0)), w1), gate = sigmoid(W2 relu(W1 mean(res)))   (CAB, cod.py:436-451)
// Forward 4 launches (conv + PReLU epilogue writing pre-activation and activation; conv; pooled sums; gate apply + residual) instead
// of 5; backward: gate backward, input-gradient convolution of w1 with the PReLU backward (and the slope gradient) in its epilogue,
// input-gradient convolution of w0 with the skip gradient added in its epilogue - no PReLU launches and no autograd add for the
// fork of x.  Weight gradients: the deferred batched launch of §4b (shared destination for the 4 calls of a module) or immediate.
inline Tensor flipped_weight(const Tensor& w, int Z, int Co, int Ci) {
  Tensor wt;
  const bool cache = deferring() && Z == 1;
  if (cache) {
    std::lock_guard<std::mutex> lk(g_pending_mu);
    auto it = g_conv_flip.find(w.data_ptr());
    if (it != g_conv_flip.end()) wt = it->second;
  }
  if (!wt.defined()) {
    wt = at::empty({Z, Ci, 3, 3, Co}, w.options());
    check(dgtd_conv3x3_flip(w.data_ptr(), wt.data_ptr(), Z, Co, Ci, stream()), "dgtd_conv3x3_flip");
    if (cache) {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      g_conv_flip.emplace(w.data_ptr(), wt);
    }
  }
  return wt;
}

struct CabFn : public torch::autograd::Function<CabFn> {
  static Tensor forward(AutogradContext* ctx, const Tensor& x_, const Tensor& w0_, const Tensor& w1_, const Tensor& a, const Tensor& cw1,
                        const Tensor& cw2) {
    Tensor x = x_.contiguous(at::MemoryFormat::ChannelsLast);
    Tensor w0 = w0_.contiguous(at::MemoryFormat::ChannelsLast), w1 = w1_.contiguous(at::MemoryFormat::ChannelsLast);
    TORCH_CHECK(x.is_cuda() && is16(x.scalar_type()) && w0.scalar_type() == x.scalar_type() && w1.scalar_type() == x.scalar_type() && a.numel() == 1,
                "dgtd cab: bf16 / fp16 tensors of one dtype on the HIP device and a single PReLU slope");
    const int B = (int)x.size(0), C = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3);
    const int64_t R = cw1.size(0), HW = (int64_t)H * W;
    Tensor a32 = f32(a);
    Tensor t0 = at::empty_like(x), t1 = at::empty_like(x), res = at::empty_like(x), out = at::empty_like(x);
    check(dgtd_conv3x3_fwd_ex(x.data_ptr(), nullptr, w0.data_ptr(), nullptr, t1.data_ptr(), t0.data_ptr(), nullptr, nullptr, a32.data_ptr<float>(), nullptr,
                              1, B, H, W, C, C, 2, 0, code(x), stream()), "dgtd_conv3x3_fwd_ex (conv + PReLU)");
    check(dgtd_conv3x3_fwd(t1.data_ptr(), nullptr, w1.data_ptr(), nullptr, res.data_ptr(), 1, B, H, W, C, C, 0, 0, code(x), stream()), "dgtd_conv3x3_fwd");
    Tensor w1f = f32(cw1.reshape({R, C})), w2f = f32(cw2.reshape({C, R}));
    Tensor stats = at::empty({2 * B * C + B * R + 64 * B * C}, x.options().dtype(at::kFloat));
    check(dgtd_ca_gate_fwd(res.data_ptr(), x.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), out.data_ptr(), stats.data_ptr<float>(), B, (int)HW,
                           C, (int)R, code(x), stream()), "dgtd_ca_gate_fwd");
    ctx->save_for_backward({x, w0, w1, t0, t1, res, a32, w1f, w2f, stats});
    const ConvKey key{B, H, W, C, C, (int)code(x)};
    ctx->saved_data["defer0"] = conv_register(w0_, w0, key);
    ctx->saved_data["defer1"] = conv_register(w1_, w1, key);
    note_leaf(ctx, "leaf_w0", w0_); note_leaf(ctx, "leaf_w1", w1_); note_leaf(ctx, "leaf_a", a);
    note_leaf(ctx, "leaf_cw1", cw1); note_leaf(ctx, "leaf_cw2", cw2);
    ctx->saved_data["meta"] = std::vector<int64_t>{st_id(a), st_id(cw1), (a.requires_grad() && !a.grad_fn()) ? (int64_t)(intptr_t)a.data_ptr() : (int64_t)0,
                                                   x_.requires_grad(), (is_leaf(cw1) && is_leaf(cw2) && cw2.scalar_type() == cw1.scalar_type())
                                                                           ? (int64_t)(intptr_t)cw1.data_ptr() : (int64_t)0};
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list gr) {
    auto saved = ctx->get_saved_variables();
    const Tensor &x = saved[0], &w0 = saved[1], &w1 = saved[2], &t0 = saved[3], &t1 = saved[4], &res = saved[5], &a32 = saved[6], &w1f = saved[7],
                 &w2f = saved[8], &stats = saved[9];
    const auto m = ctx->saved_data["meta"].toIntVector();
    const auto a_dt = (at::ScalarType)m[0], cw_dt = (at::ScalarType)m[1];
    const void* akey = (const void*)(intptr_t)m[2];
    const int B = (int)x.size(0), C = (int)x.size(1), H = (int)x.size(2), W = (int)x.size(3);
    const int64_t R = w1f.size(0), HW = (int64_t)H * W;
    Tensor g = (gr[0].scalar_type() == x.scalar_type() ? gr[0] : gr[0].to(x.scalar_type())).contiguous(at::MemoryFormat::ChannelsLast);
    // (1) through the gate: dres, the CA weight gradients; the skip gradient is g itself.  A module called several times per step (the
    //     decoder iterations) writes one { dw1 | dw2 } row per call into its accumulator buffer; the first call hands autograd the sums
    Tensor dres = at::empty_like(x);
    Tensor small = at::empty({2 * R * C + B * C + 64 * B * C + B * 2 * R * C}, x.options().dtype(at::kFloat));
    float* sp = small.data_ptr<float>();
    float* row = sp;                                 // where this call's { dw1 | dw2 } go
    Tensor dcw1, dcw2;
    const void* ckey = (const void*)(intptr_t)m[4];
    bool ca_shared = ckey && cw_dt == at::kFloat && deferring() && g_shared_ok.load(std::memory_order_relaxed);
    if (ca_shared) {
      queue_final_flush();
      bool have;
      { std::lock_guard<std::mutex> lk(g_pending_mu); have = g_ca_acc.count(ckey) > 0; }
      if (!have && (leaf_has_grad(ctx, "leaf_cw1") || leaf_has_grad(ctx, "leaf_cw2"))) ca_shared = false;
    }
    if (ca_shared) {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      auto it = g_ca_acc.find(ckey);
      if (it == g_ca_acc.end()) {
        dcw1 = at::empty({R, C, 1, 1}, x.options().dtype(at::kFloat));
        dcw2 = at::empty({C, R, 1, 1}, x.options().dtype(at::kFloat));
        // one row PER SAMPLE and call: the gate backward leaves the batch sum to the flush (dgtd_ca_gate_bwd_rows), B rows per call
        CaAcc acc{at::empty({(int64_t)CA_MAX_CALLS * B, 2 * R * C}, x.options().dtype(at::kFloat)), 0, (int)(R * C), dcw1.data_ptr(), dcw2.data_ptr(), B};
        it = g_ca_acc.emplace(ckey, acc).first;
      }
      if (it->second.count < CA_MAX_CALLS && it->second.rows == B) row = it->second.buf.data_ptr<float>() + (int64_t)(it->second.count++) * B * 2 * R * C;
      else ca_shared = false;                        // more calls than rows: this one goes through autograd's own accumulation
    }
    if (ca_shared)
      check(dgtd_ca_gate_bwd_rows(g.data_ptr(), res.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), stats.data_ptr<float>(), dres.data_ptr(), row,
                                  sp + 2 * R * C, B, (int)HW, C, (int)R, code(x), stream()), "dgtd_ca_gate_bwd_rows");
    else
      check(dgtd_ca_gate_bwd(g.data_ptr(), res.data_ptr(), w1f.data_ptr<float>(), w2f.data_ptr<float>(), stats.data_ptr<float>(), dres.data_ptr(), row,
                             row + R * C, sp + 2 * R * C, B, (int)HW, C, (int)R, code(x), stream()), "dgtd_ca_gate_bwd");
    if (!ca_shared) {
      dcw1 = small.narrow(0, 0, R * C).view({R, C, 1, 1});
      dcw2 = small.narrow(0, R * C, R * C).view({C, R, 1, 1});
      if (cw_dt != at::kFloat) { dcw1 = dcw1.to(cw_dt); dcw2 = dcw2.to(cw_dt); }
    }
    // (2) slope accumulator: the ONE PReLU of the decoder (cod.py:686) is shared by every CAB - one fp32 scalar per step (shared deferral),
    //     or a private one for this call
    bool share = akey && deferring() && g_shared_ok.load(std::memory_order_relaxed);
    if (share) {
      bool have;
      queue_final_flush();
      { std::lock_guard<std::mutex> lk(g_pending_mu); have = g_prelu_acc.count(akey) > 0; }
      if (!have && leaf_has_grad(ctx, "leaf_a")) share = false;
    }
    Tensor acc, da;
    if (share) {
      std::lock_guard<std::mutex> lk(g_pending_mu);
      auto it = g_prelu_acc.find(akey);
      if (it == g_prelu_acc.end()) {
        acc = at::zeros({1}, x.options().dtype(at::kFloat));
        da = at::empty({1}, x.options().dtype(a_dt));
        g_prelu_acc.emplace(akey, acc);
        g_pending.push_back(PendingReduce{dgtd_reduce_entry{acc.data_ptr<float>(), 1, 1, nullptr, 0, da.data_ptr(), (int32_t)code(da), 0, 0, nullptr}, acc});
      } else {
        acc = it->second;             // later calls add into the same scalar and hand autograd nothing
      }
    } else {
      acc = at::zeros({1}, x.options().dtype(at::kFloat));
    }
    // (3) input gradient of the second convolution, taken through the PReLU in its epilogue (dt0 = gradient w.r.t. the pre-activation)
    Tensor dt0 = at::empty_like(x);
    Tensor wt1 = flipped_weight(w1, 1, C, C);
    check(dgtd_conv3x3_fwd_ex(dres.data_ptr(), nullptr, wt1.data_ptr(), nullptr, dt0.data_ptr(), nullptr, nullptr, t0.data_ptr(), a32.data_ptr<float>(),
                              acc.data_ptr<float>(), 1, B, H, W, C, C, 3, 0, code(x), stream()), "dgtd_conv3x3_fwd_ex (input gradient through PReLU)");
    if (!share) da = a_dt == at::kFloat ? acc : acc.to(a_dt);
    // (4) input gradient of the first convolution + the skip gradient
    Tensor dx;
    if (m[3]) {
      dx = at::empty_like(x);
      Tensor wt0 = flipped_weight(w0, 1, C, C);
      check(dgtd_conv3x3_fwd_ex(dt0.data_ptr(), nullptr, wt0.data_ptr(), nullptr, dx.data_ptr(), nullptr, g.data_ptr(), nullptr, nullptr, nullptr, 1, B, H, W,
                                C, C, 0, 0, code(x), stream()), "dgtd_conv3x3_fwd_ex (input gradient + skip)");
    }
    // (5) weight gradients
    const ConvKey key{B, H, W, C, C, (int)code(x)};
    auto wgrad = [&](const char* dkey, const char* lkey, const Tensor& xin, const Tensor& dy, const Tensor& w) {
      Tensor dw, db;
      const bool parked = ctx->saved_data[dkey].toBool() && deferring() && (conv_dest_exists(w) || !leaf_has_grad(ctx, lkey)) &&
                          conv_park(xin, dy, Tensor(), w, key, false, dw, db);
      if (!parked) {
        dw = at::empty_like(w);
        Tensor ws = at::empty({dgtd_conv3x3_wgrad_workspace(1, B, H, W, C, C)}, x.options().dtype(at::kByte));
        check(dgtd_conv3x3_wgrad(xin.data_ptr(), dy.data_ptr(), nullptr, dw.data_ptr(), nullptr, ws.data_ptr(), 1, B, H, W, C, C, 0, code(x), stream()),
              "dgtd_conv3x3_wgrad");
      }
      return dw;
    };
    Tensor dw1 = wgrad("defer1", "leaf_w1", t1, dres, w1);
    Tensor dw0 = wgrad("defer0", "leaf_w0", x, dt0, w0);
    return {dx, dw0, dw1, da, dcw1, dcw2};
  }
};

Tensor layer_norm(const Tensor& x, const Tensor& w, const Tensor& b, double eps) { return LayerNormFn::apply(x, w, b, eps); }
std::tuple<Tensor, Tensor> layer_norm_fork(const Tensor& x, const Tensor& w, const Tensor& b, double eps) {
  auto r = LayerNormForkFn::apply(x, w, b, eps);
  return {r[0], r[1]};
}
Tensor sra_attention(const Tensor& q, const Tensor& kv, int64_t heads, double scale) { return SraAttnFn::apply(q, kv, heads, scale); }
Tensor dwconv_nhwc(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, bool gelu) { return DwConvFn::apply(x, w, b, gelu); }
std::tuple<Tensor, Tensor> dwconv_fork(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b) {
  auto r = DwConvForkFn::apply(x, w, b);
  return {r[0], r[1]};
}
Tensor scale_residual(const Tensor& x, const Tensor& y, const c10::optional<Tensor>& s, const c10::optional<Tensor>& gamma) {
  return ScaleResidualFn::apply(x, y, s, gamma);
}
Tensor linear(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, int64_t dt_code) { return LinearFn::apply(x, w, b, dt_code); }
Tensor linear_gelu(const Tensor& x, const Tensor& w, const Tensor& b, int64_t dt_code) { return LinearGeluFn::apply(x, w, b, dt_code); }
Tensor linear_residual(const Tensor& h, const Tensor& w, const Tensor& b, const Tensor& x, const c10::optional<Tensor>& s,
                       const c10::optional<Tensor>& gamma, int64_t dt_code) {
  return LinearResidualFn::apply(h, w, b, x, s, gamma, dt_code);
}
Tensor mlp_residual(const Tensor& v, const Tensor& w1, const Tensor& b1, const Tensor& w2, const Tensor& b2, const Tensor& x,
                    const c10::optional<Tensor>& s, const c10::optional<Tensor>& gamma, int64_t dt_code) {
  return MlpResidualFn::apply(v, w1, b1, w2, b2, x, s, gamma, dt_code);
}
bool gemm_ok(int64_t M, int64_t N, int64_t K, int64_t dt_code) { return own_ok(from_code(dt_code), M, N, K); }
Tensor cab(const Tensor& x, const Tensor& w0, const Tensor& w1, const Tensor& a, const Tensor& cw1, const Tensor& cw2) {
  return CabFn::apply(x, w0, w1, a, cw1, cw2);
}
Tensor conv3x3(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, bool relu) { return Conv3x3Fn::apply(x, w, b, relu); }
Tensor conv3x3_cl(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& b, bool relu) { return Conv3x3ClFn::apply(x, w, b, relu); }
Tensor prelu(const Tensor& x, const Tensor& a) { return PReLUFn::apply(x, a); }
Tensor ca_gate(const Tensor& res, const Tensor& x, const Tensor& w1, const Tensor& w2) { return CAGateFn::apply(res, x, w1, w2); }
Tensor bilinear_resize(const Tensor& x, int64_t oh, int64_t ow, bool align) { return BilinearFn::apply(x, oh, ow, align); }
Tensor sam(const Tensor& xh, const Tensor& xl, const Tensor& w1, const Tensor& w2, const Tensor& v1, const Tensor& v2) {
  return SamFn::apply(xh, xl, w1, w2, v1, v2);
}
Tensor batch_norm(const Tensor& x, const Tensor& gamma, const Tensor& beta, Tensor running_mean, Tensor running_var, Tensor num_batches, bool training,
                  double momentum, double eps) {
  return BatchNormFn::apply(x, gamma, beta, running_mean, running_var, num_batches, training, momentum, eps);
}

}  // namespace

TORCH_LIBRARY(dgtd, m) {
  m.def("layer_norm(Tensor x, Tensor weight, Tensor bias, float eps) -> Tensor", &layer_norm);
  m.def("layer_norm_fork(Tensor(a) x, Tensor weight, Tensor bias, float eps) -> (Tensor, Tensor(a))", &layer_norm_fork);
  m.def("sra_attention(Tensor q, Tensor kv, int heads, float scale) -> Tensor", &sra_attention);
  m.def("dwconv_nhwc(Tensor x, Tensor weight, Tensor? bias, bool gelu) -> Tensor", &dwconv_nhwc);
  m.def("dwconv_fork(Tensor(a) x, Tensor weight, Tensor? bias) -> (Tensor, Tensor(a))", &dwconv_fork);
  m.def("scale_residual(Tensor x, Tensor y, Tensor? s, Tensor? gamma) -> Tensor", &scale_residual);
  m.def("linear(Tensor x, Tensor weight, Tensor? bias, int dtype_code) -> Tensor", &linear);
  m.def("linear_gelu(Tensor x, Tensor weight, Tensor bias, int dtype_code) -> Tensor", &linear_gelu);
  m.def("linear_residual(Tensor h, Tensor weight, Tensor bias, Tensor x, Tensor? s, Tensor? gamma, int dtype_code) -> Tensor", &linear_residual);
  m.def("mlp_residual(Tensor v, Tensor w1, Tensor b1, Tensor w2, Tensor b2, Tensor x, Tensor? s, Tensor? gamma, int dtype_code) -> Tensor", &mlp_residual);
  m.def("gemm_ok(int M, int N, int K, int dtype_code) -> bool", &gemm_ok);
  m.def("cab(Tensor x, Tensor w0, Tensor w1, Tensor a, Tensor cw1, Tensor cw2) -> Tensor", &cab);
  m.def("conv3x3(Tensor x, Tensor weight, Tensor? bias, bool relu) -> Tensor", &conv3x3);
  m.def("conv3x3_cl(Tensor x, Tensor weight, Tensor? bias, bool relu) -> Tensor", &conv3x3_cl);
  m.def("prelu(Tensor x, Tensor a) -> Tensor", &prelu);
  m.def("ca_gate(Tensor res, Tensor x, Tensor w1, Tensor w2) -> Tensor", &ca_gate);
  m.def("bilinear_resize(Tensor x, int oh, int ow, bool align) -> Tensor", &bilinear_resize);
  m.def("sam(Tensor xh, Tensor xl, Tensor w1, Tensor w2, Tensor v1, Tensor v2) -> Tensor", &sam);
  m.def("batch_norm(Tensor x, Tensor weight, Tensor bias, Tensor(a!) running_mean, Tensor(b!) running_var, Tensor(c!) num_batches, bool training, "
        "float momentum, float eps) -> Tensor", &batch_norm);
  m.def("set_deferred(bool on) -> ()", &set_deferred);
  m.def("flush_deferred() -> ()", &flush_deferred);
  m.def("flush_deferred_async() -> ()", &flush_deferred_async);
  m.def("pending_reductions() -> int", &pending_reductions);
  m.def("flushed_reductions() -> int", &flushed_reductions);
  m.def("set_shared_deferral(bool on) -> ()", &set_shared_deferral);
  m.def("arena_hint(int group, int idx, int count) -> ()", &arena_hint);
  m.def("arena_roles(int out, int grad_a, int grad_b) -> ()", &arena_roles);
  m.def("arena_release(int group) -> ()", &arena_release);
  m.def("arena_bytes() -> int", &arena_bytes);
  m.def("arena_pin(int group, bool on) -> ()", &arena_pin);
}

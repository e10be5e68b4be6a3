// gemm.hip — the Linear layers of both trunks as ONE hand-written MFMA GEMM with this model's epilogues (gfx950, wave = 64).
//
//   D[M,N] = A[M,K] · B[N,K]^T   (both operands K-contiguous: nn.Linear's x [tokens, in] and weight [out, in])
//
// replaces the library GEMM + the elementwise passes around it:
//   * bias                                  q / kv / fc1 / sr / 1x1 convs                      twig/model/cod.py:900-921, :852
//   * bias + exact-erf GELU, writing BOTH the pre-activation (saved for the backward) and h    convnext_Block pwconv1 + act, :1097-1098
//   * bias + x + s[b]·gamma[c]·y            pwconv2 + layer scale + DropPath + residual        :1099-1116; attn.proj / Mlp.fc2 + residual, :958-959
//   * ⊙ gelu'(pre) + bias-gradient column partials  (input gradient of pwconv2 → gradient of the pre-activation)   grad of :1097-1099
// The input-gradient GEMMs  dX[M,K] = dY[M,N] · W[N,K]  run through the same kernel on a TRANSPOSED copy of the weight (W^T [K,N] is
// K-contiguous in the reduction dim N): weights change once per step, the copies are refreshed by ONE batched launch per step
// (dgtd_transpose_batched), so one kernel shape serves forward and backward.  Weight gradients (dY^T X, reduction over tokens) stay
// with the batched library GEMM of the deferred phase (csrc_torch/bindings.cpp).
//
// Kernel: 128 x BN x 64 tiles (BN = 128 or 64), 4 waves, wave tile 64x64 / 32x64 of v_mfma_f32_32x32x16_{bf16,f16}; operands staged
// global -> LDS with 16-byte LDS-DMA (global_load_lds_dwordx4, no VGPR round trip), two LDS stages, the next tile in flight across
// the barrier (counted vmcnt, raw s_barrier); the LDS image is the lane-linear DMA image with the 16-byte chunk index XOR-swizzled by
// (row >> 1) & 7 ON THE GLOBAL SOURCE ADDRESS, which makes every ds_read_b128 fragment read conflict-free (128-byte rows: two rows per
// 256-byte bank row).  MFMA orientation D^T = B·A^T so a lane owns ONE output row and 4-column runs; the fp32 accumulators go through
// LDS once and every global access of the epilogue (outputs, residual, saved pre-activation) is a coalesced 16-byte row-major access,
// all epilogue arithmetic in fp32 with ONE rounding per output.  blockIdx -> tile mapping is XCD-aware (the column tiles of one row
// panel of A run on one XCD's L2).  MFMA-bound above K ~ 512, HBM-bound below: 2·M·N·K flop over (M·K + N·K + n_out·M·N + ...)·e bytes.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BK = 64;
enum { EPI_BIAS = 0, EPI_GELU = 1, EPI_RESIDUAL = 2, EPI_GELU_BWD = 3 };

struct GemmArgs {
  const void* A;          // [M,K]
  const void* B;          // [N,K]
  const void* bias;       // [N] in T or NULL
  void* D;                // EPI_BIAS: out; EPI_GELU: pre-activation (NULL: not stored); EPI_RESIDUAL: out; EPI_GELU_BWD: dpre
  void* D2;               // EPI_GELU: h = gelu(pre); EPI_RESIDUAL: y = acc + bias (NULL: not stored)
  const void* X;          // EPI_RESIDUAL: residual x [M,N]; EPI_GELU_BWD: saved pre-activation [M,N]
  const float* s;         // EPI_RESIDUAL: per-sample DropPath scale [M / rows_per_sample] or NULL
  const float* gamma;     // EPI_RESIDUAL: layer scale [N] or NULL
  float* colsum;          // EPI_GELU_BWD: per-row-tile column partials [M/128][N] or NULL
  int M, N, K, tiles_n;
  int64_t rows_per_sample;
  int dbg;                // tools only (DGTD_GEMM_DBG): 1 = one k-step instead of K / 64, 2 = no global stores in the epilogue
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// one 16-byte LDS-DMA per lane: LDS destination = wave-uniform base + lane * 16
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((glb_void_ptr)gsrc, (lds_void_ptr)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NSTAGE = 2: the next tile's DMA is issued at the top of a step, two barriers per step, 64 KB of LDS (two workgroups per CU cover each
// other's stalls).  NSTAGE = 3: the DMA of tile kt + 2 is issued in slices BEHIND the MFMA groups of step kt (its issue slots hide
// under the matrix pipe), two tiles in flight, ONE barrier per step, 96 KB of LDS.
template <typename T, int BN, int EPI, int NSTAGE>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const GemmArgs g) {
  typedef typename Vec16<T>::type V8;
  constexpr int WM = BN == 128 ? 64 : 32;                  // wave tile rows: 2x2 waves of 64x64, or 4x1 waves of 32x64
  constexpr int MI = WM / 32, NI = 2;                       // 32x32 MFMA tiles per wave
  constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGE = A_BYTES + B_BYTES;
  constexpr int SROW = BN * 4 + 16;                         // fp32 staging row stride in bytes (+16: conflict-free 16-byte column writes)
  constexpr int LDS_BYTES = NSTAGE * STAGE > BM * SROW ? NSTAGE * STAGE : BM * SROW;
  constexpr int A_PER_WAVE = (BM / 8) / 4, B_PER_WAVE = (BN / 8) / 4;   // 1-KiB DMA pieces (8 rows x 128 B) per wave and stage
  __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];       // the ONLY LDS object (a second one de-pipelines the DMA waits)

  // XCD-aware tile order: blocks b and b + 8 share an XCD (its L2); give every XCD a contiguous run of tiles, column tiles fastest,
  // so the BN-wide tiles of one 128-row panel of A are served from one L2 (bijective for any grid size)
  const int nblk = gridDim.x, bid = blockIdx.x;
  const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7;
  const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
  const int tm = t / g.tiles_n, tn = t - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform: LDS-DMA bases become scalar arithmetic
  const int K = g.K;

  const T* Ag = (const T*)g.A + (size_t)m0 * K;
  const T* Bg = (const T*)g.B + (size_t)n0 * K;
  // per-lane source offsets of the DMA pieces (elements, without the k offset): row-in-piece = lane / 8, swizzled chunk = lane % 8
  int a_off[A_PER_WAVE], b_off[B_PER_WAVE];
#pragma unroll
  for (int p = 0; p < A_PER_WAVE; ++p) {
    const int row = (wave * A_PER_WAVE + p) * 8 + (lane >> 3);
    a_off[p] = row * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
#pragma unroll
  for (int p = 0; p < B_PER_WAVE; ++p) {
    const int row = (wave * B_PER_WAVE + p) * 8 + (lane >> 3);
    b_off[p] = row * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
  }
  constexpr int PIECES = A_PER_WAVE + B_PER_WAVE;
  // pieces [p0, p1) of this wave's share of tile kt -> stage st (A pieces first); the k offset rides on the UNIFORM base pointer
  auto stage_part = [&](int kt, int st, int p0, int p1) {
    char* sa = lds + st * STAGE;
    char* sb = sa + A_BYTES;
    const T* Ak = Ag + kt * BK;
    const T* Bk = Bg + kt * BK;
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      if (p < p0 || p >= p1) continue;
      if (p < A_PER_WAVE) glds16(Ak + a_off[p], sa + (wave * A_PER_WAVE + p) * 1024);
      else glds16(Bk + b_off[p - A_PER_WAVE], sb + (wave * B_PER_WAVE + (p - A_PER_WAVE)) * 1024);
    }
  };
  auto stage = [&](int kt, int st) { stage_part(kt, st, 0, PIECES); };

  const int wm = BN == 128 ? (wave >> 1) : wave, wn = BN == 128 ? (wave & 1) : 0;
  const int r = lane & 31, h = lane >> 5;
  // fragment read offsets inside a stage: row * 128 + ((2 kk + h) ^ ((row >> 1) & 7)) * 16
  int a_row[MI], a_sw[MI], b_row[NI], b_sw[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) { const int row = wm * WM + mi * 32 + r; a_row[mi] = row * 128; a_sw[mi] = (row >> 1) & 7; }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) { const int row = wn * 64 + ni * 32 + r; b_row[ni] = row * 128; b_sw[ni] = (row >> 1) & 7; }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  const int nk = (g.dbg & 1) ? 1 : K / BK;
  // one k-slice group: fragments double-buffered in registers (the reads of slice kk + 1 fly while the MFMAs of slice kk issue)
  V8 af[2][MI], bf[2][NI];
  auto frags = [&](const char* sa, const char* sb, int kk, int set) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) af[set][mi] = *reinterpret_cast<const V8*>(sa + a_row[mi] + (((2 * kk + h) ^ a_sw[mi]) << 4));
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bf[set][ni] = *reinterpret_cast<const V8*>(sb + b_row[ni] + (((2 * kk + h) ^ b_sw[ni]) << 4));
  };
  auto mfmas = [&](int set) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma16(bf[set][ni], af[set][mi], acc[mi][ni]);   // D^T tile: lane & 31 = output ROW
    __builtin_amdgcn_s_setprio(0);
  };
  if (NSTAGE == 2) {
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
      const int st = kt & 1;
      if (kt + 1 < nk) {
        stage(kt + 1, st ^ 1);                             // its buffer was last read before the closing barrier of step kt - 1
        wait_vmcnt<PIECES>();                              // all but the pieces just issued: tile kt has landed (this wave's share)
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();                        // ... and everybody else's share
      const char* sa = lds + st * STAGE;
      const char* sb = sa + A_BYTES;
      frags(sa, sb, 0, 0);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        if (kk + 1 < BK / 16) frags(sa, sb, kk + 1, (kk + 1) & 1);
        mfmas(kk & 1);
      }
      __builtin_amdgcn_s_barrier();                        // every wave is done reading stage st before step kt + 1 refills it
    }
  } else {
    // prologue: tiles 0 and 1 in flight
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    int st = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<PIECES>(); else wait_vmcnt<0>();   // tile kt landed (this wave's share); tile kt + 1 may still fly
      __builtin_amdgcn_s_barrier();                        // everybody's share landed AND everybody finished reading step kt - 1's stage,
                                                           // which is the stage tile kt + 2 is about to overwrite
      const char* sa = lds + st * STAGE;
      const char* sb = sa + A_BYTES;
      const int st2 = st == 0 ? 2 : st - 1;                // (kt + 2) % 3
      const bool more = kt + 2 < nk;
      frags(sa, sb, 0, 0);
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        if (kk + 1 < BK / 16) frags(sa, sb, kk + 1, (kk + 1) & 1);
        mfmas(kk & 1);
        // a quarter of the DMA of tile kt + 2, issued behind this group's MFMAs
        if (more) stage_part(kt + 2, st2, (PIECES * kk) / (BK / 16), (PIECES * (kk + 1)) / (BK / 16));
      }
      st = st == 2 ? 0 : st + 1;
    }
    __builtin_amdgcn_s_barrier();                          // all fragment reads done before the staging tile overwrites the stages
  }

  // the epilogue's tensor operand (residual x / saved pre-activation): every 16-byte chunk this thread will need is requested NOW, so
  // the HBM latency runs under the accumulator staging instead of once per row pass
  constexpr int CPR = BN / 8, RPP = 256 / CPR, NPASS = BM / RPP;   // chunks per row, rows per pass, passes
  const int cc = tid % CPR, r0 = tid / CPR;
  const int ncol = n0 + cc * 8;
  V8 xin[NPASS];
  if (EPI == EPI_RESIDUAL || EPI == EPI_GELU_BWD) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) xin[i] = *reinterpret_cast<const V8*>((const T*)g.X + (size_t)(m0 + r0 + i * RPP) * g.N + ncol);
  }

  // ---- accumulators -> fp32 staging tile [BM][BN] in LDS (the operand stages are dead after the closing barrier)
  // lane (r, h) holds, for output row m = r of its 32-row block, columns 8 q + 4 h + {0..3} (registers 4q .. 4q+3)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      char* base = lds + (wm * WM + mi * 32 + r) * SROW + (wn * 64 + ni * 32 + 4 * h) * 4;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * qd + e];
        *reinterpret_cast<f32x4*>(base + qd * 32) = v;
      }
    }
  __syncthreads();

  // ---- row-major epilogue: thread = one 8-column chunk, rows tid / CPR + i * RPP
  float bias[8], gam[8], csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { bias[e] = 0.f; gam[e] = 1.f; csum[e] = 0.f; }
  if (g.bias) {
    const V8 bv = *reinterpret_cast<const V8*>((const T*)g.bias + ncol);
#pragma unroll
    for (int e = 0; e < 8; ++e) bias[e] = (float)bv[e];
  }
  if (EPI == EPI_RESIDUAL && g.gamma) {
#pragma unroll
    for (int e = 0; e < 8; ++e) gam[e] = g.gamma[ncol + e];
  }
  const bool store = !(g.dbg & 2);
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int row = r0 + i * RPP;
    const char* sp = lds + row * SROW + cc * 32;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(sp), hi = *reinterpret_cast<const f32x4*>(sp + 16);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = lo[e] + bias[e]; v[4 + e] = hi[e] + bias[4 + e]; }
    const size_t o = (size_t)(m0 + row) * g.N + ncol;
    V8 out;
    if (EPI == EPI_BIAS) {
#pragma unroll
      for (int e = 0; e < 8; ++e) out[e] = (T)v[e];
      if (store) *reinterpret_cast<V8*>((T*)g.D + o) = out;
      else asm volatile("" ::"v"(out));
    } else if (EPI == EPI_GELU) {
      V8 pre;
#pragma unroll
      for (int e = 0; e < 8; ++e) { pre[e] = (T)v[e]; out[e] = (T)gelu_fast((float)pre[e]); }   // GELU of the STORED pre-activation: what the backward recomputes from
      if (g.D && store) *reinterpret_cast<V8*>((T*)g.D + o) = pre;
      if (store) *reinterpret_cast<V8*>((T*)g.D2 + o) = out;
      else asm volatile("" ::"v"(out), "v"(pre));
    } else if (EPI == EPI_RESIDUAL) {
      const V8 xv = xin[i];
      const float sc = g.s ? g.s[(m0 + row) / g.rows_per_sample] : 1.f;
      V8 y;
#pragma unroll
      for (int e = 0; e < 8; ++e) { y[e] = (T)v[e]; out[e] = (T)fmaf(sc * gam[e], (float)y[e], (float)xv[e]); }   // from the STORED y, like the backward
      if (g.D2 && store) *reinterpret_cast<V8*>((T*)g.D2 + o) = y;
      if (store) *reinterpret_cast<V8*>((T*)g.D + o) = out;
      else asm volatile("" ::"v"(out), "v"(y));
    } else {                                                 // EPI_GELU_BWD
      const V8 pv = xin[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        out[e] = (T)(v[e] * gelu_grad_fast((float)pv[e]));
        csum[e] += (float)out[e];                            // the bias gradient sums the gradient the weight-gradient GEMM will read
      }
      if (store) *reinterpret_cast<V8*>((T*)g.D + o) = out;
      else asm volatile("" ::"v"(out));
    }
  }
  if (EPI == EPI_GELU_BWD && g.colsum) {
    __syncthreads();                                         // staging tile fully consumed
    float* part = reinterpret_cast<float*>(lds);             // [RPP][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) part[r0 * BN + cc * 8 + e] = csum[e];
    __syncthreads();
    if (tid < BN) {
      float sum = 0.f;
#pragma unroll 4
      for (int j = 0; j < RPP; ++j) sum += part[j * BN + tid];
      g.colsum[(size_t)tm * g.N + n0 + tid] = sum;
    }
  }
}

// out[c][r] = in[r][c] for a batch of [rows][cols] 16-bit matrices (the once-per-step transposed weight copies): 64x64 tiles through LDS.
// rows % 8 == 0 and cols % 8 == 0 (every Linear of the model): 16-byte global loads and stores on both sides; the LDS tile is
// [source column][source row] with the 8-row chunk index XOR-ed by (column >> 3), which spreads the eight lanes that scatter one
// source row's chunks over eight banks (a plain 16-byte-aligned row stride puts them all on one).  Other shapes: 2-byte accesses.
struct TrTable { const void* src[64]; void* dst[64]; int rows[64], cols[64], first[65]; int count; };
__global__ __launch_bounds__(256) void transpose_batched_kernel(TrTable t) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];
  const int blk = blockIdx.x;
  int lo = 0, hi = t.count;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t.first[mid] <= blk) lo = mid; else hi = mid; }
  const int e = lo, local = blk - t.first[e];
  const int R = t.rows[e], C = t.cols[e], tc = (C + 63) / 64;
  const int tr0 = (local / tc) * 64, tc0 = (local % tc) * 64;
  const unsigned short* s = (const unsigned short*)t.src[e];
  unsigned short* d = (unsigned short*)t.dst[e];
  if (((R | C) & 7) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
    typedef unsigned short us8 __attribute__((ext_vector_type(8)));
    for (int i = threadIdx.x; i < 512; i += 256) {          // 64 source rows x 8 chunks of 8 columns
      const int r = i >> 3, ch = i & 7;
      if (tr0 + r < R && tc0 + ch * 8 < C) {
        const us8 v = *reinterpret_cast<const us8*>(s + (size_t)(tr0 + r) * C + tc0 + ch * 8);
        const int pos = r ^ (ch << 3);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[ch * 8 + j][pos] = v[j];
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256) {          // 64 destination rows (source columns) x 8 chunks of 8 source rows
      const int c = i >> 3, q = i & 7;
      if (tc0 + c < C && tr0 + q * 8 < R) {
        const us8 v = *reinterpret_cast<const us8*>(&tile[c][(q ^ (c >> 3)) * 8]);
        *reinterpret_cast<us8*>(d + (size_t)(tc0 + c) * R + tr0 + q * 8) = v;
      }
    }
    return;
  }
  const int x = threadIdx.x & 63, y0 = threadIdx.x >> 6;
  for (int y = y0; y < 64; y += 4)
    if (tr0 + y < R && tc0 + x < C) tile[y][x] = s[(size_t)(tr0 + y) * C + tc0 + x];
  __syncthreads();
  for (int y = y0; y < 64; y += 4)
    if (tc0 + y < C && tr0 + x < R) d[(size_t)(tc0 + y) * R + tr0 + x] = tile[x][y];
}

// A/B knobs (read once): DGTD_GEMM_STAGES = 2 | 3 (pipeline form), DGTD_GEMM_WIDE_MIN = fewest 128-wide tiles for which the wide tile is taken
static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
template <typename T, int EPI>
int launch(const GemmArgs& a0, hipStream_t st) {
  static const int force_stages = env_int("DGTD_GEMM_STAGES", 0), wide_min = env_int("DGTD_GEMM_WIDE_MIN", 192), dbg = env_int("DGTD_GEMM_DBG", 0);
  GemmArgs a = a0;
  a.dbg = dbg;
  const int tiles_m = a.M / BM;
  // 128-wide column tiles when they alone fill the chip, 64-wide otherwise (N not a multiple of 128, or too few tiles)
  const bool wide = a.N % 128 == 0 && (int64_t)tiles_m * (a.N / 128) >= wide_min;
  a.tiles_n = a.N / (wide ? 128 : 64);
  // measured (tools/bench_gemm_own.py, profiles/r03_gemm_variants.txt): with at most one wide tile per CU no second workgroup covers a
  // workgroup's stalls, and the deeper 3-stage pipeline wins (8192x512x2048: 22.6 vs 25.3 us); with more tiles two co-resident
  // 2-stage workgroups per CU do (4096^3: 893 vs 761 TF/s)
  const int stages = force_stages ? force_stages : ((wide && tiles_m * a.tiles_n <= 256) ? 3 : 2);
  const dim3 grid(tiles_m * a.tiles_n), block(256);
  if (wide && stages == 3) hipLaunchKernelGGL((gemm_tn_kernel<T, 128, EPI, 3>), grid, block, 0, st, a);
  else if (wide) hipLaunchKernelGGL((gemm_tn_kernel<T, 128, EPI, 2>), grid, block, 0, st, a);
  else if (stages == 3) hipLaunchKernelGGL((gemm_tn_kernel<T, 64, EPI, 3>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_tn_kernel<T, 64, EPI, 2>), grid, block, 0, st, a);
  DGTD_CHECK_LAUNCH("gemm_tn");
  return 0;
}

template <int EPI>
int dispatch(const GemmArgs& a, dgtd_dtype dt, hipStream_t st) {
  if (dt == DGTD_F16) return launch<f16_t, EPI>(a, st);
  return launch<bf16_t, EPI>(a, st);
}

int check_common(const void* a, const void* b, const void* d, int M, int N, int K, dgtd_dtype dt, const char* name) {
  DGTD_REQUIRE(a && b && d, "%s: null operand", name);
  DGTD_REQUIRE(DGTD_IS_HALF(dt), "%s: bf16 / fp16 only (dtype %d)", name, (int)dt);
  DGTD_REQUIRE(dgtd_gemm_supported(M, N, K, dt), "%s: unsupported shape M=%d N=%d K=%d (M %% 128, N %% 64, K %% 64 must be 0)", name, M, N, K);
  DGTD_REQUIRE(((uintptr_t)a | (uintptr_t)b | (uintptr_t)d) % 16 == 0, "%s: operands must be 16-byte aligned", name);
  return 0;
}

}  // namespace

extern "C" int dgtd_gemm_supported(int M, int N, int K, dgtd_dtype dt) {
  return DGTD_IS_HALF(dt) && M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 64 == 0 && K % 64 == 0 &&
         (int64_t)M * K < (1ll << 31) && (int64_t)N * K < (1ll << 31) ? 1 : 0;
}

// roofline label of a call: 2 M N K flop over e (M K + N K + maps M N) bytes (maps = the [M,N] tensors the epilogue reads or writes); above the
// ridge (2.5 PF / 8 TB/s = 312 flop/B) the call is priced against the MFMA peak, below it against HBM.  With K = 512 and two [M,N] maps
// (the ConvNeXt / Mlp expansions) the intensity is 2 K / (2 e) = 256 flop/B: those calls are HBM-bound by their activation traffic.
#define GEMM_PROF(s, maps, ...)                                                                                        \
  const double prof_flops_ = 2.0 * M * N * K, prof_bytes_ = 2.0 * ((double)M * K + (double)N * K + (double)(maps) * M * N); \
  const bool prof_mfma_ = prof_flops_ > 312.5 * prof_bytes_;                                                           \
  DGTD_PROF(s, prof_mfma_ ? DGTD_MFMA : DGTD_HBM, prof_mfma_ ? prof_flops_ : prof_bytes_, __VA_ARGS__)

extern "C" int dgtd_gemm_bias(const void* a, const void* b, const void* bias, void* d, int M, int N, int K, dgtd_dtype dt, dgtd_stream s) {
  GEMM_PROF(s, 1, "dgtd_gemm_bias[M=%d,N=%d,K=%d]", M, N, K);
  if (int rc = check_common(a, b, d, M, N, K, dt, "gemm_bias")) return rc;
  GemmArgs g{a, b, bias, d, nullptr, nullptr, nullptr, nullptr, nullptr, M, N, K, 0, 1};
  return dispatch<EPI_BIAS>(g, dt, (hipStream_t)s);
}

extern "C" int dgtd_gemm_bias_gelu(const void* a, const void* b, const void* bias, void* pre, void* h, int M, int N, int K, dgtd_dtype dt,
                                   dgtd_stream s) {
  GEMM_PROF(s, pre ? 2 : 1, "dgtd_gemm_bias_gelu[M=%d,N=%d,K=%d]", M, N, K);
  if (int rc = check_common(a, b, h, M, N, K, dt, "gemm_bias_gelu")) return rc;
  DGTD_REQUIRE(!pre || (uintptr_t)pre % 16 == 0, "gemm_bias_gelu: pre must be 16-byte aligned");
  GemmArgs g{a, b, bias, pre, h, nullptr, nullptr, nullptr, nullptr, M, N, K, 0, 1};
  return dispatch<EPI_GELU>(g, dt, (hipStream_t)s);
}

extern "C" int dgtd_gemm_bias_residual(const void* a, const void* b, const void* bias, const void* x, const float* scale, const float* gamma,
                                       void* y, void* out, int M, int N, int K, int64_t rows_per_sample, dgtd_dtype dt, dgtd_stream s) {
  GEMM_PROF(s, y ? 3 : 2, "dgtd_gemm_bias_residual[M=%d,N=%d,K=%d]", M, N, K);
  if (int rc = check_common(a, b, out, M, N, K, dt, "gemm_bias_residual")) return rc;
  DGTD_REQUIRE(x && (uintptr_t)x % 16 == 0 && (!y || (uintptr_t)y % 16 == 0), "gemm_bias_residual: x / y must be 16-byte aligned");
  DGTD_REQUIRE(!scale || (rows_per_sample > 0 && M % rows_per_sample == 0), "gemm_bias_residual: bad rows_per_sample");
  GemmArgs g{a, b, bias, out, y, x, scale, gamma, nullptr, M, N, K, 0, scale ? rows_per_sample : 1};
  return dispatch<EPI_RESIDUAL>(g, dt, (hipStream_t)s);
}

extern "C" int64_t dgtd_gemm_gelu_bwd_workspace(int M, int N) { return (int64_t)(M / BM) * N * 4; }

extern "C" int dgtd_gemm_gelu_bwd(const void* dy, const void* w_t, const void* pre, void* dpre, void* colsum_ws, int* nblocks, int M, int N,
                                  int K, dgtd_dtype dt, dgtd_stream s) {
  GEMM_PROF(s, 2, "dgtd_gemm_gelu_bwd[M=%d,N=%d,K=%d]", M, N, K);
  if (int rc = check_common(dy, w_t, dpre, M, N, K, dt, "gemm_gelu_bwd")) return rc;
  DGTD_REQUIRE(pre && (uintptr_t)pre % 16 == 0, "gemm_gelu_bwd: pre must be 16-byte aligned");
  GemmArgs g{dy, w_t, nullptr, dpre, nullptr, pre, nullptr, nullptr, (float*)colsum_ws, M, N, K, 0, 1};
  if (nblocks) *nblocks = M / BM;
  return dispatch<EPI_GELU_BWD>(g, dt, (hipStream_t)s);
}

extern "C" int dgtd_transpose_batched(const void* const* src, void* const* dst, const int* rows, const int* cols, int n, dgtd_dtype dt,
                                      dgtd_stream s) {
  DGTD_REQUIRE(n >= 0 && (n == 0 || (src && dst && rows && cols)) && DGTD_IS_HALF(dt), "transpose_batched: bad arguments");
  double bytes = 0;
  for (int i = 0; i < n; ++i) bytes += 4.0 * rows[i] * cols[i];
  DGTD_PROF(s, DGTD_HBM, bytes, "dgtd_transpose_batched[n=%d]", n);
  for (int b0 = 0; b0 < n; b0 += 64) {
    TrTable t;
    const int m = n - b0 < 64 ? n - b0 : 64;
    t.count = m;
    int blocks = 0;
    for (int i = 0; i < m; ++i) {
      DGTD_REQUIRE(src[b0 + i] && dst[b0 + i] && rows[b0 + i] > 0 && cols[b0 + i] > 0, "transpose_batched: bad entry %d", b0 + i);
      t.src[i] = src[b0 + i]; t.dst[i] = dst[b0 + i]; t.rows[i] = rows[b0 + i]; t.cols[i] = cols[b0 + i];
      t.first[i] = blocks;
      blocks += ((rows[b0 + i] + 63) / 64) * ((cols[b0 + i] + 63) / 64);
    }
    for (int i = m; i <= 64; ++i) t.first[i] = blocks;
    hipLaunchKernelGGL(transpose_batched_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, t);
    DGTD_CHECK_LAUNCH("transpose_batched");
  }
  return 0;
}

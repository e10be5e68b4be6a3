// attn_bwd.hip — backward of the fused spatial-reduction attention (dQ, dK, dV), head_dim 64.
// Gradient of twig/model/cod.py:913-917 with P recomputed from Q, K and the forward's log-sum-exp.
//
// Structure (key-stationary; K/V are short): a workgroup owns one (batch, head) and a chunk
// of 32-row query tiles; wave w owns keys [32w, 32w+32) of the current 256-key slice and keeps dK^T/dV^T of
// those keys in accumulator registers while the workgroup sweeps the query tiles.  Per query tile every
// wave computes its S / dP tiles in BOTH orientations (rows = query and rows = key) so that each of the
// three gradient products sums over the accumulator ROW index and can take the accumulator straight back
// as an MFMA operand (CDNA4 guide §3): no score tile is ever transposed through memory.
//   dV[key][d] += sum_q P[q][key]  dO[q][d]      (rows = q tile as the A operand, Z = X^T B)
//   dK[key][d] += sum_q dS[q][key] Q[q][d]
//   dQ^T[d][q] += sum_k K[k][d]   dS^T[k][q]     (rows = key tile as the B operand, Y = A X)
// dQ partials of the waves are summed with LDS atomics; dK/dV leave the workgroup as fp32 global atomics
// (one 128-B segment per half-wave, the full-rate shape) into a caller-zeroed fp32 buffer.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr int SLICE = 256;  // keys per pass over the query chunk (8 waves x 32 keys)

// delta[b][h][n] = sum_d dO[b][n][h*64+d] * O[b][n][h*64+d]
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* __restrict__ o, const T* __restrict__ dout,
                                                         float* __restrict__ delta, int B, int N, int heads) {
  typedef typename Vec16<T>::type VT;
  constexpr int V = Vec16<T>::N, LPR = 64 / V;  // lanes per (row, head)
  const int64_t total = (int64_t)B * N * heads;
  const int64_t gid = ((int64_t)blockIdx.x * 256 + threadIdx.x);
  const int64_t item = gid / LPR;
  const int sub = (int)(gid % LPR);
  float s = 0.f;
  int64_t bn = 0; int hd = 0;
  if (item < total) {
    bn = item / heads; hd = (int)(item % heads);
    const size_t off = ((size_t)bn * heads + hd) * 64 + sub * V;
    VT a = *reinterpret_cast<const VT*>(o + off);
    VT g = *reinterpret_cast<const VT*>(dout + off);
#pragma unroll
    for (int j = 0; j < V; ++j) s += (float)a[j] * (float)g[j];
  }
  s = group_sum(s, LPR);
  if (item < total && sub == 0) {
    const int64_t b = bn / N, n = bn % N;
    delta[((size_t)b * heads + hd) * N + n] = s;
  }
}

// ------------------------------------------------------------------------------------------ bf16
// Two main launches (plus the tiny delta kernel), no LDS atomics, no transposed copies anywhere:
//   dK/dV : workgroup = up to 8 waves = 256 keys of one (batch, head) and a chunk of query tiles.  Wave w keeps the K/V row
//           fragments of its 32 keys in registers and dK^T/dV^T in accumulators.  Each 32-query Q/dO tile is staged ONCE per
//           workgroup into LDS (coalesced 16-B copies, double buffered: one barrier per tile; the next tile's global loads are
//           in flight during the MFMAs).  Row fragments come from ds_read_b128, column fragments (query on the k axis) from the
//           transposing read ds_read_b64_tr_b16 of the same row-major tile.  P and dS (rows = query) go back into the MFMA as A
//           operands (Z = X^T B).
//   dQ    : the forward kernel's shape: K and V staged once in LDS (row-major), lane = query, S^T/dP^T with rows = key, dS^T
//           fed back as the B operand and K^T fragments taken with ds_read_b64_tr_b16 (Y = A X).  lse and delta are lane-local.
template <typename T>
struct DkdvSmem {
  static constexpr int QS = 72;
  T q[2][32][QS];
  T g[2][32][QS];
  float lse2[2][32];
  float dl[2][32];
};

constexpr int KGROUP = 128;   // keys per dK/dV workgroup (4 waves x 32)
constexpr int64_t DKDV_MAX_WGS = 1024;                       // workgroups that may carry a partial slab
constexpr int64_t DKDV_SLAB_BYTES = DKDV_MAX_WGS * 65536;    // 64 MB

template <typename T>
__global__ __launch_bounds__(256, 2) void sra_bwd_dkdv_bf16(const T* __restrict__ q, const T* __restrict__ kv,
                                                            const T* __restrict__ dout, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, float* __restrict__ dkv,
                                                            int N, int Nkv, int heads, int kgroups, float scale, int qch,
                                                            int64_t slab_stride) {
  typedef T bf16_t;                                  // T = bf16_t or f16_t
  typedef typename Vec16<T>::type bf16x8;
  __shared__ __attribute__((aligned(16))) DkdvSmem<T> sm;
  constexpr int QS = DkdvSmem<T>::QS;
  const int C = heads * 64;
  const int b = blockIdx.z, hd = blockIdx.y / kgroups, kg = blockIdx.y % kgroups;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const float sl2 = scale * LOG2E;
  const bf16_t* kvb = kv + (size_t)b * Nkv * 2 * C + hd * 64;
  const size_t sbase = ((size_t)b * heads + hd) * N;
  const int qt_begin = blockIdx.x * qch;
  const int qt_end = min(qt_begin + qch, (N + 31) / 32);
  const int ktile0 = kg * KGROUP + wave * 32;       // first key of this wave
  const bool wave_active = ktile0 < Nkv;
  const int key = ktile0 + r;
  const bool kok = key < Nkv;

  // staging role of this thread: chunks tid and tid + 256 of the 512 16-byte chunks {Q tile | dO tile}
  auto stage_load = [&](int q0, bf16x8 (&v)[2], float& sv) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int row = tid >> 3, ch = tid & 7;      // k = 0: Q, k = 1: dO
      if (q0 + row < N) v[k] = *reinterpret_cast<const bf16x8*>((k ? dout : q) + ((size_t)b * N + q0 + row) * C + hd * 64 + ch * 8);
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) v[k][j] = (bf16_t)0.f;
    }
    if (tid < 64) {   // lanes 0-31: lse (log2 domain), lanes 32-63: delta
      const int row = tid & 31;
      const bool ok = (q0 + row) < N;
      sv = tid < 32 ? (ok ? lse[sbase + q0 + row] * LOG2E : INFINITY) : (ok ? delta[sbase + q0 + row] : 0.f);
    }
  };
  auto stage_store = [&](int buf, const bf16x8 (&v)[2], float sv) {
    const int row = tid >> 3, ch = tid & 7;
    *reinterpret_cast<bf16x8*>(&sm.q[buf][row][ch * 8]) = v[0];
    *reinterpret_cast<bf16x8*>(&sm.g[buf][row][ch * 8]) = v[1];
    if (tid < 32) sm.lse2[buf][tid] = sv;
    else if (tid < 64) sm.dl[buf][tid - 32] = sv;
  };

  bf16x8 kf[4], vf[4];                              // lane (key r, half h): K/V[key][16s+8h .. +7]
  {
    const bf16_t* kp = kvb + (size_t)key * 2 * C + 8 * h;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (kok) { kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s); vf[s] = *reinterpret_cast<const bf16x8*>(kp + C + 16 * s); }
      else
#pragma unroll
        for (int j = 0; j < 8; ++j) { kf[s][j] = (bf16_t)0.f; vf[s][j] = (bf16_t)0.f; }
    }
  }
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f; }

  bf16x8 st[2];
  float sv = 0.f;
  stage_load(qt_begin * 32, st, sv);
  stage_store(0, st, sv);
  __syncthreads();
  int cur = 0;
#pragma unroll 1
  for (int qt = qt_begin; qt < qt_end; ++qt) {
    const bool more = (qt + 1) < qt_end;
    if (more) stage_load((qt + 1) * 32, st, sv);     // global loads in flight during this tile's MFMAs
    if (wave_active) {
      f32x16 sA, pA;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sA[i] = 0.f; pA[i] = 0.f; }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 qf = *reinterpret_cast<const bf16x8*>(&sm.q[cur][r][16 * s + 8 * h]);
        const bf16x8 gf = *reinterpret_cast<const bf16x8*>(&sm.g[cur][r][16 * s + 8 * h]);
        sA = mfma16(qf, kf[s], sA);   // S[q][key]
        pA = mfma16(gf, vf[s], pA);   // dP[q][key]
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int qi = mfma_row(i, h);
        const float p = kok ? __builtin_amdgcn_exp2f(fmaf(sA[i], sl2, -sm.lse2[cur][qi])) : 0.f;
        sA[i] = p;                                   // P
        pA[i] = p * (pA[i] - sm.dl[cur][qi]) * scale;   // dS
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 pf, df;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pf[j] = (bf16_t)sA[8 * s2 + j]; df[j] = (bf16_t)pA[8 * s2 + j]; }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          // column fragments: lane (d = nb*32 + r, half h), element j <-> query 16*s2 + 8*(j>>2) + 4h + (j&3)
          const bf16x8 gb = lds_tr_frag(&sm.g[cur][0][0], QS, 16 * s2, nb * 32, lane);
          const bf16x8 qb = lds_tr_frag(&sm.q[cur][0][0], QS, 16 * s2, nb * 32, lane);
          dv[nb] = mfma16(pf, gb, dv[nb]);  // dV[key][d]
          dk[nb] = mfma16(df, qb, dk[nb]);  // dK[key][d]
        }
      }
    }
    if (more) stage_store(cur ^ 1, st, sv);          // buffer cur^1 was last read before the previous barrier
    __syncthreads();
    cur ^= 1;
  }
  if (wave_active) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = ktile0 + mfma_row(i, h);
        if (kk < Nkv) {
          // lanes r: 128-B contiguous segments.  slab_stride == 0: this workgroup swept every query, the sum is complete and goes
          // straight to dkv; otherwise the query chunks write PLAIN partial slabs [chunk][B][Nkv][2C] that dkdv_reduce_kernel sums
          // (no fp32 atomics: 512 workgroups x 64 KB of atomic adds cost ~25 us of a 63 us launch at the chip-wide atomic rate)
          float* p = dkv + (size_t)blockIdx.x * slab_stride + ((size_t)b * Nkv + kk) * 2 * C + hd * 64 + nb * 32 + r;
          p[0] = dk[nb][i];
          p[C] = dv[nb][i];
        }
      }
  }
}

// dkv[i] = sum over the query chunks of slab[c][i]  (float4 body; n is a multiple of 4: 2C = 128 heads floats per key)
__global__ __launch_bounds__(256) void dkdv_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dkv, int64_t n4, int chunks,
                                                          int64_t stride4) {
  const f32x4* s4 = reinterpret_cast<const f32x4*>(slab);
  f32x4* o4 = reinterpret_cast<f32x4*>(dkv);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 acc = s4[i];
    for (int c = 1; c < chunks; ++c) {
      const f32x4 v = s4[i + (int64_t)c * stride4];
      acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
    }
    o4[i] = acc;
  }
}

template <typename T, int KCH>
__global__ __launch_bounds__(256, 2) void sra_bwd_dq_bf16(const T* __restrict__ q, const T* __restrict__ kv, const T* __restrict__ o,
                                                          const T* __restrict__ dout, const float* __restrict__ lse,
                                                          float* __restrict__ delta, T* __restrict__ dq,
                                                          int N, int Nkv, int heads, float scale, int qtw) {
  typedef T bf16_t;
  typedef typename Vec16<T>::type bf16x8;
  typedef typename Vec8<T>::type bf16x4;
  constexpr int KS = 72, NT = KCH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + KCH * KS;
  const int C = heads * 64;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const float sl2 = scale * LOG2E;
  const bf16_t* kvb = kv + (size_t)b * Nkv * 2 * C + hd * 64;
  const int nchunks = (Nkv + KCH - 1) / KCH;
  for (int t = 0; t < qtw; ++t) {
    const int q0 = ((blockIdx.x * qtw + t) * 4 + wave) * 32;
    const bool qok = (q0 + r) < N;
    bf16x8 qf[4], gf[4];                            // B operands: lane (query r, half h)
    {
      const size_t off = ((size_t)b * N + q0 + r) * C + hd * 64 + 8 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (qok) { qf[s] = *reinterpret_cast<const bf16x8*>(q + off + 16 * s); gf[s] = *reinterpret_cast<const bf16x8*>(dout + off + 16 * s); }
        else
#pragma unroll
          for (int j = 0; j < 8; ++j) { qf[s][j] = (bf16_t)0.f; gf[s][j] = (bf16_t)0.f; }
      }
    }
    const size_t so = ((size_t)b * heads + hd) * N + q0 + r;
    const float l2 = qok ? lse[so] * LOG2E : INFINITY;
    // delta = rowsum(dO * O) of this lane's query: the lane pair (h = 0, 1) holds the dO row already (gf), O is read the same way; the
    // separate delta launch and its second pass over dO / O are gone.  Written out for the dK/dV kernel, which runs after this one.
    float dl = 0.f;
    if (qok) {
      const size_t off = ((size_t)b * N + q0 + r) * C + hd * 64 + 8 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 of = *reinterpret_cast<const bf16x8*>(o + off + 16 * s);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)of[j] * (float)gf[s][j];
      }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (qok && h == 0) delta[so] = dl;
    f32x16 dqa[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dqa[0][i] = 0.f; dqa[1][i] = 0.f; }
    for (int c = 0; c < nchunks; ++c) {
      const int k0 = c * KCH;
      const int kn = min(KCH, Nkv - k0);
      if (!(nchunks == 1 && t > 0)) {
        __syncthreads();
        for (int i = tid; i < KCH * 8; i += 256) {
          int key = i >> 3, ch = i & 7;
          bf16x8 kk, vv;
          if (key < kn) {
            const bf16_t* p = kvb + (size_t)(k0 + key) * 2 * C + ch * 8;
            kk = *reinterpret_cast<const bf16x8*>(p);
            vv = *reinterpret_cast<const bf16x8*>(p + C);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { kk[j] = (bf16_t)0.f; vv[j] = (bf16_t)0.f; }
          }
          *reinterpret_cast<bf16x8*>(Ks + key * KS + ch * 8) = kk;
          *reinterpret_cast<bf16x8*>(Vs + key * KS + ch * 8) = vv;
        }
        __syncthreads();
      }
      const int ntiles = (kn + 31) >> 5;
#pragma unroll 1
      for (int kt = 0; kt < ntiles; ++kt) {
        f32x16 sT, pT;
#pragma unroll
        for (int i = 0; i < 16; ++i) { sT[i] = 0.f; pT[i] = 0.f; }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bf16x8 ka = *reinterpret_cast<const bf16x8*>(Ks + (kt * 32 + r) * KS + 16 * s + 8 * h);
          const bf16x8 va = *reinterpret_cast<const bf16x8*>(Vs + (kt * 32 + r) * KS + 16 * s + 8 * h);
          sT = mfma16(ka, qf[s], sT);   // S^T[key][q]
          pT = mfma16(va, gf[s], pT);   // dP^T[key][q]
        }
        const bool ragged = (kn & 31) && (kt == ntiles - 1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float p = __builtin_amdgcn_exp2f(fmaf(sT[i], sl2, -l2));
          if (ragged && kt * 32 + mfma_row(i, h) >= kn) p = 0.f;
          pT[i] = p * (pT[i] - dl) * scale;          // dS^T
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          bf16x8 df;
#pragma unroll
          for (int j = 0; j < 8; ++j) df[j] = (bf16_t)pT[8 * s2 + j];
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const bf16x8 ka = lds_tr_frag(Ks, KS, kt * 32 + 16 * s2, nb * 32, lane);      // K^T fragment
            dqa[nb] = mfma16(ka, df, dqa[nb]);  // dQ^T[d][q]
          }
        }
      }
    }
    if (qok) {
      bf16_t* op = dq + ((size_t)b * N + q0 + r) * C + hd * 64;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (bf16_t)dqa[nb][4 * g + j];
          *reinterpret_cast<bf16x4*>(op + nb * 32 + 8 * g + 4 * h) = w;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------ fp32
// LDS (floats): Qr[32][65], Gr[32][65] (row-major, padded: conflict-free column reads), Ks[SLICE][64],
//               lse2[32], dlt[32], dqs[32][64]
struct BwdSmemF32 {
  static constexpr int QS = 65;
  static constexpr size_t off_Qr = 0;
  static constexpr size_t off_Gr = off_Qr + 32 * QS * 4;
  static constexpr size_t off_Ks = off_Gr + 32 * QS * 4;
  static constexpr size_t off_lse = off_Ks + (size_t)SLICE * 64 * 4;
  static constexpr size_t off_dlt = off_lse + 32 * 4;
  static constexpr size_t off_dq = off_dlt + 32 * 4;
  static constexpr size_t total = off_dq + 64 * 33 * 4;
};

__global__ __launch_bounds__(512) void sra_bwd_f32(const float* __restrict__ q, const float* __restrict__ kv,
                                                   const float* __restrict__ dout, const float* __restrict__ lse,
                                                   const float* __restrict__ delta, float* __restrict__ dq,
                                                   float* __restrict__ dkv, int N, int Nkv, int heads, float scale,
                                                   int qch) {
  typedef BwdSmemF32 L;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Qr = reinterpret_cast<float*>(smem + L::off_Qr);
  float* Gr = reinterpret_cast<float*>(smem + L::off_Gr);
  float* Ks = reinterpret_cast<float*>(smem + L::off_Ks);
  float* lse2 = reinterpret_cast<float*>(smem + L::off_lse);
  float* dlt = reinterpret_cast<float*>(smem + L::off_dlt);
  float* dqs = reinterpret_cast<float*>(smem + L::off_dq);

  const int C = heads * 64;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const float sl2 = scale * LOG2E;
  const float* kvb = kv + (size_t)b * Nkv * 2 * C + hd * 64;
  const int qt_begin = blockIdx.x * qch;
  const int qt_end = min(qt_begin + qch, (N + 31) / 32);
  const int nslices = (Nkv + SLICE - 1) / SLICE;

  for (int sl = 0; sl < nslices; ++sl) {
    const int k0 = sl * SLICE;
    const int kn = min(SLICE, Nkv - k0);
    const bool wave_active = wave * 32 < kn;
    __syncthreads();
    for (int i = tid; i < SLICE * 16; i += nthr) {
      int key = i >> 4, ch = i & 15;
      f32x4 kk;
      if (key < kn) kk = *reinterpret_cast<const f32x4*>(kvb + (size_t)(k0 + key) * 2 * C + ch * 4);
      else { kk[0] = kk[1] = kk[2] = kk[3] = 0.f; }
      *reinterpret_cast<f32x4*>(Ks + key * 64 + ch * 4) = kk;
    }
    __syncthreads();
    // stationary K/V values of this lane's key: element tt <-> d = 2*tt + h
    float kf[32], vf[32];
    {
      const bool kok = (wave * 32 + r) < kn;
      const float* kp = kvb + (size_t)(k0 + wave * 32 + r) * 2 * C;
#pragma unroll
      for (int tt = 0; tt < 32; tt += 2) {
        f32x4 a, v;
        if (kok) { a = *reinterpret_cast<const f32x4*>(kp + 2 * tt); v = *reinterpret_cast<const f32x4*>(kp + C + 2 * tt); }
        else { a[0] = a[1] = a[2] = a[3] = 0.f; v = a; }
        kf[tt] = h ? a[1] : a[0]; kf[tt + 1] = h ? a[3] : a[2];
        vf[tt] = h ? v[1] : v[0]; vf[tt + 1] = h ? v[3] : v[2];
      }
    }
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = dk[1][i] = dv[0][i] = dv[1][i] = 0.f; }

    for (int qt = qt_begin; qt < qt_end; ++qt) {
      const int q0 = qt * 32;
      __syncthreads();
      for (int i = tid; i < 1024; i += nthr) {
        const int which = i >> 9, j16 = i & 511, row = j16 >> 4, ch = j16 & 15;
        const float* src = (which ? dout : q) + ((size_t)b * N + q0 + row) * C + hd * 64 + ch * 4;
        f32x4 v;
        if (q0 + row < N) v = *reinterpret_cast<const f32x4*>(src);
        else { v[0] = v[1] = v[2] = v[3] = 0.f; }
        float* dst = (which ? Gr : Qr) + row * L::QS + ch * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[j] = v[j];
      }
      for (int i = tid; i < 32; i += nthr) {
        const bool ok = (q0 + i) < N;
        const size_t o = ((size_t)b * heads + hd) * N + q0 + i;
        lse2[i] = ok ? lse[o] * LOG2E : INFINITY;
        dlt[i] = ok ? delta[o] : 0.f;
      }
      for (int i = tid; i < 64 * 33; i += nthr) dqs[i] = 0.f;
      __syncthreads();

      if (wave_active) {
        const bool key_ok_lane = (wave * 32 + r) < kn;
        const float* qrow = Qr + r * L::QS + h;   // Q[q0+r][2tt+h]
        const float* grow = Gr + r * L::QS + h;
        // ===== orientation 1: rows = query, lane = key
        {
          f32x16 sA, pA;
#pragma unroll
          for (int i = 0; i < 16; ++i) { sA[i] = 0.f; pA[i] = 0.f; }
#pragma unroll
          for (int tt = 0; tt < 32; ++tt) {
            sA = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[2 * tt], kf[tt], sA, 0, 0, 0);
            pA = __builtin_amdgcn_mfma_f32_32x32x2f32(grow[2 * tt], vf[tt], pA, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int qi = mfma_row(i, h);
            float p = key_ok_lane ? exp2f(sA[i] * sl2 - lse2[qi]) : 0.f;
            sA[i] = p;
            pA[i] = p * (pA[i] - dlt[qi]) * scale;
          }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int qi = mfma_row(i, h);
            const float* gp = Gr + qi * L::QS + r;
            const float* qp = Qr + qi * L::QS + r;
            dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[i], gp[0], dv[0], 0, 0, 0);
            dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[i], gp[32], dv[1], 0, 0, 0);
            dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(pA[i], qp[0], dk[0], 0, 0, 0);
            dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(pA[i], qp[32], dk[1], 0, 0, 0);
          }
        }
        // ===== orientation 2: rows = key, lane = query
        {
          f32x16 sB, pB;
#pragma unroll
          for (int i = 0; i < 16; ++i) { sB[i] = 0.f; pB[i] = 0.f; }
#pragma unroll
          for (int tt = 0; tt < 32; ++tt) {
            sB = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tt], qrow[2 * tt], sB, 0, 0, 0);
            pB = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[tt], grow[2 * tt], pB, 0, 0, 0);
          }
          const float l2 = lse2[r], dl = dlt[r];
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const bool kok = (wave * 32 + mfma_row(i, h)) < kn;
            float p = kok ? exp2f(sB[i] * sl2 - l2) : 0.f;
            pB[i] = p * (pB[i] - dl) * scale;
          }
          f32x16 dqa[2];
#pragma unroll
          for (int i = 0; i < 16; ++i) { dqa[0][i] = 0.f; dqa[1][i] = 0.f; }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float* kp = Ks + (wave * 32 + mfma_row(i, h)) * 64 + r;   // K[key][d = nb*32 + r]
            dqa[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[0], pB[i], dqa[0], 0, 0, 0);
            dqa[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[32], pB[i], dqa[1], 0, 0, 0);
          }
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&dqs[(nb * 32 + mfma_row(i, h)) * 33 + r], dqa[nb][i]);
        }
      }
      __syncthreads();
      for (int i = tid; i < 32 * 16; i += nthr) {
        const int row = i >> 4, ch = i & 15;
        if (q0 + row < N) {
          float* dst = dq + ((size_t)b * N + q0 + row) * C + hd * 64 + ch * 4;
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = dqs[(ch * 4 + j) * 33 + row];
          if (sl > 0) { f32x4 prev = *reinterpret_cast<const f32x4*>(dst); o += prev; }
          *reinterpret_cast<f32x4*>(dst) = o;
        }
      }
    }
    if (wave_active) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = wave * 32 + mfma_row(i, h);
          if (key < kn) {
            float* p = dkv + ((size_t)b * Nkv + k0 + key) * 2 * C + hd * 64 + nb * 32 + r;
            atomicAdd(p, dk[nb][i]);
            atomicAdd(p + C, dv[nb][i]);
          }
        }
    }
  }
}

}  // namespace

// workspace = delta [B,h,N] fp32 (256-byte aligned) + the dK/dV partial slabs of the query chunks (16-bit paths; bounded: every
// workgroup owns 128 keys x 64 channels x {K, V} x 4 B = 64 KB of one slab and at most ~2 x 512 workgroups carry a slab)
extern "C" int64_t dgtd_sra_attn_bwd_workspace(int B, int N, int heads) {
  return (((int64_t)B * N * heads * 4 + 255) / 256) * 256 + DKDV_SLAB_BYTES;
}

extern "C" int dgtd_sra_attn_bwd(const void* q, const void* kv, const void* out, const void* dout, const float* lse,
                                 void* dq, float* dkv_f32, void* workspace, int B, int N, int Nkv, int heads, float scale,
                                 dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_MFMA, 10.0 * B * heads * (double)N * Nkv * 64, "dgtd_sra_attn_bwd[B=%d,N=%d,Nkv=%d,h=%d]", B, N, Nkv, heads);
  DGTD_REQUIRE(B > 0 && N > 0 && Nkv > 0 && heads > 0, "sra_attn_bwd: bad sizes B=%d N=%d Nkv=%d heads=%d", B, N, Nkv, heads);
  DGTD_REQUIRE(dt == DGTD_F32 || DGTD_IS_HALF(dt), "sra_attn_bwd: bad dtype %d", (int)dt);
  hipStream_t st = (hipStream_t)s;
  float* delta = (float*)workspace;
  const int64_t items = (int64_t)B * N * heads;
  const int qtiles = (int)cdiv(N, 32);
  if (DGTD_IS_HALF(dt)) {
    // (1) dQ, forward-shaped launch; computes delta = rowsum(dO * O) on the way and writes it for (2)
    {
      int qtw = 1;
      while (qtw < 8 && cdiv(N, 128) * B * heads / (qtw * 2) >= 512) qtw *= 2;
      dim3 grid((unsigned)cdiv(N, 128 * qtw), heads, B);
      if (Nkv <= 64) {
        constexpr int KCH = 64;
        DGTD_DISPATCH_HALF(dt, hipLaunchKernelGGL((sra_bwd_dq_bf16<T_, KCH>), grid, dim3(256), (size_t)2 * KCH * 72 * 2, st, (const T_*)q, (const T_*)kv,
                           (const T_*)out, (const T_*)dout, lse, delta, (T_*)dq, N, Nkv, heads, scale, qtw));
      } else {
        constexpr int KCH = 256;
        DGTD_DISPATCH_HALF(dt, hipLaunchKernelGGL((sra_bwd_dq_bf16<T_, KCH>), grid, dim3(256), (size_t)2 * KCH * 72 * 2, st, (const T_*)q, (const T_*)kv,
                           (const T_*)out, (const T_*)dout, lse, delta, (T_*)dq, N, Nkv, heads, scale, qtw));
      }
      DGTD_CHECK_LAUNCH("sra_attn_bwd_dq");
    }
    // (2) dK/dV: workgroup = 128 keys (4 waves) x a chunk of query tiles.  Keys are split across workgroups; queries are split as far as
    // needed to put ~2 workgroups on every CU.  With a single chunk the result is stored plainly into dkv; with several, every chunk
    // writes a plain partial slab and ONE reduce launch sums them (was: fp32 atomics into a caller-zeroed buffer).
    const int kgroups = (int)cdiv(Nkv, KGROUP);
    DGTD_REQUIRE((int64_t)heads * kgroups <= 65535, "sra_attn_bwd: heads*kgroups too large for the grid");
    static const int64_t wg_env = getenv("DGTD_DKDV_WGS") ? atol(getenv("DGTD_DKDV_WGS")) : 0;
    const int64_t cols = (int64_t)B * heads * kgroups;
    const int64_t wg_target = wg_env ? std::min<int64_t>(wg_env, DKDV_MAX_WGS) : (cols <= 32 ? 512 : 256);
    int nq = (int)std::min<int64_t>(qtiles, std::max<int64_t>(1, cdiv(wg_target, cols)));
    const int64_t dkv_elems = (int64_t)B * Nkv * 2 * heads * 64;
    float* slab = (float*)((char*)workspace + (((int64_t)B * N * heads * 4 + 255) / 256) * 256);
    while (nq > 1 && (int64_t)nq * dkv_elems * 4 > DKDV_SLAB_BYTES) --nq;      // the slabs must fit the workspace
    const int qch = (int)cdiv(qtiles, nq);
    const int nqc = (int)cdiv(qtiles, qch);
    DGTD_DISPATCH_HALF(dt, hipLaunchKernelGGL(sra_bwd_dkdv_bf16<T_>, dim3(nqc, heads * kgroups, B), dim3(256), 0, st, (const T_*)q, (const T_*)kv,
                       (const T_*)dout, lse, (const float*)delta, nqc > 1 ? slab : dkv_f32, N, Nkv, heads, kgroups, scale, qch,
                       nqc > 1 ? dkv_elems : (int64_t)0));
    DGTD_CHECK_LAUNCH("sra_attn_bwd_dkdv");
    if (nqc > 1) {
      const int64_t n4 = dkv_elems / 4;
      hipLaunchKernelGGL(dkdv_reduce_kernel, dim3((unsigned)std::min<int64_t>(cdiv(n4, 256), 2048)), dim3(256), 0, st, (const float*)slab, dkv_f32, n4, nqc,
                         n4);
      DGTD_CHECK_LAUNCH("sra_attn_bwd_dkdv_reduce");
    }
  } else {
    hipLaunchKernelGGL((attn_delta_kernel<float>), dim3((unsigned)cdiv(items * 16, 256)), dim3(256), 0, st, (const float*)out, (const float*)dout, delta, B, N, heads);
    DGTD_CHECK_LAUNCH("attn_delta");
    // query tiles per workgroup: enough workgroups to fill 256 CUs, few enough dK/dV flushes
    int qch = 1;
    while (qch < 32 && cdiv(qtiles, qch * 2) * B * heads >= 512) qch *= 2;
    const int nwaves = (int)cdiv(std::min(Nkv, SLICE), 32);
    dim3 grid((unsigned)cdiv(qtiles, qch), heads, B), block(64 * nwaves);
    hipLaunchKernelGGL(sra_bwd_f32, grid, block, BwdSmemF32::total, st, (const float*)q, (const float*)kv, (const float*)dout, lse, delta, (float*)dq, dkv_f32, N, Nkv, heads, scale, qch);
  }
  DGTD_CHECK_LAUNCH("sra_attn_bwd");
  return 0;
}

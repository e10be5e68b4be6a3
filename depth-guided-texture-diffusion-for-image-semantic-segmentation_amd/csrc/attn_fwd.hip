// attn_fwd.hip — fused spatial-reduction attention forward: out = softmax(Q K^T * scale) V, head_dim 64.
// Replaces twig/model/cod.py:913-917.  K/V per (batch, head) are short (N_kv = (S/32)^2),
// so a workgroup stages a whole K/V chunk in LDS once and streams Q tiles past it; the score tile never
// leaves registers.
//
// MFMA mapping (wave64, 32x32 tiles; layouts from the CDNA4 guide §3):
//   S^T = K . Q^T   : A = K rows (key on the MFMA row), B = Q^T (query on the lane)  -> each lane holds one
//                     query column; the softmax over keys is lane-local + one cross-half shuffle.
//   O^T = V^T . P^T : the S^T accumulator (rows = key) is fed straight back as the B operand ("accumulator
//                     tile as the next MFMA's operand"); A = V^T fragments.  O^T keeps the query on the lane,
//                     so the online-softmax rescale and the final 1/l are lane-local too.
// bf16 I/O : v_mfma_f32_32x32x16_bf16, fp32 accumulate, P rounded to bf16 for the second product; V^T fragments by
//            ds_read_b64_tr_b16 from a row-major V tile.
// fp32 I/O : v_mfma_f32_32x32x2_f32 (exact fp32) — the parity-mode kernel.
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// ------------------------------------------------------------------------------------------ bf16
// LDS image: Ks[KCH][72], Vs[KCH][72] — both row-major (row = key), 16-B staging copies only.  The V^T fragments of the second
// product come from the transposing LDS read (ds_read_b64_tr_b16), so nothing is transposed at staging time.
// KCH keys are staged per LDS chunk; the score tile is processed KREG keys at a time (online softmax between units) so the
// kernel stays at <= 256 registers -> 2 waves per SIMD: one wave's softmax VALU overlaps the other's MFMA.
template <typename T, int KCH, int KREG>
__global__ __launch_bounds__(256, 2) void sra_fwd_bf16(const T* __restrict__ q, const T* __restrict__ kv,
                                                    T* __restrict__ out, float* __restrict__ lse,
                                                    int N, int Nkv, int heads, float scale_log2e, int qtw) {
  typedef T bf16_t;                                  // T = bf16_t or f16_t: same kernel on v_mfma_f32_32x32x16_bf16 / _f16
  typedef typename Vec16<T>::type bf16x8;
  typedef typename Vec8<T>::type bf16x4;
  constexpr int KS = 72, NT = KREG / 32, NU = KCH / KREG;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Vs = Ks + KCH * KS;
  const int C = heads * 64;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const bf16_t* kvb = kv + (size_t)b * Nkv * 2 * C + hd * 64;
  const int nchunks = (Nkv + KCH - 1) / KCH;

  for (int t = 0; t < qtw; ++t) {
    const int q0 = ((blockIdx.x * qtw + t) * 4 + wave) * 32;
    const bool qok = (q0 + r) < N;
    // Q^T fragments (B operand): lane (query r, half h), k-step s -> Q[q][16s+8h .. +7]
    bf16x8 qf[4];
    {
      const bf16_t* qp = q + ((size_t)b * N + q0 + r) * C + hd * 64 + 8 * h;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (qok) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
        else
#pragma unroll
          for (int j = 0; j < 8; ++j) qf[s][j] = (bf16_t)0.f;
      }
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }

    for (int c = 0; c < nchunks; ++c) {
      const int k0 = c * KCH;
      const int kn = min(KCH, Nkv - k0);          // valid keys in this chunk
      if (!(nchunks == 1 && t > 0)) {
        __syncthreads();
        for (int i = tid; i < KCH * 8; i += 256) {   // K and V rows, 16-B copies; zero-fill keys >= kn
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
#pragma unroll 1
      for (int u = 0; u < NU; ++u) {
      const int ubase = u * KREG;                 // first key of this register unit inside the LDS chunk
      if (ubase >= kn) break;
      const int ntiles = (min(kn - ubase, KREG) + 31) >> 5;
      const bool first_unit = (c == 0 && u == 0);
      // ---- S^T tiles: rows = key, col(lane) = query
      f32x16 sc[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < ntiles) {
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            bf16x8 a = *reinterpret_cast<const bf16x8*>(Ks + (ubase + kt * 32 + r) * KS + 16 * s + 8 * h);
            acc = mfma16(a, qf[s], acc);
          }
          if ((kn & 31) && ubase + kt * 32 + 32 > kn) {   // ragged last tile only: padded keys leave the softmax
#pragma unroll
            for (int i = 0; i < 16; ++i)
              if (ubase + kt * 32 + mfma_row(i, h) >= kn) acc[i] = -INFINITY;
          }
          sc[kt] = acc;
        }
      }
      // ---- softmax in the log2 domain: one max, one fma + v_exp_f32 + one add per score
      float mc = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
        if (kt < ntiles)
#pragma unroll
          for (int i = 0; i < 16; ++i) mc = fmaxf(mc, sc[kt][i]);
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64)) * scale_log2e;     // scale > 0 commutes with max
      const float m_new = fmaxf(m_run, mc);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on the first chunk (m_run = -inf)
      m_run = m_new;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
        if (kt < ntiles)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sc[kt][i], scale_log2e, -m_new));
            sc[kt][i] = p;
            ls += p;
          }
      l_run = l_run * alpha + ls;
      if (!first_unit) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
      }
      // ---- O^T += V^T . P^T  (A = V^T fragment by transposing LDS read, B = P^T straight from the accumulator)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < ntiles) {
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pb;
#pragma unroll
            for (int j = 0; j < 8; ++j) pb[j] = (bf16_t)sc[kt][8 * s2 + j];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              const bf16x8 a = lds_tr_frag(Vs, KS, ubase + kt * 32 + 16 * s2, nb * 32, lane);
              o[nb] = mfma16(a, pb, o[nb]);
            }
          }
        }
      }
      }  // register unit
    }
    // ---- finalize: lane = query
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qok) {
      if (h == 0) lse[((size_t)b * heads + hd) * N + q0 + r] = m_run * LN2 + logf(l_tot);
      bf16_t* op = out + ((size_t)b * N + q0 + r) * C + hd * 64;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = (bf16_t)(o[nb][4 * g + j] * inv);
          *reinterpret_cast<bf16x4*>(op + nb * 32 + 8 * g + 4 * h) = w;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------ fp32
// LDS image: Ks[KCH][65] fp32 (padded: column reads across keys are conflict-free), Vs[KCH][64] fp32.
template <int KCH>
__global__ __launch_bounds__(256) void sra_fwd_f32(const float* __restrict__ q, const float* __restrict__ kv,
                                                   float* __restrict__ out, float* __restrict__ lse,
                                                   int N, int Nkv, int heads, float scale_log2e, int qtw) {
  constexpr int KS = 65, NT = KCH / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* Ks = reinterpret_cast<float*>(smem);
  float* Vs = Ks + KCH * KS;
  const int C = heads * 64;
  const int b = blockIdx.z, hd = blockIdx.y;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const float* kvb = kv + (size_t)b * Nkv * 2 * C + hd * 64;
  const int nchunks = (Nkv + KCH - 1) / KCH;

  for (int t = 0; t < qtw; ++t) {
    const int q0 = ((blockIdx.x * qtw + t) * 4 + wave) * 32;
    const bool qok = (q0 + r) < N;
    // B operand of mfma_32x32x2: B[k = h][col = query r] for k-step tt -> Q[q][2*tt + h]
    float qv[32];
    {
      const float* qp = q + ((size_t)b * N + q0 + r) * C + hd * 64;
#pragma unroll
      for (int tt = 0; tt < 32; tt += 2) {
        f32x4 v4;
        if (qok) v4 = *reinterpret_cast<const f32x4*>(qp + 2 * tt);
        else { v4[0] = v4[1] = v4[2] = v4[3] = 0.f; }
        qv[tt] = h ? v4[1] : v4[0];
        qv[tt + 1] = h ? v4[3] : v4[2];
      }
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }

    for (int c = 0; c < nchunks; ++c) {
      const int k0 = c * KCH;
      const int kn = min(KCH, Nkv - k0);
      if (!(nchunks == 1 && t > 0)) {
        __syncthreads();
        for (int i = tid; i < KCH * 16; i += 256) {
          int key = i >> 4, ch = i & 15;
          f32x4 kk, vv;
          if (key < kn) {
            const float* p = kvb + (size_t)(k0 + key) * 2 * C + ch * 4;
            kk = *reinterpret_cast<const f32x4*>(p);
            vv = *reinterpret_cast<const f32x4*>(p + C);
          } else { kk[0] = kk[1] = kk[2] = kk[3] = 0.f; vv = kk; }
#pragma unroll
          for (int j = 0; j < 4; ++j) Ks[key * KS + ch * 4 + j] = kk[j];
          *reinterpret_cast<f32x4*>(Vs + key * 64 + ch * 4) = vv;
        }
        __syncthreads();
      }
      const int ntiles = (kn + 31) >> 5;
      f32x16 sc[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < ntiles) {
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
          const float* kp = Ks + (kt * 32 + r) * KS + h;
#pragma unroll
          for (int tt = 0; tt < 32; ++tt) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kp[2 * tt], qv[tt], acc, 0, 0, 0);
          sc[kt] = acc;
        }
      }
      float mc = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < ntiles) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            float v = sc[kt][i] * scale_log2e;
            if (kt * 32 + mfma_row(i, h) >= kn) v = -INFINITY;
            sc[kt][i] = v;
            mc = fmaxf(mc, v);
          }
        }
      }
      mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
      const float m_new = fmaxf(m_run, mc);
      const float alpha = exp2f(m_run - m_new);
      m_run = m_new;
      float ls = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < ntiles) {
#pragma unroll
          for (int i = 0; i < 16; ++i) { float p = exp2f(sc[kt][i] - m_new); sc[kt][i] = p; ls += p; }
        }
      }
      l_run = l_run * alpha + ls;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
      // O^T[d][q] += sum_key V[key][d] * P^T[key][q]; k-step i pairs lane-half h with key mfma_row(i, h)
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        if (kt < ntiles) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const float* vp = Vs + (kt * 32 + mfma_row(i, h)) * 64 + r;
            o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[0], sc[kt][i], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vp[32], sc[kt][i], o[1], 0, 0, 0);
          }
        }
      }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    if (qok) {
      if (h == 0) lse[((size_t)b * heads + hd) * N + q0 + r] = m_run * LN2 + logf(l_tot);
      float* op = out + ((size_t)b * N + q0 + r) * C + hd * 64;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 w;
#pragma unroll
          for (int j = 0; j < 4; ++j) w[j] = o[nb][4 * g + j] * inv;
          *reinterpret_cast<f32x4*>(op + nb * 32 + 8 * g + 4 * h) = w;
        }
    }
  }
}

int pick_qtw(int N, int bh) {
  // q tiles of 128 rows per workgroup-iteration; aim for >= ~4 workgroups per CU, amortise staging otherwise
  int64_t groups = cdiv(N, 128);
  int qtw = 1;
  while (qtw < 8 && groups * bh / (qtw * 2) >= 512) qtw *= 2;   // ~2 workgroups per CU; K/V staging amortised over qtw tiles
  return qtw;
}

}  // namespace

extern "C" int dgtd_sra_attn_fwd(const void* q, const void* kv, void* out, float* lse, int B, int N, int Nkv, int heads,
                                 float scale, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_MFMA, 4.0 * B * heads * (double)N * Nkv * 64, "dgtd_sra_attn_fwd[B=%d,N=%d,Nkv=%d,h=%d]", B, N, Nkv, heads);
  DGTD_REQUIRE(B > 0 && N > 0 && Nkv > 0 && heads > 0, "sra_attn_fwd: bad sizes B=%d N=%d Nkv=%d heads=%d", B, N, Nkv, heads);
  DGTD_REQUIRE(heads <= 65535 && B <= 65535, "sra_attn_fwd: grid limits");
  const int qtw = pick_qtw(N, B * heads);
  dim3 grid((unsigned)cdiv(N, 128 * qtw), heads, B), block(256);
  const float sl2 = scale * LOG2E;
  if (DGTD_IS_HALF(dt)) {
    if (Nkv <= 64) {
      constexpr int KCH = 64;
      size_t lds = (size_t)(2 * KCH * 72) * 2;
      DGTD_DISPATCH_HALF(dt, hipLaunchKernelGGL((sra_fwd_bf16<T_, KCH, 64>), grid, block, lds, (hipStream_t)s, (const T_*)q, (const T_*)kv, (T_*)out, lse, N, Nkv, heads, sl2, qtw));
    } else {
      constexpr int KCH = 256;
      size_t lds = (size_t)(2 * KCH * 72) * 2;
      DGTD_DISPATCH_HALF(dt, hipLaunchKernelGGL((sra_fwd_bf16<T_, KCH, 128>), grid, block, lds, (hipStream_t)s, (const T_*)q, (const T_*)kv, (T_*)out, lse, N, Nkv, heads, sl2, qtw));
    }
  } else if (dt == DGTD_F32) {
    if (Nkv <= 64) {
      constexpr int KCH = 64;
      size_t lds = (size_t)(KCH * 65 + KCH * 64) * 4;
      hipLaunchKernelGGL((sra_fwd_f32<KCH>), grid, block, lds, (hipStream_t)s, (const float*)q, (const float*)kv, (float*)out, lse, N, Nkv, heads, sl2, qtw);
    } else {
      constexpr int KCH = 128;
      size_t lds = (size_t)(KCH * 65 + KCH * 64) * 4;
      hipLaunchKernelGGL((sra_fwd_f32<KCH>), grid, block, lds, (hipStream_t)s, (const float*)q, (const float*)kv, (float*)out, lse, N, Nkv, heads, sl2, qtw);
    }
  } else {
    DGTD_FAIL(2, "sra_attn_fwd: bad dtype %d", (int)dt);
  }
  DGTD_CHECK_LAUNCH("sra_attn_fwd");
  return 0;
}

// conv3x3.hip — dense 3x3 convolution (stride 1, zero padding 1) on NHWC bf16 maps with small channel counts (24..96), as
// implicit GEMM on v_mfma_f32_32x32x16_bf16.  Serves the 16 prompt decoders (ShapePropDecoder, twig/model/cod.py:1216-1226: two
// conv3x3 24->24 + ReLU per decoder at S/4) and the CAB bodies of the Hitnet decoder (cod.py:441-446: conv3x3 C->C, C in {32,64,96}).
// Z independent convolutions (own weights / outputs, own or shared input) run in ONE launch (blockIdx.z).
//
//   forward      y[z,b,h,w,co] = act( sum_{ky,kx,ci} x[z|0,b,h+ky-1,w+kx-1,ci] * w[z,co,ky,kx,ci] + bias[z,co] )
//   backward/x   the same kernel on dy with the transposed + spatially flipped kernel (conv3x3_flip_kernel); the ReLU of the
//                forward is undone while the dy tile is staged (value kept where the saved forward output is > 0)
//   backward/w   dw[z,co,ky,kx,ci] = sum_{b,h,w} dy[z,b,h,w,co] * x[z|0,b,h+ky-1,w+kx-1,ci]: pixels are the MFMA K dimension; both
//                operands come from pixel-major LDS tiles through the transposing read (ds_read_b64_tr_b16), per-workgroup partial
//                sums are reduced in a fixed order by conv3x3_wgrad_reduce_kernel, which also emits the bias gradient.
//
// Roofline: HBM.  Algorithmic bytes per convolution and direction = 2 B * B*H*W * (Ci + Co) (+ the mask map in the backward);
// arithmetic intensity 9*Ci*Co/(Ci+Co) flop/B = 108 (24->24) .. 432 (96->96), far below the bf16 MFMA ridge (~300 flop/B only
// reached by the 96-channel case), so the tiles are sized for load/store coalescing, not for MFMA occupancy.
#include "common.h"

namespace {

// T = bf16_t or f16_t (same kernels, v_mfma_f32_32x32x16_bf16 / _f16)
template <typename T>
__device__ __forceinline__ typename Vec16<T>::type zero8() {
  typename Vec16<T>::type v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (T)0.f;
  return v;
}

// Extra epilogue operands of conv3x3_fwd_kernel (dgtd_conv3x3_fwd_ex): everything the Hitnet CAB needs around its two convolutions
// without a separate elementwise launch.  All tensors share y's layout [Z][B][H][W][Co].
//   act 0: none   1: ReLU   2: PReLU forward  - y2 (optional) receives the pre-activation, y = pre > 0 ? pre : slope * pre
//   act 3: PReLU backward - the convolution output is the gradient w.r.t. the PReLU OUTPUT; ref = the saved pre-activation:
//          y = ref > 0 ? v : slope * v, and sum(v * ref over ref <= 0) is added atomically to *sgrad (one add per workgroup)
//   add: y += add (the gradient of a skip connection that forks off the convolution's input), applied last
struct ConvEpi { const void* add; const void* ref; void* y2; const float* slope; float* sgrad; int act; };

template <typename T>
__device__ __forceinline__ typename Vec16<T>::type load_masked(const T* __restrict__ src, const T* __restrict__ mask, size_t o) {
  typedef typename Vec16<T>::type V8;
  V8 v = *reinterpret_cast<const V8*>(src + o);
  if (mask) {
    const V8 m = *reinterpret_cast<const V8*>(mask + o);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)m[e] > 0.f ? v[e] : (T)0.f;
  }
  return v;
}

// ------------------------------------------------------------------------------------------------ forward / backward-data
// Workgroup = 4 waves; tile = TWX columns x TH = 4*MT*(32/TWX) rows of output pixels; wave w owns MT m-tiles of 32 pixels
// ((32/TWX) rows x TWX columns each).  The (TH+2) x (TWX+2) input halo tile is staged in LDS as [channel group of 8][pixel]
// (16-byte elements): the MFMA B fragment of lane (pixel, k-half) is one conflict-free ds_read_b128.
// K runs tap by tap; within a tap an MFMA step covers channel groups 2j (lanes 0-31) and 2j+1 (lanes 32-63).
// A = kernel rows (co): the [Co][CI] slice of each tap is copied to LDS with coalesced loads (fetched one or two taps ahead into
// registers) - reading the fragments straight from global memory costs one cache line per lane and was the bottleneck.  D[co][pixel].
template <typename T, int CI, int NT, int MT, int TWX>
__global__ __launch_bounds__(256) void conv3x3_fwd_kernel(const T* __restrict__ x, const T* __restrict__ mask,
                                                          const T* __restrict__ w, const T* __restrict__ bias,
                                                          T* __restrict__ y, int H, int W, int Co_full, int relu, int tiles_w,
                                                          long x_zs, long w_zs, long y_zs, int nsplit, ConvEpi epi) {
  constexpr int G = CI / 8, RW = 32 / TWX, TH = 4 * MT * RW, LW = TWX + 2, LP = (TH + 2) * LW;
  constexpr int COP = NT * 32, GP = G | 1, WSZ = COP * GP, NWR = (COP * G + 255) / 256;
  constexpr int WB = ((size_t)G * LP + 2 * WSZ) * 16 <= 65536 ? 2 : 1;   // double-buffer the kernel slices when LDS allows
  typedef typename Vec16<T>::type bf16x8;
  typedef typename Vec8<T>::type bf16x4;
  typedef T bf16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16x8* tile = reinterpret_cast<bf16x8*>(smem);   // [G][LP]
  bf16x8* wbuf = tile + G * LP;                      // [WB][COP][GP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  // blockIdx.z = convolution z x output-channel part: with few pixel tiles (the 96- and 64-channel CAB convolutions at 64^2 and
  // 32^2 give 256 / 64 workgroups) the NT 32-channel output tiles go to separate workgroups instead of one wave's registers:
  // three times the workgroups per CU, the input tile is re-read from L2
  const int tw = blockIdx.x % tiles_w, th = blockIdx.x / tiles_w, b = blockIdx.y, z = blockIdx.z / nsplit;
  const int co_base = (blockIdx.z % nsplit) * (NT * 32);
  const int Co = min(Co_full - co_base, NT * 32);        // output channels of this workgroup
  const int h0 = th * TH, w0 = tw * TWX;
  const size_t img = (size_t)b * H * W;
  const bf16_t* xb = x + (size_t)z * x_zs + img * CI;
  const bf16_t* mb = mask ? mask + (size_t)z * x_zs + img * CI : nullptr;
  const bf16_t* wz = w + (size_t)z * w_zs + (size_t)co_base * 9 * CI;

  // kernel slice of one tap, [co][GP] 16-byte chunks in LDS (GP odd: the A-fragment reads of 16 consecutive lanes hit 16 different
  // 16-byte bank groups); fetched from global with channel-group-fastest indexing (CI*2 contiguous bytes per output channel)
  bf16x8 wr[NWR];
  auto wfetch = [&](int tap) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int c = tid + i * 256;
      const int co = c / G, g = c % G;
      wr[i] = (c < COP * G && co < Co) ? *reinterpret_cast<const bf16x8*>(wz + ((size_t)co * 9 + tap) * CI + g * 8) : zero8<T>();
    }
  };
  auto wstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NWR; ++i) {
      const int c = tid + i * 256;
      if (c < COP * G) wbuf[buf * WSZ + (c / G) * GP + c % G] = wr[i];
    }
  };
  wfetch(0);
  for (int c = tid; c < LP * G; c += 256) {
    const int pix = c / G, g = c % G;
    const int pr = pix / LW, pc = pix % LW;
    const int h = h0 - 1 + pr, ww = w0 - 1 + pc;
    bf16x8 v = zero8<T>();
    if (h >= 0 && h < H && ww >= 0 && ww < W) v = load_masked(xb, mb, ((size_t)h * W + ww) * CI + g * 8);
    tile[g * LP + pix] = v;
  }
  if (WB == 2) {
    wstore(0);
    wfetch(1);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  const int pr = l31 / TWX, pc = l31 % TWX;
  int basepix[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) basepix[mt] = ((wave * MT + mt) * RW + pr) * LW + pc;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (WB == 2) {
      __syncthreads();                            // slice `tap` (and at tap 0 the input tile) is in LDS; everyone is done with tap-1
      if (tap + 1 < 9) wstore((tap + 1) & 1);
      if (tap + 2 < 9) wfetch(tap + 2);
    } else {
      if (tap > 0) __syncthreads();               // everyone is done reading slice tap-1
      wstore(0);
      __syncthreads();
      if (tap + 1 < 9) wfetch(tap + 1);
    }
    const bf16x8* wb = wbuf + (WB == 2 ? (tap & 1) * WSZ : 0);
    const int tapoff = (tap / 3) * LW + tap % 3;
#pragma unroll
    for (int js = 0; js < (G + 1) / 2; ++js) {
      const bool valid = 2 * js + 1 < G || !half;                    // odd G: the upper half of the last step is empty
      const int g = valid ? 2 * js + half : 2 * js;
      bf16x8 wf[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wf[nt] = valid ? wb[(nt * 32 + l31) * GP + g] : zero8<T>();
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const bf16x8 xf = tile[g * LP + basepix[mt] + tapoff];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma16(wf[nt], xf, acc[mt][nt]);
      }
    }
  }

  // epilogue: lane holds pixel l31 of each m-tile; registers 4q..4q+3 = output channels 8q + 4*half + {0..3}
  const bf16_t* bz = bias ? bias + (size_t)z * Co_full + co_base : nullptr;
  const size_t ybase = (size_t)z * y_zs + img * Co_full + co_base;
  bf16_t* yb = y + ybase;
  const bf16_t* addb = epi.add ? (const bf16_t*)epi.add + ybase : nullptr;
  const bf16_t* refb = epi.ref ? (const bf16_t*)epi.ref + ybase : nullptr;
  bf16_t* y2b = epi.y2 ? (bf16_t*)epi.y2 + ybase : nullptr;
  const int act = relu ? 1 : epi.act;
  const float slope = (act >= 2) ? epi.slope[0] : 0.f;
  float sg = 0.f;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int h = h0 + (wave * MT + mt) * RW + pr, ww = w0 + pc;
    if (h >= H || ww >= W) continue;
    const size_t po = ((size_t)h * W + ww) * Co_full;
    bf16_t* yp = yb + po;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = nt * 32 + 8 * q + 4 * half;
        if (co >= Co) continue;
        bf16x4 o, rv, av;
        if (act == 3) rv = *reinterpret_cast<const bf16x4*>(refb + po + co);
        if (addb) av = *reinterpret_cast<const bf16x4*>(addb + po + co);
        if (act == 2 && y2b) {
          bf16x4 pre;
#pragma unroll
          for (int e = 0; e < 4; ++e) pre[e] = (bf16_t)(acc[mt][nt][4 * q + e] + (bz ? (float)bz[co + e] : 0.f));
          *reinterpret_cast<bf16x4*>(y2b + po + co) = pre;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[mt][nt][4 * q + e] + (bz ? (float)bz[co + e] : 0.f);
          if (act == 1) v = fmaxf(v, 0.f);
          else if (act == 2) { v = (float)(bf16_t)v; v = v > 0.f ? v : slope * v; }           // PReLU of the ROUNDED pre-activation (what y2 holds)
          else if (act == 3) {
            const float r = (float)rv[e];
            v = (float)(bf16_t)v;                                                               // the gradient tensor the unfused path stored
            if (!(r > 0.f)) { sg += v * r; v *= slope; }
          }
          if (addb) v += (float)av[e];
          o[e] = (bf16_t)v;
        }
        *reinterpret_cast<bf16x4*>(yp + co) = o;
      }
  }
  if (act == 3 && epi.sgrad) {      // slope gradient: one atomic per workgroup (uniform branch; the LDS tiles are dead here)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    sg = wave_sum(sg);
    if (lane == 0) red[wave] = sg;
    __syncthreads();
    if (tid == 0) atomicAdd(epi.sgrad, red[0] + red[1] + red[2] + red[3]);
  }
}

// w [Z][Co][3][3][Ci] -> wt [Z][Ci][3][3][Co], taps flipped: wt[z][ci][ky][kx][co] = w[z][co][2-ky][2-kx][ci]
__global__ __launch_bounds__(256) void conv3x3_flip_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ wt, int Co, int Ci, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int co = (int)(i % Co);
  long r = i / Co;
  const int tap = (int)(r % 9);
  r /= 9;
  const int ci = (int)(r % Ci);
  const long z = r / Ci;
  wt[i] = w[((z * Co + co) * 9 + (8 - tap)) * Ci + ci];
}

// ------------------------------------------------------------------------------------------------ backward / weights
constexpr int lds_stride_for(int c) {   // row stride (elements) of a pixel-major tile: >= roundup32(c) and == 32 mod 64, so that the
  int s = (c + 31) / 32 * 32;           // 4 rows x 64 B a 32-lane half reads per transposing load fall into disjoint bank groups
  return (s % 64 == 0) ? s + 32 : s;
}

// grid (P pixel splits, ceil(Co/32) output-channel tiles x NS filter-row groups, Z).  Workgroup: 32 output channels x the
// (9/NS)*ceil(CI/32) (tap, 32-channel block) tiles of its filter rows, distributed round-robin over the 4 waves; loops over its
// share of TH x TW pixel tiles, K = 16 pixels per MFMA.  NS = 3 when CI > 32 (more workgroups for the same partial-sum volume).
// partial [Z][MTt][P][32 co][9][CB*32 ci] fp32 (+ bias partial [Z][MTt][P][32], written by the ky-group 0 workgroups).
// The operands of convolution z come from a by-value table: the Z convolutions of one call (strided views of one tensor) or the
// convolutions of SEPARATE calls gathered by the deferred weight-gradient phase (dgtd_conv3x3_wgrad_batched).  Several table entries
// may feed the same weight (a module called several times per step): slot[z] is the weight, pbase[z] the first of this entry's P
// partial rows among the Ptot rows of that weight, and the reduce kernel sums all of them - the per-call gradients are never formed.
constexpr int WG_MAX = 32;
struct WgTab { const void* x[WG_MAX]; const void* dy[WG_MAX]; const void* mask[WG_MAX]; int slot[WG_MAX]; int pbase[WG_MAX]; };
struct WgOut { void* dw[WG_MAX]; void* db[WG_MAX]; };

template <typename T, int CI, int TH, int TW>
__global__ __launch_bounds__(256) void conv3x3_wgrad_kernel(WgTab tab, float* __restrict__ partial,
                                                            float* __restrict__ partial_b, int B, int H, int W, int Co,
                                                            int tiles_w, int tiles_h, int Ptot) {
  constexpr int G = CI / 8, CB = (CI + 31) / 32, XS = lds_stride_for(CI), LW = TW + 2, LP = (TH + 2) * LW, NPIX = TH * TW;
  constexpr int NS = CI > 32 ? 3 : 1, NTN = 9 * CB / NS, TPW = (NTN + 3) / 4, KS = NPIX / 16, KPR = TW / 16;
  typedef typename Vec16<T>::type bf16x8;
  typedef T bf16_t;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  bf16_t* xt = reinterpret_cast<bf16_t*>(smem);          // [LP][XS]
  bf16_t* dt = xt + LP * XS;                             // [NPIX][32]
  __shared__ float bred[8][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int p = blockIdx.x, P = gridDim.x, mt = blockIdx.y / NS, kg = blockIdx.y % NS, MTt = gridDim.y / NS, z = blockIdx.z;
  const int ntiles = B * tiles_h * tiles_w;
  const bf16_t* __restrict__ x = (const bf16_t*)tab.x[z];
  const bf16_t* __restrict__ dy = (const bf16_t*)tab.dy[z];
  const bf16_t* __restrict__ mask = (const bf16_t*)tab.mask[z];
  const int slot = tab.slot[z], prow = tab.pbase[z] + p;
  f32x16 acc[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float bsum = 0.f;
  if (CI % 32) {   // pad columns of the x tile are read by the transposing loads: keep them finite
    for (int c = tid; c < LP * (CB * 32 - CI) / 8; c += 256) {
      const int pix = c / ((CB * 32 - CI) / 8), g = c % ((CB * 32 - CI) / 8);
      *reinterpret_cast<bf16x8*>(xt + pix * XS + CI + g * 8) = zero8<T>();
    }
  }
  // The next pixel tile is fetched into registers while the MFMAs of the current one run (one workgroup per CU: nothing else
  // would hide the global-load latency), then moved to LDS between two barriers.
  constexpr int NX = (LP * G + 255) / 256, ND = (NPIX * 4 + 255) / 256;
  bf16x8 xr[NX], dr[ND];
  auto fetch = [&](int t) {
    const int tw = t % tiles_w, th = (t / tiles_w) % tiles_h, b = t / (tiles_w * tiles_h);
    const int h0 = th * TH, w0 = tw * TW;
    const size_t img = (size_t)b * H * W;
    const bf16_t* xb = x + img * CI;
    const bf16_t* db = dy + img * Co;
    const bf16_t* mb = mask ? mask + img * Co : nullptr;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = tid + i * 256;
      const int pix = c / G, g = c % G;
      const int pr = pix / LW, pc = pix % LW;
      const int h = h0 - 1 + pr, ww = w0 - 1 + pc;
      xr[i] = zero8<T>();
      if (c < LP * G && h >= 0 && h < H && ww >= 0 && ww < W) xr[i] = *reinterpret_cast<const bf16x8*>(xb + ((size_t)h * W + ww) * CI + g * 8);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int c = tid + i * 256;
      const int pix = c >> 2, q = c & 3;
      const int h = h0 + pix / TW, ww = w0 + pix % TW, co = mt * 32 + q * 8;
      dr[i] = zero8<T>();
      if (c < NPIX * 4 && h < H && ww < W && co < Co) dr[i] = load_masked(db, mb, ((size_t)h * W + ww) * Co + co);
    }
  };
  if (p < ntiles) fetch(p);
  for (int t = p; t < ntiles; t += P) {
    __syncthreads();   // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int c = tid + i * 256;
      if (c < LP * G) *reinterpret_cast<bf16x8*>(xt + (c / G) * XS + (c % G) * 8) = xr[i];
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int c = tid + i * 256;
      if (c < NPIX * 4) *reinterpret_cast<bf16x8*>(dt + (c >> 2) * 32 + (c & 3) * 8) = dr[i];
    }
    __syncthreads();
    if (t + P < ntiles) fetch(t + P);
    if (kg == 0) {   // bias gradient: thread (co = tid & 31, slice = tid >> 5) sums every 8th pixel of the dy tile
      const int co = tid & 31, sl = tid >> 5;
      float s = 0.f;
      for (int pix = sl; pix < NPIX; pix += 8) s += (float)dt[pix * 32 + co];
      bsum += s;
    }
#pragma unroll 2
    for (int ks = 0; ks < KS; ++ks) {
      const int r = ks / KPR, c0 = (ks % KPR) * 16;
      const bf16x8 a = lds_tr_frag(dt, 32, r * TW + c0, 0, lane);
#pragma unroll
      for (int i = 0; i < TPW; ++i) {
        const int tl = wave + 4 * i;
        if (tl < NTN) {
          const int tap = kg * (9 / NS) + tl / CB, cb = tl % CB;
          const bf16x8 bfr = lds_tr_frag(xt, XS, (r + tap / 3) * LW + c0 + tap % 3, cb * 32, lane);
          acc[i] = mfma16(a, bfr, acc[i]);
        }
      }
    }
  }
  float* pp = partial + ((((size_t)slot * MTt + mt) * Ptot + prow) * 32) * 9 * (CB * 32);
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int tl = wave + 4 * i;
    if (tl < NTN) {
      const int tap = kg * (9 / NS) + tl / CB, cb = tl % CB;
#pragma unroll
      for (int r = 0; r < 16; ++r) pp[((size_t)mfma_row(r, half) * 9 + tap) * (CB * 32) + cb * 32 + l31] = acc[i][r];
    }
  }
  bred[tid >> 5][tid & 31] = bsum;
  __syncthreads();
  if (kg == 0 && tid < 32) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += bred[k][tid];
    partial_b[(((size_t)slot * MTt + mt) * Ptot + prow) * 32 + tid] = s;
  }
}

// dw [Z][Co][9][CI] bf16 and db [Z][Co] bf16 from the partials (fixed summation order)
template <typename T>
__global__ __launch_bounds__(256) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ partial_b,
                                                                   WgOut out, int Z, int Co, int CI, int CB, int MTt, int P) {
  const long n = (long)Z * Co * 9 * CI;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const int ci = (int)(i % CI);
    long r = i / CI;
    const int tap = (int)(r % 9);
    r /= 9;
    const int co = (int)(r % Co);
    const long z = r / Co;
    const int mt = co >> 5, cl = co & 31;
    const float* pp = partial + (((size_t)z * MTt + mt) * P * 32 + cl) * 9 * (CB * 32) + (size_t)tap * (CB * 32) + ci;
    const size_t ps = (size_t)32 * 9 * (CB * 32);
    float s0 = 0.f, s1 = 0.f;
    int q = 0;
    for (; q + 7 < P; q += 8) {      // 8 independent loads in flight, fixed summation order
      const float v0 = pp[q * ps], v1 = pp[(q + 1) * ps], v2 = pp[(q + 2) * ps], v3 = pp[(q + 3) * ps];
      const float v4 = pp[(q + 4) * ps], v5 = pp[(q + 5) * ps], v6 = pp[(q + 6) * ps], v7 = pp[(q + 7) * ps];
      s0 += (v0 + v1) + (v2 + v3);
      s1 += (v4 + v5) + (v6 + v7);
    }
    for (; q < P; ++q) s0 += pp[q * ps];
    ((T*)out.dw[z])[i - z * ((long)Co * 9 * CI)] = (T)(s0 + s1);
  }
  if (i < (long)Z * Co && out.db[i / Co]) {
    const int co = (int)(i % Co);
    const long z = i / Co;
    const float* pb = partial_b + (((size_t)z * MTt + (co >> 5)) * P) * 32 + (co & 31);
    float s = 0.f;
    for (int q = 0; q < P; ++q) s += pb[(size_t)q * 32];
    ((T*)out.db[z])[co] = (T)s;
  }
}

struct FwdGeom { int mt, twx; };

inline FwdGeom fwd_geom(int Z, int B, int H, int W, int Ci) {
  FwdGeom g;
  g.twx = W >= 32 ? 32 : 16;
  const int rw = 32 / g.twx;
  // the largest tile that still gives >= 1024 workgroups (4 per CU), within 64 KB of LDS for the halo tile
  const int mt_cap = Ci <= 32 ? 4 : (Ci <= 64 ? 2 : 1);
  g.mt = 1;
  for (int mt = mt_cap; mt > 1; mt >>= 1) {
    const long wgs = (long)Z * B * cdiv(H, 4 * mt * rw) * cdiv(W, g.twx);
    if (wgs >= 1024) { g.mt = mt; break; }
  }
  return g;
}

template <typename T, int CI, int NT, int MT, int TWX>
int launch_fwd(const void* x, const void* mask, const void* w, const void* bias, void* y, int Z, int B, int H, int W, int Co, int relu,
               int shared_x, hipStream_t st, const ConvEpi& epi, int nsplit = 1) {
  constexpr int RW = 32 / TWX, TH = 4 * MT * RW, LP = (TH + 2) * (TWX + 2), G = CI / 8, WSZ = NT * 32 * (G | 1);
  constexpr size_t lds = ((size_t)G * LP + ((((size_t)G * LP + 2 * WSZ) * 16 <= 65536) ? 2 : 1) * WSZ) * 16;
  static_assert(lds <= 65536, "halo tile + kernel slices exceed 64 KB of LDS");
  const int tiles_w = (int)cdiv(W, TWX), tiles_h = (int)cdiv(H, TH);
  const long plane = (long)B * H * W;
  hipLaunchKernelGGL((conv3x3_fwd_kernel<T, CI, NT, MT, TWX>), dim3(tiles_w * tiles_h, B, Z * nsplit), dim3(256), lds, st, (const T*)x,
                     (const T*)mask, (const T*)w, (const T*)bias, (T*)y, H, W, Co, relu, tiles_w,
                     shared_x ? 0L : plane * CI, (long)Co * 9 * CI, plane * Co, nsplit, epi);
  DGTD_CHECK_LAUNCH("conv3x3_fwd");
  return 0;
}

template <typename T, int CI, int NT>
int dispatch_fwd_geom(const void* x, const void* mask, const void* w, const void* bias, void* y, int Z, int B, int H, int W, int Co,
                      int relu, int shared_x, hipStream_t st, const ConvEpi& epi) {
  const FwdGeom g = fwd_geom(Z, B, H, W, CI);
  if constexpr (NT > 1) {   // fewer than 2 workgroups per CU: one workgroup per 32-channel output tile
    const long wgs = (long)Z * B * cdiv(H, 4 * (32 / g.twx)) * cdiv(W, g.twx);
    static const bool split_on = !(getenv("DGTD_CONV3X3_SPLIT") && getenv("DGTD_CONV3X3_SPLIT")[0] == '0');
    if (split_on && wgs < 512) {
      if (g.twx == 16) return launch_fwd<T, CI, 1, 1, 16>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi, NT);
      return launch_fwd<T, CI, 1, 1, 32>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi, NT);
    }
  }
  if (g.twx == 16) return launch_fwd<T, CI, NT, 1, 16>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi);
  if constexpr (CI <= 32) { if (g.mt == 4) return launch_fwd<T, CI, NT, 4, 32>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi); }
  if constexpr (CI <= 64) { if (g.mt >= 2) return launch_fwd<T, CI, NT, 2, 32>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi); }
  return launch_fwd<T, CI, NT, 1, 32>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi);
}

bool supported(int Ci, int Co) {
  auto ok = [](int c) { return c == 24 || c == 32 || c == 64 || c == 96; };
  return ok(Ci) && ok(Co);
}

// n table entries feeding nslots weights (entry i -> slot[i]; every slot is fed by the same number of entries, n / nslots)
template <typename T, int CI, int TH, int TW>
int launch_wgrad_tab(const void* const* x, const void* const* dy, const void* const* mask, const int* slot, int n, void* const* dw,
                     void* const* db, int nslots, void* ws, int B, int H, int W, int Co, int P, hipStream_t st) {
  constexpr int CB = (CI + 31) / 32, XS = lds_stride_for(CI), LP = (TH + 2) * (TW + 2);
  constexpr size_t lds = ((size_t)LP * XS + (size_t)TH * TW * 32) * 2;
  static_assert(lds <= 65536, "wgrad tiles exceed 64 KB of LDS");
  DGTD_REQUIRE(n > 0 && n <= WG_MAX && nslots > 0 && n % nslots == 0, "conv3x3_wgrad: %d convolutions / %d weights per launch (at most %d, equal shares)",
               n, nslots, WG_MAX);
  const int tiles_w = (int)cdiv(W, TW), tiles_h = (int)cdiv(H, TH), MTt = (int)cdiv(Co, 32), per = n / nslots, Ptot = per * P;
  WgTab tab;
  WgOut out;
  int seen[WG_MAX];
  for (int i = 0; i < WG_MAX; ++i) { seen[i] = 0; out.dw[i] = nullptr; out.db[i] = nullptr; }
  for (int i = 0; i < WG_MAX; ++i) {
    const int j = i < n ? i : 0;
    tab.x[i] = x[j]; tab.dy[i] = dy[j]; tab.mask[i] = mask ? mask[j] : nullptr;
    if (i < n) {
      DGTD_REQUIRE(slot[i] >= 0 && slot[i] < nslots && seen[slot[i]] < per, "conv3x3_wgrad: bad weight slot %d of entry %d", slot[i], i);
      tab.slot[i] = slot[i]; tab.pbase[i] = seen[slot[i]]++ * P;
    } else { tab.slot[i] = 0; tab.pbase[i] = 0; }
  }
  for (int z = 0; z < nslots; ++z) { out.dw[z] = dw[z]; out.db[z] = db ? db[z] : nullptr; }
  float* partial = (float*)ws;
  float* partial_b = partial + (size_t)nslots * MTt * Ptot * 32 * 9 * (CB * 32);
  hipLaunchKernelGGL((conv3x3_wgrad_kernel<T, CI, TH, TW>), dim3(P, MTt * (CI > 32 ? 3 : 1), n), dim3(256), lds, st, tab, partial, partial_b, B, H, W,
                     Co, tiles_w, tiles_h, Ptot);
  DGTD_CHECK_LAUNCH("conv3x3_wgrad");
  const long cnt = (long)nslots * Co * 9 * CI;
  hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel<T>, dim3((int)cdiv(cnt, 256)), dim3(256), 0, st, (const float*)partial, (const float*)partial_b, out,
                     nslots, Co, CI, CB, MTt, Ptot);
  DGTD_CHECK_LAUNCH("conv3x3_wgrad_reduce");
  return 0;
}

template <typename T, int CI, int TH, int TW>
int launch_wgrad(const void* x, const void* dy, const void* mask, void* dw, void* db, void* ws, int Z, int B, int H, int W, int Co,
                 int shared_x, int P, hipStream_t st) {
  DGTD_REQUIRE(Z <= WG_MAX, "conv3x3_wgrad: Z=%d convolutions per call (at most %d)", Z, WG_MAX);
  const size_t plane = (size_t)B * H * W;
  const void *xs[WG_MAX], *dys[WG_MAX], *ms[WG_MAX];
  void *dws[WG_MAX], *dbs[WG_MAX];
  int slots[WG_MAX];
  for (int z = 0; z < Z; ++z) {
    xs[z] = (const T*)x + (shared_x ? 0 : (size_t)z * plane * CI);
    dys[z] = (const T*)dy + (size_t)z * plane * Co;
    ms[z] = mask ? (const T*)mask + (size_t)z * plane * Co : nullptr;
    dws[z] = (T*)dw + (size_t)z * Co * 9 * CI;
    dbs[z] = db ? (T*)db + (size_t)z * Co : nullptr;
    slots[z] = z;
  }
  return launch_wgrad_tab<T, CI, TH, TW>(xs, dys, mask ? ms : nullptr, slots, Z, dws, db ? dbs : nullptr, Z, ws, B, H, W, Co, P, st);
}

inline int wgrad_splits(int Z, int B, int H, int W, int Ci, int Co) {
  const int tw = W >= 32 ? 32 : 16, th = Ci >= 64 ? 4 : 8;
  const long ntiles = (long)B * cdiv(H, th) * cdiv(W, tw);
  const long ygroups = (long)Z * cdiv(Co, 32) * (Ci > 32 ? 3 : 1);
  // ~2 workgroups per CU; 96 -> 96 channels (9 output tiles x 3 filter-row groups per pixel split) runs better with ~4 (measured:
  // 16 convolutions at 64x64 343 -> 304 us; every other geometry of the model is slower with more partial sums)
  static const long wgs_env = getenv("DGTD_WGRAD_WGS") ? atol(getenv("DGTD_WGRAD_WGS")) : 0;
  const long wgs = wgs_env ? wgs_env : (Ci >= 96 && Co >= 96 ? 1024 : 512);
  const long want = std::max<long>(1, wgs / ygroups);
  // the partial sums written (and re-read by the reduce kernel) should stay below the bytes of the two input maps
  const long in_bytes = 2L * ((long)Z * B * H * W * (Ci + Co)), part_bytes = (long)Z * cdiv(Co, 32) * 32 * 9 * ((Ci + 31) / 32 * 32) * 4;
  const long cap = std::max<long>(std::max<long>(4, in_bytes / part_bytes), (16L << 20) / part_bytes);   // small maps: up to 16 MB of partials
  return (int)std::max<long>(1, std::min<long>(std::min<long>(ntiles, want), std::min<long>(cap, 256)));
}

}  // namespace

extern "C" int dgtd_conv3x3_supported(int Ci, int Co, int H, int W) { return supported(Ci, Co) && H >= 1 && W >= 16 && W % 16 == 0; }

static int conv3x3_fwd_impl(const void* x, const void* mask, const void* w, const void* bias, void* y, int Z, int B, int H, int W, int Ci, int Co,
                            int relu, int shared_x, dgtd_dtype dt, dgtd_stream s, const ConvEpi& epi) {
  // roofline label: 18 Ci Co flop per pixel over 2 (Ci + Co) bytes; above the ridge (2.5 PF / 8 TB/s = 312 flop/B; 96 -> 96: 432) the
  // call is priced against the MFMA peak, below it against HBM
  const int extra = (epi.add ? 1 : 0) + (epi.ref ? 1 : 0) + (epi.y2 ? 1 : 0);
  const double conv_bytes = 2.0 * B * H * W * ((shared_x ? 1.0 : (double)Z) * Ci * (mask ? 2 : 1) + (double)Z * Co * (1 + extra));
  const double conv_flops = 18.0 * Z * B * H * W * (double)Ci * Co;
  const bool conv_mfma = conv_flops > 312.5 * conv_bytes;
  // the epilogue operands change the bytes (and with them the side of the ridge), so they are part of the key
  DGTD_PROF(s, conv_mfma ? DGTD_MFMA : DGTD_HBM, conv_mfma ? conv_flops : conv_bytes, extra ? "dgtd_conv3x3_fwd[Z=%d,%dx%d,%d->%d,+%d]" : "dgtd_conv3x3_fwd[Z=%d,%dx%d,%d->%d]",
            Z, H, W, Ci, Co, extra);
  DGTD_REQUIRE(Z > 0 && B > 0 && H > 0 && W > 0, "conv3x3_fwd: bad sizes");
  DGTD_REQUIRE(DGTD_IS_HALF(dt), "conv3x3_fwd: dtype %d (the kernel is bf16 / fp16 only)", (int)dt);
  DGTD_REQUIRE(dgtd_conv3x3_supported(Ci, Co, H, W), "conv3x3_fwd: unsupported geometry Ci=%d Co=%d H=%d W=%d", Ci, Co, H, W);
  DGTD_REQUIRE(!(shared_x && mask), "conv3x3_fwd: a mask needs its own input per convolution");
  DGTD_REQUIRE(epi.act >= 0 && epi.act <= 3 && !(relu && epi.act > 1), "conv3x3_fwd: bad activation %d", epi.act);
  DGTD_REQUIRE(epi.act < 2 || epi.slope, "conv3x3_fwd: PReLU needs the slope");
  DGTD_REQUIRE(epi.act != 3 || epi.ref, "conv3x3_fwd: the PReLU backward epilogue needs the saved pre-activation");
  hipStream_t st = (hipStream_t)s;
  const int nt = (int)cdiv(Co, 32);
#define DGTD_CONV_CASE(CI_, NT_) if (Ci == CI_ && nt == NT_) DGTD_DISPATCH_HALF(dt, return (dispatch_fwd_geom<T_, CI_, NT_>(x, mask, w, bias, y, Z, B, H, W, Co, relu, shared_x, st, epi)));
  DGTD_CONV_CASE(24, 1) DGTD_CONV_CASE(24, 2) DGTD_CONV_CASE(24, 3)
  DGTD_CONV_CASE(32, 1) DGTD_CONV_CASE(32, 2) DGTD_CONV_CASE(32, 3)
  DGTD_CONV_CASE(64, 1) DGTD_CONV_CASE(64, 2) DGTD_CONV_CASE(64, 3)
  DGTD_CONV_CASE(96, 1) DGTD_CONV_CASE(96, 2) DGTD_CONV_CASE(96, 3)
#undef DGTD_CONV_CASE
  DGTD_FAIL(2, "conv3x3_fwd: no kernel for Ci=%d Co=%d", Ci, Co);
}

extern "C" int dgtd_conv3x3_fwd(const void* x, const void* mask, const void* w, const void* bias, void* y, int Z, int B, int H, int W,
                                int Ci, int Co, int relu, int shared_x, dgtd_dtype dt, dgtd_stream s) {
  return conv3x3_fwd_impl(x, mask, w, bias, y, Z, B, H, W, Ci, Co, relu, shared_x, dt, s, ConvEpi{nullptr, nullptr, nullptr, nullptr, nullptr, 0});
}

extern "C" int dgtd_conv3x3_fwd_ex(const void* x, const void* mask, const void* w, const void* bias, void* y, void* y2, const void* add, const void* ref,
                                   const float* slope, float* slope_grad, int Z, int B, int H, int W, int Ci, int Co, int act, int shared_x,
                                   dgtd_dtype dt, dgtd_stream s) {
  return conv3x3_fwd_impl(x, mask, w, bias, y, Z, B, H, W, Ci, Co, 0, shared_x, dt, s, ConvEpi{add, ref, y2, slope, slope_grad, act});
}

extern "C" int dgtd_conv3x3_flip(const void* w, void* wt, int Z, int Co, int Ci, dgtd_stream s) {
  DGTD_REQUIRE(Z > 0 && Co > 0 && Ci > 0, "conv3x3_flip: bad sizes");
  const long n = (long)Z * Co * 9 * Ci;
  hipLaunchKernelGGL(conv3x3_flip_kernel, dim3((int)cdiv(n, 256)), dim3(256), 0, (hipStream_t)s, (const uint16_t*)w, (uint16_t*)wt, Co, Ci, n);
  DGTD_CHECK_LAUNCH("conv3x3_flip");
  return 0;
}

extern "C" int64_t dgtd_conv3x3_wgrad_workspace(int Z, int B, int H, int W, int Ci, int Co) {
  const int P = wgrad_splits(Z, B, H, W, Ci, Co), CB = (Ci + 31) / 32, MTt = (int)cdiv(Co, 32);
  return (int64_t)Z * MTt * P * 32 * (9 * CB * 32 + 1) * (int64_t)sizeof(float);
}

extern "C" int64_t dgtd_conv3x3_wgrad_batched_workspace(int n, int B, int H, int W, int Ci, int Co) {
  return dgtd_conv3x3_wgrad_workspace(n, B, H, W, Ci, Co);      // n * P partial row sets in total, however they are shared out
}

extern "C" int dgtd_conv3x3_wgrad_batched(const void* const* x, const void* const* dy, const void* const* mask, const int* slot, int n,
                                          void* const* dw, void* const* db, int nslots, void* workspace, int B, int H, int W, int Ci, int Co,
                                          dgtd_dtype dt, dgtd_stream s) {
  DGTD_REQUIRE(n > 0 && nslots > 0 && x && dy && slot && dw && workspace, "conv3x3_wgrad_batched: no convolutions");
  DGTD_PROF(s, DGTD_HBM, 2.0 * n * B * H * W * ((double)Ci + (double)Co * (mask ? 2 : 1)), "dgtd_conv3x3_wgrad_batched[n%d/%d,%dx%d,%d->%d]", n, nslots, H, W, Ci, Co);
  DGTD_REQUIRE(B > 0 && H > 0 && W > 0, "conv3x3_wgrad_batched: bad sizes");
  DGTD_REQUIRE(DGTD_IS_HALF(dt), "conv3x3_wgrad_batched: dtype %d (the kernel is bf16 / fp16 only)", (int)dt);
  DGTD_REQUIRE(dgtd_conv3x3_supported(Ci, Co, H, W), "conv3x3_wgrad_batched: unsupported geometry Ci=%d Co=%d H=%d W=%d", Ci, Co, H, W);
  hipStream_t st = (hipStream_t)s;
  const int P = wgrad_splits(n, B, H, W, Ci, Co);
  const bool wide = W >= 32;
#define DGTD_WG_CASE(CI_, TH_) if (Ci == CI_) DGTD_DISPATCH_HALF(dt, return wide ? (launch_wgrad_tab<T_, CI_, TH_, 32>(x, dy, mask, slot, n, dw, db, nslots, workspace, B, H, W, Co, P, st)) \
                                                         : (launch_wgrad_tab<T_, CI_, TH_, 16>(x, dy, mask, slot, n, dw, db, nslots, workspace, B, H, W, Co, P, st)));
  DGTD_WG_CASE(24, 8) DGTD_WG_CASE(32, 8) DGTD_WG_CASE(64, 4) DGTD_WG_CASE(96, 4)
#undef DGTD_WG_CASE
  DGTD_FAIL(2, "conv3x3_wgrad_batched: no kernel for Ci=%d", Ci);
}

extern "C" int dgtd_conv3x3_wgrad(const void* x, const void* dy, const void* mask, void* dw, void* db, void* workspace, int Z, int B,
                                  int H, int W, int Ci, int Co, int shared_x, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 2.0 * B * H * W * ((shared_x ? 1.0 : (double)Z) * Ci + (double)Z * Co * (mask ? 2 : 1)), "dgtd_conv3x3_wgrad[Z=%d,%dx%d,%d->%d]", Z, H, W, Ci, Co);
  DGTD_REQUIRE(Z > 0 && B > 0 && H > 0 && W > 0, "conv3x3_wgrad: bad sizes");
  DGTD_REQUIRE(DGTD_IS_HALF(dt), "conv3x3_wgrad: dtype %d (the kernel is bf16 / fp16 only)", (int)dt);
  DGTD_REQUIRE(dgtd_conv3x3_supported(Ci, Co, H, W), "conv3x3_wgrad: unsupported geometry Ci=%d Co=%d H=%d W=%d", Ci, Co, H, W);
  hipStream_t st = (hipStream_t)s;
  const int P = wgrad_splits(Z, B, H, W, Ci, Co);
  const bool wide = W >= 32;
#define DGTD_WG_CASE(CI_, TH_) if (Ci == CI_) DGTD_DISPATCH_HALF(dt, return wide ? (launch_wgrad<T_, CI_, TH_, 32>(x, dy, mask, dw, db, workspace, Z, B, H, W, Co, shared_x, P, st)) \
                                                         : (launch_wgrad<T_, CI_, TH_, 16>(x, dy, mask, dw, db, workspace, Z, B, H, W, Co, shared_x, P, st)));
  DGTD_WG_CASE(24, 8) DGTD_WG_CASE(32, 8) DGTD_WG_CASE(64, 4) DGTD_WG_CASE(96, 4)
#undef DGTD_WG_CASE
  DGTD_FAIL(2, "conv3x3_wgrad: no kernel for Ci=%d", Ci);
}

// ms_deform_attn.hip — multi-scale deformable attention sampling, forward and backward, for gfx950.
// Replaces the reference's own native op twig/ops (MSDA.ms_deform_attn_forward / _backward, twig/ops/src/ms_deform_attn.h:20-60;
// semantics = ms_deform_attn_core_pytorch, twig/ops/functions/ms_deform_attn_func.py:49-71):
//   out[n,q,m,:] = sum_{l,p} attn[n,q,m,l,p] * bilinear(value_l[n,:,m,:], loc[n,q,m,l,p])      (zero padding, align_corners=False)
// value [N,S,M,D] (S = sum_l H_l*W_l, level l starts at level_start[l]); spatial_shapes [L,2] = (H_l, W_l) int64;
// loc [N,Lq,M,L,P,2] = (x, y) in [0,1]; attn [N,Lq,M,L,P]; out [N,Lq,M*D].  Pixel coordinates: h = y*H - 0.5, w = x*W - 0.5.
// Gather-bound (HBM / L2).  Forward: one thread per 16 BYTES of output channels (4 fp32 / 2 fp64) when D allows it - the D/4 threads
// of a (n,q,m) share the sampling geometry (broadcast loads) and each corner is one 16-byte load per thread, a quarter of the load
// and address instructions of the one-channel-per-thread mapping (kept for odd D).  Backward: one workgroup per (n,q,m); value gradients by atomics (4 corners per
// sample), the location / weight gradients need a sum over d: wave shuffle + LDS across the waves of the workgroup instead of the
// reference's seven shared-memory reduction variants (ms_deform_im2col_cuda.cuh:302-920).  The backward deliberately keeps one
// channel per lane: the atomic units work per 128-byte line and instruction, and D consecutive lanes cover each corner's line with ONE
// atomic instruction; a 4-channels-per-lane variant (8 triples per wave) issued 4x the line requests and ran 3.5x slower (measured).
#include "common.h"

namespace {

template <typename S> struct Geom { int h0, w0; S lh, lw; bool ok, in00, in01, in10, in11; };

template <typename S>
__device__ __forceinline__ Geom<S> geom(S loc_x, S loc_y, int H, int W) {
  Geom<S> g;
  const S h = loc_y * (S)H - (S)0.5, w = loc_x * (S)W - (S)0.5;
  g.ok = h > (S)-1 && w > (S)-1 && h < (S)H && w < (S)W;
  const S hf = floor(h), wf = floor(w);
  g.h0 = (int)hf; g.w0 = (int)wf;
  g.lh = h - hf; g.lw = w - wf;
  const bool h0ok = g.h0 >= 0, h1ok = g.h0 + 1 <= H - 1, w0ok = g.w0 >= 0, w1ok = g.w0 + 1 <= W - 1;
  g.in00 = h0ok && w0ok; g.in01 = h0ok && w1ok; g.in10 = h1ok && w0ok; g.in11 = h1ok && w1ok;
  return g;
}

template <typename S>
__global__ __launch_bounds__(256) void msda_fwd_kernel(const S* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart, const S* __restrict__ loc,
                                                       const S* __restrict__ attn, S* __restrict__ out, int N, int Sv, int M, int D,
                                                       int L, int Lq, int P) {
  const int64_t total = (int64_t)N * Lq * M * D;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % D);
    int64_t r = i / D;                       // (n, q, m) flattened
    const int m = (int)(r % M);
    const int64_t nq = r / M;
    const int n = (int)(nq / Lq);
    const S* vb = value + ((size_t)n * Sv * M + m) * D + d;       // + s * M * D
    const S* lp = loc + (size_t)r * L * P * 2;
    const S* ap = attn + (size_t)r * L * P;
    S acc = 0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const S* vl = vb + (size_t)lstart[l] * M * D;
      for (int p = 0; p < P; ++p) {
        const Geom<S> g = geom<S>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W);
        if (!g.ok) continue;
        const size_t rs = (size_t)M * D;
        const S v00 = g.in00 ? vl[((size_t)g.h0 * W + g.w0) * rs] : (S)0, v01 = g.in01 ? vl[((size_t)g.h0 * W + g.w0 + 1) * rs] : (S)0;
        const S v10 = g.in10 ? vl[((size_t)(g.h0 + 1) * W + g.w0) * rs] : (S)0, v11 = g.in11 ? vl[((size_t)(g.h0 + 1) * W + g.w0 + 1) * rs] : (S)0;
        const S s = ((S)1 - g.lh) * (((S)1 - g.lw) * v00 + g.lw * v01) + g.lh * (((S)1 - g.lw) * v10 + g.lw * v11);
        acc += ap[l * P + p] * s;
      }
    }
    out[i] = acc;
  }
}

// V channels per thread (V * sizeof(S) == 16): same sampling arithmetic, vector corner loads
template <typename S, int V>
__global__ __launch_bounds__(256) void msda_fwd_vec_kernel(const S* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lstart, const S* __restrict__ loc,
                                                           const S* __restrict__ attn, S* __restrict__ out, int N, int Sv, int M, int D,
                                                           int L, int Lq, int P) {
  typedef S vec __attribute__((ext_vector_type(V)));
  const int DG = D / V;
  const int64_t total = (int64_t)N * Lq * M * DG;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i % DG) * V;
    const int64_t r = i / DG;                // (n, q, m) flattened
    const int m = (int)(r % M);
    const int n = (int)((r / M) / Lq);
    const S* vb = value + ((size_t)n * Sv * M + m) * D + d;
    const S* lp = loc + (size_t)r * L * P * 2;
    const S* ap = attn + (size_t)r * L * P;
    const size_t rs = (size_t)M * D;
    vec acc = (vec)(S)0;
    for (int l = 0; l < L; ++l) {
      const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
      const S* vl = vb + (size_t)lstart[l] * rs;
      for (int p = 0; p < P; ++p) {
        const Geom<S> g = geom<S>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W);
        if (!g.ok) continue;
        const vec zero = (vec)(S)0;
        const S* c00 = vl + ((size_t)g.h0 * W + g.w0) * rs;
        const vec v00 = g.in00 ? *reinterpret_cast<const vec*>(c00) : zero, v01 = g.in01 ? *reinterpret_cast<const vec*>(c00 + rs) : zero;
        const vec v10 = g.in10 ? *reinterpret_cast<const vec*>(c00 + (size_t)W * rs) : zero;
        const vec v11 = g.in11 ? *reinterpret_cast<const vec*>(c00 + (size_t)W * rs + rs) : zero;
        // same association as the scalar kernel: (1-lh)*((1-lw)*v00 + lw*v01) + lh*((1-lw)*v10 + lw*v11)
        const vec s = ((S)1 - g.lh) * (((S)1 - g.lw) * v00 + g.lw * v01) + g.lh * (((S)1 - g.lw) * v10 + g.lw * v11);
        acc += ap[l * P + p] * s;
      }
    }
    *reinterpret_cast<vec*>(out + r * D + d) = acc;
  }
}

template <typename S> __device__ __forceinline__ S wave_sum_t(S v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one workgroup per (n, q, m); threads stride over d
template <typename S>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const S* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart, const S* __restrict__ loc,
                                                       const S* __restrict__ attn, const S* __restrict__ gout, S* __restrict__ gvalue,
                                                       S* __restrict__ gloc, S* __restrict__ gattn, int N, int Sv, int M, int D, int L,
                                                       int Lq, int P) {
  __shared__ S red[3][4];
  const int64_t r = blockIdx.x;              // (n, q, m)
  const int m = (int)(r % M);
  const int n = (int)((r / M) / Lq);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const size_t rs = (size_t)M * D;
  const S* vb = value + ((size_t)n * Sv * M + m) * D;
  S* gvb = gvalue + ((size_t)n * Sv * M + m) * D;
  const S* go = gout + (size_t)r * D;
  const S* lp = loc + (size_t)r * L * P * 2;
  const S* ap = attn + (size_t)r * L * P;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const size_t lo = (size_t)lstart[l] * rs;
    for (int p = 0; p < P; ++p) {
      const Geom<S> g = geom<S>(lp[(l * P + p) * 2], lp[(l * P + p) * 2 + 1], H, W);
      const S a = ap[l * P + p];
      S sw = 0, sh = 0, sa = 0;
      if (g.ok) {                              // workgroup-uniform
        const size_t o00 = lo + ((size_t)g.h0 * W + g.w0) * rs, o01 = o00 + rs, o10 = o00 + (size_t)W * rs, o11 = o10 + rs;
        const S w00 = ((S)1 - g.lh) * ((S)1 - g.lw), w01 = ((S)1 - g.lh) * g.lw, w10 = g.lh * ((S)1 - g.lw), w11 = g.lh * g.lw;
        for (int d = tid; d < D; d += blockDim.x) {
          const S gd = go[d], ga = gd * a;
          const S v00 = g.in00 ? vb[o00 + d] : (S)0, v01 = g.in01 ? vb[o01 + d] : (S)0;
          const S v10 = g.in10 ? vb[o10 + d] : (S)0, v11 = g.in11 ? vb[o11 + d] : (S)0;
          if (g.in00) atomicAdd(gvb + o00 + d, ga * w00);
          if (g.in01) atomicAdd(gvb + o01 + d, ga * w01);
          if (g.in10) atomicAdd(gvb + o10 + d, ga * w10);
          if (g.in11) atomicAdd(gvb + o11 + d, ga * w11);
          sa += gd * (w00 * v00 + w01 * v01 + w10 * v10 + w11 * v11);
          sh += ga * (((S)1 - g.lw) * (v10 - v00) + g.lw * (v11 - v01));
          sw += ga * (((S)1 - g.lh) * (v01 - v00) + g.lh * (v11 - v10));
        }
      }
      sw = wave_sum_t(sw); sh = wave_sum_t(sh); sa = wave_sum_t(sa);
      __syncthreads();                         // red[] of the previous point has been consumed
      if (lane == 0) { red[0][wave] = sw; red[1][wave] = sh; red[2][wave] = sa; }
      __syncthreads();
      if (tid == 0) {
        S tw = 0, th = 0, ta = 0;
        for (int k = 0; k < nw; ++k) { tw += red[0][k]; th += red[1][k]; ta += red[2][k]; }
        gloc[((size_t)r * L * P + l * P + p) * 2] = tw * (S)W;
        gloc[((size_t)r * L * P + l * P + p) * 2 + 1] = th * (S)H;
        gattn[(size_t)r * L * P + l * P + p] = ta;
      }
    }
  }
}

bool check_sizes(int N, int S, int M, int D, int L, int Lq, int P) { return N > 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq > 0 && P > 0; }

}  // namespace

extern "C" int dgtd_ms_deform_attn_fwd(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const void* sampling_loc, const void* attn_weight, void* out, int N, int S, int M, int D, int L,
                                       int Lq, int P, dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (double)dgtd_esize(dt) * N * Lq * M * D * (4.0 * L * P + 1), "dgtd_ms_deform_attn_fwd[N=%d,Lq=%d,M=%d,D=%d,L=%d,P=%d]", N, Lq, M, D, L, P);
  DGTD_REQUIRE(check_sizes(N, S, M, D, L, Lq, P), "ms_deform_attn_fwd: bad sizes");
  const int V = dt == DGTD_F32 ? 4 : 2;
  if ((dt == DGTD_F32 || dt == DGTD_F64) && D % V == 0 && ((uintptr_t)value % 16 == 0) && ((uintptr_t)out % 16 == 0)) {
    const int64_t vtotal = (int64_t)N * Lq * M * (D / V);
    const int vgrid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(vtotal, 256), 65536));
    if (dt == DGTD_F32) hipLaunchKernelGGL((msda_fwd_vec_kernel<float, 4>), dim3(vgrid), dim3(256), 0, (hipStream_t)s, (const float*)value, spatial_shapes, level_start_index, (const float*)sampling_loc, (const float*)attn_weight, (float*)out, N, S, M, D, L, Lq, P);
    else hipLaunchKernelGGL((msda_fwd_vec_kernel<double, 2>), dim3(vgrid), dim3(256), 0, (hipStream_t)s, (const double*)value, spatial_shapes, level_start_index, (const double*)sampling_loc, (const double*)attn_weight, (double*)out, N, S, M, D, L, Lq, P);
    DGTD_CHECK_LAUNCH("ms_deform_attn_fwd");
    return 0;
  }
  const int64_t total = (int64_t)N * Lq * M * D;
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(total, 256), 65536));
  if (dt == DGTD_F32) hipLaunchKernelGGL(msda_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)value, spatial_shapes, level_start_index, (const float*)sampling_loc, (const float*)attn_weight, (float*)out, N, S, M, D, L, Lq, P);
  else if (dt == DGTD_F64) hipLaunchKernelGGL(msda_fwd_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const double*)value, spatial_shapes, level_start_index, (const double*)sampling_loc, (const double*)attn_weight, (double*)out, N, S, M, D, L, Lq, P);
  else DGTD_FAIL(2, "ms_deform_attn_fwd: dtype %d (float32 or float64; the reference casts half inputs to float32)", (int)dt);
  DGTD_CHECK_LAUNCH("ms_deform_attn_fwd");
  return 0;
}

extern "C" int dgtd_ms_deform_attn_bwd(const void* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                       const void* sampling_loc, const void* attn_weight, const void* grad_out, void* grad_value,
                                       void* grad_sampling_loc, void* grad_attn_weight, int N, int S, int M, int D, int L, int Lq, int P,
                                       dgtd_dtype dt, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, (double)dgtd_esize(dt) * N * Lq * M * D * (8.0 * L * P + 1), "dgtd_ms_deform_attn_bwd[N=%d,Lq=%d,M=%d,D=%d,L=%d,P=%d]", N, Lq, M, D, L, P);
  DGTD_REQUIRE(check_sizes(N, S, M, D, L, Lq, P), "ms_deform_attn_bwd: bad sizes");
  const int64_t blocks = (int64_t)N * Lq * M;
  DGTD_REQUIRE(blocks < (1LL << 31), "ms_deform_attn_bwd: too many (n, q, m) triples");
  const int threads = (int)std::min<int64_t>(256, cdiv(D, 64) * 64);
  if (dt == DGTD_F32) hipLaunchKernelGGL(msda_bwd_kernel<float>, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)s, (const float*)value, spatial_shapes, level_start_index, (const float*)sampling_loc, (const float*)attn_weight, (const float*)grad_out, (float*)grad_value, (float*)grad_sampling_loc, (float*)grad_attn_weight, N, S, M, D, L, Lq, P);
  else if (dt == DGTD_F64) hipLaunchKernelGGL(msda_bwd_kernel<double>, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)s, (const double*)value, spatial_shapes, level_start_index, (const double*)sampling_loc, (const double*)attn_weight, (const double*)grad_out, (double*)grad_value, (double*)grad_sampling_loc, (double*)grad_attn_weight, N, S, M, D, L, Lq, P);
  else DGTD_FAIL(2, "ms_deform_attn_bwd: dtype %d (float32 or float64)", (int)dt);
  DGTD_CHECK_LAUNCH("ms_deform_attn_bwd");
  return 0;
}

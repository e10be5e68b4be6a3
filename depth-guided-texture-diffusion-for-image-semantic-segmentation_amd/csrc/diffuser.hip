// diffuser.hip — the depth-guided texture diffuser front end (fwd + bwd), fp32 throughout.
// Replaces twig/model/cod.py:1295-1298 (nearest 12x12 sample of the FFT high-pass image -> 1x1 conv 3->1176 ->
// sigmoid; depth -> 1x1 conv 1->24 -> bilinear 12x12), MessagePassing.forward cod.py:1189-1208 (random-walk
// normalisation, 4 zero-padded 7x7 propagation steps, 1x1 conv 24->3, bilinear 12->S) and the `+ image` of cod.py:1302.
//
// Facts used:
//   * the 24 latent channels propagate independently (a per-pixel dynamic depthwise conv) — only the final 1x1
//     conv mixes them -> one workgroup per (image, channel), its 49x144 normalised weights live in LDS (28 KB);
//   * bilinear_12(conv1x1(depth)) == conv1x1(bilinear_12(depth)) (bilinear weights sum to 1), so the reference's
//     [B,24,S,S] fp32 depth embedding (201 MB at batch 8, 512^2) is never materialised: 144 samples per image;
//   * reference tensor order: regressor channel = c*49 + ky*7 + kx = F.unfold's (C*k*k, L) order (cod.py:1193,1204).
// The state kernel is latency-bound (tiny); the tail kernel is HBM-bound: algorithmic bytes = 2*4*B*3*S^2.
#include "common.h"

namespace {

constexpr int G = 12, P = 144, K = 7, T = 49, LAT = 24, STEPS = 4;

// PyTorch upsample index rules (aten/src/ATen/native/UpSample.h): scale = (float)in / out
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) { return min((int)floorf(dst * scale), in - 1); }
__device__ __forceinline__ void bilinear_src(int dst, float scale, int in, int& i0, int& i1, float& l1) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

struct DiffSmem {
  float xx[3][P];
  float d12[P];
  float Wn[T][P];     // normalised propagation weights
  float inv[P];       // 1 / (sum_t W + eps)
  float st[STEPS + 1][P];
};

// shared forward body: fills s.xx, s.d12, s.Wn, s.inv, s.st[0..4]
__device__ void diffuser_forward_body(DiffSmem& s, const float* __restrict__ x_hp, const float* __restrict__ depth,
                                      const float* __restrict__ reg_w, const float* __restrict__ reg_b,
                                      const float* __restrict__ enc_w, const float* __restrict__ enc_b, int S, int b, int c) {
  const int tid = threadIdx.x;
  const float scale = (float)S / (float)G;
  if (tid < P) {
    const int py = tid / G, px = tid % G;
    const int sy = nearest_src(py, scale, S), sx = nearest_src(px, scale, S);
#pragma unroll
    for (int i = 0; i < 3; ++i) s.xx[i][tid] = x_hp[(((size_t)b * 3 + i) * S + sy) * S + sx];
    int y0, y1, x0, x1; float ly, lx;
    bilinear_src(py, scale, S, y0, y1, ly);
    bilinear_src(px, scale, S, x0, x1, lx);
    const float* d = depth + (size_t)b * S * S;
    const float v = (1.f - ly) * ((1.f - lx) * d[(size_t)y0 * S + x0] + lx * d[(size_t)y0 * S + x1]) +
                    ly * ((1.f - lx) * d[(size_t)y1 * S + x0] + lx * d[(size_t)y1 * S + x1]);
    s.d12[tid] = v;
    s.st[0][tid] = enc_w[c] * v + enc_b[c];
  }
  __syncthreads();
  for (int i = tid; i < T * P; i += blockDim.x) {
    const int t = i / P, p = i % P;
    const float* w = reg_w + (size_t)(c * T + t) * 3;
    const float z = w[0] * s.xx[0][p] + w[1] * s.xx[1][p] + w[2] * s.xx[2][p] + reg_b[c * T + t];
    s.Wn[t][p] = 1.f / (1.f + expf(-z));
  }
  __syncthreads();
  if (tid < P) {
    float sum = 0.f;
    for (int t = 0; t < T; ++t) sum += s.Wn[t][tid];
    s.inv[tid] = 1.f / (sum + 1e-5f);
  }
  __syncthreads();
  for (int i = tid; i < T * P; i += blockDim.x) s.Wn[i / P][i % P] *= s.inv[i % P];
  __syncthreads();
  for (int step = 0; step < STEPS; ++step) {
    if (tid < P) {
      const int py = tid / G, px = tid % G;
      float acc = 0.f;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int yy = py + ky - K / 2;
        if (yy < 0 || yy >= G) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int xq = px + kx - K / 2;
          if (xq < 0 || xq >= G) continue;
          acc += s.Wn[ky * K + kx][tid] * s.st[step][yy * G + xq];
        }
      }
      s.st[step + 1][tid] = acc;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void diffuser_fwd_kernel(const float* __restrict__ x_hp, const float* __restrict__ depth,
                                                           const float* __restrict__ reg_w, const float* __restrict__ reg_b,
                                                           const float* __restrict__ enc_w, const float* __restrict__ enc_b,
                                                           float* __restrict__ x4, int S) {
  __shared__ DiffSmem s;
  const int c = blockIdx.x, b = blockIdx.y;
  diffuser_forward_body(s, x_hp, depth, reg_w, reg_b, enc_w, enc_b, S, b, c);
  if (threadIdx.x < P) x4[((size_t)b * LAT + c) * P + threadIdx.x] = s.st[STEPS][threadIdx.x];
}

// backward w.r.t. the parameters (inputs carry no gradient).  d_* are accumulated with atomics (zeroed by the caller).
__global__ __launch_bounds__(256) void diffuser_bwd_kernel(const float* __restrict__ x_hp, const float* __restrict__ depth,
                                                           const float* __restrict__ reg_w, const float* __restrict__ reg_b,
                                                           const float* __restrict__ enc_w, const float* __restrict__ enc_b,
                                                           const float* __restrict__ g4, float* __restrict__ d_reg_w,
                                                           float* __restrict__ d_reg_b, float* __restrict__ d_enc_w,
                                                           float* __restrict__ d_enc_b, int S) {
  __shared__ DiffSmem s;
  __shared__ float dW[T][P];
  __shared__ float g[2][P];
  __shared__ float red[2][4];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  diffuser_forward_body(s, x_hp, depth, reg_w, reg_b, enc_w, enc_b, S, b, c);
  for (int i = tid; i < T * P; i += blockDim.x) dW[i / P][i % P] = 0.f;
  if (tid < P) g[0][tid] = g4[((size_t)b * LAT + c) * P + tid];
  __syncthreads();
  int cur = 0;
  for (int step = STEPS - 1; step >= 0; --step) {
    if (tid < P) {
      const int py = tid / G, px = tid % G;
      const float gp = g[cur][tid];
      float acc = 0.f;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int t = ky * K + kx;
          // dWn[t][p] += g_{s+1}[p] * x_s[p + off(t)]
          const int yy = py + ky - K / 2, xq = px + kx - K / 2;
          if (yy >= 0 && yy < G && xq >= 0 && xq < G) dW[t][tid] += gp * s.st[step][yy * G + xq];
          // g_s[q] = sum_t Wn[t][q - off(t)] * g_{s+1}[q - off(t)]   (gather form)
          const int sy = py - (ky - K / 2), sx = px - (kx - K / 2);
          if (sy >= 0 && sy < G && sx >= 0 && sx < G) acc += s.Wn[t][sy * G + sx] * g[cur][sy * G + sx];
        }
      }
      g[cur ^ 1][tid] = acc;
    }
    __syncthreads();
    cur ^= 1;
  }
  // initial state x0 = enc_w[c] * d12 + enc_b[c]
  {
    float a = (tid < P) ? g[cur][tid] * s.d12[tid] : 0.f;
    float bsum = (tid < P) ? g[cur][tid] : 0.f;
    a = wave_sum(a); bsum = wave_sum(bsum);
    if ((tid & 63) == 0) { red[0][tid >> 6] = a; red[1][tid >> 6] = bsum; }
    __syncthreads();
    if (tid == 0) {
      atomicAdd(&d_enc_w[c], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
      atomicAdd(&d_enc_b[c], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
  }
  // through the normalisation Wn = W * inv and the sigmoid: dz = dW_raw * W (1 - W)
  if (tid < P) {
    float dot = 0.f;
    for (int t = 0; t < T; ++t) dot += dW[t][tid] * s.Wn[t][tid];
    const float iv = s.inv[tid];
    for (int t = 0; t < T; ++t) {
      const float wn = s.Wn[t][tid];
      const float w = wn / iv;                 // raw sigmoid output
      const float dwraw = (dW[t][tid] - dot) * iv;
      dW[t][tid] = dwraw * w * (1.f - w);      // dz
    }
  }
  __syncthreads();
  if (tid < T * 4) {
    const int t = tid >> 2, i = tid & 3;  // i < 3: weight column i; i == 3: bias
    float acc = 0.f;
    for (int p = 0; p < P; ++p) acc += dW[t][p] * (i < 3 ? s.xx[i][p] : 1.f);
    if (i < 3) atomicAdd(&d_reg_w[(size_t)(c * T + t) * 3 + i], acc);
    else atomicAdd(&d_reg_b[c * T + t], acc);
  }
}

// ---- tail: e2 = conv1x1_{24->3}(x4) on the 12x12 grid, out = bilinear_{12->S}(e2) + image
__global__ __launch_bounds__(256) void diffuse_tail_fwd_kernel(const float* __restrict__ x4, const float* __restrict__ cw,
                                                               const float* __restrict__ cb, const float* __restrict__ image,
                                                               float* __restrict__ out, int S) {
  __shared__ float e2[P];
  const int bo = blockIdx.y, b = bo / 3, o = bo % 3, tid = threadIdx.x;
  if (tid < P) {
    float acc = cb[o];
    for (int c = 0; c < LAT; ++c) acc += cw[o * LAT + c] * x4[((size_t)b * LAT + c) * P + tid];
    e2[tid] = acc;
  }
  __syncthreads();
  const float scale = (float)G / (float)S;
  const size_t plane = (size_t)bo * S * S;
  const int q4 = S / 4;  // float4 per row
  for (int i = blockIdx.x * blockDim.x + tid; i < S * q4; i += gridDim.x * blockDim.x) {
    const int y = i / q4, xq = (i % q4) * 4;
    int y0, y1; float ly;
    bilinear_src(y, scale, G, y0, y1, ly);
    f32x4 im = *reinterpret_cast<const f32x4*>(image + plane + (size_t)y * S + xq);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int x0, x1; float lx;
      bilinear_src(xq + j, scale, G, x0, x1, lx);
      const float v = (1.f - ly) * ((1.f - lx) * e2[y0 * G + x0] + lx * e2[y0 * G + x1]) +
                      ly * ((1.f - lx) * e2[y1 * G + x0] + lx * e2[y1 * G + x1]);
      r[j] = v + im[j];
    }
    *reinterpret_cast<f32x4*>(out + plane + (size_t)y * S + xq) = r;
  }
}

// ge2[b][o][cell] = sum over pixels of bilinear weight(pixel -> cell) * gout   (gather per cell)
__global__ __launch_bounds__(256) void diffuse_tail_bwd_cells_kernel(const float* __restrict__ gout, float* __restrict__ ge2, int S) {
  __shared__ float red[4];
  const int cell = blockIdx.x, bo = blockIdx.y, cy = cell / G, cx = cell % G, tid = threadIdx.x;
  const float scale = (float)G / (float)S;
  // pixel range that can touch this cell: src in (c-1, c+1)  ->  dst in ((c-0.5)/scale - 0.5, (c+1.5)/scale - 0.5); clamp handles borders
  const int lo_y = max(0, (int)floorf((cy - 0.5f) / scale - 0.5f) - 1), hi_y = min(S - 1, (int)ceilf((cy + 1.5f) / scale - 0.5f) + 1);
  const int lo_x = max(0, (int)floorf((cx - 0.5f) / scale - 0.5f) - 1), hi_x = min(S - 1, (int)ceilf((cx + 1.5f) / scale - 0.5f) + 1);
  const int ny = hi_y - lo_y + 1, nx = hi_x - lo_x + 1;
  const float* gp = gout + (size_t)bo * S * S;
  float acc = 0.f;
  for (int i = tid; i < ny * nx; i += blockDim.x) {
    const int y = lo_y + i / nx, x = lo_x + i % nx;
    int y0, y1, x0, x1; float ly, lx;
    bilinear_src(y, scale, G, y0, y1, ly);
    bilinear_src(x, scale, G, x0, x1, lx);
    float wy = (y0 == cy ? 1.f - ly : 0.f) + (y1 == cy ? ly : 0.f);
    float wx = (x0 == cx ? 1.f - lx : 0.f) + (x1 == cx ? lx : 0.f);
    const float w = wy * wx;
    if (w != 0.f) acc += w * gp[(size_t)y * S + x];
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) ge2[(size_t)bo * P + cell] = red[0] + red[1] + red[2] + red[3];
}

// g4 = cw^T ge2 ; d_cw += ge2 x4^T ; d_cb += sum ge2     (one workgroup per image)
__global__ __launch_bounds__(256) void diffuse_tail_bwd_mix_kernel(const float* __restrict__ ge2, const float* __restrict__ x4,
                                                                   const float* __restrict__ cw, float* __restrict__ g4,
                                                                   float* __restrict__ d_cw, float* __restrict__ d_cb) {
  __shared__ float ge[3][P];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < 3 * P; i += blockDim.x) ge[i / P][i % P] = ge2[(size_t)b * 3 * P + i];
  __syncthreads();
  for (int i = tid; i < LAT * P; i += blockDim.x) {
    const int c = i / P, p = i % P;
    g4[(size_t)b * LAT * P + i] = cw[c] * ge[0][p] + cw[LAT + c] * ge[1][p] + cw[2 * LAT + c] * ge[2][p];
  }
  if (tid < 3 * LAT) {
    const int o = tid / LAT, c = tid % LAT;
    float acc = 0.f;
    for (int p = 0; p < P; ++p) acc += ge[o][p] * x4[((size_t)b * LAT + c) * P + p];
    atomicAdd(&d_cw[o * LAT + c], acc);
  } else if (tid < 3 * LAT + 3) {
    const int o = tid - 3 * LAT;
    float acc = 0.f;
    for (int p = 0; p < P; ++p) acc += ge[o][p];
    atomicAdd(&d_cb[o], acc);
  }
}

}  // namespace

extern "C" int dgtd_diffuser_fwd(const float* x_hp, const float* depth, const float* reg_w, const float* reg_b,
                                 const float* enc_w, const float* enc_b, float* x4, int B, int S, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 4.0 * B * (4 * 144 + 24 * 144), "dgtd_diffuser_fwd[B=%d,S=%d]", B, S);
  DGTD_REQUIRE(B > 0 && S >= G, "diffuser_fwd: bad sizes B=%d S=%d", B, S);
  hipLaunchKernelGGL(diffuser_fwd_kernel, dim3(LAT, B), dim3(256), 0, (hipStream_t)s, x_hp, depth, reg_w, reg_b, enc_w, enc_b, x4, S);
  DGTD_CHECK_LAUNCH("diffuser_fwd");
  return 0;
}

extern "C" int dgtd_diffuser_bwd(const float* x_hp, const float* depth, const float* reg_w, const float* reg_b,
                                 const float* enc_w, const float* enc_b, const float* g4, float* d_reg_w, float* d_reg_b,
                                 float* d_enc_w, float* d_enc_b, int B, int S, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 4.0 * B * (4 * 144 + 2 * 24 * 144), "dgtd_diffuser_bwd[B=%d,S=%d]", B, S);
  DGTD_REQUIRE(B > 0 && S >= G, "diffuser_bwd: bad sizes B=%d S=%d", B, S);
  hipLaunchKernelGGL(diffuser_bwd_kernel, dim3(LAT, B), dim3(256), 0, (hipStream_t)s, x_hp, depth, reg_w, reg_b, enc_w, enc_b, g4,
                     d_reg_w, d_reg_b, d_enc_w, d_enc_b, S);
  DGTD_CHECK_LAUNCH("diffuser_bwd");
  return 0;
}

extern "C" int dgtd_diffuse_tail_fwd(const float* x4, const float* cw, const float* cb, const float* image, float* out,
                                     int B, int S, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 2.0 * 4 * B * 3 * S * S, "dgtd_diffuse_tail_fwd[B=%d,S=%d]", B, S);
  DGTD_REQUIRE(B > 0 && S >= 4 && S % 4 == 0, "diffuse_tail_fwd: S=%d must be a positive multiple of 4", S);
  const int gx = (int)std::min<int64_t>(cdiv((int64_t)S * (S / 4), 256), 256);
  hipLaunchKernelGGL(diffuse_tail_fwd_kernel, dim3(gx, 3 * B), dim3(256), 0, (hipStream_t)s, x4, cw, cb, image, out, S);
  DGTD_CHECK_LAUNCH("diffuse_tail_fwd");
  return 0;
}

extern "C" int64_t dgtd_diffuse_tail_bwd_workspace(int B) { return (int64_t)B * 3 * P * sizeof(float); }

extern "C" int dgtd_diffuse_tail_bwd(const float* gout, const float* x4, const float* cw, float* g4, float* d_cw, float* d_cb,
                                     void* workspace, int B, int S, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 4.0 * B * 3 * S * S, "dgtd_diffuse_tail_bwd[B=%d,S=%d]", B, S);
  DGTD_REQUIRE(B > 0 && S >= G, "diffuse_tail_bwd: bad sizes B=%d S=%d", B, S);
  float* ge2 = (float*)workspace;
  hipLaunchKernelGGL(diffuse_tail_bwd_cells_kernel, dim3(P, 3 * B), dim3(256), 0, (hipStream_t)s, gout, ge2, S);
  DGTD_CHECK_LAUNCH("diffuse_tail_bwd_cells");
  hipLaunchKernelGGL(diffuse_tail_bwd_mix_kernel, dim3(B), dim3(256), 0, (hipStream_t)s, (const float*)ge2, x4, cw, g4, d_cw, d_cb);
  DGTD_CHECK_LAUNCH("diffuse_tail_bwd_mix");
  return 0;
}

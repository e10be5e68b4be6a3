// loss.hip — fused structure loss of the five deep-supervision heads (fp32).
// Replaces, for training: the x8 bilinear up-sampling of the 1-channel head maps (twig/model/cod.py:796, :806; align_corners=False)
// and cod.cal_loss (cod.py:76-85) called five times (cod.py:137-142): weit = 1 + 5|avgpool31(gt) - gt|, weighted BCE-with-logits +
// weighted soft IoU per sample, mean over the batch, mixed with weights (0, .2, .4, .6) for P1[0..3] and 1 for P2.
//
// The edge weight depends on the label only -> computed ONCE (the reference evaluates the 31x31 box filter five times), the five
// full-resolution logit maps are never materialised: every label pixel bilinearly samples the five low-resolution maps on the fly.
// HBM-bound: algorithmic bytes = label read (4 B/px) + weight write/read (8 B/px) + the low-res maps (L2 resident).
//   box   : separable 31-tap running sums through LDS (32x32 output tile, 62x62 input tile)
//   fwd   : per (map k, sample b): A = sum w*bce, I = sum w*sig*gt, U = sum w*(sig+gt); per b: W = sum w   (wave shuffles + atomics)
//   bwd   : gather per low-res pixel over the <= 16x16 label pixels it touches (transpose of the bilinear sampling)
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int NMAP = 5, RAD = 15, TILE = 32, HALO = TILE + 2 * RAD;

__device__ __forceinline__ void bil_src(int dst, float scale, int in, int& i0, int& i1, float& l1) {
  float src = scale * (dst + 0.5f) - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  i1 = i0 + ((i0 < in - 1) ? 1 : 0);
  l1 = src - (float)i0;
}

// weit[b][y][x] = 1 + 5*|boxmean31(gt) - gt|  (zero padding, divisor 961: F.avg_pool2d count_include_pad=True); wsum[b] += sum weit
__global__ __launch_bounds__(256) void loss_weight_kernel(const float* __restrict__ gt, float* __restrict__ weit,
                                                          float* __restrict__ wsum, int S) {
  __shared__ float in[HALO][HALO + 1];
  __shared__ float hs[HALO][TILE + 1];
  __shared__ float red[4];
  const int b = blockIdx.z, ty0 = blockIdx.y * TILE, tx0 = blockIdx.x * TILE, tid = threadIdx.x;
  const float* g = gt + (size_t)b * S * S;
  for (int i = tid; i < HALO * HALO; i += 256) {
    const int yy = i / HALO, xx = i % HALO, y = ty0 + yy - RAD, x = tx0 + xx - RAD;
    in[yy][xx] = (y >= 0 && y < S && x >= 0 && x < S) ? g[(size_t)y * S + x] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < HALO * TILE; i += 256) {     // horizontal 31-tap sums
    const int yy = i / TILE, xx = i % TILE;
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * RAD + 1; ++k) s += in[yy][xx + k];
    hs[yy][xx] = s;
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = tid; i < TILE * TILE; i += 256) {     // vertical 31-tap sums
    const int yy = i / TILE, xx = i % TILE, y = ty0 + yy, x = tx0 + xx;
    if (y < S && x < S) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 2 * RAD + 1; ++k) s += hs[yy + k][xx];
      const float w = 1.f + 5.f * fabsf(s * (1.f / 961.f) - in[yy + RAD][xx + RAD]);
      weit[(size_t)b * S * S + (size_t)y * S + x] = w;
      acc += w;
    }
  }
  acc = wave_sum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) atomicAdd(&wsum[b], red[0] + red[1] + red[2] + red[3]);
}

// sums[k][b][3] += {A, I, U} over this block's pixels.  lo: [NMAP][B][hs][hs] low-res logits.
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ lo, const float* __restrict__ gt,
                                                       const float* __restrict__ weit, float* __restrict__ sums, int B, int S, int hs) {
  __shared__ float red[4][NMAP * 3];
  const int b = blockIdx.y, tid = threadIdx.x;
  const float scale = (float)hs / (float)S;
  float acc[NMAP][3];
#pragma unroll
  for (int k = 0; k < NMAP; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0.f;
  const size_t plane = (size_t)b * S * S;
  for (int i = blockIdx.x * 256 + tid; i < S * S; i += gridDim.x * 256) {
    const int y = i / S, x = i % S;
    int y0, y1, x0, x1; float ly, lx;
    bil_src(y, scale, hs, y0, y1, ly);
    bil_src(x, scale, hs, x0, x1, lx);
    const float w = weit[plane + i], t = gt[plane + i];
    const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
    for (int k = 0; k < NMAP; ++k) {
      const float* m = lo + ((size_t)k * B + b) * hs * hs;
      const float z = w00 * m[y0 * hs + x0] + w01 * m[y0 * hs + x1] + w10 * m[y1 * hs + x0] + w11 * m[y1 * hs + x1];
      const float e = expf(-fabsf(z));
      const float bce = fmaxf(z, 0.f) - z * t + log1pf(e);
      const float sg = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      acc[k][0] += w * bce;
      acc[k][1] += w * sg * t;
      acc[k][2] += w * (sg + t);
    }
  }
#pragma unroll
  for (int k = 0; k < NMAP; ++k)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float v = wave_sum(acc[k][j]);
      if ((tid & 63) == 0) red[tid >> 6][k * 3 + j] = v;
    }
  __syncthreads();
  if (tid < NMAP * 3) atomicAdd(&sums[((size_t)(tid / 3) * B + b) * 3 + tid % 3], red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

// loss = sum_k mix[k] * mean_b( A/W + 1 - (I+1)/(U-I+1) )   ; one thread
__global__ void loss_finish_kernel(const float* __restrict__ sums, const float* __restrict__ wsum, const float* __restrict__ mix,
                                   float* __restrict__ loss, int B) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float tot = 0.f;
  for (int k = 0; k < NMAP; ++k) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* p = sums + ((size_t)k * B + b) * 3;
      s += p[0] / wsum[b] + 1.f - (p[1] + 1.f) / (p[2] - p[1] + 1.f);
    }
    tot += mix[k] * s / (float)B;
  }
  loss[0] = tot;
}

// dlo[k][b][cy][cx] = gout * mix[k]/B * sum_pixels bilinear_weight(pixel -> cell) * dL_k/dz(pixel), all five maps in one gather
__global__ __launch_bounds__(64) void loss_bwd_kernel(const float* __restrict__ lo, const float* __restrict__ gt,
                                                      const float* __restrict__ weit, const float* __restrict__ sums,
                                                      const float* __restrict__ wsum, const float* __restrict__ mix,
                                                      const float* __restrict__ gout, float* __restrict__ dlo, int B, int S, int hs) {
  const int cell = blockIdx.x, cy = cell / hs, cx = cell % hs, b = blockIdx.y, lane = threadIdx.x;
  const float scale = (float)hs / (float)S;
  const float Wb = wsum[b];
  float Nn[NMAP], Dn[NMAP], acc[NMAP];
#pragma unroll
  for (int k = 0; k < NMAP; ++k) {
    const float* p3 = sums + ((size_t)k * B + b) * 3;
    Nn[k] = p3[1] + 1.f; Dn[k] = p3[2] - p3[1] + 1.f; acc[k] = 0.f;
  }
  // label pixels that can touch this cell: src in (c-1, c+1)
  const int lo_y = max(0, (int)floorf((cy - 0.5f) / scale - 0.5f) - 1), hi_y = min(S - 1, (int)ceilf((cy + 1.5f) / scale - 0.5f) + 1);
  const int lo_x = max(0, (int)floorf((cx - 0.5f) / scale - 0.5f) - 1), hi_x = min(S - 1, (int)ceilf((cx + 1.5f) / scale - 0.5f) + 1);
  const int ny = hi_y - lo_y + 1, nx = hi_x - lo_x + 1;
  const size_t plane = (size_t)b * S * S;
  for (int i = lane; i < ny * nx; i += 64) {
    const int y = lo_y + i / nx, x = lo_x + i % nx;
    int y0, y1, x0, x1; float ly, lx;
    bil_src(y, scale, hs, y0, y1, ly);
    bil_src(x, scale, hs, x0, x1, lx);
    const float wy = (y0 == cy ? 1.f - ly : 0.f) + (y1 == cy ? ly : 0.f);
    const float wx = (x0 == cx ? 1.f - lx : 0.f) + (x1 == cx ? lx : 0.f);
    const float wc = wy * wx;
    if (wc != 0.f) {
      const float w = weit[plane + (size_t)y * S + x], t = gt[plane + (size_t)y * S + x];
      const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
      for (int k = 0; k < NMAP; ++k) {
        if (mix[k] == 0.f) continue;                          // a map the loss does not weigh has a zero gradient (stage 0: 0.2 * 0, cod.py:139)
        const float* m = lo + ((size_t)k * B + b) * hs * hs;
        const float z = w00 * m[y0 * hs + x0] + w01 * m[y0 * hs + x1] + w10 * m[y1 * hs + x0] + w11 * m[y1 * hs + x1];
        const float e = expf(-fabsf(z));
        const float sg = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
        const float ds = sg * (1.f - sg);
        acc[k] += wc * (w * (sg - t) / Wb - w * ds * (t * Dn[k] - Nn[k] * (1.f - t)) / (Dn[k] * Dn[k]));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NMAP; ++k) {
    const float v = wave_sum(acc[k]);
    if (lane == 0) dlo[((size_t)k * B + b) * hs * hs + cell] = gout[0] * mix[k] / (float)B * v;
  }
}

// The same for the x8 up-sampling the model uses (S == 8 hs, cod.py:796 / :806), tiled: a workgroup owns 8 x 8 cells.  The label pixels
// that touch them are the 72 x 72 block starting 4 pixels before the tile; the derivative dL_k/dz of every such pixel is evaluated ONCE
// (the gather above evaluates every pixel once per cell it touches: 4x, behind 9 taps of index arithmetic) into LDS, then four threads
// per cell sum their 16 x 16 window with the bilinear weights - the same weights, the same order per cell on every run (deterministic).
constexpr int LT = 8, LP = LT * 8 + 8;             // cells per tile side, pixels per tile side (72)
__global__ __launch_bounds__(256) void loss_bwd_tiled_kernel(const float* __restrict__ lo, const float* __restrict__ gt,
                                                             const float* __restrict__ weit, const float* __restrict__ sums,
                                                             const float* __restrict__ wsum, const float* __restrict__ mix,
                                                             const float* __restrict__ gout, float* __restrict__ dlo, int B, int S, int hs) {
  __shared__ float g[LP][LP + 1];
  const int b = blockIdx.z, cy0 = blockIdx.y * LT, cx0 = blockIdx.x * LT, tid = threadIdx.x;
  const int py0 = cy0 * 8 - 4, px0 = cx0 * 8 - 4;
  const float scale = (float)hs / (float)S, Wb = wsum[b];
  const size_t plane = (size_t)b * S * S;
  const int cell = tid >> 2, part = tid & 3, cy = cy0 + cell / LT, cx = cx0 + cell % LT;
  // the cell's window = pixel rows 8 cy - 4 .. 8 cy + 11 = tile rows 8 (cy - cy0) .. + 15 (columns likewise); this thread takes 4 of the 16
  // rows.  Its bilinear weights are the same for every map (0 outside the image or where the pixel does not touch the cell)
  const int ry = (cy - cy0) * 8, rx = (cx - cx0) * 8;
  float wys[4], wxs[16];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = py0 + ry + part * 4 + r;
    int y0, y1; float ly;
    bil_src(y, scale, hs, y0, y1, ly);
    wys[r] = (y >= 0 && y < S) ? (y0 == cy ? 1.f - ly : 0.f) + (y1 == cy ? ly : 0.f) : 0.f;
  }
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int x = px0 + rx + c;
    int x0, x1; float lx;
    bil_src(x, scale, hs, x0, x1, lx);
    wxs[c] = (x >= 0 && x < S) ? (x0 == cx ? 1.f - lx : 0.f) + (x1 == cx ? lx : 0.f) : 0.f;
  }
  for (int k = 0; k < NMAP; ++k) {
    const float mk = mix[k];
    float acc = 0.f;
    if (mk != 0.f) {                                 // a map the loss does not weigh has a zero gradient (stage 0: 0.2 * 0, cod.py:139)
      const float* p3 = sums + ((size_t)k * B + b) * 3;
      const float Nn = p3[1] + 1.f, Dn = p3[2] - p3[1] + 1.f;
      const float* m = lo + ((size_t)k * B + b) * hs * hs;
      for (int p = tid; p < LP * LP; p += 256) {      // (the taps are recomputed per map: a few integer ops against 20 live arrays of 21)
        const int py = p / LP, px = p - py * LP, y = py0 + py, x = px0 + px;
        float gv = 0.f;
        if (y >= 0 && y < S && x >= 0 && x < S) {
          int y0, y1, x0, x1; float ly, lx;
          bil_src(y, scale, hs, y0, y1, ly);
          bil_src(x, scale, hs, x0, x1, lx);
          const float w = weit[plane + (size_t)y * S + x], t = gt[plane + (size_t)y * S + x];
          const float z = (1.f - ly) * (1.f - lx) * m[y0 * hs + x0] + (1.f - ly) * lx * m[y0 * hs + x1] + ly * (1.f - lx) * m[y1 * hs + x0] +
                          ly * lx * m[y1 * hs + x1];
          const float e = expf(-fabsf(z));
          const float sg = z >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
          const float ds = sg * (1.f - sg);
          gv = w * (sg - t) / Wb - w * ds * (t * Dn - Nn * (1.f - t)) / (Dn * Dn);
        }
        g[py][px] = gv;
      }
      __syncthreads();
      if (cy < hs && cx < hs) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float row = 0.f;
#pragma unroll
          for (int c = 0; c < 16; ++c) row += wxs[c] * g[ry + part * 4 + r][rx + c];
          acc += wys[r] * row;
        }
      }
      __syncthreads();                               // g is rewritten for the next map
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (part == 0 && cy < hs && cx < hs) dlo[((size_t)k * B + b) * hs * hs + cy * hs + cx] = gout[0] * mk / (float)B * acc;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ SSIM value (no gradient)
// loss3 of cod.forward (cod.py:143-144): SSIM(minmax(x_hp), image) with SSIM._ssim (cod.py:330-348): reflect-pad 1, five 3x3 mean
// filters, clamp((1 - n/d)/2, 0, 1), mean over channels then over everything (= the mean over all elements).  The min-max
// normalisation e = (x_hp - min) / (max - min + 1e-8) is global, so: pass 1 = min / max of x_hp, pass 2 = the SSIM map's sum.
// The reference runs ~30 elementwise launches on [B,3,S,S] fp32 tensors for this scalar.
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// mm[0] = ordered-uint min, mm[1] = ordered-uint max (initialised to 0xffffffff / 0 by the host-side memset pattern below)
__global__ __launch_bounds__(256) void ssim_minmax_kernel(const float* __restrict__ x, unsigned* __restrict__ mm, long n) {
  float lo = INFINITY, hi = -INFINITY;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += (long)gridDim.x * 1024) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
    lo = fminf(fminf(lo, v[0]), fminf(fminf(v[1], v[2]), v[3]));
    hi = fmaxf(fmaxf(hi, v[0]), fmaxf(fmaxf(v[1], v[2]), v[3]));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[n - 1 - threadIdx.x]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o, 64)); hi = fmaxf(hi, __shfl_xor(hi, o, 64)); }
  // one atomic pair per WORKGROUP: 8192 same-address device atomics (one pair per wave of 2048 workgroups) serialised into 190 us
  __shared__ float wl[4], wh[4];
  if ((threadIdx.x & 63) == 0) { wl[threadIdx.x >> 6] = lo; wh[threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    lo = fminf(fminf(wl[0], wl[1]), fminf(wl[2], wl[3]));
    hi = fmaxf(fmaxf(wh[0], wh[1]), fmaxf(wh[2], wh[3]));
    atomicMin(mm, f2ord(lo)); atomicMax(mm + 1, f2ord(hi));
  }
}

__device__ __forceinline__ int reflect1(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// one thread per pixel of one [S,S] plane; acc[0] += sum of the clamped SSIM map (double: 6 M terms)
__global__ __launch_bounds__(256) void ssim_map_kernel(const float* __restrict__ xh, const float* __restrict__ img, const unsigned* __restrict__ mm,
                                                       double* __restrict__ acc, int S) {
  __shared__ double red[4];
  const float mn = ord2f(mm[0]), mx = ord2f(mm[1]);
  const float inv = 1.f / (mx - mn + 1e-8f);
  const size_t plane = (size_t)blockIdx.y * S * S;
  const float* xp = xh + plane;
  const float* yp = img + plane;
  float local = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < S * S; p += gridDim.x * 256) {
    const int py = p / S, px = p % S;
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = reflect1(py + dy, S);
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = reflect1(px + dx, S);
        const float a = (xp[(size_t)yy * S + xx] - mn) * inv, b = yp[(size_t)yy * S + xx];
        sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
      }
    }
    const float k = 1.f / 9.f, C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float mu_x = sx * k, mu_y = sy * k;
    const float sig_x = sxx * k - mu_x * mu_x, sig_y = syy * k - mu_y * mu_y, sig_xy = sxy * k - mu_x * mu_y;
    const float nn = (2.f * mu_x * mu_y + C1) * (2.f * sig_xy + C2);
    const float dd = (mu_x * mu_x + mu_y * mu_y + C1) * (sig_x + sig_y + C2);
    local += fminf(fmaxf((1.f - nn / dd) * 0.5f, 0.f), 1.f);
  }
  double d = (double)wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, red[0] + red[1] + red[2] + red[3]);
}

__global__ void ssim_finish_kernel(const double* __restrict__ acc, float* __restrict__ out, double inv_n) { out[0] = (float)(acc[0] * inv_n); }

extern "C" int64_t dgtd_ssim_workspace(void) { return 16; }

// x_hp, image fp32 [B,C,S,S] (NCHW contiguous); out fp32 [1] = SSIM._ssim(minmax(x_hp), image) (cod.py:143-144, :330-348).
// workspace: dgtd_ssim_workspace() bytes.
extern "C" int dgtd_ssim_value(const float* x_hp, const float* image, float* out, void* workspace, int B, int C, int S, dgtd_stream s) {
  DGTD_REQUIRE(B > 0 && C > 0 && S > 1 && x_hp && image && out && workspace, "ssim_value: bad arguments");
  DGTD_PROF(s, DGTD_HBM, 3.0 * 4 * B * C * S * S, "dgtd_ssim_value[B=%d,C=%d,S=%d]", B, C, S);
  hipStream_t st = (hipStream_t)s;
  unsigned* mm = (unsigned*)workspace;
  double* acc = (double*)((char*)workspace + 8);
  hipError_t e = hipMemsetAsync(mm, 0xff, 4, st);
  if (e == hipSuccess) e = hipMemsetAsync((char*)workspace + 4, 0, 12, st);
  if (e != hipSuccess) DGTD_FAIL(3, "ssim_value: memset failed: %s", hipGetErrorString(e));
  const long n = (long)B * C * S * S;
  hipLaunchKernelGGL(ssim_minmax_kernel, dim3((int)std::min<long>(cdiv(n, 1024), 512)), dim3(256), 0, st, x_hp, mm, n);
  DGTD_CHECK_LAUNCH("ssim_minmax");
  // one fp64 atomic per workgroup on ONE address: 6144 of them (256 slices x 24 planes) serialised into most of the kernel's 82 us;
  // 96 slices per plane (2304 atomics) measured best: 105 -> 66 us for the whole value (tools/ssim_time.py)
  static const int slices = getenv("DGTD_SSIM_SLICES") ? atoi(getenv("DGTD_SSIM_SLICES")) : 96;
  hipLaunchKernelGGL(ssim_map_kernel, dim3((int)std::max<long>(1, std::min<long>(cdiv((long)S * S, 256), slices)), B * C), dim3(256), 0, st, x_hp, image, (const unsigned*)mm, acc, S);
  DGTD_CHECK_LAUNCH("ssim_map");
  hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(1), 0, st, (const double*)acc, out, 1.0 / (double)n);
  DGTD_CHECK_LAUNCH("ssim_finish");
  return 0;
}

extern "C" int64_t dgtd_seg_loss_workspace(int B, int S) { return ((int64_t)B * S * S + (int64_t)5 * B * 3 + B) * sizeof(float); }

// workspace layout: weit [B,S,S] | sums [5,B,3] | wsum [B]   (kept by the caller between fwd and bwd)
extern "C" int dgtd_seg_loss_fwd(const float* lo, const float* label, const float* mix, float* loss, void* workspace, int B, int S,
                                 int hs, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 4.0 * B * S * S * 4, "dgtd_seg_loss_fwd[B=%d,S=%d]", B, S);
  DGTD_REQUIRE(B > 0 && S > 0 && hs > 0, "seg_loss_fwd: bad sizes B=%d S=%d hs=%d", B, S, hs);
  hipStream_t st = (hipStream_t)s;
  float* weit = (float*)workspace;
  float* sums = weit + (size_t)B * S * S;
  float* wsum = sums + (size_t)NMAP * B * 3;
  hipError_t e = hipMemsetAsync(sums, 0, ((size_t)NMAP * B * 3 + B) * sizeof(float), st);
  if (e != hipSuccess) DGTD_FAIL(3, "seg_loss_fwd: memset failed: %s", hipGetErrorString(e));
  const int tiles = (int)cdiv(S, TILE);
  hipLaunchKernelGGL(loss_weight_kernel, dim3(tiles, tiles, B), dim3(256), 0, st, label, weit, wsum, S);
  DGTD_CHECK_LAUNCH("loss_weight");
  const int gx = (int)std::min<int64_t>(cdiv((int64_t)S * S, 256 * 4), 256);
  hipLaunchKernelGGL(loss_fwd_kernel, dim3(gx, B), dim3(256), 0, st, lo, label, (const float*)weit, sums, B, S, hs);
  DGTD_CHECK_LAUNCH("loss_fwd");
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)sums, (const float*)wsum, mix, loss, B);
  DGTD_CHECK_LAUNCH("loss_finish");
  return 0;
}

extern "C" int dgtd_seg_loss_bwd(const float* lo, const float* label, const float* mix, const float* gout, float* dlo,
                                 const void* workspace, int B, int S, int hs, dgtd_stream s) {
  DGTD_PROF(s, DGTD_HBM, 4.0 * B * S * S * 2 * 4, "dgtd_seg_loss_bwd[B=%d,S=%d]", B, S);
  DGTD_REQUIRE(B > 0 && S > 0 && hs > 0, "seg_loss_bwd: bad sizes B=%d S=%d hs=%d", B, S, hs);
  const float* weit = (const float*)workspace;
  const float* sums = weit + (size_t)B * S * S;
  const float* wsum = sums + (size_t)NMAP * B * 3;
  static const bool tiled = !(getenv("DGTD_LOSS_BWD_TILED") && getenv("DGTD_LOSS_BWD_TILED")[0] == '0');
  if (tiled && S == 8 * hs && B <= 65535) {          // the model's x8 up-sampling: every pixel's derivative evaluated once per tile
    const int tiles = (hs + LT - 1) / LT;
    hipLaunchKernelGGL(loss_bwd_tiled_kernel, dim3(tiles, tiles, B), dim3(256), 0, (hipStream_t)s, lo, label, weit, sums, wsum, mix, gout, dlo, B, S, hs);
  } else {
    hipLaunchKernelGGL(loss_bwd_kernel, dim3(hs * hs, B), dim3(64), 0, (hipStream_t)s, lo, label, weit, sums, wsum, mix, gout, dlo, B, S, hs);
  }
  DGTD_CHECK_LAUNCH("loss_bwd");
  return 0;
}

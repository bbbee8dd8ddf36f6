// HBM-bound glue layers of the U-Net forward (NHWC, channels padded to 16) and the time-embedding
// tower.  Reference: models.py:134-135 (MaxPool2d(2), bilinear x2 align_corners=True),
// models.py:205-224 (upsample + concat, final upsample + 1x1 head), models.py:15-39,175-185,66-77
// (sinusoidal embedding, time / condition MLPs, per-block time projection).
#include "dt_conv_epilogue.h"

namespace dt {

// enc1.conv1 (reference models.py:61-63 on the C <= 3 channel image): conv3x3 + folded BN + ReLU + time bias as a
// direct fp32 convolution.  K = 9C <= 27 is far too short for the matrix cores (the former im2col + K = 32 GEMM spent
// 28 + 6 us per teacher forward on 0.1 GFLOP); here a thread keeps the 9C x 4 weights of its channel quad in registers,
// the zero-bordered image sits in LDS, and the value is computed ONCE per image and stored for every pass with that
// pass's time-bias row (the passes of a CFG step share x and differ only in the embedding).  HBM-bound on the store.
// A thread computes PX consecutive pixels of a picture row from one (PX + 2)-wide window per (channel, dy): the
// one-pixel form issued 27 LDS reads per stored float4 and was bound by them (25 us for a 67 MB store).
template <int C, int PX>
__global__ __launch_bounds__(256) void first_conv_kernel(const float *__restrict__ x, const float *__restrict__ wf,
                                                         const float *__restrict__ scale, const float *__restrict__ shift,
                                                         const float *__restrict__ tb, int tb_stride, int tb_div,
                                                         float *__restrict__ out, int B, int n_pass, int H, int W, int cp,
                                                         int parts) {
  extern __shared__ float img[];                       // [C][H+2][W+2], zero border
  const int tid = threadIdx.x, b = blockIdx.x / parts, part = blockIdx.x - b * parts;
  const int Wp = W + 2, plane = (H + 2) * Wp, HW = H * W;
  const int QN = cp >> 2, PPI = 256 / QN;              // channel quads per pixel, pixel groups per sweep of the workgroup
  const int q = tid % QN, pl = tid / QN;
  f32x4 w[9 * C];                                      // requested before the image: both latencies overlap
#pragma unroll
  for (int k = 0; k < 9 * C; ++k) w[k] = *reinterpret_cast<const f32x4 *>(wf + (size_t)k * cp + q * 4);
  const f32x4 sc = *reinterpret_cast<const f32x4 *>(scale + q * 4), sh = *reinterpret_cast<const f32x4 *>(shift + q * 4);
  for (int i = tid; i < C * plane; i += 256) {
    const int c = i / plane, r = i - c * plane;
    const int y = r / Wp - 1, xx = r - (y + 1) * Wp - 1;
    img[i] = (y >= 0 && y < H && xx >= 0 && xx < W) ? x[((size_t)(b * C + c) * H + y) * W + xx] : 0.f;
  }
  __syncthreads();
  if (pl >= PPI) return;
  const int GW = W / PX, G = H * GW;                   // groups of PX pixels along x (the launcher picks PX | W)
  const int per = (G + parts - 1) / parts;
  const int hi = (part + 1) * per < G ? (part + 1) * per : G;
  for (int g = part * per + pl; g < hi; g += PPI) {
    const int y = g / GW, x0 = (g - y * GW) * PX;
    f32x4 acc[PX];
#pragma unroll
    for (int px = 0; px < PX; ++px) acc[px] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per pixel the accumulation order is (channel, dy, dx), one fma per term: the order of the one-pixel form
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        float v[PX + 2];
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) v[j] = img[c * plane + (y + dy) * Wp + x0 + j];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int px = 0; px < PX; ++px)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[px][e] = fmaf(v[px + dx], w[c * 9 + dy * 3 + dx][e], acc[px][e]);
      }
#pragma unroll
    for (int px = 0; px < PX; ++px)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[px][e] = fmaxf(acc[px][e] * sc[e] + sh[e], 0.f);
    const int pix = y * W + x0;
    for (int pass = 0; pass < n_pass; ++pass) {
      const int bt = pass * B + b;
      const f32x4 t = tb ? *reinterpret_cast<const f32x4 *>(tb + (size_t)(bt / tb_div) * tb_stride + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int px = 0; px < PX; ++px)
        *reinterpret_cast<f32x4 *>(out + ((size_t)bt * HW + pix + px) * cp + q * 4) = acc[px] + t;
    }
  }
}

template <int PX>
static void first_conv_dispatch(int C, int grid, size_t lds, hipStream_t s, const float *x, const float *wf, const float *scale,
                                const float *shift, const float *tb, int tb_stride, int tb_div, float *out, int B, int n_pass,
                                int H, int W, int cout_p, int parts) {
  if (C == 1) first_conv_kernel<1, PX><<<grid, 256, lds, s>>>(x, wf, scale, shift, tb, tb_stride, tb_div, out, B, n_pass, H, W, cout_p, parts);
  else if (C == 2) first_conv_kernel<2, PX><<<grid, 256, lds, s>>>(x, wf, scale, shift, tb, tb_stride, tb_div, out, B, n_pass, H, W, cout_p, parts);
  else first_conv_kernel<3, PX><<<grid, 256, lds, s>>>(x, wf, scale, shift, tb, tb_stride, tb_div, out, B, n_pass, H, W, cout_p, parts);
}

int launch_first_conv(const float *x, const float *wf, const float *scale, const float *shift, const float *tb, int tb_stride,
                      int tb_div, float *out, int B, int n_pass, int C, int H, int W, int cout, int cout_p, hipStream_t s) {
  if (C < 1 || C > 3 || cout_p % 4 || cout_p > 1024) return DT_E_SHAPE;
  const size_t lds = (size_t)C * (H + 2) * (W + 2) * sizeof(float);
  if (lds > 64 * 1024) return DT_E_SHAPE;
  const int parts = B >= 1024 ? 1 : (B >= 256 ? 4 : 8);            // enough workgroups to cover the chip at small batches
  ProfileScope prof(KC_FIRST_CONV, 2.0 * n_pass * B * H * W * (double)cout * 9 * C, 4.0 * B * H * W * (C + (double)n_pass * cout_p), s);
  if (W % 4 == 0) first_conv_dispatch<4>(C, B * parts, lds, s, x, wf, scale, shift, tb, tb_stride, tb_div, out, B, n_pass, H, W, cout_p, parts);
  else first_conv_dispatch<1>(C, B * parts, lds, s, x, wf, scale, shift, tb, tb_stride, tb_div, out, B, n_pass, H, W, cout_p, parts);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// conv1.weight OIHW [cout][C][3][3] -> wf[k = c*9 + tap][cout_p] (zero on padding channels)
__global__ void pack_first_conv_kernel(const float *__restrict__ w, float *__restrict__ wf, int cout, int C, int cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * C * cp) return;
  const int k = i / cp, n = i - k * cp;
  wf[i] = n < cout ? w[(size_t)n * 9 * C + k] : 0.f;
}

int launch_pack_first_conv(const float *w_oihw, float *wf, int cout, int C, int cout_p, hipStream_t s) {
  pack_first_conv_kernel<<<(9 * C * cout_p + 255) / 256, 256, 0, s>>>(w_oihw, wf, cout, C, cout_p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// 2x2 max pooling, stride 2, float4 over channels
__global__ void maxpool_kernel(const float4 *__restrict__ in, float4 *__restrict__ out, int Bt, int H, int W,
                               int c4) {
  const int Ho = H >> 1, Wo = W >> 1;
  const size_t total = (size_t)Bt * Ho * Wo * c4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = i % c4;
    size_t r = i / c4;
    const int xo = r % Wo; r /= Wo;
    const int yo = r % Ho;
    const int b = r / Ho;
    const float4 *p = in + (((size_t)b * H + 2 * yo) * W + 2 * xo) * c4 + c;
    const float4 a = p[0], bq = p[c4], cq = p[(size_t)W * c4], d = p[(size_t)W * c4 + c4];
    float4 o;
    o.x = fmaxf(fmaxf(a.x, bq.x), fmaxf(cq.x, d.x));
    o.y = fmaxf(fmaxf(a.y, bq.y), fmaxf(cq.y, d.y));
    o.z = fmaxf(fmaxf(a.z, bq.z), fmaxf(cq.z, d.z));
    o.w = fmaxf(fmaxf(a.w, bq.w), fmaxf(cq.w, d.w));
    out[i] = o;
  }
}

int launch_maxpool(const float *in, float *out, int Bt, int H, int W, int cp, hipStream_t s) {
  const size_t total = (size_t)Bt * (H / 2) * (W / 2) * (cp / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  ProfileScope prof(KC_POOL, 0.0, 4.0 * Bt * H * W * cp * 1.25, s);
  maxpool_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), Bt,
                                         H, W, cp / 4);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// out[b][y][x][0:c1p] = bilinear_x2(lo[b]), out[...][c1p:c1p+c2p] = skip[b][y][x]  (torch.cat dim=1)
// 32-bit index math (the launcher checks the element count), one float4 of channels per thread.
__global__ __launch_bounds__(256) void upcat_kernel(const float4 *__restrict__ lo, const float4 *__restrict__ skip,
                                                    float4 *__restrict__ out, int Bt, int h, int w, int c1q, int c2q) {
  const unsigned H = 2 * h, W = 2 * w, cq = c1q + c2q;
  const unsigned cw = skip ? cq : (unsigned)c1q;  // skip == nullptr: only the upsampled channels are written
  const unsigned total = (unsigned)Bt * H * W * cw;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned pixel = i / cw, c = i - pixel * cw;
    const unsigned row = pixel / W, x = pixel - row * W;
    const unsigned b = row / H, y = row - b * H;
    float4 o;
    if (c >= (unsigned)c1q) {
      o = skip[(size_t)pixel * c2q + (c - c1q)];
    } else {
      int y0, y1, x0, x1;
      float wy0, wy1, wx0, wx1;
      bilinear_src((int)y, h, (int)H, y0, y1, wy0, wy1);
      bilinear_src((int)x, w, (int)W, x0, x1, wx0, wx1);
      const float4 *base = lo + (size_t)b * h * w * c1q + c;
      const float4 v00 = base[(y0 * w + x0) * c1q], v01 = base[(y0 * w + x1) * c1q];
      const float4 v10 = base[(y1 * w + x0) * c1q], v11 = base[(y1 * w + x1) * c1q];
      o.x = wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
      o.y = wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
      o.z = wy0 * (wx0 * v00.z + wx1 * v01.z) + wy1 * (wx0 * v10.z + wx1 * v11.z);
      o.w = wy0 * (wx0 * v00.w + wx1 * v01.w) + wy1 * (wx0 * v10.w + wx1 * v11.w);
    }
    out[(size_t)pixel * cq + c] = o;
  }
}

int launch_upcat(const float *lo, const float *skip, float *out, int Bt, int h, int w, int c1p, int c2p,
                 hipStream_t s) {
  const size_t total = (size_t)Bt * 4 * h * w * ((c1p + (skip ? c2p : 0)) / 4);
  if (total >= (1ull << 32)) return DT_E_SHAPE;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  ProfileScope prof(KC_UPCAT, 0.0, 4.0 * Bt * h * w * (c1p + 4.0 * c1p + (skip ? 8.0 * c2p : 0.0)), s);
  upcat_kernel<<<blocks, 256, 0, s>>>(reinterpret_cast<const float4 *>(lo), reinterpret_cast<const float4 *>(skip),
                                       reinterpret_cast<float4 *>(out), Bt, h, w, c1p / 4, c2p / 4);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// Head (models.py:221-224): eps = conv1x1(bilinear_x2(d)) + bias.  Both maps are linear and the bilinear
// weights sum to one, so eps = bilinear_x2(conv1x1(d) + bias): the 1x1 conv runs at the LOW resolution
// (4x fewer dot products, one NHWC pixel per wave) and only C channels are upsampled.
// Kernel 1: eight lanes per low-res pixel (eight pixels per wave); a lane walks float4 number l, l+8, ... of the
// pixel's channel row (128 contiguous bytes per group and step), the C partial dot products are reduced with three
// xor-shuffles inside the group.  lowres[b][y][x][4].
__global__ __launch_bounds__(256) void head_kernel(const float *__restrict__ lo, const float *__restrict__ wf,
                                                   const float *__restrict__ bias, float *__restrict__ lowres,
                                                   size_t n_pix, int cp, int C, int c_real) {
  const int sub = threadIdx.x & 7;
  const size_t g0 = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 3;
  const size_t n_groups = ((size_t)gridDim.x * blockDim.x) >> 3;
  const int c4 = cp >> 2;
  for (size_t base = g0 - (g0 & 7); base < n_pix; base += n_groups) {     // a wave's eight groups advance together
    const size_t pix = base + (g0 & 7);
    float part[4] = {0.f, 0.f, 0.f, 0.f};
    if (pix < n_pix) {
      const float4 *p4 = reinterpret_cast<const float4 *>(lo + pix * cp);
      for (int q = sub; q < c4; q += 8) {
        const float4 v = p4[q];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = q * 4 + e;
          if (k < c_real) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
              if (c < C) part[c] = fmaf(wf[c * c_real + k], vv[e], part[c]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int off = 4; off > 0; off >>= 1) part[c] += __shfl_xor(part[c], off, 64);
    if (sub == 0 && pix < n_pix) {
      float4 o = make_float4(part[0] + bias[0], 0.f, 0.f, 0.f);
      if (C > 1) o.y = part[1] + bias[1];
      if (C > 2) o.z = part[2] + bias[2];
      *reinterpret_cast<float4 *>(lowres + pix * 4) = o;
    }
  }
}

// Kernel 2: eps[b][c][Y][X] (NCHW) = bilinear_x2(lowres)[b][Y][X][c], one thread per output element
__global__ void head_upsample_kernel(const float *__restrict__ lowres, float *__restrict__ eps, int Bt, int h, int w,
                                     int C) {
  const int H = 2 * h, W = 2 * w;
  const size_t total = (size_t)Bt * C * H * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = i % W;
    size_t r = i / W;
    const int y = r % H; r /= H;
    const int c = r % C;
    const size_t b = r / C;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilinear_src(y, h, H, y0, y1, wy0, wy1);
    bilinear_src(x, w, W, x0, x1, wx0, wx1);
    const float *base = lowres + b * h * w * 4 + c;
    const float v00 = base[((size_t)y0 * w + x0) * 4], v01 = base[((size_t)y0 * w + x1) * 4];
    const float v10 = base[((size_t)y1 * w + x0) * 4], v11 = base[((size_t)y1 * w + x1) * 4];
    eps[i] = bilinear_blend(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
  }
}

// Bilinear resize of NCHW images with align_corners=True (torch.nn.functional.interpolate as the reference calls it for a
// student that runs at another resolution: utils/trajectory_manager.py:153-163,298-305, analysis/metrics/trajectory_metrics.py:40-52).
// One thread per four consecutive output pixels of a row (16-byte stores); HBM-bound.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float *__restrict__ in, float *__restrict__ out, int planes,
                                                              int h, int w, int H, int W) {
  const int Wq = (W + 3) >> 2;
  const size_t total = (size_t)planes * H * Wq;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int xq = i % Wq;
    size_t r = i / Wq;
    const int y = r % H;
    const size_t pl = r / H;
    int y0, y1;
    float wy0, wy1;
    bilinear_src(y, h, H, y0, y1, wy0, wy1);
    const float *base = in + pl * (size_t)h * w;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = xq * 4 + j;
      v[j] = 0.f;
      if (x < W) {
        int x0, x1;
        float wx0, wx1;
        bilinear_src(x, w, W, x0, x1, wx0, wx1);
        v[j] = bilinear_blend(base[(size_t)y0 * w + x0], base[(size_t)y0 * w + x1], base[(size_t)y1 * w + x0], base[(size_t)y1 * w + x1],
                              wy0, wy1, wx0, wx1);
      }
    }
    float *o = out + (pl * H + y) * (size_t)W + xq * 4;
    if (W % 4 == 0) *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
    else
      for (int j = 0; j < 4 && xq * 4 + j < W; ++j) o[j] = v[j];
  }
}

int launch_resize_bilinear(const float *in, float *out, int planes, int h, int w, int H, int W, hipStream_t s) {
  if (!in || !out) return DT_E_NULL;
  if (planes < 1 || h < 1 || w < 1 || H < 1 || W < 1) return DT_E_SHAPE;
  const size_t total = (size_t)planes * H * ((W + 3) / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  ProfileScope prof(KC_POOL, 0.0, 4.0 * planes * ((double)h * w + (double)H * W), s);
  resize_bilinear_kernel<<<blocks, 256, 0, s>>>(in, out, planes, h, w, H, W);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

int launch_head(const float *lo, const float *wf, const float *bias, float *lowres, int Bt, int h, int w, int cp, int C, int c_real,
                hipStream_t s) {
  if (C > 3) return DT_E_SHAPE;
  const size_t n_pix = (size_t)Bt * h * w;
  const size_t blocks = (n_pix + 31) / 32;             // 8 lanes per pixel
  ProfileScope prof(KC_HEAD, 2.0 * n_pix * C * c_real, 4.0 * n_pix * (c_real + 4.0), s);
  head_kernel<<<(int)(blocks < 8192 ? blocks : 8192), 256, 0, s>>>(lo, wf, bias, lowres, n_pix, cp, C, c_real);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

int launch_head_upsample(const float *lowres, float *eps, int Bt, int h, int w, int C, hipStream_t s) {
  const size_t n_pix = (size_t)Bt * h * w;
  const size_t total = n_pix * 4 * C;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  ProfileScope prof(KC_HEAD_UP, 0.0, 4.0 * n_pix * (4.0 + 4.0 * C), s);
  head_upsample_kernel<<<blocks, 256, 0, s>>>(lowres, eps, Bt, h, w, C);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---------------------------------------------------------------------------------------------
// Time-embedding tower.  Every output neuron is one wave-level dot product (lanes stride the input axis -> coalesced
// weight rows, __shfl_down reduction); a wave runs four of them at a time (independent accumulators hide the load
// and shuffle latency), a workgroup has sixteen waves, and the per-block projections of one (t, cond) row -- the
// long stage, sum of the blocks' channel counts -- are cut into `slices` workgroups that each recompute the two
// short stages.  (One four-wave workgroup per row took 0.57 ms for the 100 rows of a 50-step loop: 480 dependent
// dot products per wave.)
template <int G>
__device__ inline void wave_dots(const float *__restrict__ w, int ld, const float *vec, int n, int lane, int count, float (&res)[G]) {
  float s[G];
#pragma unroll
  for (int j = 0; j < G; ++j) s[j] = 0.f;
  for (int k = lane; k < n; k += 64) {
    const float v = vec[k];
#pragma unroll
    for (int j = 0; j < G; ++j)
      if (j < count) s[j] = fmaf(w[(size_t)j * ld + k], v, s[j]);
  }
#pragma unroll
  for (int j = 0; j < G; ++j) {
    for (int off = 32; off > 0; off >>= 1) s[j] += __shfl_down(s[j], off, 64);
    res[j] = __shfl(s[j], 0, 64);
  }
}

__global__ __launch_bounds__(1024) void time_bias_kernel(const TembWeights tw, const int32_t *__restrict__ t,
                                                         const float *__restrict__ cond,
                                                         const uint8_t *__restrict__ present, float *__restrict__ out,
                                                         int slices) {
  extern __shared__ float sm[];   // emb[D] | hid[D] | temb[D]
  constexpr int G = 4;
  const int D = tw.D, row = blockIdx.x / slices, slice = blockIdx.x - row * slices;
  float *emb = sm, *hid = sm + D, *temb = sm + 2 * D;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n_waves = blockDim.x >> 6;
  const float tv = (float)t[row];
  for (int k = tid; k < D; k += blockDim.x) {
    float v = 0.f;                                  // odd D: trailing zero (models.py:33-36)
    if (k < tw.half) v = sinf(tv * tw.freqs[k]);
    else if (k < 2 * tw.half) v = cosf(tv * tw.freqs[k - tw.half]);
    emb[k] = v;
  }
  const bool has_cond = cond != nullptr && (present == nullptr || present[row] != 0);
  const float cv = has_cond ? cond[row] : 0.f;
  for (int k = tid; k < D; k += blockDim.x) hid[k] = fmaxf(fmaf(tw.wc0[k], cv, tw.bc0[k]), 0.f);
  __syncthreads();
  for (int o = wave * G; o < D; o += n_waves * G) {
    const int cnt = D - o < G ? D - o : G;
    float a[G], b[G];
    wave_dots<G>(tw.w1 + (size_t)o * D, D, emb, D, lane, cnt, a);
    if (has_cond) wave_dots<G>(tw.wc2 + (size_t)o * D, D, hid, D, lane, cnt, b);
    if (lane == 0)
      for (int j = 0; j < cnt; ++j) {
        float v = fmaxf(a[j] + tw.b1[o + j], 0.f);
        if (has_cond) v += b[j] + tw.bc2[o + j];
        temb[o + j] = v;
      }
  }
  __syncthreads();
  float *orow = out + (size_t)row * tw.tb_stride;
  const int per = (tw.tb_cols + slices - 1) / slices;
  const int lo = slice * per, hi = lo + per < tw.tb_cols ? lo + per : tw.tb_cols;
  for (int o = lo + wave * G; o < hi; o += n_waves * G) {
    const int cnt = hi - o < G ? hi - o : G;
    float a[G];
    wave_dots<G>(tw.wt + (size_t)o * D, D, temb, D, lane, cnt, a);
    if (lane == 0)
      for (int j = 0; j < cnt; ++j) orow[o + j] = fmaxf(a[j] + tw.bt[o + j], 0.f);
  }
}

// Class biases of enc1.conv2 (reference models.py:66-79: h = relu(bn1(conv1 x)) + b, then conv2(h)).  conv2 is linear and
// b is constant over the picture, so conv2(a + b) = conv2(a) + sum over the taps that fall INSIDE the picture of
// W2[n][ci][tap] b[ci]: nine per-channel vectors (top / middle / bottom row x left / middle / right column), which is what
// lets the CFG passes of a step -- same x, different b -- share ONE conv2 launch over the B images.
// out row layout: [tb_cols projected channels | 9 x c0p class biases], class = 3 * (y == 0 ? 0 : y == H-1 ? 2 : 1) + same for x.
__global__ __launch_bounds__(1024) void tap_bias_kernel(const TembWeights tw, float *__restrict__ out) {
  extern __shared__ float bsm[];            // the row's enc1 time bias [c0p] | per-tap sums [9][c0p]
  const int c0p = tw.c0p;
  float *tsum = bsm + c0p;
  float *orow = out + (size_t)blockIdx.x * tw.tb_stride;
  for (int i = threadIdx.x; i < c0p; i += blockDim.x) bsm[i] = orow[i];      // enc1 is block 0: columns [0, c0p)
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * c0p; i += blockDim.x) {                  // (tap, n): consecutive threads read consecutive n
    const int tap = i / c0p, n = i - tap * c0p;
    const float *w = tw.w2t + (size_t)tap * c0p * c0p + n;
    float a = 0.f;
    for (int ci = 0; ci < c0p; ++ci) a = fmaf(w[(size_t)ci * c0p], bsm[ci], a);
    tsum[i] = a;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * c0p; i += blockDim.x) {                  // (class, n)
    const int cls = i / c0p, n = i - cls * c0p;
    const int cy = cls / 3, cx = cls - 3 * cy;
    float a = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool in = !(cy == 0 && dy < 0) && !(cy == 2 && dy > 0) && !(cx == 0 && dx < 0) && !(cx == 2 && dx > 0);
      if (in) a += tsum[tap * c0p + n];
    }
    orow[tw.tb_cols + i] = a;
  }
}

int launch_time_bias(const TembWeights &tw, const int32_t *t, const float *cond, const uint8_t *present, int rows,
                     float *out, hipStream_t s) {
  if (rows <= 0) return DT_OK;
  const int slices = rows >= 512 ? 1 : (rows >= 64 ? 4 : 8);      // few rows: spread the long stage over more CUs
  ProfileScope prof(KC_TIME_BIAS, 2.0 * rows * tw.D * (2.0 * tw.D + tw.tb_cols) + (tw.w2t ? 18.0 * rows * tw.c0p * tw.c0p : 0.0),
                    4.0 * rows * tw.tb_stride, s);
  time_bias_kernel<<<rows * slices, 1024, 3 * tw.D * sizeof(float), s>>>(tw, t, cond, present, out, slices);
  DT_LAUNCH_CHECK();
  if (tw.w2t) {
    tap_bias_kernel<<<rows, 1024, 10 * tw.c0p * sizeof(float), s>>>(tw, out);
    DT_LAUNCH_CHECK();
  }
  return DT_OK;
}

// conv2.weight OIHW [cout][cin][3][3] -> w2t[tap][ci][cp] (zero on padding rows / columns)
__global__ void pack_tap_major_kernel(const float *__restrict__ w, float *__restrict__ w2t, int cout, int cin, int cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * cp * cp) return;
  const int n = i % cp, ci = (i / cp) % cp, tap = i / (cp * cp);
  w2t[i] = (n < cout && ci < cin) ? w[((size_t)n * cin + ci) * 9 + tap] : 0.f;
}

int launch_pack_tap_major(const float *w_oihw, float *w2t, int cout, int cin, int cp, hipStream_t s) {
  pack_tap_major_kernel<<<(9 * cp * cp + 255) / 256, 256, 0, s>>>(w_oihw, w2t, cout, cin, cp);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// copy a Linear's rows [out][in] and bias into a zero-padded [out_p][in] / [out_p] slab
__global__ void pack_linear_rows_kernel(const float *w, const float *b, float *wp, float *bp, int out, int in, int out_p) {
  const size_t total = (size_t)out_p * in;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int o = i / in;
    wp[i] = o < out ? w[i] : 0.f;
    if (i % in == 0) bp[o] = o < out ? b[o] : 0.f;
  }
}

int launch_pack_linear_rows(const float *w, const float *b, float *wp, float *bp, int out, int in, int out_p,
                            hipStream_t s) {
  const size_t total = (size_t)out_p * in;
  pack_linear_rows_kernel<<<(int)((total + 255) / 256), 256, 0, s>>>(w, b, wp, bp, out, in, out_p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

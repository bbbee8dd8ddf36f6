// Implicit-GEMM 3x3 / 1x1 convolution on fp32 MFMA for gfx950 (CDNA4).
//
// Replaces the ATen conv2d + eval-BatchNorm + ReLU (+time-bias add, +residual add) sequence of
// reference models.py:59-83.  GEMM view: M = B*H*W output pixels (NHWC rows), N = out channels,
// K = taps * in channels.  Exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain.
//
// Tiling: a 256-thread workgroup (4 wave64, one per SIMD) owns a BM x BN output tile and walks K in
// chunks of 16 channels of one tap.  Per chunk the A tile [BM][16] (activations, gathered with the
// tap's pixel shift and zero padding) and the B tile [BN][16] (weights, pre-tiled at pack time so the
// copy is linear) are staged through LDS with register prefetch + double buffering (one barrier per
// chunk).  Both tiles keep K contiguous per row (64 B) so a lane fetches FOUR consecutive k of its
// row with one ds_read_b128; the 16-B slots of a row are XOR-swizzled by (row>>2)&3, which makes each
// of ds_read_b128's 16-lane groups hit 16 distinct slots of the 256-B bank row (conflict-free).
// The MFMA k-pairing is permuted accordingly (lane half h supplies k = 8s+4h+i for MFMA i of set s)
// identically for A and B, so the sum over k is complete and the result layout is the standard one:
// acc register r of lane l = D[row (r&3)+8*(r>>2)+4*(l>>5)][col l&31], i.e. a register is a 128-B run
// of consecutive channels for two pixel rows -> coalesced NHWC stores.
#include <cstdlib>

#include "dt_conv_epilogue.h"

namespace dt {

template <int BM, int BN>
__global__ __launch_bounds__(256, 4) void conv_gemm_kernel(const ConvParams p) {
  constexpr int WN = 2;                         // 2x2 waves
  constexpr int MI = BM / 64, NI = BN / 64;     // 32x32 MFMA tiles per wave in m / n
  constexpr int A_PER = BM / 64, B_PER = BN / 64;  // float4 staged per thread
  constexpr int STAGE = (BM + BN) * 16;         // floats per LDS stage
  constexpr int LDS_FLOATS = 2 * STAGE > epilogue_stage_floats<BN>() ? 2 * STAGE : epilogue_stage_floats<BN>();
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HW = p.H * p.W;
  const int CC = p.cin_p >> 4;                  // 16-channel chunks per tap

  // ---- per-thread staging coordinates (fixed for the whole K walk)
  int a_off[A_PER], a_off2[A_PER], a_y[A_PER], a_x[A_PER];
  bool a_ok[A_PER];
  int a_lds[A_PER];
#pragma unroll
  for (int j = 0; j < A_PER; ++j) {
    const int q = tid + j * 256, row = q >> 2, pslot = q & 3;
    const int lslot = pslot ^ ((row >> 2) & 3);
    const int m = m0 + row;
    a_ok[j] = m < p.M;
    const int mm = a_ok[j] ? m : 0;
    const int b = mm / HW, rem = mm - b * HW;
    a_y[j] = rem / p.W;
    a_x[j] = rem - a_y[j] * p.W;
    a_off[j] = mm * p.cin_p + lslot * 4;       // M*cin_p < 2^31 is checked by the host
    a_off2[j] = mm * p.cin2_p + lslot * 4;     // fused skip walk (same pixel, other tensor)
    a_lds[j] = q * 4;
  }
  const float *wbase = p.w + (size_t)n0 * 16 + tid * 4;

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // row / swizzle terms of this lane's fragment reads
  int a_row[MI], b_row[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) a_row[mi] = wm * (MI * 32) + mi * 32 + l31;
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) b_row[ni] = wn * (NI * 32) + ni * 32 + l31;

  // Software pipeline, one barrier per chunk: iteration `it` issues the global loads of chunk it+1
  // into registers, runs the MFMAs of chunk it from LDS stage it&1, then parks the registers in the
  // other stage.  it == -1 is the prologue (loads chunk 0, no compute).
  // split-K: grid.z slices the (tap, channel chunk) walk into equal runs of whole chunks
  const int n_main = (p.tap_hi - p.tap_lo) * CC / p.splits;
  const int n_iter = n_main + (p.in2 ? (p.cin2_p >> 4) : 0);   // main walk, then the fused 1x1 skip walk
  int tap = p.tap_lo + (blockIdx.z * n_main) / CC, cc = (blockIdx.z * n_main) % CC;
  for (int it = -1; it < n_iter; ++it) {
    const bool more = it + 1 < n_iter;
    f32x4 ra[A_PER], rb[B_PER];
    if (more && it + 1 >= n_main) {
      const int c2 = it + 1 - n_main;
#pragma unroll
      for (int j = 0; j < A_PER; ++j)
        ra[j] = a_ok[j] ? *reinterpret_cast<const f32x4 *>(p.in2 + a_off2[j] + c2 * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
      const float *wt = p.w2 + (size_t)n0 * 16 + tid * 4 + (size_t)c2 * p.n_p * 16;
#pragma unroll
      for (int j = 0; j < B_PER; ++j) rb[j] = *reinterpret_cast<const f32x4 *>(wt + j * 1024);
    } else if (more) {
      int dy = 0, dx = 0;
      if (p.ksize == 3) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
      const int shift = (dy * p.W + dx) * p.cin_p + cc * 16;
#pragma unroll
      for (int j = 0; j < A_PER; ++j) {
        const int yy = a_y[j] + dy, xx = a_x[j] + dx;
        const bool ok = a_ok[j] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        ra[j] = ok ? *reinterpret_cast<const f32x4 *>(p.in + a_off[j] + shift) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const float *wt = wbase + (size_t)(tap * p.ccw + cc) * p.n_p * 16;
#pragma unroll
      for (int j = 0; j < B_PER; ++j) rb[j] = *reinterpret_cast<const f32x4 *>(wt + j * 1024);
      if (++cc == CC) { cc = 0; ++tap; }
    }
    if (it >= 0) {
      if (it == n_main && p.in2) conv_midpoint<MI, NI>(p, acc, n0, wn, l31);
      const float *A = lds + (it & 1) * STAGE, *B = A + BM * 16;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        f32x4 fa[MI], fb[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          fa[mi] = *reinterpret_cast<const f32x4 *>(A + a_row[mi] * 16 + (((2 * s + half) ^ ((a_row[mi] >> 2) & 3)) << 2));
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          fb[ni] = *reinterpret_cast<const f32x4 *>(B + b_row[ni] * 16 + (((2 * s + half) ^ ((b_row[ni] >> 2) & 3)) << 2));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi][i], fb[ni][i], acc[mi][ni], 0, 0, 0);
            }
      }
    }
    if (more) {
      float *A = lds + ((it + 1) & 1) * STAGE, *B = A + BM * 16;
#pragma unroll
      for (int j = 0; j < A_PER; ++j) *reinterpret_cast<f32x4 *>(A + a_lds[j]) = ra[j];
#pragma unroll
      for (int j = 0; j < B_PER; ++j) *reinterpret_cast<f32x4 *>(B + (tid + j * 256) * 4) = rb[j];
    }
    __syncthreads();
  }

  conv_epilogue<MI, NI>(p, acc, lds, m0, n0, wm, wn, half, l31);
}

int launch_splitk_epilogue(const ConvParams &p, hipStream_t s);

int launch_conv(const ConvParams &p, hipStream_t s) {
  if (!p.in || !p.w || !p.scale || !p.shift || !p.out) return DT_E_NULL;
  if (p.cin_p % 16 || p.cout_p % 16 || p.n_p % kNPad || p.M <= 0) return DT_E_SHAPE;
  if ((long long)p.M * p.cin_p >= (1ll << 31) || (long long)p.M * p.cout_p * (p.n_dup > 1 ? p.n_dup : 1) >= (1ll << 31)) return DT_E_SHAPE;
  int bm = p.bm, bn = p.bn;
  if (!bm || !bn) {
    const ConvChoice c = heuristic_choice(p.M, p.n_p, 1);
    bm = c.bm; bn = c.bn;
  }
  if ((bm != 64 && bm != 128 && !(bm == 256 && bn == 64 && p.prec >= 3)) || (bn != 64 && bn != 128) || p.n_p % bn || p.prec == 2 ||
      p.prec < 0 || p.prec > 5)
    return DT_E_ARG;
  const bool tall_m = bm == 128, wide_n = bn == 128;
  dim3 grid((p.M + bm - 1) / bm, p.n_p / bn, p.splits);
  if (p.splits < 1 || (p.splits > 1 && !p.slab)) return DT_E_ARG;
  if (p.ccw < (p.cin_p >> 4) || (p.in2 && p.ccw2 < (p.cin2_p >> 4))) return DT_E_ARG;   // weight chunks per tap of the pack in use
  if (p.prec >= 3 ? !chunks_fit(p.cin_p >> 4, p.ccw, p.splits * strip_kc(p.prec, bm, bn)) : (((p.tap_hi - p.tap_lo) * (p.cin_p >> 4)) % p.splits != 0)) return DT_E_ARG;
  // algorithmic flops: what the reference's conv2d does on the unpadded shape (all ksize^2 taps)
  if (p.in2 && (p.splits != 1 || !p.w2 || !p.bias2 || p.cin2_p % 16)) return DT_E_ARG;
  const double flops = 2.0 * p.M * (double)p.cout_real * ((double)p.cin_real * p.ksize * p.ksize + (p.in2 ? p.cin2_real : 0));
  if (p.prec >= 3) {
    ProfileScope prof(p.prec == 5 ? (tall_m ? KC_CONVS_K128x64 : (wide_n ? KC_CONVS_K64x128 : KC_CONVS_K64x64)) : bm == 256 ? KC_CONVS_256x64 : tall_m ? (wide_n ? KC_CONVS_128x128 : KC_CONVS_128x64) : (wide_n ? KC_CONVS_64x128 : KC_CONVS_64x64),
                      flops, 4.0 * p.M * ((double)p.cin_real + p.cout_real), s);
    const int st = launch_conv_strip(p, bm, bn, p.prec, s);
    if (st) return st;
  } else if (p.prec == 1) {
    ProfileScope prof(tall_m ? (wide_n ? KC_CONVB_128x128 : KC_CONVB_128x64) : (wide_n ? KC_CONVB_64x128 : KC_CONVB_64x64),
                      flops, 4.0 * p.M * ((double)p.cin_real + p.cout_real), s);
    const int st = launch_conv_bf16x6(p, bm, bn, s);
    if (st) return st;
  } else {
    ProfileScope prof(tall_m ? (wide_n ? KC_CONV_128x128 : KC_CONV_128x64) : (wide_n ? KC_CONV_64x128 : KC_CONV_64x64),
                      flops, 4.0 * p.M * ((double)p.cin_real + p.cout_real), s);
    if (tall_m && wide_n) conv_gemm_kernel<128, 128><<<grid, 256, 0, s>>>(p);
    else if (tall_m) conv_gemm_kernel<128, 64><<<grid, 256, 0, s>>>(p);
    else if (wide_n) conv_gemm_kernel<64, 128><<<grid, 256, 0, s>>>(p);
    else conv_gemm_kernel<64, 64><<<grid, 256, 0, s>>>(p);
    DT_LAUNCH_CHECK();
  }
  if (p.splits > 1) return launch_splitk_epilogue(p, s);
  return DT_OK;
}

// Default tile + tap-split choice when a layer shape has not been autotuned: the widest N tile the
// padded channel count allows, 128 rows once that still gives two workgroups per CU, and tap groups
// so that small spatial levels (M = batch*4*4, batch*2*2 rows) still put >= 1.5 workgroups on every CU.
ConvChoice heuristic_choice(int M, int n_p, int taps) {
  ConvChoice c;
  c.bn = n_p % 128 == 0 ? 128 : 64;
  const long long blocks128 = (long long)((M + 127) / 128) * (n_p / c.bn);
  c.bm = blocks128 >= 512 ? 128 : 64;
  const long long blocks = (long long)((M + c.bm - 1) / c.bm) * (n_p / c.bn);
  c.splits = 1;
  c.prec = 0;
  if (taps == 9 && M <= kSplitMaxRows && blocks < 384) c.splits = blocks * 3 >= 384 ? 3 : 9;
  c.fuse = c.splits == 1;   // fold a block's 1x1 skip into conv2 whenever conv2 is not tap-split
  return c;
}

// sum of the split-K slabs in z order + the layer epilogue, float4 over channels.  S = compile-time slab count
// (all S loads of an element are issued before the first add); S = 0 walks p.splits at run time.
// POOL: a thread owns a 2x2 window of one picture (four rows of the output) and also writes their maximum to
// p.pool_out (the encoder's max pool, folded in so that split layers do not need a separate pooling launch).
template <int S>
__device__ inline float4 splitk_finish(const ConvParams &p, unsigned m, int n, size_t slab_stride) {
  const size_t o = (size_t)m * p.cout_p + n;
  float4 a;
  if (S > 0) {
    float4 part[S > 0 ? S : 1];
#pragma unroll
    for (int z = 0; z < S; ++z) part[z] = *reinterpret_cast<const float4 *>(p.slab + z * slab_stride + o);
    a = part[0];
#pragma unroll
    for (int z = 1; z < S; ++z) { a.x += part[z].x; a.y += part[z].y; a.z += part[z].z; a.w += part[z].w; }
  } else {
    a = *reinterpret_cast<const float4 *>(p.slab + o);
    for (int z = 1; z < p.splits; ++z) {
      const float4 b = *reinterpret_cast<const float4 *>(p.slab + z * slab_stride + o);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
  }
  const float4 sc = *reinterpret_cast<const float4 *>(p.scale + n), sh = *reinterpret_cast<const float4 *>(p.shift + n);
  float4 v = make_float4(a.x * sc.x + sh.x, a.y * sc.y + sh.y, a.z * sc.z + sh.z, a.w * sc.w + sh.w);
  if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  if (p.tb) {
    const float4 t = *reinterpret_cast<const float4 *>(p.tb + (size_t)(m / (unsigned)p.m_per_tb) * p.tb_stride + n);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (p.add) {
    const float4 t = *reinterpret_cast<const float4 *>(p.add + o);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  *reinterpret_cast<float4 *>(p.out + o) = v;
  return v;
}

template <int S, bool POOL>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvParams p) {
  // 32-bit index math (launch_conv guarantees M * cout_p < 2^31)
  const unsigned c4 = (unsigned)p.cout_p >> 2;
  const size_t slab_stride = (size_t)p.M * p.cout_p;
  if (!POOL) {
    const unsigned total = (unsigned)p.M * c4;
    const bool pow2 = (c4 & (c4 - 1)) == 0;
    const int sh4 = 31 - __builtin_clz(c4);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const unsigned m = pow2 ? i >> sh4 : i / c4;
      splitk_finish<S>(p, m, (int)(i - m * c4) * 4, slab_stride);
    }
  } else {
    const unsigned Wo = (unsigned)p.W >> 1, Ho = (unsigned)p.H >> 1, HW = (unsigned)(p.H * p.W);
    const unsigned total = ((unsigned)p.M >> 2) * c4;                    // one thread per (2x2 window, channel quad)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
      const unsigned wq = i / c4;
      const int n = (int)(i - wq * c4) * 4;
      const unsigned xo = wq % Wo, r = wq / Wo;
      const unsigned yo = r % Ho, b = r / Ho;
      const unsigned m00 = b * HW + 2 * yo * (unsigned)p.W + 2 * xo;
      const float4 a = splitk_finish<S>(p, m00, n, slab_stride), bq = splitk_finish<S>(p, m00 + 1, n, slab_stride);
      const float4 c = splitk_finish<S>(p, m00 + p.W, n, slab_stride), d = splitk_finish<S>(p, m00 + p.W + 1, n, slab_stride);
      float4 mx;
      mx.x = fmaxf(fmaxf(a.x, bq.x), fmaxf(c.x, d.x)); mx.y = fmaxf(fmaxf(a.y, bq.y), fmaxf(c.y, d.y));
      mx.z = fmaxf(fmaxf(a.z, bq.z), fmaxf(c.z, d.z)); mx.w = fmaxf(fmaxf(a.w, bq.w), fmaxf(c.w, d.w));
      *reinterpret_cast<float4 *>(p.pool_out + (size_t)wq * p.cout_p + n) = mx;
    }
  }
}

template <bool POOL>
static int launch_splitk_epilogue_t(const ConvParams &p, hipStream_t s) {
  const size_t total = (size_t)(POOL ? p.M / 4 : p.M) * (p.cout_p / 4);
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  ProfileScope prof(KC_SPLITK_EPILOGUE, 0.0, 4.0 * p.M * p.cout_p * (p.splits + 1.0 + (POOL ? 0.25 : 0.0)), s);
  switch (p.splits) {
    case 2: splitk_epilogue_kernel<2, POOL><<<blocks, 256, 0, s>>>(p); break;
    case 3: splitk_epilogue_kernel<3, POOL><<<blocks, 256, 0, s>>>(p); break;
    case 4: splitk_epilogue_kernel<4, POOL><<<blocks, 256, 0, s>>>(p); break;
    case 8: splitk_epilogue_kernel<8, POOL><<<blocks, 256, 0, s>>>(p); break;
    case 9: splitk_epilogue_kernel<9, POOL><<<blocks, 256, 0, s>>>(p); break;
    default: splitk_epilogue_kernel<0, POOL><<<blocks, 256, 0, s>>>(p); break;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

int launch_splitk_epilogue(const ConvParams &p, hipStream_t s) {
  return p.pool_out ? launch_splitk_epilogue_t<true>(p, s) : launch_splitk_epilogue_t<false>(p, s);
}

// ---------------------------------------------------------------------------------------------
// Weight re-tiling (runs once per model in dt_unet_create).
// wp[((kc*n_p + n)*16) + 4*(slot ^ ((n>>2)&3)) + e] = W[n][c(k)][tap(k)],  k = kc*16 + 4*slot + e,
// k = tap*cin_p + cp; padded input channel cp maps to real channel c through the concat split
// (channels [0,split_c) live at [0,split_c) and channels [split_c,cin) at [split_cp, ...)).
__global__ void pack_conv_kernel(const float *__restrict__ w, float *__restrict__ wp, int cout, int cin, int ksize,
                                 int cin_p, int n_p, int split_c, int split_cp, size_t total) {
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int pos = idx & 15;
    const size_t rowi = idx >> 4;
    const int n = rowi % n_p;
    const int kc = rowi / n_p;
    const int pslot = pos >> 2, e = pos & 3;
    const int lslot = pslot ^ ((n >> 2) & 3);
    const int k = kc * 16 + lslot * 4 + e;
    const int tap = k / cin_p, cp = k - tap * cin_p;
    int c = -1;
    if (cp < split_cp) { if (cp < split_c) c = cp; }
    else { const int cc = split_c + (cp - split_cp); if (cc < cin) c = cc; }
    float v = 0.f;
    if (n < cout && c >= 0) v = w[((size_t)n * cin + c) * (ksize * ksize) + tap];
    wp[idx] = v;
  }
}

int launch_pack_conv(const float *w, float *wp, int cout, int cin, int ksize, int cin_p, int n_p, int split_c,
                     int split_cp, hipStream_t s) {
  const size_t total = (size_t)ksize * ksize * cin_p * n_p;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  pack_conv_kernel<<<blocks, 256, 0, s>>>(w, wp, cout, cin, ksize, cin_p, n_p, split_c, split_cp, total);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// eval BatchNorm folded to y = acc*scale + shift (reference models.py:62-63 / F.batch_norm eval):
// scale = g / sqrt(var + 1e-5), shift = (conv_bias - mean) * scale + beta.  With g == nullptr the
// layer is a plain conv: scale = 1, shift = conv_bias.  Padding channels get 0 / 0.
__global__ void fold_bn_kernel(const float *cb, const float *g, const float *b, const float *mean, const float *var,
                               float *scale, float *shift, int cout, int n_p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_p) return;
  float sc = 0.f, sh = 0.f;
  if (n < cout) {
    if (g) {
      sc = g[n] / sqrtf(var[n] + 1e-5f);
      sh = (cb[n] - mean[n]) * sc + b[n];
    } else {
      sc = 1.f;
      sh = cb[n];
    }
  }
  scale[n] = sc;
  shift[n] = sh;
}

int launch_fold_bn(const float *cb, const float *g, const float *b, const float *mean, const float *var, float *scale,
                   float *shift, int cout, int n_p, hipStream_t s) {
  fold_bn_kernel<<<(n_p + 255) / 256, 256, 0, s>>>(cb, g, b, mean, var, scale, shift, cout, n_p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// enc1 skip weights [cout][C][1][1] + bias -> w3[n_p][4] = (w0, w1, w2, bias), zero on padding
__global__ void pack_res3_kernel(const float *w, const float *b, float *w3, int cout, int C, int n_p) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= n_p) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n < cout) {
    v.x = w[n * C];
    if (C > 1) v.y = w[n * C + 1];
    if (C > 2) v.z = w[n * C + 2];
    v.w = b[n];
  }
  *reinterpret_cast<float4 *>(w3 + 4 * n) = v;
}

int launch_pack_res3(const float *w, const float *b, float *w3, int cout, int C, int n_p, hipStream_t s) {
  if (C > 3) return DT_E_SHAPE;
  pack_res3_kernel<<<(n_p + 255) / 256, 256, 0, s>>>(w, b, w3, cout, C, n_p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

// Implicit-GEMM convolution with fp32-accurate arithmetic on the bf16 matrix cores ("split-bf16").
//
// gfx950 has no reduced-precision fast path for fp32 inputs (no xf32/TF32); its exact fp32 MFMA runs at
// 1/16 of the bf16 rate.  This kernel keeps fp32 ACCURACY while using v_mfma_f32_32x32x16_bf16:
// every fp32 operand is split EXACTLY into three bf16 planes
//      a = a1 + a2 + a3,   a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)
// (round-to-nearest at each step; the residuals are exact in fp32 and the third one fits in 8 bits, so
// the three planes carry all 24 significand bits), and a*b is evaluated as the six plane products whose
// magnitude can reach 2^-24 of |a*b|:
//      a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1)          [dropped: a2b3, a3b2, a3b3 <= 2^-26 |ab|]
// Each bf16 x bf16 product is exact in the fp32 accumulator, so the result differs from an exact-product
// fp32 accumulation by < 2^-25 relative per term -- below fp32's own rounding.  Six bf16 MFMAs (6 x 32
// cycles for 32x32x16) replace eight fp32 MFMAs (8 x 64 cycles): 2.67x fewer matrix-pipe cycles.
//
// Weights are split once at pack time ([K/16][3 planes][N_pad][16 k], LDS-ready).  Activations stay fp32
// in HBM; a thread splits the 8 consecutive k it stages (2 x dwordx4 -> 3 x ds_write_b128, ~45 VALU).
// Tiling, tap walk, split-K and epilogue are those of dt_conv.hip.  LDS: per stage 3 planes x (BM+BN) rows
// x 32 B; the two 16-B halves of a row are swapped on rows with bit 3 set, which makes ds_read_b128's
// 16-lane groups conflict-free (rows 32 B apart would otherwise collide two-way).
#include "dt_conv_epilogue.h"

namespace dt {

// ABL != 0 builds are timing experiments (tools/ablate.py) and produce wrong results by design:
// 1 no barrier, 2 no LDS fragment reads after the first chunk, 3 no MFMA, 4 no global loads, 5 no split VALU
// Wave layout: WM x WN waves (64*WM*WN threads) tile the BM x BN block; each wave owns (BM/WM) x (BN/WN).
// Only 2x2 waves of <= 64x64 are instantiated: 256-row tiles (2x4 waves of 128x64, 4x2 of 64x64, 2x2 of 128x64)
// were measured at 81 / 145 / 103 TF/s against 179 for 128x128 and are not built.
template <int BM, int BN, int ABL = 0, int WM = 2, int WN = 2>
__global__ __launch_bounds__(64 * WM * WN, BM * BN > 128 * 128 ? 2 : 3) void conv_gemm_bf16x6_kernel(const ConvParams p) {
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
  constexpr int PLANE_A = BM * 16, PLANE_B = BN * 16;            // bf16 elements per plane per stage
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);                  // bf16 elements per stage
  __shared__ __attribute__((aligned(16))) __bf16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HW = p.H * p.W;
  const int CC = p.cin_p >> 4;

  // ---- A staging: item -> (row, k-half): 8 consecutive k = 32 contiguous bytes of one pixel; AP items per thread
  constexpr int NT = 64 * WM * WN;
  constexpr int AP = (BM * 2 + NT - 1) / NT, BP = (BN * 2 + NT - 1) / NT;
  bool a_ok[AP];
  int a_y[AP], a_x[AP], a_off[AP], a_off2[AP], a_lds[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int item = tid + i * NT;
    const int rowi = item >> 1, hh = item & 1;
    const int m = m0 + rowi;
    a_ok[i] = item < BM * 2 && m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int b = mm / HW, rem = mm - b * HW;
    a_y[i] = rem / p.W; a_x[i] = rem - a_y[i] * p.W;
    a_off[i] = mm * p.cin_p + hh * 8;
    a_off2[i] = mm * p.cin2_p + hh * 8;                               // fused skip walk
    a_lds[i] = rowi * 16 + ((hh ^ ((rowi >> 3) & 1)) << 3);          // element offset inside a plane
  }
  // ---- B staging: the three plane tiles are contiguous [BN][16] bf16 runs in the packed weights
  const __bf16 *wbase = reinterpret_cast<const __bf16 *>(p.w) + (size_t)n0 * 16 + tid * 8;
  const size_t w_plane = (size_t)p.n_p * 16;                       // elements between planes of one chunk

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  int a_frag[MI], b_frag[NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * (MI * 32) + mi * 32 + l31;
    a_frag[mi] = row * 16 + ((half ^ ((row >> 3) & 1)) << 3);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = wn * (NI * 32) + ni * 32 + l31;
    b_frag[ni] = row * 16 + ((half ^ ((row >> 3) & 1)) << 3);
  }

  // split-K: grid.z slices the (tap, channel chunk) walk into equal runs of whole chunks
  const int n_main = (p.tap_hi - p.tap_lo) * CC / p.splits;
  const int n_iter = n_main + (p.in2 ? (p.cin2_p >> 4) : 0);       // main walk, then the fused 1x1 skip walk
  int tap = p.tap_lo + (blockIdx.z * n_main) / CC, cc = (blockIdx.z * n_main) % CC;
  for (int it = -1; it < n_iter; ++it) {
    const bool more = it + 1 < n_iter;
    f32x4 ra0[AP], ra1[AP];
    u32x4 rb[BP][3];
#pragma unroll
    for (int i = 0; i < AP; ++i) { ra0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ra1[i] = ra0[i]; }
    if (ABL == 4) {
      ra0[0] = f32x4{1.f, 2.f, 3.f, 4.f}; ra1[0] = ra0[0];
      rb[0][0] = rb[0][1] = rb[0][2] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
      if (more && ++cc == CC) { cc = 0; ++tap; }
    } else if (more && it + 1 >= n_main) {
      const int c2 = it + 1 - n_main;
#pragma unroll
      for (int i = 0; i < AP; ++i)
        if (a_ok[i]) {
          const float *src = p.in2 + a_off2[i] + c2 * 16;
          ra0[i] = *reinterpret_cast<const f32x4 *>(src);
          ra1[i] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
      const __bf16 *wt = reinterpret_cast<const __bf16 *>(p.w2) + (size_t)n0 * 16 + tid * 8 + (size_t)c2 * 3 * w_plane;
#pragma unroll
      for (int i = 0; i < BP; ++i)
        if (tid + i * NT < BN * 2) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) rb[i][pl] = *reinterpret_cast<const u32x4 *>(wt + i * NT * 8 + pl * w_plane);
        }
    } else if (more) {
      int dy = 0, dx = 0;
      if (p.ksize == 3) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int yy = a_y[i] + dy, xx = a_x[i] + dx;
        if (a_ok[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) {
          const float *src = p.in + a_off[i] + (dy * p.W + dx) * p.cin_p + cc * 16;
          ra0[i] = *reinterpret_cast<const f32x4 *>(src);
          ra1[i] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
      }
      const __bf16 *wt = wbase + (size_t)(tap * p.ccw + cc) * 3 * w_plane;
#pragma unroll
      for (int i = 0; i < BP; ++i)
        if (tid + i * NT < BN * 2) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) rb[i][pl] = *reinterpret_cast<const u32x4 *>(wt + i * NT * 8 + pl * w_plane);
        }
      if (++cc == CC) { cc = 0; ++tap; }
    }
    if (it >= 0) {
      if (it == n_main && p.in2) conv_midpoint<MI, NI>(p, acc, n0, wn, l31);
      const __bf16 *A = lds + (it & 1) * STAGE, *B = A + 3 * PLANE_A;
      bf16x8 fb[NI][3];
      u32x4 fake = {(unsigned)lane * 0x01010101u, 0x3f803f80u, (unsigned)it, 0x3c003c00u};   // ABL == 2 only
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fb[ni][pl] = ABL == 2 ? __builtin_bit_cast(bf16x8, fake) : *reinterpret_cast<const bf16x8 *>(B + pl * PLANE_B + b_frag[ni]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        bf16x8 fa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fa[pl] = ABL == 2 ? __builtin_bit_cast(bf16x8, fake) : *reinterpret_cast<const bf16x8 *>(A + pl * PLANE_A + a_frag[mi]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          // smallest terms first so their sum is formed before it meets the large partial sums
          f32x16 c = acc[mi][ni];
          if (ABL == 3) {              // keep the fragments live, skip the matrix work
            asm volatile("" :: "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fb[ni][0]), "v"(fb[ni][1]), "v"(fb[ni][2]));
            continue;
          }
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[ni][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[ni][0], c, 0, 0, 0);
          acc[mi][ni] = c;
        }
      }
    }
    if (more) {
      __bf16 *A = lds + ((it + 1) & 1) * STAGE, *B = A + 3 * PLANE_A;
#pragma unroll
      for (int i = 0; i < AP; ++i)
        if (tid + i * NT < BM * 2) {
          bf16x8 p1, p2, p3;
          if (ABL == 5) {                // no split VALU: park raw bits
            p1 = __builtin_bit_cast(bf16x8, ra0[i]); p2 = __builtin_bit_cast(bf16x8, ra1[i]); p3 = p1;
          } else
          split8(ra0[i], ra1[i], p1, p2, p3);
          *reinterpret_cast<bf16x8 *>(A + a_lds[i]) = p1;
          *reinterpret_cast<bf16x8 *>(A + PLANE_A + a_lds[i]) = p2;
          *reinterpret_cast<bf16x8 *>(A + 2 * PLANE_A + a_lds[i]) = p3;
        }
#pragma unroll
      for (int i = 0; i < BP; ++i)
        if (tid + i * NT < BN * 2) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(B + pl * PLANE_B + (tid + i * NT) * 8) = rb[i][pl];
        }
    }
    if (ABL != 1) __syncthreads();
  }
  static_assert(2 * STAGE * sizeof(__bf16) >= epilogue_stage_floats<BN>() * sizeof(float), "epilogue stage");
  conv_epilogue<MI, NI>(p, acc, reinterpret_cast<float *>(lds), m0, n0, wm, wn, half, l31);
}

int launch_conv_bf16x6(const ConvParams &p, int bm, int bn, hipStream_t s) {
  dim3 grid((p.M + bm - 1) / bm, p.n_p / bn, p.splits);
#ifdef DT_TOOLS   // ablation instantiations (wrong results by design) exist only in a tools build (build.py --tools)
  if (p.ablate && bm == 128 && bn == 128) {     // timing experiments (tools/ablate.py)
    switch (p.ablate) {
      case 1: conv_gemm_bf16x6_kernel<128, 128, 1><<<grid, 256, 0, s>>>(p); break;
      case 2: conv_gemm_bf16x6_kernel<128, 128, 2><<<grid, 256, 0, s>>>(p); break;
      case 3: conv_gemm_bf16x6_kernel<128, 128, 3><<<grid, 256, 0, s>>>(p); break;
      case 4: conv_gemm_bf16x6_kernel<128, 128, 4><<<grid, 256, 0, s>>>(p); break;
      case 5: conv_gemm_bf16x6_kernel<128, 128, 5><<<grid, 256, 0, s>>>(p); break;
      default: return DT_E_ARG;
    }
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
#endif
  if (bm == 128 && bn == 128) conv_gemm_bf16x6_kernel<128, 128><<<grid, 256, 0, s>>>(p);
  else if (bm == 128) conv_gemm_bf16x6_kernel<128, 64><<<grid, 256, 0, s>>>(p);
  else if (bn == 128) conv_gemm_bf16x6_kernel<64, 128><<<grid, 256, 0, s>>>(p);
  else conv_gemm_bf16x6_kernel<64, 64><<<grid, 256, 0, s>>>(p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight split + re-tiling (once per model): wp is bf16 [K/16][3][n_p][16], element k = 8*hh + j of row n
// stored at half (hh ^ ((n>>3)&1)); k = tap*cin_p + cp with the same concat mapping as pack_conv_kernel.
__global__ void pack_conv_bf16x3_kernel(const float *__restrict__ w, __bf16 *__restrict__ wp, int cout, int cin,
                                        int ksize, int cin_p, int cin_w, int n_p, int split_c, int split_cp, size_t total) {
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    // idx enumerates (kc, n, phys position 0..15); the three planes are written together
    const int pos = idx & 15;
    const size_t rowi = idx >> 4;
    const int n = rowi % n_p;
    const size_t kc = rowi / n_p;
    const int phh = pos >> 3, j = pos & 7;
    const int hh = phh ^ ((n >> 3) & 1);
    const int k = (int)kc * 16 + hh * 8 + j;
    const int tap = k / cin_w, cp = k - tap * cin_w;       // (channels [cin_p, cin_w) of a tap: zero chunks, see ConvParams::ccw)
    int c = -1;
    if (cp < split_cp) { if (cp < split_c) c = cp; }
    else if (cp < cin_p) { const int cc = split_c + (cp - split_cp); if (cc < cin) c = cc; }
    float v = 0.f;
    if (n < cout && c >= 0) v = w[((size_t)n * cin + c) * (ksize * ksize) + tap];
    const __bf16 a1 = (__bf16)v;
    const float r1 = v - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const __bf16 a3 = (__bf16)(r1 - (float)a2);
    const size_t base = (kc * 3 * n_p + n) * 16 + pos;
    const size_t plane = (size_t)n_p * 16;
    wp[base] = a1; wp[base + plane] = a2; wp[base + 2 * plane] = a3;
  }
}

int launch_pack_conv_bf16x3(const float *w, void *wp, int cout, int cin, int ksize, int cin_p, int cin_w, int n_p, int split_c,
                            int split_cp, hipStream_t s) {
  if (cin_w < cin_p || cin_w % 16) return DT_E_ARG;
  const size_t total = (size_t)ksize * ksize * cin_w * n_p;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  pack_conv_bf16x3_kernel<<<blocks, 256, 0, s>>>(w, reinterpret_cast<__bf16 *>(wp), cout, cin, ksize, cin_p, cin_w, n_p,
                                                 split_c, split_cp, total);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

// Split-bf16 implicit-GEMM convolution fed by LDS-DMA (global_load_lds) from PRE-SPLIT activations.
//
// Same arithmetic as dt_conv_bf16.hip (exact 3-plane bf16 split, six plane products per chunk on
// v_mfma_f32_32x32x16_bf16, fp32 accumulate), but the activation planes are produced once by the layer
// that writes the tensor (conv epilogues, pool, upsample+concat all emit a bf16 "planes" twin
// [M][C/16][3][16] next to the fp32 NHWC tensor) instead of being re-split by every consumer tile.
// Both operand tiles are then pure copies, so they go global -> LDS with `global_load_lds_dwordx4`:
// no staging VGPRs, no VALU split, no ds_write in the K loop -- a wave's instruction stream per chunk is
// 3-6 DMA issues, 12 ds_read_b128 and 24 MFMAs.
//
// LDS-DMA writes `M0 base + lane*16 B`, i.e. one wave-instruction fills 1 KiB = 32 rows x 32 B of one plane
// linearly; the per-lane GLOBAL address is free, so the half-swap swizzle of the LDS image (rows with bit 3
// set keep their two 16-B halves exchanged, see dt_conv_bf16.hip) is applied on the source side, and
// out-of-image taps / rows beyond M read a zero page.  Pipeline: at the top of iteration i every wave
// waits for its own DMAs of chunk i (vmcnt), the workgroup barrier makes all of them visible and also
// retires every read of the other stage, then the DMAs of chunk i+1 are issued into that stage and fly
// during the MFMAs of chunk i.  One barrier per chunk.  (Implemented as a 3-stage ring with the DMAs two
// chunks ahead and a counted vmcnt, so an L2 miss has two chunk times to land.)
#include "dt_conv_epilogue.h"

namespace dt {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv_gemm_bf16x6_dma_kernel(const ConvParams p) {
  constexpr int WN = 2;
  constexpr int MI = BM / 64, NI = BN / 64;
  constexpr int PLANE_A = BM * 16, PLANE_B = BN * 16;            // bf16 elements per plane per stage
  constexpr int STAGE = 3 * (PLANE_A + PLANE_B);
  constexpr int GA = BM / 32, GB = BN / 32;                        // 32-row groups (one DMA instr per group and plane)
  constexpr int NSTAGE = 3;                                       // ring: chunk i lives in stage i % 3
  __shared__ __attribute__((aligned(16))) __bf16 lds[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HW = p.H * p.W;
  const int CC = p.cin_p >> 4;

  // ---- DMA roles.  A row group g (32 rows) is owned by wave g; B row groups are spread so that every
  // wave issues a similar number of DMAs: 128x128 -> 3 A + 3 B each, 64x64 -> waves 0,1 feed A, 2,3 feed B.
  const bool a_owner = wave < GA;
  const int b_group = GA + GB <= 4 ? wave - GA : wave;             // which B row group this wave feeds (may be out of range)
  const bool b_owner = b_group >= 0 && b_group < GB;
  // lane -> (row inside the group, physical 16-B half); the logical half is un-swizzled on the source side
  const int d_row = lane >> 1, d_half = lane & 1;
  const int a_rowi = wave * 32 + d_row;
  const int a_m = m0 + a_rowi;
  const bool a_ok = a_owner && a_m < p.M;
  const int a_mm = a_ok ? a_m : 0;
  const int a_b = a_mm / HW, a_rem = a_mm - a_b * HW;
  const int a_y = a_rem / p.W, a_x = a_rem - a_y * p.W;
  const int a_lhalf = d_half ^ ((a_rowi >> 3) & 1);
  const __bf16 *in_pl = reinterpret_cast<const __bf16 *>(p.in_pl);
  const __bf16 *in2_pl = reinterpret_cast<const __bf16 *>(p.in2_pl);
  const __bf16 *zero = reinterpret_cast<const __bf16 *>(p.zero);
  const __bf16 *wbase = reinterpret_cast<const __bf16 *>(p.w) + (size_t)(n0 + b_group * 32) * 16 + lane * 8;
  const __bf16 *wbase2 = reinterpret_cast<const __bf16 *>(p.w2) + (size_t)(n0 + b_group * 32) * 16 + lane * 8;
  const size_t w_plane = (size_t)p.n_p * 16;

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
  const int n_iter = n_main + (p.in2_pl ? (p.cin2_p >> 4) : 0);    // main walk, then the fused 1x1 skip walk
  int tap = p.tap_lo + (blockIdx.z * n_main) / CC, cc = (blockIdx.z * n_main) % CC, next = 0;

  // issue the DMAs of chunk `next` into LDS stage `stage`
  auto issue = [&](int stage) __attribute__((always_inline)) {
    __bf16 *A = lds + stage * STAGE, *B = A + 3 * PLANE_A;
    const __bf16 *asrc = zero, *bsrc;
    if (next >= n_main) {
      const int c2 = next - n_main;
      if (a_ok) asrc = in2_pl + ((size_t)a_mm * (p.cin2_p >> 4) + c2) * 48 + a_lhalf * 8;
      bsrc = wbase2 + (size_t)c2 * 3 * w_plane;
    } else {
      int dy = 0, dx = 0;
      if (p.ksize == 3) { dy = tap / 3 - 1; dx = tap - (tap / 3) * 3 - 1; }
      const int yy = a_y + dy, xx = a_x + dx;
      if (a_ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
        asrc = in_pl + ((size_t)(a_mm + dy * p.W + dx) * CC + cc) * 48 + a_lhalf * 8;
      bsrc = wbase + (size_t)(tap * CC + cc) * 3 * w_plane;
      if (++cc == CC) { cc = 0; ++tap; }
    }
    ++next;
    const bool a_zero = asrc == zero;
    if (a_owner) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        __builtin_amdgcn_global_load_lds(a_zero ? asrc : asrc + pl * 16, A + pl * PLANE_A + wave * 512, 16, 0, 0);
    }
    if (b_owner) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        __builtin_amdgcn_global_load_lds(bsrc + pl * w_plane, B + pl * PLANE_B + b_group * 512, 16, 0, 0);
    }
  };

  // Ring of three stages, DMAs two chunks ahead.  vmcnt counts this wave's outstanding DMA instructions in
  // issue order, so "all but the youngest chunk's" = vmcnt(n_dma) retires chunk `it` while chunk it+1 stays
  // in flight across the barrier (raw s_barrier: __syncthreads() would drain vmcnt to 0).
  const int n_dma = (a_owner ? 3 : 0) + (b_owner ? 3 : 0);
  issue(0);
  if (n_iter > 1) issue(1);
  for (int it = 0; it < n_iter; ++it) {
    if (it + 1 < n_iter) {
      if (n_dma == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (n_dma == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();          // chunk `it` is complete for every wave; reads of stage (it+2)%3 are retired
    asm volatile("" ::: "memory");
    if (it + 2 < n_iter) issue((it + 2) % NSTAGE);
    if (it == n_main && p.in2_pl) conv_midpoint<MI, NI>(p, acc, n0, wn, l31);
    const __bf16 *A = lds + (it % NSTAGE) * STAGE, *B = A + 3 * PLANE_A;
    bf16x8 fb[NI][3];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fb[ni][pl] = *reinterpret_cast<const bf16x8 *>(B + pl * PLANE_B + b_frag[ni]);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      bf16x8 fa[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[pl] = *reinterpret_cast<const bf16x8 *>(A + pl * PLANE_A + a_frag[mi]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        f32x16 c = acc[mi][ni];
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
  conv_epilogue<MI, NI>(p, acc, m0, n0, wm, wn, half, l31);
}

int launch_conv_bf16x6_dma(const ConvParams &p, int bm, int bn, hipStream_t s) {
  if (!p.in_pl || !p.zero || (p.in2 && !p.in2_pl)) return DT_E_NULL;
  dim3 grid((p.M + bm - 1) / bm, p.n_p / bn, p.splits);
  if (bm == 128 && bn == 128) conv_gemm_bf16x6_dma_kernel<128, 128><<<grid, 256, 0, s>>>(p);
  else if (bm == 128) conv_gemm_bf16x6_dma_kernel<128, 64><<<grid, 256, 0, s>>>(p);
  else if (bn == 128) conv_gemm_bf16x6_dma_kernel<64, 128><<<grid, 256, 0, s>>>(p);
  else conv_gemm_bf16x6_dma_kernel<64, 64><<<grid, 256, 0, s>>>(p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

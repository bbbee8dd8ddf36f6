// 3x3 convolution on the split-bf16 arithmetic of dt_conv_bf16.hip with the activation tile staged ONCE per
// 16-channel chunk and reused by all nine taps ("strip" kernel).
//
// The plain implicit-GEMM kernel re-stages its BM x 16 activation tile for every (tap, chunk): nine global
// loads, nine 3-way bf16 splits and nine LDS writes of (nearly) the same pixels.  Here a workgroup keeps the
// tile's rows PLUS a halo of W+1 pixels on either side in LDS as three bf16 planes:
//      strip row s  <->  pixel m0 - (W+1) + s,      s in [0, BM + 2(W+1))
// and tap (dy, dx) of output row r reads strip row r + (W+1) + dy*W + dx -- the same LDS image at a shifted row
// (halo W instead of W+1 when the tile covers whole picture rows, see strip_halo).
// Taps that fall outside the picture (zero padding, also across picture boundaries inside a tile) read one of
// eight all-zero rows instead (the one with the same bank as the real row would have: no conflicts); validity
// is a 9-bit mask per fragment row, computed once.
// Per chunk: one strip load/split/write, then nine steps that only stage the 3 x BN x 16 weight tile of their
// tap (double buffered) and run 24 MFMAs per wave.  ds_read_b128 stays conflict-free at any shift because the
// half-swap swizzle is a function of the strip row (rows 8 apart always differ in it).
//
// (Measured and not kept: weight fragments straight from global memory -- no weight LDS traffic, no barrier per
// tap, double-buffered strip -- 3-8 % slower (181 vs 186, 149 vs 156, 171 vs 181 TF/s); the weight tile through
// global_load_lds into a ring of three, prefetched two taps ahead with counted vmcnt -- 1-5 % slower.)
//
// Split-K runs over channel chunks (grid.z slabs, summed in z order by splitk_epilogue_kernel); the block's
// fused 1x1 skip walk follows as single-tap chunks over in2 / w2, as in the plain kernel.
#include <mutex>
#include <type_traits>

#include "dt_conv_epilogue.h"

namespace dt {

extern __shared__ __attribute__((aligned(16))) __bf16 strip_lds[];

// Halo rows either side of the tile.  The corner taps reach W + 1 pixels back / ahead; when the tile starts at x = 0 and
// ends at x = W - 1 (BM a multiple of W) those two reads are out-of-picture taps of the first / last row and go to the
// zero rows anyway, so W rows are enough -- which is what lets the K = 32 tile fit twice per CU at W = 16.
__host__ __device__ inline int strip_halo(int W, int bm) { return bm % W == 0 ? W : W + 1; }

// ABL != 0: timing experiments (wrong results): 1 no per-tap barrier, 2 no weight staging, 3 no MFMA,
// 4 no fragment reads after the first step, 5 no strip re-staging
// KC = 16-channel chunks staged and multiplied per step (1 or 2): KC = 2 halves the barriers and doubles the
// MFMAs between them at twice the LDS footprint and staging registers (2 waves/SIMD instead of 3).
// WK = waves that share one output tile and split the step's KC chunks between them (1, 2 or 4): small layers need
// small workgroup tiles to fill the chip, and a 64 x 64 tile cut 2 x 2 leaves each wave 32 x 32 (6 fragment reads per
// 6 MFMAs, ~100 TF/s).  With WK = 4 every wave keeps a 64 x 64 accumulator (12 reads per 24 MFMAs) over a quarter of
// the channels and the four partial tiles are summed in wave order in the staged epilogue -- split-K without slabs.
template <int BM, int BN, int ABL = 0, int KC = 1, int WK = 1>
__global__ __launch_bounds__(256, KC == 1 ? 3 : 2) void conv_strip_bf16x6_kernel(const ConvParams p) {
  // four waves: 2 x 2 over the tile, or 4 x 1 for the 256 x 64 tile (a 64 x 64 wave tile -- 12 fragment reads per 24
  // MFMAs, like the 128 x 128 tile -- for layers with 64 output channels, where 128 x 64 leaves each wave 64 x 32),
  // or WM x 1 x WK with the K split
  constexpr int WN = WK > 1 ? BN / 64 : (BM == 256 ? 1 : 2), WM = 4 / (WN * WK), NT = 256;
  constexpr int MI = BM / (32 * WM), NI = BN / (32 * WN);
  constexpr int KW = KC / WK;                                      // chunks of a step one wave multiplies
  static_assert(KC % WK == 0 && MI * 32 * WM == BM && NI * 32 * WN == BN && (WK == 1 || (MI == 2 && NI == 2)), "wave layout");
  constexpr int PLANE_B = BN * 16, STAGE_B = 3 * PLANE_B;         // bf16 elements
  constexpr int AP = BM == 256 ? 3 : (KC == 4 ? 1 : 2);            // strip items (row, k-half) per thread (K = 64 steps: rows of at most 31 px)
  const int halo = strip_halo(p.W, BM);
  const int R = BM + 2 * halo;                                     // strip rows
  const int RZ = (R + 7) & ~7;                                     // 8 all-zero rows start here (multiple of 8)
  const int PLANE_A = (RZ + 8) * 16;
  __bf16 *As = strip_lds;                                          // [KC][3][RZ+8][16]
  __bf16 *Bs = strip_lds + KC * 3 * PLANE_A;                       // [2][KC][3][BN][16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: chunk and tile offsets stay in scalar registers
  const int wk = wave % WK, wm = (wave / WK) / WN, wn = (wave / WK) % WN;
  const int half = lane >> 5, l31 = lane & 31;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int HW = p.H * p.W;
  const int CC = p.cin_p >> 4;

  // ---- strip staging: item -> (strip row, k-half), 32 contiguous bytes of one pixel
  bool s_in[AP], s_ok[AP], s_core[AP];
  int s_off[AP], s_off2[AP], s_offb[AP], s_lds[AP];
#pragma unroll
  for (int i = 0; i < AP; ++i) {
    const int item = tid + i * NT;
    const int srow = item >> 1, hh = item & 1;
    const int m = m0 - halo + srow;
    s_in[i] = item < 2 * R;
    s_ok[i] = s_in[i] && m >= 0 && m < p.M;
    s_core[i] = s_ok[i] && srow >= halo && srow < halo + BM;       // rows the single-tap skip walk needs
    const int mm = s_ok[i] ? m : 0;
    s_off[i] = mm * p.cin_p + hh * 8;
    s_off2[i] = mm * p.cin2_p + hh * 8;
    s_offb[i] = mm * p.b_stride + hh * 8;                             // two-source block input (decoder concat)
    s_lds[i] = srow * 16 + ((hh ^ ((srow >> 3) & 1)) << 3);
  }
  // ---- weight staging: the three plane tiles of one (tap, chunk) are contiguous [BN][16] bf16 runs
  // (with the K split and BN = 64 both thread halves stage: even / odd (chunk, plane) tiles)
  constexpr bool B_ALL = WK > 1 && BN * 4 == NT;
  constexpr int NB = B_ALL ? KC * 3 / 2 : KC * 3, SB = B_ALL ? 2 : 1;
  static_assert(!B_ALL || (KC * 3) % 2 == 0, "weight staging");
  const bool b_thread = B_ALL || tid < BN * 2;
  const int bt = B_ALL ? (tid & (BN * 2 - 1)) : tid, bsub = B_ALL ? tid / (BN * 2) : 0;
  const size_t w_plane = (size_t)p.n_p * 16;
  const __bf16 *wbase = reinterpret_cast<const __bf16 *>(p.w) + (size_t)n0 * 16 + bt * 8 + bsub * w_plane;
  const __bf16 *wbase2 = reinterpret_cast<const __bf16 *>(p.w2) + (size_t)n0 * 16 + bt * 8 + bsub * w_plane;
  const int b_lds = bsub * PLANE_B + bt * 8;

  // ---- fragment rows: strip row of the centre tap and the 9-bit tap-validity mask
  int a_row[MI];
  unsigned a_mask[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row_i = wm * (MI * 32) + mi * 32 + l31;
    const int m = m0 + row_i;
    a_row[mi] = row_i + halo;
    unsigned mask = 0;
    if (m < p.M) {
      const int rem = m % HW;
      const int y = rem / p.W, x = rem - y * p.W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1u << t;
      }
    }
    a_mask[mi] = mask;
  }
  int b_frag[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int row = wn * (NI * 32) + ni * 32 + l31;
    b_frag[ni] = row * 16 + ((half ^ ((row >> 3) & 1)) << 3);
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  // steps-worth of KC channel chunks in this z slice; the walk may run past the layer's CC chunks (odd chunk counts): the
  // weight pack holds zero chunks there (p.ccw per tap, checked by the launcher) and the activation loads are skipped
  const int n_main = (CC + p.splits * KC - 1) / (p.splits * KC);
  const int cc0 = blockIdx.z * n_main * KC;
  const int CC2 = p.cin2_p >> 4;
  const int n_chunks = n_main + (p.in2 ? (CC2 + KC - 1) / KC : 0);

  f32x4 sa0[KC][AP], sa1[KC][AP];
  u32x4 rb[NB];
  auto load_strip = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      sa0[kk][i] = f32x4{0.f, 0.f, 0.f, 0.f}; sa1[kk][i] = sa0[kk][i];
      if (ch < n_main) {
        const int cc = cc0 + ch * KC + kk;
        if (s_ok[i] && cc < CC) {
          const float *src = (p.in_b && cc >= p.cc_a) ? p.in_b + s_offb[i] + (cc - p.cc_a) * 16 : p.in + s_off[i] + cc * 16;
          sa0[kk][i] = *reinterpret_cast<const f32x4 *>(src);
          sa1[kk][i] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
      } else if (s_core[i] && (ch - n_main) * KC + kk < CC2) {
        const int c2 = (ch - n_main) * KC + kk;
        const float *src = (p.in2_b && c2 >= p.cc_a) ? p.in2_b + s_offb[i] + (c2 - p.cc_a) * 16 : p.in2 + s_off2[i] + c2 * 16;
        sa0[kk][i] = *reinterpret_cast<const f32x4 *>(src);
        sa1[kk][i] = *reinterpret_cast<const f32x4 *>(src + 4);
      }
    }
  };
  auto write_strip = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int kk = 0; kk < KC; ++kk)
#pragma unroll
    for (int i = 0; i < AP; ++i)
      if (s_in[i]) {
        bf16x8 p1, p2, p3;
        split8(sa0[kk][i], sa1[kk][i], p1, p2, p3);
        __bf16 *A = As + kk * 3 * PLANE_A;
        *reinterpret_cast<bf16x8 *>(A + s_lds[i]) = p1;
        *reinterpret_cast<bf16x8 *>(A + PLANE_A + s_lds[i]) = p2;
        *reinterpret_cast<bf16x8 *>(A + 2 * PLANE_A + s_lds[i]) = p3;
      }
  };
  // The weight tiles are visited in a fixed order, so a running pointer replaces per-step index arithmetic:
  // next tap of the chunk group (+tap_stride), first tap of the next group, the skip weights, next skip group.
  const size_t tap_stride = (size_t)p.ccw * 3 * w_plane, group_stride = (size_t)KC * 3 * w_plane;
  const __bf16 *wrun = wbase + (size_t)cc0 * 3 * w_plane;
  enum { ADV_NONE = 0, ADV_TAP, ADV_GROUP, ADV_SKIP0, ADV_SKIP };
  auto load_b = [&](int adv) __attribute__((always_inline)) {
    if (adv == ADV_TAP) wrun += tap_stride;
    else if (adv == ADV_GROUP) wrun = wrun - 8 * tap_stride + group_stride;
    else if (adv == ADV_SKIP0) wrun = wbase2;
    else if (adv == ADV_SKIP) wrun += group_stride;
    if (b_thread) {
      const __bf16 *wt = ABL == 6 ? wbase : wrun;                  // ABL 6: every tile re-reads the first one (cache-hot)
#pragma unroll
      for (int i = 0; i < NB; ++i)                                 // tile i = (chunk, plane): consecutive chunks of one tap are 3 planes apart
        rb[i] = *reinterpret_cast<const u32x4 *>(wt + (SB * i) * w_plane);
    }
  };
  auto write_b = [&](int stage) __attribute__((always_inline)) {
    if (b_thread) {
      __bf16 *B = Bs + stage * KC * STAGE_B;
#pragma unroll
      for (int i = 0; i < NB; ++i) *reinterpret_cast<u32x4 *>(B + (SB * i) * PLANE_B + b_lds) = rb[i];
    }
  };

  // ---- K-split kernels: weight fragments straight from global memory.  With the step's chunks split across the waves a
  // weight tile is multiplied by ONE wave (64 x 64 tile) or two (128 x 64, 64 x 128): staging it through LDS shares nothing
  // and costs a barrier per tap.  The packed tile in global memory IS the LDS image (half-swap swizzle included), so a lane's
  // B fragment is one 16-byte load; the fragments of step s + 1 are requested before the MFMAs of step s, and the only
  // barriers left are the two around a strip restage.
  constexpr bool DB = WK > 1 && ABL == 0;
  bf16x8 fbn[KW][NI][3];
  auto load_bd = [&](int adv) __attribute__((always_inline)) {
    if (adv == ADV_TAP) wrun += tap_stride;
    else if (adv == ADV_GROUP) wrun = wrun - 8 * tap_stride + group_stride;
    else if (adv == ADV_SKIP0) wrun = wbase2;
    else if (adv == ADV_SKIP) wrun += group_stride;
    const __bf16 *wt = wrun - (bt * 8 + bsub * w_plane);           // the tile's first element (wrun carries this thread's staging offset)
#pragma unroll
    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          fbn[kw][ni][pl] = *reinterpret_cast<const bf16x8 *>(wt + ((wk * KW + kw) * 3 + pl) * w_plane + b_frag[ni]);
  };

  const long long tl0 = p.ablate == 8 ? wall_clock64() : 0;      // timeline diagnostic (tools/block_timeline.py)
  // ---- prologue: strip of chunk 0, weights of step 0, the zero row
  load_strip(0);
  if (DB) load_bd(ADV_NONE); else load_b(ADV_NONE);
  if (tid < 48 * KC) *reinterpret_cast<u32x4 *>(As + (tid >> 4) * PLANE_A + RZ * 16 + (tid & 15) * 8) = u32x4{0u, 0u, 0u, 0u};
  write_strip();
  if (!DB) write_b(0);
  __syncthreads();

  // ABL == 9 (diagnostic build, tools/tap_phases.py): s_memtime stamps split every step into
  //   [0] fragment reads + MFMA issue, [1] wait for the next tap's weight loads, [2] their LDS writes, [3] barrier wait
  // summed per wave in scalar registers; the real kernel executes none of this.
  unsigned long long ph[4] = {0, 0, 0, 0}, ts = 0;
  auto stamp = [&](int seg) __attribute__((always_inline)) {
    if (ABL != 9) return;
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (seg >= 0) ph[seg] += t - ts;
    ts = t;
  };
  stamp(-1);
  // One step = one tap of one (KC-chunk) group.  TT is the tap as a compile-time constant: the nine taps of a main
  // chunk are unrolled, so shifts, mask bits and the whole walk bookkeeping fold away and only the chunk loop is
  // left as scalar control.
  const long long tl_pro = p.ablate == 8 ? wall_clock64() : 0;
  int step = 0;
  bool stage_next_strip = true;            // false: the next step is a skip step that takes its activations straight from global memory
  auto do_step = [&](auto TT, int ch, bool first_tap, bool last_tap, bool next_chunk, int adv) __attribute__((always_inline)) {
    constexpr int tt = decltype(TT)::value;
    const bool more = !last_tap || next_chunk;
    if (first_tap && next_chunk && stage_next_strip && ABL != 5) load_strip(ch + 1);   // lands while this chunk's taps run
    bf16x8 fbc[KW][NI][3];                                         // DB: this step's weight fragments (requested during the previous step)
    if (DB) {
#pragma unroll
      for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fbc[kw][ni][pl] = fbn[kw][ni][pl];
      if (more) load_bd(adv);
    } else if (more && ABL != 2) load_b(adv);
    {
      int wv = p.W;
      asm volatile("" : "+s"(wv));           // likewise: do not keep nine precomputed shifts in SGPRs
      const int shift = (tt / 3 - 1) * wv + (tt % 3 - 1);
      int a_e[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        // an out-of-picture tap reads zero row RZ + (srow & 7) at the same physical half (a row's 16-B slot in the 256-B
        // bank row depends on srow & 7 and, through the half swap, on bit 3 -- which `half ^ ...` below keeps): the bank a lane hits is
        // the one its in-picture read would hit, so any mix of the two stays conflict-free
        int row0 = a_row[mi];
        asm volatile("" : "+v"(row0));       // keeps the nine taps' addresses from being hoisted out of the chunk loop (18+ VGPRs)
        const int srow = row0 + shift;
        const int lrow = ((a_mask[mi] >> tt) & 1u) ? srow : RZ + (srow & 7);
        a_e[mi] = lrow * 16 + ((half ^ ((srow >> 3) & 1)) << 3);
      }
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int kk = wk * KW + kw;                                // this wave's chunk of the step (all of them without the K split)
        const __bf16 *A = As + kk * 3 * PLANE_A;
        const __bf16 *B = Bs + ((step & 1) * KC + kk) * STAGE_B;
        bf16x8 fb[NI][3];
        const u32x4 fake = {(unsigned)lane * 0x01010101u + (unsigned)a_e[0], 0x3f803f80u, (unsigned)step, 0x3c003c00u};   // ABL == 4 only
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            fb[ni][pl] = DB ? fbc[kw][ni][pl]
                            : ((ABL == 4) ? __builtin_bit_cast(bf16x8, fake) : *reinterpret_cast<const bf16x8 *>(B + pl * PLANE_B + b_frag[ni]));
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          bf16x8 fa[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            fa[pl] = (ABL == 4) ? __builtin_bit_cast(bf16x8, fake) : *reinterpret_cast<const bf16x8 *>(A + pl * PLANE_A + a_e[mi]);
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
    }
    stamp(0);
    if (ABL == 9) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(1); }
    if (!DB && more && ABL != 2) write_b((step + 1) & 1);
    stamp(2);
    const bool restage = last_tap && next_chunk && stage_next_strip && ABL != 5;
    if (restage) {
      __syncthreads();                                             // every wave is done with this chunk's strip
      write_strip();
    }
    if (DB ? restage : (ABL != 1 || last_tap)) __syncthreads();
    stamp(3);
    ++step;
  };

  // ---- fused 1x1 skip walk with the activations taken straight from global memory (wave tiles of at most 2 x 32 rows x
  // chunks per step): a lane's A fragment of a 32 x 32 x 16 MFMA is 8 consecutive channels of ONE pixel, so two float4 loads
  // and the three-way split give it without LDS.  The strip path restaged the tile per step behind two barriers with the loads
  // of step s + 1 issued only one (single-tap) step ahead: 25-40 us per layer for 2-4 GFLOP.  Here the loads of step s + 2 are
  // issued when step s has consumed its registers (two register sets, the strip staging registers being idle), the weight
  // tile keeps its double buffer and a step has one barrier.
  constexpr bool DIRECT = MI * KW <= 2 && KC >= 2 && ABL == 0;   // (K = 16 steps: 3 waves per SIMD leave no registers for the two sets)
  f32x4 ga0[KW][MI][2], ga1[KW][MI][2];
  const int n_skip = p.in2 ? (CC2 + KC - 1) / KC : 0;
  auto load_a = [&](f32x4 (&g)[KW][MI][2], int s2) __attribute__((always_inline)) {
#pragma unroll
    for (int kw = 0; kw < KW; ++kw)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int c2 = s2 * KC + wk * KW + kw;
        const int m = m0 + wm * (MI * 32) + mi * 32 + l31;
        g[kw][mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; g[kw][mi][1] = g[kw][mi][0];
        if (m < p.M && c2 < CC2) {
          const float *src = (p.in2_b && c2 >= p.cc_a) ? p.in2_b + (size_t)m * p.b_stride + (c2 - p.cc_a) * 16 + half * 8
                                                       : p.in2 + (size_t)m * p.cin2_p + c2 * 16 + half * 8;
          g[kw][mi][0] = *reinterpret_cast<const f32x4 *>(src);
          g[kw][mi][1] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
      }
  };
  auto skip_step = [&](f32x4 (&g)[KW][MI][2], int s2) __attribute__((always_inline)) {
    const bool more = s2 + 1 < n_skip;
    bf16x8 fbc[KW][NI][3];
    if (DB) {
#pragma unroll
      for (int kw = 0; kw < KW; ++kw)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) fbc[kw][ni][pl] = fbn[kw][ni][pl];
      if (more) load_bd(ADV_SKIP);
    } else if (more) load_b(ADV_SKIP);
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
      const int kk = wk * KW + kw;
      const __bf16 *B = Bs + ((step & 1) * KC + kk) * STAGE_B;
      bf16x8 fb[NI][3];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) fb[ni][pl] = DB ? fbc[kw][ni][pl] : *reinterpret_cast<const bf16x8 *>(B + pl * PLANE_B + b_frag[ni]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        bf16x8 fa[3];
        split8(g[kw][mi][0], g[kw][mi][1], fa[0], fa[1], fa[2]);
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
    if (s2 + 2 < n_skip) load_a(g, s2 + 2);
    if (!DB) {
      if (more) write_b((step + 1) & 1);
      __syncthreads();
    }
    ++step;
  };

  // Two workgroups share a CU (and each SIMD) on the big layers, and the SIMD's arbiter favours the older wave: left
  // alone, the older workgroup finishes its loop well before the younger one, which then runs the rest by itself at one
  // wave per SIMD.  Blocks b and b + 256 are the pair that usually shares a CU (round-robin dispatch; a guess that only
  // affects speed): they take the high priority in alternate chunk groups.
  // (Only for grids of at most two workgroups per CU: with three resident ones two would share a priority -- measured
  // 11 % slower on enc1.conv2's 1024 workgroups -- and larger grids refill CUs at arbitrary times anyway.)
  const int prio_grp = ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1;
  const bool use_prio = p.ablate != 7 && gridDim.x * gridDim.y * gridDim.z <= 512;
  for (int ch = 0; ch < n_main; ++ch) {                            // 3x3 walk: nine unrolled taps per chunk group
    if (use_prio) {
      if ((ch ^ prio_grp) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    // (with the K split the fused skip walk starts from a fresh prologue after the mid-kernel reduction below)
    const bool next_chunk = ch + 1 < ((WK > 1 && p.in2) ? n_main : n_chunks);
    if (DIRECT && WK == 1 && p.in2 && ch + 1 == n_main) {          // the skip walk's first two steps load during this chunk's taps
      stage_next_strip = false;
      load_a(ga0, 0);
      if (n_skip > 1) load_a(ga1, 1);
    }
    do_step(std::integral_constant<int, 0>{}, ch, true, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 1>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 2>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 3>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 4>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 5>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 6>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 7>{}, ch, false, false, next_chunk, ADV_TAP);
    do_step(std::integral_constant<int, 8>{}, ch, false, true, next_chunk, ch + 1 < n_main ? ADV_GROUP : ADV_SKIP0);
    if (ch + 1 == n_main && p.in2) {
      if constexpr (WK == 1) {
        conv_midpoint<MI, NI>(p, acc, n0, wn, l31);
      } else {
        // BN + ReLU apply to the COMPLETE 3x3 sum: the WK partial tiles meet in LDS (free after the last step's barrier), wave
        // 0 of each group continues with relu(bn(sum)) and the others with zero, and the skip walk restarts the pipeline
        constexpr int P = BN + 4, COPY = WM * 32 * P;
        float *red = reinterpret_cast<float *>(strip_lds);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          if (mi || DB) __syncthreads();                             // (DB: the last step ended without a barrier; the strip is still being read)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              red[wk * COPY + (wm * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * P + wn * (NI * 32) + ni * 32 + l31] = acc[mi][ni][r];
          __syncthreads();
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float *src = red + (wm * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * P + wn * (NI * 32) + ni * 32 + l31;
              float v = src[0];
#pragma unroll
              for (int c = 1; c < WK; ++c) v += src[c * COPY];
              acc[mi][ni][r] = wk == 0 ? v : 0.f;
            }
        }
        if (wk == 0) conv_midpoint<MI, NI>(p, acc, n0, wn, l31);
        __syncthreads();
        if (DB) load_bd(ADV_SKIP0); else load_b(ADV_SKIP0);
        if (DIRECT) {
          load_a(ga0, 0);
          if (n_skip > 1) load_a(ga1, 1);
        } else {
          load_strip(n_main);
          if (tid < 48 * KC) *reinterpret_cast<u32x4 *>(As + (tid >> 4) * PLANE_A + RZ * 16 + (tid & 15) * 8) = u32x4{0u, 0u, 0u, 0u};
          write_strip();
        }
        if (!DB) write_b(step & 1);
        __syncthreads();
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
  if (DIRECT) {                                                    // fused 1x1 skip walk, activations from global memory
    for (int s2 = 0; s2 < n_skip; s2 += 2) {
      skip_step(ga0, s2);
      if (s2 + 1 < n_skip) skip_step(ga1, s2 + 1);
    }
  } else {
    for (int ch = n_main; ch < n_chunks; ++ch)                     // fused 1x1 skip walk through the strip: centre tap only
      do_step(std::integral_constant<int, 4>{}, ch, true, true, ch + 1 < n_chunks, ADV_SKIP);
  }
  const long long tl1 = p.ablate == 8 ? wall_clock64() : 0;
  conv_epilogue<MI, NI, WM, WK>(p, acc, reinterpret_cast<float *>(strip_lds), m0, n0, wm, wn, half, l31, wk);
  if (ABL == 9) {                          // per-wave phase sums (cycles) + step count into the unused split-K slab
    if (lane == 0) {
      unsigned long long *rec = reinterpret_cast<unsigned long long *>(p.slab) + 8 * (4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + wave);
      rec[0] = ph[0]; rec[1] = ph[1]; rec[2] = ph[2]; rec[3] = ph[3]; rec[4] = (unsigned long long)step;
    }
  }
  if (p.ablate == 8 && p.splits == 1) {   // (start, prologue end, loop end, end) per workgroup into the unused split-K slab
    __syncthreads();
    if (tid == 0) {
      long long *rec = reinterpret_cast<long long *>(p.slab) + 4 * (blockIdx.x + gridDim.x * blockIdx.y);
      rec[0] = tl0; rec[1] = tl_pro; rec[2] = tl1; rec[3] = wall_clock64();
    }
  }
}

// chunks per step of a strip arithmetic code: 3 -> 1, 4 -> 2, 5 (K split across the waves) -> 4 waves x 1 chunk on the
// 64 x 64 tile, 2 waves x 1 chunk on the 128 x 64 and 64 x 128 tiles
int strip_kc(int prec, int bm, int bn) { return prec == 5 ? (bm == 64 && bn == 64 ? 4 : 2) : (prec == 4 ? 2 : 1); }

static size_t strip_lds_bytes(int W, int bm, int bn, int prec) {
  const int kc = strip_kc(prec, bm, bn);
  const int R = bm + 2 * strip_halo(W, bm);
  // the strip's three planes per chunk; the double-buffered weight tile unless the waves take their fragments from global memory (K split)
  const size_t loop = (size_t)kc * ((size_t)3 * (((R + 7) & ~7) + 8) * 16 + (prec == 5 ? 0 : (size_t)2 * 3 * bn * 16)) * sizeof(__bf16);
  // the staged epilogue reuses the same LDS: WM * 32 rows per copy, one copy per K-split wave (128 rows in every K-split layout but 64 x 128)
  const size_t stage = (size_t)(bm == 256 || (prec == 5 && bn == 64) ? 128 : 64) * (bn + 4) * sizeof(float);
  return loop > stage ? loop : stage;
}

bool strip_admissible(int W, int bm, int bn, int prec) {
  if (W + 1 > 64) return false;                                    // 2 (3) strip items per thread cover BM + 2(W+1) <= 256 (384) rows
  if (bm == 256 && (bn != 64 || prec == 5)) return false;          // the 4 x 1 wave layout exists for 64-column tiles only
  if (prec == 5 && (bm > 128 || (bm == 128 && bn == 128))) return false;
  if (strip_kc(prec, bm, bn) == 4 && W + 1 > 32) return false;     // one strip item per thread: BM + 2(W+1) <= 128 rows
  return strip_lds_bytes(W, bm, bn, prec) <= (prec == 3 ? 65536u : 98304u);
}

int launch_conv_strip(const ConvParams &p_in, int bm, int bn, int prec, hipStream_t s) {
  ConvParams p = p_in;
  const int kc = strip_kc(prec, bm, bn);
  if (p.ksize != 3 || p.tap_lo != 0 || p.tap_hi != 9 || p.splits < 1 || prec < 3 || prec > 5) return DT_E_ARG;
  if (prec == 5 && (bm > 128 || (bm == 128 && bn == 128))) return DT_E_ARG;
  if (kc == 4 && p.W + 1 > 32) return DT_E_SHAPE;
  if (!chunks_fit(p.cin_p >> 4, p.ccw, p.splits * kc) || (p.in2 && !chunks_fit(p.cin2_p >> 4, p.ccw2, kc))) return DT_E_ARG;
  if (p.W + 1 > 64) return DT_E_SHAPE;                             // 2 (3) strip items per thread cover BM + 2(W+1) <= 256 (384) rows
  if (bm == 256 && bn != 64) return DT_E_ARG;
  dim3 grid((p.M + bm - 1) / bm, p.n_p / bn, p.splits);
  size_t lds = strip_lds_bytes(p.W, bm, bn, prec);
  p.dup_stage2 = 0;
  if (p.n_dup == 2 && p.pool_out) {   // a second epilogue stage behind the first (and its K-split copies), if it fits this launch's LDS class
    const size_t rows1 = bm == 256 || kc == 4 || (prec == 5 && bm == 128) ? 128 : 64;         // WK * WM * 32
    const size_t rows2 = bm == 256 ? 128 : (bm == 128 ? 64 : 32);                              // WM * 32
    const size_t need = (rows1 + (prec == 5 ? rows2 : rows1)) * (bn + 4) * sizeof(float);
    if (need <= (prec == 3 ? 65536u : 98304u)) {
      p.dup_stage2 = 1;
      if (need > lds) lds = need;
    }
  }
#ifdef DT_TOOLS   // ablation instantiations (wrong results by design) exist only in a tools build (build.py --tools)
  if (p.ablate && p.ablate != 7 && p.ablate != 8 && bm == 128 && bn == 128) {     // timing experiments (tools/ablate.py)
    switch (p.ablate) {
      case 1: conv_strip_bf16x6_kernel<128, 128, 1><<<grid, 256, lds, s>>>(p); break;
      case 2: conv_strip_bf16x6_kernel<128, 128, 2><<<grid, 256, lds, s>>>(p); break;
      case 3: conv_strip_bf16x6_kernel<128, 128, 3><<<grid, 256, lds, s>>>(p); break;
      case 4: conv_strip_bf16x6_kernel<128, 128, 4><<<grid, 256, lds, s>>>(p); break;
      case 5: conv_strip_bf16x6_kernel<128, 128, 5><<<grid, 256, lds, s>>>(p); break;
      case 6: conv_strip_bf16x6_kernel<128, 128, 6><<<grid, 256, lds, s>>>(p); break;
      case 9: conv_strip_bf16x6_kernel<128, 128, 9><<<grid, 256, lds, s>>>(p); break;
      default: return DT_E_ARG;
    }
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
#endif
  if (prec >= 4) {
    static std::once_flag attr_once;   // 128x128 needs 80 KB of dynamic LDS; launches come from several host threads
    static int attr_status = DT_OK;
    std::call_once(attr_once, [] {
      const void *fns[8] = {reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<64, 128, 0, 2, 2>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<64, 64, 0, 4, 4>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<128, 64, 0, 2, 2>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<256, 64, 0, 2>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<128, 128, 0, 2>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<128, 64, 0, 2>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<64, 128, 0, 2>),
                            reinterpret_cast<const void *>(&conv_strip_bf16x6_kernel<64, 64, 0, 2>)};
      for (const void *f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
        if (e != hipSuccess) attr_status = (int)e;
      }
    });
    if (attr_status != DT_OK) return attr_status;
    if (lds > 98304) return DT_E_SHAPE;
    if (prec == 5 && bn == 128) conv_strip_bf16x6_kernel<64, 128, 0, 2, 2><<<grid, 256, lds, s>>>(p);
    else if (prec == 5 && bm == 64) conv_strip_bf16x6_kernel<64, 64, 0, 4, 4><<<grid, 256, lds, s>>>(p);
    else if (prec == 5) conv_strip_bf16x6_kernel<128, 64, 0, 2, 2><<<grid, 256, lds, s>>>(p);
    else if (bm == 256) conv_strip_bf16x6_kernel<256, 64, 0, 2><<<grid, 256, lds, s>>>(p);
    else if (bm == 128 && bn == 128) conv_strip_bf16x6_kernel<128, 128, 0, 2><<<grid, 256, lds, s>>>(p);
    else if (bm == 128) conv_strip_bf16x6_kernel<128, 64, 0, 2><<<grid, 256, lds, s>>>(p);
    else if (bn == 128) conv_strip_bf16x6_kernel<64, 128, 0, 2><<<grid, 256, lds, s>>>(p);
    else conv_strip_bf16x6_kernel<64, 64, 0, 2><<<grid, 256, lds, s>>>(p);
    DT_LAUNCH_CHECK();
    return DT_OK;
  }
  if (lds > 65536) return DT_E_SHAPE;
  if (bm == 256) conv_strip_bf16x6_kernel<256, 64><<<grid, 256, lds, s>>>(p);
  else if (bm == 128 && bn == 128) conv_strip_bf16x6_kernel<128, 128><<<grid, 256, lds, s>>>(p);
  else if (bm == 128) conv_strip_bf16x6_kernel<128, 64><<<grid, 256, lds, s>>>(p);
  else if (bn == 128) conv_strip_bf16x6_kernel<64, 128><<<grid, 256, lds, s>>>(p);
  else conv_strip_bf16x6_kernel<64, 64><<<grid, 256, lds, s>>>(p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

// Software-pipelined strip kernel: the 3x3 convolution of dt_conv_strip.hip (split-bf16 arithmetic, activation
// strip staged once per 16-channel chunk and reused by all nine taps) with the per-tap latency chain taken off
// the matrix pipe's critical path.
//
// Why (tools/tap_phases.py, s_memtime stamps in the plain strip kernel, one workgroup per CU): a tap's 24 MFMAs
// (768 cycles of matrix pipe) took 1308 cycles from the barrier to the last MFMA issue, because the address math,
// the twelve ds_read_b128 of the tap's fragments and their LDS latency all sit in front of the first MFMA and
// nothing else runs on the SIMD meanwhile; LDS writes of the next weight tile (179) and the barrier (285) follow.
// Here, while the MFMAs of tap t run, the wave already reads the fragments of tap t+1 into a second register set:
//   * weights live in a ring of THREE tap tiles and are staged two taps ahead (tile t+2 is written at the end of
//     tap t), so that tile t+1 is visible to every wave while tap t computes;
//   * the strip is double buffered: chunk c+1 is written (split into its three bf16 planes) in the middle of chunk
//     c, so a chunk boundary costs no barrier pair and no restaging stall, only one exposed fragment read;
//   * nine taps = three turns of the ring, so ring slots, register sets and tap shifts are all compile-time
//     constants in the unrolled chunk body.
// One barrier per tap remains (it publishes weight tile t+2).  LDS: 2 strips + 3 weight tiles = 72 KB for a
// 128 x 128 tile at W = 16 (two workgroups per CU); registers: two fragment sets, 2 waves per SIMD.
//
// Split-K over channel chunks (grid.z slabs) and the fused 1x1 skip walk are those of dt_conv_strip.hip; the skip
// walk (single-tap chunks over in2 / w2) is not pipelined, it alternates the two strip buffers instead.
#include <mutex>
#include <type_traits>

#include "dt_conv_epilogue.h"

namespace dt {

extern __shared__ __attribute__((aligned(16))) __bf16 pipe_lds[];

// STAMP: diagnostic build (tools/tap_phases.py): s_memtime sums per wave: [0] MFMA block, [1] weight wait,
// [2] LDS writes, [3] barrier
template <int BM, int BN, bool STAMP = false>
__global__ __launch_bounds__(256, 2) void conv_strip_pipe_bf16x6_kernel(const ConvParams p) {
  constexpr int WN = 2, NT = 256;
  constexpr int MI = BM / 64, NI = BN / 64;
  constexpr int PLANE_B = BN * 16, STAGE_B = 3 * PLANE_B;         // bf16 elements
  constexpr int AP = 2;                                            // strip items (row, k-half) per thread
  constexpr bool ALL_B = BN * 2 == NT;                             // every thread stages 16 B of each weight plane
  const int halo = p.W + 1;
  const int R = BM + 2 * halo;                                     // strip rows
  const int RZ = (R + 7) & ~7;                                     // 16 all-zero rows start here (multiple of 8)
  const int PLANE_A = (RZ + 16) * 16;
  const int STRIP = 3 * PLANE_A;                                   // one strip buffer: [3][RZ+16][16]
  __bf16 *As = pipe_lds;                                           // [2][3][RZ+16][16]
  __bf16 *Bs = pipe_lds + 2 * STRIP;                               // [3 slots][3][BN][16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
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
    s_offb[i] = mm * p.b_stride + hh * 8;                          // two-source block input (decoder concat)
    s_lds[i] = srow * 16 + ((hh ^ ((srow >> 3) & 1)) << 3);
  }
  // ---- weight staging: the three plane tiles of one (tap, chunk) are contiguous [BN][16] bf16 runs
  const bool b_thread = ALL_B || tid < BN * 2;
  const __bf16 *wbase = reinterpret_cast<const __bf16 *>(p.w) + (size_t)n0 * 16 + tid * 8;
  const __bf16 *wbase2 = reinterpret_cast<const __bf16 *>(p.w2) + (size_t)n0 * 16 + tid * 8;
  const size_t w_plane = (size_t)p.n_p * 16;
  const size_t tap_stride = (size_t)CC * 3 * w_plane, chunk_stride = 3 * w_plane;

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

  const int n_main = CC / p.splits;                                // channel chunks of this z slice
  const int cc0 = blockIdx.z * n_main;
  const int n_skip = p.in2 ? (p.cin2_p >> 4) : 0;

  f32x4 sa0[AP], sa1[AP];
  u32x4 rb[3];
  // chunk `ch` of the walk: main chunks first, then the skip chunks (core rows only)
  auto load_strip = [&](int ch) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      sa0[i] = f32x4{0.f, 0.f, 0.f, 0.f}; sa1[i] = sa0[i];
      if (ch < n_main) {
        if (s_ok[i]) {
          const int cc = cc0 + ch;
          const float *src = (p.in_b && cc >= p.cc_a) ? p.in_b + s_offb[i] + (cc - p.cc_a) * 16 : p.in + s_off[i] + cc * 16;
          sa0[i] = *reinterpret_cast<const f32x4 *>(src);
          sa1[i] = *reinterpret_cast<const f32x4 *>(src + 4);
        }
      } else if (s_core[i]) {
        const int c2 = ch - n_main;
        const float *src = (p.in2_b && c2 >= p.cc_a) ? p.in2_b + s_offb[i] + (c2 - p.cc_a) * 16 : p.in2 + s_off2[i] + c2 * 16;
        sa0[i] = *reinterpret_cast<const f32x4 *>(src);
        sa1[i] = *reinterpret_cast<const f32x4 *>(src + 4);
      }
    }
  };
  auto write_strip = [&](__bf16 *A) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AP; ++i)
      if (s_in[i]) {
        bf16x8 p1, p2, p3;
        split8(sa0[i], sa1[i], p1, p2, p3);
        *reinterpret_cast<bf16x8 *>(A + s_lds[i]) = p1;
        *reinterpret_cast<bf16x8 *>(A + PLANE_A + s_lds[i]) = p2;
        *reinterpret_cast<bf16x8 *>(A + 2 * PLANE_A + s_lds[i]) = p3;
      }
  };
  auto load_b = [&](const __bf16 *wt) __attribute__((always_inline)) {
    if (b_thread) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) rb[pl] = *reinterpret_cast<const u32x4 *>(wt + pl * w_plane);
    }
  };
  auto write_b = [&](int slot) __attribute__((always_inline)) {
    if (b_thread) {
      __bf16 *B = Bs + slot * STAGE_B;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(B + pl * PLANE_B + tid * 8) = rb[pl];
    }
  };

  unsigned long long ph[4] = {0, 0, 0, 0}, ts = 0;
  auto stamp = [&](int seg) __attribute__((always_inline)) {
    if (!STAMP) return;
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (seg >= 0) ph[seg] += t - ts;
    ts = t;
  };

  const long long tl0 = p.ablate == 8 ? wall_clock64() : 0;      // timeline diagnostic (tools/block_timeline.py)
  // ---- prologue: strip of chunk 0 (buffer 0), weight tiles of taps 0 and 1 (slots 0, 1), the zero rows of both buffers
  const __bf16 *wchunk = wbase + (size_t)cc0 * chunk_stride;      // tap 0 of the current main chunk
  load_strip(0);
  if (n_main > 0) {
    load_b(wchunk);
    write_b(0);
    load_b(wchunk + tap_stride);
    write_b(1);
  } else {                                                         // (not reached by the launcher: every slice has chunks)
    load_b(wbase2);
    write_b(0);
  }
  if (tid < 192) {
    const int buf = tid / 96, r = tid - buf * 96;
    *reinterpret_cast<u32x4 *>(As + buf * STRIP + (r >> 5) * PLANE_A + RZ * 16 + (r & 31) * 8) = u32x4{0u, 0u, 0u, 0u};
  }
  write_strip(As);
  __syncthreads();

  bf16x8 fa[2][MI][3], fb[2][NI][3];                               // two fragment register sets
  // fragment reads of tap TT from strip buffer A and ring slot SLOT into register set SET
  auto read_frags = [&](auto SET, auto TT, const __bf16 *A, const __bf16 *B) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value, tt = decltype(TT)::value;
    int wv = p.W;
    asm volatile("" : "+s"(wv));             // do not keep nine precomputed shifts in SGPRs
    const int shift = (tt / 3 - 1) * wv + (tt % 3 - 1);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      // an out-of-picture tap reads zero row RZ + (srow & 15) at the same physical half: the bank a lane hits is
      // the one its in-picture read would hit, so any mix of the two stays conflict-free
      int row0 = a_row[mi];
      asm volatile("" : "+v"(row0));         // keeps the nine taps' addresses from being hoisted out of the chunk loop
      const int srow = row0 + shift;
      const int lrow = ((a_mask[mi] >> tt) & 1u) ? srow : RZ + (srow & 15);
      const int ae = lrow * 16 + ((half ^ ((srow >> 3) & 1)) << 3);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[set][mi][pl] = *reinterpret_cast<const bf16x8 *>(A + pl * PLANE_A + ae);
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fb[set][ni][pl] = *reinterpret_cast<const bf16x8 *>(B + pl * PLANE_B + b_frag[ni]);
  };
  auto mfma_block = [&](auto SET) __attribute__((always_inline)) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        // smallest terms first so their sum is formed before it meets the large partial sums
        f32x16 c = acc[mi][ni];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][2], fb[set][ni][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][1], fb[set][ni][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][0], fb[set][ni][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][1], fb[set][ni][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][0], fb[set][ni][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mi][0], fb[set][ni][0], c, 0, 0, 0);
        acc[mi][ni] = c;
      }
  };

  const bool has_skip = n_skip > 0;
  const long long tl1 = p.ablate == 8 ? wall_clock64() : 0;
  stamp(-1);
  // One main step = tap TT of chunk ch.  Entering it, the fragments of this tap are in register set TT & 1, weight
  // tile TT+1 is visible in ring slot (TT+1) % 3, and tile TT+2 is loaded and published during the step.
  auto do_tap = [&](auto TT, int ch, const __bf16 *Acur, __bf16 *Anext) __attribute__((always_inline)) {
    constexpr int tt = decltype(TT)::value;
    constexpr int cur = tt & 1, nxt = cur ^ 1;
    // hipcc's scheduler otherwise sinks the next tap's fragment reads down to their first use (one tap later), which
    // puts their latency back in front of that tap's MFMAs: nothing moves across a tap boundary
    __builtin_amdgcn_sched_barrier(0);
    const bool next_chunk = ch + 1 < n_main || has_skip;
    if (tt == 0 && next_chunk) load_strip(ch + 1);                 // lands while this chunk's taps run
    // weight tile two steps ahead: taps 2..8 of this chunk, then taps 0, 1 of the next main chunk or skip chunks 0, 1
    bool has_w = true;
    const __bf16 *wt;
    if (tt <= 6) wt = wchunk + (tt + 2) * tap_stride;
    else if (ch + 1 < n_main) wt = wchunk + chunk_stride + (tt - 7) * tap_stride;
    else { wt = wbase2 + (tt - 7) * chunk_stride; has_w = (tt - 7) < n_skip; }
    if (has_w) load_b(wt);
    if (tt < 8) read_frags(std::integral_constant<int, nxt>{}, std::integral_constant<int, (tt + 1) % 9>{}, Acur,
                           Bs + ((tt + 1) % 3) * STAGE_B);
    mfma_block(std::integral_constant<int, cur>{});
    if (tt < 8) {                              // one fragment read of the next tap behind each of the first MFMAs
#pragma unroll
      for (int i = 0; i < 3 * (MI + NI); ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 6 * MI * NI - 3 * (MI + NI), 0);
    }
    stamp(0);
    if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(1); }
    if (tt == 5 && next_chunk) write_strip(Anext);                 // its loads were issued five taps ago
    if (has_w) write_b((tt + 2) % 3);
    stamp(2);
    __syncthreads();
    stamp(3);
  };

  // Two workgroups share a CU (and each SIMD) on the big layers, and the SIMD's arbiter favours the older wave: left
  // alone, the older workgroup finishes its loop 25 % before the younger one, which then runs the rest by itself at
  // one wave per SIMD.  Blocks b and b + 256 are the pair that usually shares a CU (round-robin dispatch; a guess that
  // only affects speed): they take the high priority in alternate chunks.
  const int prio_grp = p.ablate == 7 ? 0 : ((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) >> 8) & 1;
  for (int ch = 0; ch < n_main; ++ch) {
    if (p.ablate != 7) {
      if ((ch ^ prio_grp) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    const __bf16 *Acur = As + (ch & 1) * STRIP;
    __bf16 *Anext = As + ((ch + 1) & 1) * STRIP;
    read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, Acur, Bs);   // the one exposed read per chunk
    do_tap(std::integral_constant<int, 0>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 1>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 2>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 3>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 4>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 5>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 6>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 7>{}, ch, Acur, Anext);
    do_tap(std::integral_constant<int, 8>{}, ch, Acur, Anext);
    wchunk += chunk_stride;
  }
  __builtin_amdgcn_s_setprio(0);
  if (has_skip) {
    // fused 1x1 skip walk: centre tap only, one 16-channel chunk of in2 per step; skip chunk k sits in strip buffer
    // (n_main + k) & 1 and ring slot k % 3 (chunks 0, 1 were published by the last two main taps)
    conv_midpoint<MI, NI>(p, acc, n0, wn, l31);
    int slot = 0;
    for (int k = 0; k < n_skip; ++k) {
      const int buf = (n_main + k) & 1;
      if (k + 1 < n_skip) load_strip(n_main + k + 1);
      const bool has_w = k + 2 < n_skip;
      if (has_w) load_b(wbase2 + (size_t)(k + 2) * chunk_stride);
      read_frags(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, As + buf * STRIP, Bs + slot * STAGE_B);
      mfma_block(std::integral_constant<int, 0>{});
      if (k + 1 < n_skip) write_strip(As + (buf ^ 1) * STRIP);
      const int wslot = slot == 0 ? 2 : slot - 1;                  // (slot + 2) % 3
      if (has_w) write_b(wslot);
      __syncthreads();
      slot = slot == 2 ? 0 : slot + 1;
    }
  }
  const long long tl2 = p.ablate == 8 ? wall_clock64() : 0;
  conv_epilogue<MI, NI>(p, acc, reinterpret_cast<float *>(pipe_lds), m0, n0, wm, wn, half, l31);
  if (p.ablate == 8 && p.splits == 1) {   // (start, prologue end, loop end, end) per workgroup into the unused split-K slab
    __syncthreads();
    if (tid == 0) {
      long long *rec = reinterpret_cast<long long *>(p.slab) + 4 * (blockIdx.x + gridDim.x * blockIdx.y);
      rec[0] = tl0; rec[1] = tl1; rec[2] = tl2; rec[3] = wall_clock64();
    }
  }
  if (STAMP) {                             // per-wave phase sums (cycles) + step count into the unused split-K slab
    if (lane == 0) {
      unsigned long long *rec = reinterpret_cast<unsigned long long *>(p.slab) + 8 * (4 * (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) + wave);
      rec[0] = ph[0]; rec[1] = ph[1]; rec[2] = ph[2]; rec[3] = ph[3]; rec[4] = (unsigned long long)(9 * n_main);
    }
  }
}

static size_t strip_pipe_lds_bytes(int W, int bm, int bn) {
  const int R = bm + 2 * (W + 1);
  return ((size_t)2 * 3 * (((R + 7) & ~7) + 16) * 16 + (size_t)3 * 3 * bn * 16) * sizeof(__bf16);
}

bool strip_pipe_admissible(int W, int bm, int bn) {
  if (W + 1 > 64) return false;                                    // 2 strip items per thread cover BM + 2(W+1) <= 256 rows
  return strip_pipe_lds_bytes(W, bm, bn) <= 160 * 1024;
}

int launch_conv_strip_pipe(const ConvParams &p, int bm, int bn, hipStream_t s) {
  if (p.ksize != 3 || p.tap_lo != 0 || p.tap_hi != 9 || p.splits < 1) return DT_E_ARG;
  if ((p.cin_p >> 4) % p.splits || (p.cin_p >> 4) / p.splits < 1) return DT_E_ARG;
  if (!strip_pipe_admissible(p.W, bm, bn)) return DT_E_SHAPE;
  static std::once_flag attr_once;       // more than 64 KB of dynamic LDS; launches come from several host threads
  static int attr_status = DT_OK;
  std::call_once(attr_once, [] {
    const void *fns[5] = {reinterpret_cast<const void *>(&conv_strip_pipe_bf16x6_kernel<128, 128>),
                          reinterpret_cast<const void *>(&conv_strip_pipe_bf16x6_kernel<128, 64>),
                          reinterpret_cast<const void *>(&conv_strip_pipe_bf16x6_kernel<64, 128>),
                          reinterpret_cast<const void *>(&conv_strip_pipe_bf16x6_kernel<64, 64>),
                          reinterpret_cast<const void *>(&conv_strip_pipe_bf16x6_kernel<128, 128, true>)};
    for (const void *f : fns) {
      const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) attr_status = (int)e;
    }
  });
  if (attr_status != DT_OK) return attr_status;
  dim3 grid((p.M + bm - 1) / bm, p.n_p / bn, p.splits);
  const size_t lds = strip_pipe_lds_bytes(p.W, bm, bn);
  if (p.ablate == 9 && bm == 128 && bn == 128) conv_strip_pipe_bf16x6_kernel<128, 128, true><<<grid, 256, lds, s>>>(p);
  else if (bm == 128 && bn == 128) conv_strip_pipe_bf16x6_kernel<128, 128><<<grid, 256, lds, s>>>(p);
  else if (bm == 128) conv_strip_pipe_bf16x6_kernel<128, 64><<<grid, 256, lds, s>>>(p);
  else if (bn == 128) conv_strip_pipe_bf16x6_kernel<64, 128><<<grid, 256, lds, s>>>(p);
  else conv_strip_pipe_bf16x6_kernel<64, 64><<<grid, 256, lds, s>>>(p);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

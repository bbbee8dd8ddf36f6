// Whole-forward (and whole-loop) kernel for the SMALL U-Nets of the size sweep (reference models.py:95-224 with
// size_factor <= ~0.25: dims [16,32,32,32] .. [32,64,64,64], analysis default sweep analyze_trajectory_metrics.py:38).
//
// The layered path runs such a model as ~25 launches per forward; at 16x16 a whole forward of 448 rows is 3-6 GFLOP, so
// every launch is a few microseconds of fixed cost around a few hundred nanoseconds of work (240-315 us per forward,
// 10-24 TF/s: profiles/r03_config2_breakdown.txt).  Here ONE workgroup carries G image-passes ("rows") through the
// entire network with every activation in LDS, and -- because a trajectory's images never interact -- through ALL
// timesteps of the reverse loop (forward, CFG mix, DDPM update, trajectory store): one launch per sampler call instead
// of ~27 per timestep.  The two CFG passes of an image are two rows of the same workgroup, which is what lets the update
// run in place.
//
// Arithmetic: exact fp32 on the matrix cores, v_mfma_f32_16x16x4_f32 (a k-ordered fmaf chain, MI355X_MICROARCH.md).  The
// operand roles are A = weights (row = output channel), B = activations (column = pixel): a lane then owns four
// consecutive output channels of one pixel, i.e. one 16-byte NHWC store into LDS.  One 16-channel chunk of one tap is four
// MFMAs per (16 channels x 16 pixels) tile: lane (i = l & 15, s = l >> 4) loads channels 4s..4s+3 of its row / pixel with
// ONE 16-byte read and feeds element j to MFMA j, which therefore sums channels {4s' + j}: the same set on both operands.
//   activations: LDS, [pixel][channel] fp32, pixel = (row g, y, x) flattened; a tile is 16 consecutive pixels
//   weights:     global memory (L2-resident, < 3 MB per model), packed [tap][chunk][channel tile][lane][4] so that a
//                wave's fragment is one coalesced 1 KB load; no LDS staging, no barrier inside a layer
// Out-of-picture taps read 16 zero bytes in LDS instead (per-lane 9-bit validity mask).
// Epilogues run in registers: folded BN + ReLU, time-bias row, identity residual / the block's 1x1 skip conv continued
// on the same accumulators (after relu(bn(.)) -- models.py:59-83), enc1's 3-channel skip, and the encoder's 2x2 max pool
// with lane shuffles (a tile is a 2 x 8 / 4 x 4 patch, or a row pair of tiles at W = 16).
//
// Eight waves per workgroup, two per SIMD: a wave issues in order, so its address arithmetic and epilogues only overlap
// MFMAs of the OTHER wave on its SIMD (one wave per SIMD measured 3.3-5x the MFMA time per layer).  The low-resolution
// levels (4x4 and below: half of the layers, a tenth of the arithmetic) have fewer (channel tile, pixel tile) pairs than
// waves: there the taps of a pair are split across waves (FusedOp::ks) and the partial tiles meet in LDS.
//
// The layer sequence is a small table (FusedOp) built on the host; LDS buffer placement is in fused_lds() below.
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "dt_conv_epilogue.h"
#include "dt_update_math.h"
#include "dt_fused.h"

namespace dt {

namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / 64;
constexpr int kXSlot = 768;                       // floats per image slot in LDS (C <= 3 channels of 16 x 16)

// every activation lives here; the device functions index this array directly (integer offsets, never pointers that were
// selected or stored: those decay to the generic address space and become flat loads, which also drain the global prefetches)
extern __shared__ __attribute__((aligned(16))) float sm[];

__device__ __forceinline__ f32x4 ldf4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
// The layer functions below are inlined into one loop over (timestep, layer); whatever they derive from the lane index alone is
// loop-invariant to the compiler, which hoists it out of BOTH loops and keeps it in registers for the whole kernel (measured:
// +33 / +44 / +44 VGPRs per inlined layer function, 216 in all).  Passing the index through an opaque asm pins such values to
// the layer that uses them.
__device__ __forceinline__ int pinned(int v) { asm volatile("" : "+v"(v)); return v; }
// 16-byte LDS accesses at a float offset that is a multiple of 4 (indexing by quads tells the compiler so: ds_read_b128 / ds_write_b128)
__device__ __forceinline__ f32x4 lds4(int off) { return reinterpret_cast<const f32x4 *>(sm)[off >> 2]; }
__device__ __forceinline__ void sts4(int off, f32x4 v) { reinterpret_cast<f32x4 *>(sm)[off >> 2] = v; }

// K walk of one unit (one 16-channel output tile x PB pixel tiles) over steps [i0, i1) of the layer's (tap, chunk) sequence:
// taps from tap_lo (0: a 3x3 neighbourhood, 4: the centre tap alone -- 1x1 convolutions and the 1x1-pixel level), KC
// 16-channel chunks per tap, chunks >= cca taken from the second source (decoder concat read in place; both sources have
// the pixel stride cs).  `w` points at the packed weights of step 0: one linear stream of (tap, chunk) blocks.
// The weight fragment of step i + 2 (global memory, L2) and the activation fragments of step i + 1 (LDS) are requested
// before the MFMAs of step i.  q[pb] = pixel * cs + 4 s of the lane's pixel; per step the lane adds a scalar
// (source + tap shift + chunk) and selects the zero slot for an out-of-picture tap: two VALU instructions per fragment.
template <int PB>
__device__ __forceinline__ void fused_walk(f32x4 (&acc)[PB], const float *__restrict__ w, int KC, int OT, int ot, int inA, int inB,
                                           int cs, int cca, int tap_lo, int i0, int i1, int W, const int (&q)[PB],
                                           const unsigned (&mask)[PB], int zero, int lane) {
  if (i0 >= i1) return;
  const float *wp = w + ((size_t)ot * 64 + lane) * 4 + (size_t)i0 * OT * 256;      // step i0; advances one block per load
  const int wstep = OT * 256;
  auto load_w = [&]() __attribute__((always_inline)) {
    const f32x4 v = ldf4(wp);
    wp += wstep;
    return v;
  };
  // position of the activation reads: tap = 3 (dy + 1) + (dx + 1), chunk kc; sh = dy W + dx
  int tap = tap_lo + i0 / KC, kc = i0 - (i0 / KC) * KC;
  int dx = tap - (tap / 3) * 3 - 1, sh = (tap / 3 - 1) * W + dx;
  auto read_a = [&](f32x4 (&af)[PB]) __attribute__((always_inline)) {
    const int sbase = (kc < cca ? inA + kc * 16 : inB + (kc - cca) * 16) + sh * cs;     // scalar
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) af[pb] = lds4(((mask[pb] >> tap) & 1u) ? q[pb] + sbase : zero);
    if (++kc == KC) {
      kc = 0; ++tap; ++sh;
      if (++dx > 1) { dx = -1; sh += W - 3; }
    }
  };
  // a second accumulator set for single-tile units: v_mfma_f32_16x16x4_f32 issues every 32 cycles but a dependent one waits 40
  f32x4 acc2[1];
  if (PB == 1) acc2[0] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mfmas = [&](const f32x4 &wc, const f32x4 (&ac)[PB]) __attribute__((always_inline)) {
    if (PB == 1) {
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[0], ac[0][0], acc[0], 0, 0, 0);
      acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[1], ac[0][1], acc2[0], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[2], ac[0][2], acc[0], 0, 0, 0);
      acc2[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[3], ac[0][3], acc2[0], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[j], ac[pb][j], acc[pb], 0, 0, 0);
    }
  };
  // Software pipeline with STATIC buffers, unrolled by two: the weights and activations of step i + 1 are requested before the
  // MFMAs of step i (the other wave of the SIMD covers what latency is left).  (A rotating pair of variables compiles to
  // register copies of the load that was just issued, i.e. a full wait per step; deeper prefetch cost 40+ more VGPRs.)
  f32x4 w0 = load_w(), w1;
  f32x4 a0[PB], a1[PB];
  read_a(a0);
  int left = i1 - i0;                               // steps still to multiply
  while (true) {
    if (left > 1) { w1 = load_w(); read_a(a1); }
    mfmas(w0, a0);
    if (--left == 0) break;
    if (left > 1) { w0 = load_w(); read_a(a0); }
    mfmas(w1, a1);
    if (--left == 0) break;
  }
  if (PB == 1) acc[0] += acc2[0];
}

// the lane's pixels of a unit: p = pixel index, q = its LDS offset term, mask = the 3x3 taps that fall inside the picture
template <int PB>
__device__ __forceinline__ void fused_pixels(int grp, int wsh, int npx, int cs, int lane, int (&p)[PB], int (&q)[PB], unsigned (&mask)[PB]) {
  const int W = 1 << wsh, HWm = (1 << (2 * wsh)) - 1;
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    p[pb] = (grp * PB + pb) * 16 + (lane & 15);
    q[pb] = p[pb] * cs + (lane >> 4) * 4;
    const int rem = p[pb] & HWm, y = rem >> wsh, x = rem & (W - 1);
    const unsigned my = (y > 0 ? 0x007u : 0u) | 0x038u | (y < W - 1 ? 0x1c0u : 0u);     // rows of taps whose y + dy is in range
    const unsigned mx = (x > 0 ? 0x049u : 0u) | 0x092u | (x < W - 1 ? 0x124u : 0u);     // columns
    mask[pb] = p[pb] < npx ? (my & mx) : 0u;
  }
}

// One convolution layer over the workgroup's G rows (see FusedOp).  Units (channel tile, group of PB pixel tiles, K slice)
// go round the eight waves; without a K split nothing inside needs a barrier (inputs and outputs are different LDS buffers).
template <int PB>
__device__ __forceinline__ void fused_conv(const FusedOp &op, const FusedLds &L, const float *__restrict__ fb,
                                           int G, int xshift, int C, int wave, int lane) {
  const int wsh = op.wsh, W = 1 << wsh, hwsh = 2 * wsh, HWm = (1 << hwsh) - 1;
  const int npx = G << hwsh;
  const int ntile = (npx + 15) >> 4, ngrp = (ntile + PB - 1) / PB;
  const int KS = op.ks, nunit = ngrp * op.ot * KS;
  const int nsteps = (op.tap_hi - op.tap_lo) * op.kc;
  // (with a K split the table guarantees nunit <= kWaves: one pass, so the barrier below is reached once by every wave)
  for (int u = wave; u < nunit || (KS > 1 && u == wave); u += kWaves) {
    const bool active = u < nunit;
    lane = pinned(lane);
    const int s = lane >> 4, px = lane & 15;
    const int ks = u % KS, uu = u / KS;
    const int ot = uu % op.ot, grp = uu / op.ot;
    int p[PB], q[PB];
    unsigned mask[PB];
    fused_pixels<PB>(grp, wsh, active ? npx : 0, op.cs_in, lane, p, q, mask);
    const int oc0 = ot * 16 + 4 * s;
    const bool fin = active && ks == 0;                 // this wave runs the unit's epilogue
    f32x4 acc[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
      const int i0 = (nsteps * ks) / KS, i1 = (nsteps * (ks + 1)) / KS;
      fused_walk<PB>(acc, fb + op.w + op.tap_lo * op.kc * op.ot * 256, op.kc, op.ot, ot, op.inA, op.inB, op.cs_in, op.cca, op.tap_lo,
                     i0, i1, W, q, mask, L.zero, lane);
    }
    if (KS > 1) {
      // the K slices of a unit meet in LDS: slices 1.. park their partial tiles, slice 0 adds them in slice order
      if (active && ks > 0) {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) sts4(L.scr + ((u * PB + pb) * 64 + lane) * 4, acc[pb]);
      }
      __syncthreads();
      if (fin) {
        for (int k2 = 1; k2 < KS; ++k2)
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) acc[pb] += lds4(L.scr + (((u + k2) * PB + pb) * 64 + lane) * 4);
      }
    }
    if (!fin) continue;
    // ---- epilogue: this lane holds output channels oc0 .. oc0 + 3 of pixel p[pb]; the folded BN vectors, biases and the
    // step's time-bias rows sit in LDS (L.par, L.tb)
    const f32x4 sc = lds4(L.par + op.scale + oc0), shv = lds4(L.par + op.shift + oc0);
    f32x4 v[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      v[pb] = acc[pb] * sc + shv;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[pb][e] = fmaxf(v[pb][e], 0.f);
    }
    if (op.emode == FUSED_E_SKIPCONV) {              // the block's 1x1 skip conv continues on relu(bn(conv2)): models.py:61,83
      int q2[PB];
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) q2[pb] = p[pb] * op.rcs + 4 * s;
      fused_walk<PB>(v, fb + op.wr, op.kcr, op.ot, ot, op.rA, op.rB, op.rcs, op.rcca, 4, 0, op.kcr, W, q2, mask, L.zero, lane);
      const f32x4 b2 = lds4(L.par + op.br + oc0);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) v[pb] += b2;
    } else if (op.emode == FUSED_E_TIMEBIAS) {       // h = relu(bn1(conv1 x)) + relu(time_mlp(t_emb)): models.py:66-77
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) v[pb] += lds4(L.tb + (p[pb] < npx ? p[pb] >> hwsh : 0) * L.tb_cols + op.tb_off + oc0);
    } else if (op.emode == FUSED_E_IDENTITY) {       // + x (in_ch == out_ch)
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) v[pb] += lds4(op.rA + (p[pb] < npx ? p[pb] : 0) * op.rcs + oc0);
    } else if (op.emode == FUSED_E_IMAGE) {          // enc1: + residual_conv(x) of the <= 3-channel image, from the image itself
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) {
        const int pp = p[pb] < npx ? p[pb] : 0;
        const int xr = L.x + ((pp >> 8) >> xshift) * kXSlot + (pp & 255);
        const float x0 = sm[xr], x1 = C > 1 ? sm[xr + 256] : 0.f, x2 = C > 2 ? sm[xr + 512] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const f32x4 w3 = lds4(L.par + op.wr + 4 * (oc0 + e));
          float rs = w3[3];
          rs = fmaf(x0, w3[0], rs);
          if (C > 1) rs = fmaf(x1, w3[1], rs);
          if (C > 2) rs = fmaf(x2, w3[2], rs);
          v[pb][e] += rs;
        }
      }
    }
    if (op.out >= 0) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
        if (p[pb] < npx) sts4(op.out + p[pb] * op.cs + oc0, v[pb]);
    }
    if (op.pool >= 0) {                              // MaxPool2d(2) of the block output (models.py:134,190-199)
      const int Wo = W >> 1;
      if (PB >= 2 && wsh == 4) {                     // W = 16: tiles 2r, 2r + 1 are picture rows y, y + 1
#pragma unroll
        for (int pb = 0; pb + 1 < PB; pb += 2) {
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = fmaxf(v[pb][e], v[pb + 1][e]);
            m[e] = fmaxf(a, __shfl_xor(a, 1, 64));
          }
          if (p[pb] < npx && !(px & 1)) {
            const int g = p[pb] >> 8, y = (p[pb] & 255) >> 4;
            sts4(op.pool + (g * 64 + (y >> 1) * 8 + (px >> 1)) * op.cs + oc0, m);
          }
        }
      } else {                                       // a tile holds whole 2x2 windows: partners px ^ W (y) and px ^ 1 (x)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          f32x4 m;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = fmaxf(v[pb][e], __shfl_xor(v[pb][e], W, 64));
            m[e] = fmaxf(a, __shfl_xor(a, 1, 64));
          }
          if (p[pb] < npx && !(px & (W | 1))) {
            const int g = p[pb] >> hwsh, rem = p[pb] & HWm, y = rem >> wsh, x = rem & (W - 1);
            sts4(op.pool + ((g << (hwsh - 2)) + (y >> 1) * Wo + (x >> 1)) * op.cs + oc0, m);
          }
        }
      }
    }
  }
}

// enc1.conv1 (models.py:61-63 on the C <= 3 channel image) + folded BN + ReLU + time bias, also on the matrix cores: K = 9C <= 27
// is padded to two 16-deep chunks and the activation fragment is gathered from the NCHW image in LDS -- lane (pixel, s) needs
// k = 16 chunk + 4 s + j, k = 9 c + tap: eight (channel, dy, dx) triples per lane, decoded once per layer.  (As a direct VALU
// convolution, one output channel per thread, this layer took 18.6 k cycles of a 276 k cycle step for 3 % of the arithmetic.)
__device__ __forceinline__ void fused_first(const FusedOp &op, const FusedLds &L, const float *__restrict__ fb,
                                            int G, int xshift, int C, int wave, int lane) {
  constexpr int PB = 2;
  lane = pinned(lane);
  const int s = lane >> 4, px = lane & 15;
  int koff[8], kdy[8], kdx[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int k = (i >> 2) * 16 + 4 * s + (i & 3);
    const int c = k / 9, tap = k - 9 * c;
    kdy[i] = k < 9 * C ? tap / 3 - 1 : 64;          // 64: never in range (padding of K)
    kdx[i] = tap - (tap / 3) * 3 - 1;
    koff[i] = c * 256 + kdy[i] * 16 + kdx[i];
  }
  const int ngrp = G * 16 / PB, nunit = ngrp * op.ot;
  const int wstep = op.ot * 256;
  for (int u = wave; u < nunit; u += kWaves) {
    const int ot = u % op.ot, grp = u / op.ot;
    const int oc0 = ot * 16 + 4 * s;
    const float *wp = fb + op.w + ((size_t)ot * 64 + lane) * 4;
    const f32x4 w0 = ldf4(wp), w1 = ldf4(wp + wstep);
    f32x4 acc[PB];
    int p[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) p[pb] = (grp * PB + pb) * 16 + px;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int g = p[pb] >> 8, y = (p[pb] & 255) >> 4, x = p[pb] & 15;
      const int xs = L.x + (g >> xshift) * kXSlot + y * 16 + x;
      float a[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool ok = (unsigned)(y + kdy[i]) < 16u && (unsigned)(x + kdx[i]) < 16u;
        a[i] = sm[ok ? xs + koff[i] : L.zero];
      }
      acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[j], a[j], acc[pb], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[j], a[4 + j], acc[pb], 0, 0, 0);
    }
    const f32x4 sc = lds4(L.par + op.scale + oc0), shv = lds4(L.par + op.shift + oc0);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      f32x4 v = acc[pb] * sc + shv;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      sts4(op.out + p[pb] * op.cs + oc0, v + lds4(L.tb + (p[pb] >> 8) * L.tb_cols + op.tb_off + oc0));
    }
  }
}

// bilinear x2, align_corners=True (models.py:135,205-217) of an LDS tensor [G][w][w][c] -> [G][2w][2w][c]
__device__ __forceinline__ void fused_upsample(const FusedOp &op, int G, int tid) {
  const int wsh = op.wsh, w = 1 << wsh, cq = op.kc * 4;          // kc 16-channel chunks; op.cs is the (padded) pixel stride
  const int W2 = 2 * w, n = G * W2 * W2 * cq;
  for (int i = pinned(tid); i < n; i += kThreads) {
    const int quad = i % cq, P = i / cq;
    const int g = P >> (2 * wsh + 2), rem = P & (W2 * W2 - 1), Y = rem >> (wsh + 1), X = rem & (W2 - 1);
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    bilinear_src(Y, w, W2, y0, y1, wy0, wy1);
    bilinear_src(X, w, W2, x0, x1, wx0, wx1);
    const int base = op.inA + (g << (2 * wsh)) * op.cs + quad * 4;
    const f32x4 v00 = lds4(base + (y0 * w + x0) * op.cs), v01 = lds4(base + (y0 * w + x1) * op.cs);
    const f32x4 v10 = lds4(base + (y1 * w + x0) * op.cs), v11 = lds4(base + (y1 * w + x1) * op.cs);
    sts4(op.out + P * op.cs + quad * 4, wy0 * (wx0 * v00 + wx1 * v01) + wy1 * (wx0 * v10 + wx1 * v11));
  }
}

// final 1x1 head at the LOW resolution (models.py:221-224; linear maps commute with a bilinear interpolation whose
// weights sum to one, as in head_kernel): low[g][8][8][4]; its weights [4][c0p] (+ bias [4]) sit in LDS
__device__ __forceinline__ void fused_head(const FusedOp &op, const FusedLds &L, int c0p, int G, int tid) {
  for (int i = pinned(tid); i < G * 64 * 4; i += kThreads) {
    const int c = i & 3, P = i >> 2;
    const int v = op.inA + P * op.cs, hw = L.head + c * c0p;
    float acc = 0.f;
    for (int k = 0; k < c0p; k += 4) {
      const f32x4 a = lds4(v + k), b = lds4(hw + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = fmaf(b[e], a[e], acc);
    }
    sm[L.low + i] = acc + sm[L.head + 4 * c0p + c];
  }
}

// eps_at (dt_update_math.h) on a low-resolution head output held in LDS at float offset `low`: the same arithmetic
__device__ __forceinline__ float eps_at_lds(int low, int c, int y, int x) {
  int y0, y1, xa, xb;
  float wy0, wy1, wx0, wx1;
  bilinear_src(y, 8, 16, y0, y1, wy0, wy1);
  bilinear_src(x, 8, 16, xa, xb, wx0, wx1);
  const int base = low + c;
  return bilinear_blend(sm[base + (y0 * 8 + xa) * 4], sm[base + (y0 * 8 + xb) * 4], sm[base + (y1 * 8 + xa) * 4], sm[base + (y1 * 8 + xb) * 4],
                        wy0, wy1, wx0, wx1);
}

}  // namespace

// mode FUSED_FORWARD: eps[r] = U-Net(x[image of row r]) for the batch rows blockIdx.x * G .. + G - 1
// mode FUSED_LOOP:    the workgroup's images are advanced n_steps timesteps in place (trajectory slots in global memory)
__global__ __launch_bounds__(kThreads, 4) void unet_fused_kernel(const FusedArgs a, const FusedOp *__restrict__ ops) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.G, C = a.C, E = C * 256;
  const FusedLds L = a.lds;
  const int Bt = a.B * a.n_pass - a.B_single * (a.n_pass - 1);

  // ---- which batch rows / images this workgroup carries
  int n_rows, xshift = 0, img0 = 0, n_img = 0, row0 = 0;
  if (a.mode == FUSED_FORWARD) {
    row0 = blockIdx.x * G;
    n_rows = Bt - row0 < G ? Bt - row0 : G;
    n_img = n_rows;
  } else {
    const int n_single = a.n_pass == 1 ? a.B : a.B_single;                 // images advanced with one pass
    const int wg_single = (n_single + G - 1) / G;
    if ((int)blockIdx.x < wg_single) {
      img0 = blockIdx.x * G;
      n_img = n_single - img0 < G ? n_single - img0 : G;
      n_rows = n_img;
    } else {
      xshift = 1;
      img0 = n_single + ((int)blockIdx.x - wg_single) * (G / 2);
      n_img = a.B - img0 < G / 2 ? a.B - img0 : G / 2;
      n_rows = 2 * n_img;
    }
  }
  // batch row of workgroup row g (rows are [pass 0 of all images | pass 1 of images B_single ..])
  auto batch_row = [&](int g) {
    if (a.mode == FUSED_FORWARD) return row0 + g;
    if (!xshift) return img0 + g;
    const int b = img0 + (g >> 1);
    return (g & 1) ? a.B + b - a.B_single : b;
  };
  if (tid < 16) {
    sm[L.zero + tid] = 0.f;
    const int g = tid < n_rows ? tid : 0;
    reinterpret_cast<int *>(sm + L.tbo)[tid] = (batch_row(g) / a.tb_div) * a.tb_stride;
  }
  // the head's weights [4][c0p] (zero beyond the C real rows / c0 real channels) and bias
  for (int i = tid; i < 4 * a.c0p + 4; i += kThreads) {
    float v = 0.f;
    if (i < 4 * a.c0p) {
      const int c = i / a.c0p, k = i - c * a.c0p;
      if (c < C && k < a.c0) v = a.head_w[c * a.c0 + k];
    } else if (i - 4 * a.c0p < C) {
      v = a.head_b[i - 4 * a.c0p];
    }
    sm[L.head + i] = v;
  }
  // the model's small vectors (folded BN, biases, image-skip rows): one contiguous piece of the slab
  for (int i = tid; i < a.n_par; i += kThreads) sm[L.par + i] = a.fbase[a.par + i];
  // ---- the images: x of FORWARD mode, trajectory slot 0 of LOOP mode; rows beyond the batch compute on zeros
  for (int i = tid; i < G * kXSlot; i += kThreads) {
    const int k = i / kXSlot, e = i - k * kXSlot;
    float v = 0.f;
    if (k < n_img && e < E) {
      if (a.mode == FUSED_FORWARD) {
        const int r = row0 + k;
        v = a.x[(size_t)(r < a.B ? r : r - a.B + a.B_single) * E + e];
      } else {
        v = a.traj[(size_t)(img0 + k) * E + e];
      }
    }
    sm[L.x + i] = v;
  }
  // LOOP: this thread's elements of the update (image k, element e) and their rows in the noise table
  constexpr int kUpd = 3;                            // G * 768 / kThreads elements per thread at G = 2 (launch_unet_fused checks)
  int zrow[kUpd];
#pragma unroll
  for (int r = 0; r < kUpd; ++r) {
    const int i = tid + r * kThreads, k = i / E;
    zrow[r] = (a.mode == FUSED_LOOP && i < n_img * E) ? (a.z_row ? a.z_row[img0 + k] : img0 + k) : 0;
  }
  // the time-bias rows of the workgroup's rows for step st: [G][tb_cols] (requested a step ahead: behind the update phase)
  auto load_tb = [&](int st) __attribute__((always_inline)) {
    const float *tbstep = a.tb + (size_t)st * a.tb_rows * a.tb_stride;
    const int *tbo = reinterpret_cast<const int *>(sm + L.tbo);
    for (int i = tid; i < G * L.tb_cols; i += kThreads) {
      const int g = i / L.tb_cols, c = i - g * L.tb_cols;
      sm[L.tb + i] = tbstep[tbo[g] + c];
    }
  };
  __syncthreads();                                  // (tbo is read by load_tb)
  load_tb(0);
  __syncthreads();

  const int n_iter = a.mode == FUSED_FORWARD ? 1 : a.n_steps;
  const size_t slot = (size_t)a.B * E;
  for (int st = 0; st < n_iter; ++st) {
    const bool noise = (a.noise_mask >> st) & 1ull;
    const bool dead = a.mode == FUSED_LOOP && a.rule == DT_RULE_ENGINE && !noise;   // t == 0: the prediction is never used
    // the step's noise is requested before the forward: its latency disappears behind the layers
    float zv[kUpd];
#pragma unroll
    for (int r = 0; r < kUpd; ++r) {
      const int i = tid + r * kThreads, k = i / E, e = i - k * E;
      zv[r] = (a.mode == FUSED_LOOP && noise && i < n_img * E) ? a.z[(size_t)((long long)zrow[r] + a.z_shift[st]) * E + e] : 0.f;
    }
    if (!dead) {
#ifdef DT_TOOLS
      if (a.trace && tid == 0 && st == (a.n_steps > 1 ? 1 : 0)) a.trace[blockIdx.x * 32] = __builtin_readcyclecounter();
#endif
      for (int li = 0; li < a.n_ops; ++li) {
        const FusedOp &op = ops[li];
        switch (op.kind) {
          case FUSED_OP_FIRST: fused_first(op, L, a.fbase, G, xshift, C, wave, lane); break;
          case FUSED_OP_CONV:
            if (op.pb == 2) fused_conv<2>(op, L, a.fbase, G, xshift, C, wave, lane);
            else fused_conv<1>(op, L, a.fbase, G, xshift, C, wave, lane);
            break;
          case FUSED_OP_UP: fused_upsample(op, G, tid); break;
          default: fused_head(op, L, a.c0p, G, tid); break;
        }
        __syncthreads();
#ifdef DT_TOOLS
        if (a.trace && tid == 0 && st == (a.n_steps > 1 ? 1 : 0)) a.trace[blockIdx.x * 32 + li + 1] = __builtin_readcyclecounter();
#endif
      }
    }
    if (a.mode == FUSED_FORWARD) {
      // eps[r][c][Y][X] = bilinear x2 of the row's low-resolution head output (head_upsample_kernel's arithmetic)
      for (int i = tid; i < n_rows * E; i += kThreads) {
        const int g = i / E, e = i - g * E;
        a.eps[(size_t)(row0 + g) * E + e] = eps_at_lds(L.low + g * 256, e >> 8, (e & 255) >> 4, e & 15);
      }
      return;
    }
    // ---- CFG mix + reverse-diffusion update + trajectory store (dt_update.hip's arithmetic, bit for bit)
    const StepCoef k3{a.coef[st][0], a.coef[st][1], a.coef[st][2]};
    float *next = a.traj + (size_t)(st + 1) * slot;
#pragma unroll
    for (int r = 0; r < kUpd; ++r) {
      const int i = tid + r * kThreads;
      if (i >= n_img * E) continue;
      const int k = i / E, e = i - k * E, b = img0 + k;
      const float xv = sm[L.x + k * kXSlot + e];
      float o = xv;
      if (!dead) {
        const int c = e >> 8, y = (e & 255) >> 4, x = e & 15;
        float ev = eps_at_lds(L.low + (k << xshift) * 256, c, y, x);
        if (xshift) {
          const float cv = eps_at_lds(L.low + (2 * k + 1) * 256, c, y, x);
          ev = cfg_mix(ev, cv, a.wg ? a.wg[b] : a.w_scalar);
        }
        o = (a.rule != DT_RULE_PSAMPLE && !noise) ? step1_rt(a.rule, xv, ev, 0.f, k3, false) : step1_rt(a.rule, xv, ev, zv[r], k3, noise);
      }
      next[(size_t)b * E + e] = o;
      sm[L.x + k * kXSlot + e] = o;
    }
    if (st + 1 < n_iter) load_tb(st + 1);           // (every layer of this step is behind a barrier: the rows are free)
    __syncthreads();
#ifdef DT_TOOLS
    if (a.trace && tid == 0 && st == (a.n_steps > 1 ? 1 : 0)) a.trace[blockIdx.x * 32 + 31] = __builtin_readcyclecounter();
#endif
  }
}

// ------------------------------------------------------------------------------------------------ host side

// OIHW conv weights -> [tap][chunk][channel tile][lane][4]: lane (i = l & 15, s = l >> 4) holds input channels
// chunk * 16 + 4 s .. + 3 of output channel tile * 16 + i; the padded input channels follow the activations' layout
// ([0, split_cp) <-> real [0, split_c), the rest <-> real [split_c, cin): the decoder's concat of two padded tensors)
__global__ void pack_fused_conv_kernel(const float *__restrict__ w, float *__restrict__ dst, int cout, int cin, int taps, int kc,
                                       int ot, int split_c, int split_cp) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= taps * kc * ot * 256) return;
  const int j = idx & 3, lane = (idx >> 2) & 63;
  int t = idx >> 8;
  const int o_t = t % ot; t /= ot;
  const int k_c = t % kc, tap = t / kc;
  const int oc = o_t * 16 + (lane & 15), cp = k_c * 16 + 4 * (lane >> 4) + j;
  int ci;
  if (cp < split_cp) ci = cp < split_c ? cp : -1;
  else ci = cp - split_cp + split_c < cin ? cp - split_cp + split_c : -1;
  dst[idx] = (oc < cout && ci >= 0) ? w[((size_t)oc * cin + ci) * taps + tap] : 0.f;
}

int launch_pack_fused_conv(const float *w_oihw, float *dst, int cout, int cin, int taps, int kc, int ot, int split_c, int split_cp,
                           hipStream_t s) {
  const int n = taps * kc * ot * 256;
  pack_fused_conv_kernel<<<(n + 255) / 256, 256, 0, s>>>(w_oihw, dst, cout, cin, taps, kc, ot, split_c, split_cp);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// enc1.conv1's OIHW weights [cout][C][3][3] -> [2 chunks][channel tile][lane][4]: K index k = 9 c + tap, zero beyond 9 C
__global__ void pack_fused_first_kernel(const float *__restrict__ w, float *__restrict__ dst, int cout, int C, int ot) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= 2 * ot * 256) return;
  const int j = idx & 3, lane = (idx >> 2) & 63;
  const int t = idx >> 8, o_t = t % ot, k_c = t / ot;
  const int oc = o_t * 16 + (lane & 15), k = k_c * 16 + 4 * (lane >> 4) + j;
  dst[idx] = (oc < cout && k < 9 * C) ? w[(size_t)oc * 9 * C + k] : 0.f;
}

int launch_pack_fused_first(const float *w_oihw, float *dst, int cout, int C, int ot, hipStream_t s) {
  pack_fused_first_kernel<<<(2 * ot * 256 + 255) / 256, 256, 0, s>>>(w_oihw, dst, cout, C, ot);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

bool fused_eligible(int C, int c0p, int c1p) {
  if (C < 1 || C > 3) return false;
  if (c0p % 16 || c1p % 16 || c0p > 32 || c1p > 64) return false;
  return c1p >= c0p && c1p <= 3 * c0p;             // buffer overlays of fused_lds(): U1 inside B + S0..S2 needs 16 s1 <= 64 s0
}

// pixel stride of an LDS activation tensor with c channels: c + 8 floats.  A ds_read_b128 is served in four groups of 16
// lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md) and a lane (pixel, s) reads 16 bytes at pixel * stride + 16 s: with
// stride = c (64 / 128 / 256 bytes) 4 to 16 pixels of a group land on the same banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE =
// 0.73 measured, and with eight waves reading fragments the LDS, not the matrix pipe, paced the layers); a stride of 2 mod 4
// sixteen-byte slots is conflict-free for every group and any tap shift.
int fused_stride(int c) { return c + 8; }

// LDS placement (floats) for G rows; see the lifetimes in fused_ops()
//   [x G*768][low G*256][zero 16][tb offsets 16][head][K-split scratch][parameters][time-bias rows]
//   [B: P1][C: S0 S1 S2 S3][E': P2 / H8][D: O2]     H0 overlays C + E' + D (and extends the arena where that is shorter)
FusedLds fused_lds(int G, int c0p, int c1p) {
  FusedLds L{};
  const int s0 = fused_stride(c0p), s1 = fused_stride(c1p);
  const int a_sz = 64 * G * s0, u = 16 * G * s1;
  L.x = 0;
  L.low = L.x + G * kXSlot;
  L.zero = L.low + G * 256;
  L.tbo = L.zero + 16;
  L.head = L.tbo + 16;                             // final 1x1: [4][c0p] weights + [4] bias
  L.scr = L.head + 4 * c0p + 4;                    // partial tiles of K-split units: kWaves x 64 lanes x 4
  L.par = L.scr + kWaves * 256;                    // the model's small vectors: 5 x cout_p per block + the image-skip rows
  L.tb_cols = 2 * c0p + 6 * c1p;                   // padded out-channels of the eight blocks = columns of a time-bias row in use
  L.tb = L.par + 5 * L.tb_cols + 4 * c0p;          // [G][tb_cols] rows of the current step
  L.b = L.tb + G * L.tb_cols;
  L.c = L.b + a_sz;
  L.e = L.c + 4 * u;
  L.d = L.e + a_sz;
  L.total = L.d + 4 * u;
  if (L.c + 256 * G * s0 > L.total) L.total = L.c + 256 * G * s0;      // H0 [256 G][s0]
  return L;
}

size_t fused_lds_bytes(int G, int c0p, int c1p) { return (size_t)fused_lds(G, c0p, c1p).total * sizeof(float); }

// The layer table of one forward for G rows per workgroup (host memory; the caller uploads it).  Buffers:
//   H0 enc1.conv1 out -> C (overlay) | P1 pooled enc1 -> B | H2 -> C | O2 -> D (skip of dec1, lives to the end) | P2 -> E'
//   H3 -> S1, O3 -> S2 (skip of dec2), P3 H4 O4 (skip of dec3) P4 H5 O5 -> quarters of S3
//   U3 H6 O6 -> quarters of S0 | U2 -> S1, H7 -> S0, O7 -> S3 | U1 -> B.. (4u floats), H8 -> E', O8 -> the a_sz floats before E'
int fused_ops(const FusedModel &m, int G, FusedOp *ops) {
  const int c0p = m.c0p, c1p = m.c1p;
  const FusedLds L = fused_lds(G, c0p, c1p);
  const int s0 = fused_stride(c0p), s1 = fused_stride(c1p);            // pixel strides of the c0p- / c1p-channel tensors
  const int u = 16 * G * s1, q = u / 4, gc = G * s1;
  const int S0 = L.c, S1 = L.c + u, S2 = L.c + 2 * u, S3 = L.c + 3 * u;
  const int H0 = L.c, P1 = L.b, H2 = L.c, O2 = L.d, P2 = L.e;
  const int H3 = S1, O3 = S2, P3 = S3, H4 = S3 + q, O4 = S3 + 2 * q, P4 = S3 + 3 * q, H5 = P4 + gc, O5 = P4 + 2 * gc;
  const int U3 = S0, H6 = S0 + q, O6 = S0 + 2 * q, U2 = S1, H7 = S0, O7 = S3;
  const int U1 = L.b, H8 = L.e, O8 = L.b + 4 * u;
  const int k0 = c0p / 16, k1 = c1p / 16;
  int n = 0;
  // pixel tiles per unit and K split of a level: as many units as the eight waves where the level allows it
  auto conv = [&](const FusedConvW &cw, int kc, int ot, int wsh, int inA, int inB, int cs_in, int cca, int emode,
                  int tb_off, int out, int cs, int pool) -> FusedOp & {
    FusedOp &o = ops[n++];
    o = FusedOp{};
    o.kind = FUSED_OP_CONV; o.w = cw.w; o.scale = cw.scale; o.shift = cw.shift;
    o.kc = kc; o.ot = ot; o.wsh = wsh; o.tap_lo = wsh == 0 ? 4 : 0; o.tap_hi = wsh == 0 ? 5 : 9;
    const int ntile = ((G << (2 * wsh)) + 15) / 16;
    o.pb = wsh == 4 ? 2 : ((wsh == 3 && (ntile / 2) * ot >= kWaves) ? 2 : 1);      // (two tiles at most: 128 VGPRs per wave)
    o.ks = 1;
    if (o.pb == 1) {
      const int units = ntile * ot, nsteps = (o.tap_hi - o.tap_lo) * kc;
      while (o.ks < 4 && units * o.ks * 2 <= kWaves && o.ks * 2 <= nsteps) o.ks *= 2;
    }
    o.inA = inA; o.inB = inB; o.cs_in = cs_in; o.cca = cca; o.emode = emode; o.tb_off = tb_off;
    o.out = out; o.cs = cs; o.pool = pool;
    return o;
  };
  auto up = [&](int src, int dst, int wsh) {
    FusedOp &o = ops[n++];
    o = FusedOp{};
    o.kind = FUSED_OP_UP; o.inA = src; o.out = dst; o.wsh = wsh; o.cs = s1; o.kc = k1;
  };
  const FusedBlockW *b = m.blk;
  {   // enc1.conv1
    FusedOp &o = ops[n++];
    o = FusedOp{};
    o.kind = FUSED_OP_FIRST; o.w = m.wf; o.scale = b[0].c1.scale; o.shift = b[0].c1.shift; o.tb_off = b[0].tb_off; o.out = H0; o.cs = s0;
    o.ot = k0;
  }
  {   // enc1.conv2 + image skip + pool (enc1's full-resolution output has no other reader)
    FusedOp &o = conv(b[0].c2, k0, k0, 4, H0, H0, s0, k0, FUSED_E_IMAGE, 0, -1, s0, P1);
    o.wr = m.w3;
  }
  conv(b[1].c1, k0, k1, 3, P1, P1, s0, k0, FUSED_E_TIMEBIAS, b[1].tb_off, H2, s1, -1);
  {
    FusedOp &o = conv(b[1].c2, k1, k1, 3, H2, H2, s1, k1, FUSED_E_SKIPCONV, 0, O2, s1, P2);
    o.wr = b[1].cr.w; o.br = b[1].cr.shift; o.kcr = k0; o.rA = P1; o.rB = P1; o.rcs = s0; o.rcca = k0;
  }
  const int lvl_in[3] = {P2, P3, P4}, lvl_h[3] = {H3, H4, H5}, lvl_o[3] = {O3, O4, O5}, lvl_p[3] = {P3, P4, -1};
  for (int j = 2; j <= 4; ++j) {   // enc3, enc4, bottleneck: identity residual
    const int wsh = 4 - j;         // 4x4, 2x2, 1x1
    conv(b[j].c1, k1, k1, wsh, lvl_in[j - 2], lvl_in[j - 2], s1, k1, FUSED_E_TIMEBIAS, b[j].tb_off, lvl_h[j - 2], s1, -1);
    FusedOp &o = conv(b[j].c2, k1, k1, wsh, lvl_h[j - 2], lvl_h[j - 2], s1, k1, FUSED_E_IDENTITY, 0, lvl_o[j - 2], s1, lvl_p[j - 2]);
    o.rA = lvl_in[j - 2]; o.rcs = s1;
  }
  const int d_up[3] = {U3, U2, U1}, d_src[3] = {O5, O6, O7}, d_skip[3] = {O4, O3, O2}, d_h[3] = {H6, H7, H8}, d_o[3] = {O6, O7, O8};
  for (int j = 5; j <= 7; ++j) {   // dec3, dec2, dec1: upsample, conv over [up | skip] read in place, 1x1 skip conv of the same concat
    const int d = j - 5, wsh = d + 1;
    const int cout_p = j == 7 ? c0p : c1p, ot = cout_p / 16, so = j == 7 ? s0 : s1;
    up(d_src[d], d_up[d], wsh - 1);
    conv(b[j].c1, 2 * k1, ot, wsh, d_up[d], d_skip[d], s1, k1, FUSED_E_TIMEBIAS, b[j].tb_off, d_h[d], so, -1);
    FusedOp &o = conv(b[j].c2, ot, ot, wsh, d_h[d], d_h[d], so, ot, FUSED_E_SKIPCONV, 0, d_o[d], so, -1);
    o.wr = b[j].cr.w; o.br = b[j].cr.shift; o.kcr = 2 * k1; o.rA = d_up[d]; o.rB = d_skip[d]; o.rcs = s1; o.rcca = k1;
  }
  {
    FusedOp &o = ops[n++];
    o = FusedOp{};
    o.kind = FUSED_OP_HEAD; o.inA = O8; o.cs = s0;
  }
  return n;
}

// algorithmic FLOPs (2 MAC, unpadded channels) of one forward of one row at 16 x 16
double fused_flops_per_row(int C, int c0, int c1) {
  double mac = 256.0 * c0 * 9 * C + 256.0 * c0 * c0 * 9 + 256.0 * c0 * C;            // enc1 (+ image skip)
  mac += 64.0 * c1 * (9.0 * c0 + 9.0 * c1 + c0);                                      // enc2
  mac += (16.0 + 4.0) * c1 * c1 * 18.0 + 2.0 * c1 * c1;                               // enc3, enc4, bottleneck (centre tap)
  mac += (4.0 + 16.0) * c1 * (9.0 * 2 * c1 + 9.0 * c1 + 2.0 * c1);                    // dec3, dec2
  mac += 64.0 * c0 * (9.0 * 2 * c1 + 9.0 * c0 + 2.0 * c1) + 64.0 * C * c0;            // dec1 + head
  return 2.0 * mac;
}

int launch_unet_fused(const FusedArgs &a, hipStream_t s) {
  static std::once_flag attr_once;
  static int attr_status = DT_OK;
  std::call_once(attr_once, [] {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&unet_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) attr_status = (int)e;
  });
  if (attr_status != DT_OK) return attr_status;
  if (a.G != 2 || a.n_ops < 1 || !a.ops) return DT_E_ARG;      // (the kernel's update slots per thread are sized for G = 2)
  if (a.mode == FUSED_LOOP && (a.n_steps < 1 || a.n_steps > kFusedMaxSteps)) return DT_E_ARG;
  const size_t lds = (size_t)a.lds.total * sizeof(float);
  if (lds > 160 * 1024) return DT_E_SHAPE;
  const int Bt = a.B * a.n_pass - a.B_single * (a.n_pass - 1);
  int grid;
  if (a.mode == FUSED_FORWARD) {
    grid = (Bt + a.G - 1) / a.G;
  } else {
    const int n_single = a.n_pass == 1 ? a.B : a.B_single;
    grid = (n_single + a.G - 1) / a.G + (a.B - n_single + a.G / 2 - 1) / (a.G / 2);
  }
  const int steps = a.mode == FUSED_FORWARD ? 1 : a.n_steps;
  ProfileScope prof(KC_FUSED, a.flops_per_row * Bt * steps, 4.0 * a.B * a.C * 256 * (steps + 1.0), s);
#ifdef DT_TOOLS
  if (getenv("DT_FUSED_TRACE")) {     // per-layer cycle stamps of the second step (tools/fused_check.py), averaged over the workgroups
    FusedArgs t = a;
    unsigned long long *dev = nullptr;
    DT_HIP_TRY(hipMalloc((void **)&dev, (size_t)grid * 32 * sizeof(unsigned long long)));
    DT_HIP_TRY(hipMemsetAsync(dev, 0, (size_t)grid * 32 * sizeof(unsigned long long), s));
    t.trace = dev;
    unet_fused_kernel<<<grid, kThreads, lds, s>>>(t, t.ops);
    DT_HIP_TRY(hipStreamSynchronize(s));
    std::vector<unsigned long long> h((size_t)grid * 32);
    DT_HIP_TRY(hipMemcpy(h.data(), dev, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    (void)hipFree(dev);
    FusedOp ops[kFusedMaxOps];
    DT_HIP_TRY(hipMemcpy(ops, a.ops, sizeof(FusedOp) * a.n_ops, hipMemcpyDeviceToHost));
    fprintf(stderr, "[dt_fused] grid %d, cycles per layer of one step (mean over workgroups):\n", grid);
    double total = 0;
    for (int li = 0; li <= a.n_ops; ++li) {
      double sum = 0; int n = 0;
      for (int g = 0; g < grid; ++g) {
        const unsigned long long t0 = h[(size_t)g * 32 + li], t1 = h[(size_t)g * 32 + (li == a.n_ops ? 31 : li + 1)];
        if (t0 && t1 > t0) { sum += (double)(t1 - t0); ++n; }
      }
      if (!n) continue;
      total += sum / n;
      if (li < a.n_ops) fprintf(stderr, "  op %2d kind %d wsh %d pb %d ks %d kc %d ot %d emode %d: %9.0f\n", li, ops[li].kind, ops[li].wsh, ops[li].pb, ops[li].ks, ops[li].kc, ops[li].ot, ops[li].emode, sum / n);
      else fprintf(stderr, "  update / store: %9.0f\n", sum / n);
    }
    fprintf(stderr, "  total %9.0f cycles\n", total);
    return DT_OK;
  }
#endif
  unet_fused_kernel<<<grid, kThreads, lds, s>>>(a, a.ops);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

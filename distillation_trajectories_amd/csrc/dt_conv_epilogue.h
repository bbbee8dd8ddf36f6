// Shared epilogue of the implicit-GEMM convolution kernels (fp32-MFMA and split-bf16 variants).
// Accumulator layout (both MFMA families, 32x32 tiles): register r of lane l holds
// D[row (r&3) + 8*(r>>2) + 4*(l>>5)][col l&31]; a register is therefore a 128-B run of consecutive
// channels for two pixel rows -> coalesced NHWC stores.
#pragma once
#include "dt_internal.h"

namespace dt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: plain dwordx4 loads, always promoted to VGPRs

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline float bf16_to_f32(__bf16 v) { return (float)v; }

// exact three-way split of 8 consecutive fp32 values into bf16 planes (packed 8 x bf16 = 16 B each)
__device__ inline void split8(const f32x4 lo, const f32x4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float a = i < 4 ? lo[i] : hi[i - 4];
    const __bf16 a1 = (__bf16)a;
    const float r1 = a - (float)a1;
    const __bf16 a2 = (__bf16)r1;
    const float r2 = r1 - (float)a2;
    p1[i] = a1; p2[i] = a2; p3[i] = (__bf16)r2;
  }
}

// floats of LDS the staged epilogue needs for a BN-column tile: WM * 32 rows (one 32-row accumulator tile of each
// wave row; the four waves sit WM x 4/WM over the tile) at a pitch of BN + 4 floats
template <int BN, int WM = 2>
constexpr int epilogue_stage_floats() { return WM * 32 * (BN + 4); }

// Epilogue of every convolution kernel, staged through LDS.
//
// The accumulator layout (lane = column, register = row) makes a direct epilogue store 4 bytes per lane, and the
// side inputs (residual, time bias) load the same way: measured per-workgroup timelines (tools/block_timeline.py)
// put such an epilogue at 14 us for a 128 x 128 tile on an idle chip and 28-35 us with a residual or the fused pool
// (one exposed memory latency per accumulator register).  Here a 32-row accumulator tile of each wave row goes
// through LDS once ([WM * 32][BN + 4] floats, conflict-free both ways) and comes back as float4 per lane with a whole
// row segment per 32 lanes: every side input is one batched float4 load per unit, every store is 16 bytes per lane,
// and the mode flags are tested per pass, not per element.
//
// `stage` is any LDS the kernel no longer needs (>= epilogue_stage_floats<BN>() floats); every wave of the
// workgroup must call this (it contains barriers).  Per row m and channel n:
//   v = in2 ? acc + bias2 : relu?(acc * scale + shift);  v += time bias row;  v += residual;  v += x3 skip;
//   out = v;  pool_out = 2x2 max of v (32-row tiles hold whole row pairs for W <= 16);  head_out = 1x1 head of v.
//
// Split-K (p.splits > 1, grid.z slices of the K walk): the raw sums go to this slice's slab and splitk_epilogue_kernel
// sums the slabs in z order in a second launch.  (Finishing inside the launch -- write-through slab stores, a ticket
// counter per tile, the last workgroup to arrive sums the slabs behind an agent-scope acquire -- was built and measured
// in round 2: bit-identical results, the same time at batch 256 (the acquire + the serial slab reads of one workgroup
// per tile cost what the second launch costs) and 38 % slower at batch 8, where tiles are few and splits deep.)
//
// WK > 1 (strip kernel with the K split across waves): WK waves hold partial sums of the same tile; each writes its own
// copy of the stage and a unit is the sum of the copies in wave order (deterministic).
template <int MI, int NI, int WM = 2, int WK = 1>
__device__ inline void conv_epilogue(const ConvParams &p, f32x16 (&acc)[MI][NI], float *stage, int m0, int n0, int wm, int wn,
                                     int half, int l31, int wk = 0) {
  constexpr int BN = NI * 32 * (4 / (WM * WK)), P = BN + 4, C4 = BN / 4, U = WM * C4 / 8;   // U float4 units per thread and pass
  constexpr int COPY = WM * 32 * P;
  const int tid = threadIdx.x;
  const int HW = p.H * p.W;
  // pictures are powers of two in every benchmark shape: row -> (picture, y, x) by shifts there (an integer division by a run-time
  // value is ~30 VALU instructions, and the fused pool / image-skip / class-bias paths do several per unit)
  const bool pow2 = (p.W & (p.W - 1)) == 0 && (HW & (HW - 1)) == 0;
  const int w_sh = 31 - __builtin_clz(p.W), hw_sh = 31 - __builtin_clz(HW);
  auto div_hw = [&](int m) { return pow2 ? m >> hw_sh : m / HW; };
  auto div_w = [&](int r) { return pow2 ? r >> w_sh : r / p.W; };
  const bool slab_mode = p.splits > 1;
  const size_t slab_stride = (size_t)p.M * p.cout_p;
  // ---- this thread's units of a pass: row = tid / C4 + k * (256 / C4), four consecutive channels
  const int c4 = tid % C4, row0 = tid / C4;
  const int n = n0 + c4 * 4;
  const bool n_ok = n < p.cout_p;
  const int nn = n_ok ? n : 0;
  auto unit = [&](int k) __attribute__((always_inline)) {
    const float *src = stage + (row0 + k * (256 / C4)) * P + c4 * 4;
    f32x4 v = *reinterpret_cast<const f32x4 *>(src);
#pragma unroll
    for (int c = 1; c < WK; ++c) v += *reinterpret_cast<const f32x4 *>(src + c * COPY);
    return v;
  };
  auto unit_row = [&](int mi, int k) { const int row = row0 + k * (256 / C4); return m0 + (row >> 5) * (MI * 32) + mi * 32 + (row & 31); };
  auto to_stage = [&](int mi) __attribute__((always_inline)) {
    __syncthreads();                                   // the main loop (or the previous pass) is done with this LDS
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int L = 4 * half + (r & 3) + 8 * (r >> 2);
        stage[wk * COPY + (wm * 32 + L) * P + wn * (NI * 32) + ni * 32 + l31] = acc[mi][ni][r];
      }
    __syncthreads();
  };

  if (slab_mode) {
    float *slab = p.slab + (size_t)blockIdx.z * slab_stride;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      to_stage(mi);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const int m = unit_row(mi, k);
        if (n_ok && m < p.M)
          *reinterpret_cast<f32x4 *>(slab + (size_t)m * p.cout_p + n) = unit(k);
      }
    }
    return;                                            // splitk_epilogue_kernel finishes the layer
  }

  // 2x2 max pool of the values the threads just wrote back into the stage (each unit is its thread's own): a 32-row tile
  // starts on an even picture row and holds whole row pairs (W <= 16), eight pooled pixels per tile
  // (n_src > 1: the passes of a dual-output layer, pass i staged at src + i * src_pitch and pooled to pool_out + i * out_pitch)
  auto pool_pass = [&](int mi, float *pool_out, const float *src, int n_src, int src_pitch, size_t out_pitch) __attribute__((always_inline)) {
    __syncthreads();
    const int Wo = p.W >> 1;
    for (int q0 = tid; q0 < n_src * WM * 8 * C4; q0 += 256) {
      const int pass = q0 / (WM * 8 * C4), q = q0 - pass * (WM * 8 * C4);
      const int pp = q / C4, qc = q % C4, j = pp & 7;
      const int yy = pow2 ? j >> (w_sh - 1) : j / Wo, xx = j - yy * Wo;
      const int L0 = 2 * yy * p.W + 2 * xx;
      const int m = m0 + (pp >> 3) * (MI * 32) + mi * 32 + L0;
      const int qn = n0 + qc * 4;
      if (m >= p.M || qn >= p.cout_p || (pass && m < p.dup_skip)) continue;
      const float *s0 = src + pass * src_pitch + ((pp >> 3) * 32 + L0) * P + qc * 4;
      const f32x4 a = *reinterpret_cast<const f32x4 *>(s0), b = *reinterpret_cast<const f32x4 *>(s0 + P);
      const f32x4 c = *reinterpret_cast<const f32x4 *>(s0 + p.W * P), d = *reinterpret_cast<const f32x4 *>(s0 + (p.W + 1) * P);
      f32x4 mx;
#pragma unroll
      for (int e = 0; e < 4; ++e) mx[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c[e], d[e]));
      const int bimg = div_hw(m), rem = m - bimg * HW;
      const int y = div_w(rem), x = rem - y * p.W;
      *reinterpret_cast<f32x4 *>(pool_out + pass * out_pitch + (((size_t)bimg * (p.H >> 1) + (y >> 1)) * Wo + (x >> 1)) * p.cout_p + qn) = mx;
    }
  };
  // enc1's 1x1 skip of the <= 3-channel image, recomputed from the NCHW image itself: the value added to unit (m, n..n+3)
  f32x4 w3r[4];                                        // this thread's four channels: weights of the <= 3 image channels + bias
#pragma unroll
  for (int e = 0; e < 4; ++e) w3r[e] = p.x3 ? *reinterpret_cast<const f32x4 *>(p.w3 + 4 * (nn + e)) : f32x4{0.f, 0.f, 0.f, 0.f};
  auto x3_skip = [&](int m, int) __attribute__((always_inline)) {
    f32x4 r;
    const int img = div_hw(m), pix = m - img * HW;                 // (x3_hw == H * W)
    const float *xr = p.x3 + (size_t)(img < p.x3_imgs ? img : img % p.x3_imgs) * p.x3_c * p.x3_hw + pix;
    const float x0 = xr[0], x1 = p.x3_c > 1 ? xr[p.x3_hw] : 0.f, x2 = p.x3_c > 2 ? xr[2 * p.x3_hw] : 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const f32x4 w3 = w3r[e];
      float rs = w3[3];
      rs = fmaf(x0, w3[0], rs);
      if (p.x3_c > 1) rs = fmaf(x1, w3[1], rs);
      if (p.x3_c > 2) rs = fmaf(x2, w3[2], rs);
      r[e] = rs;
    }
    return r;
  };

#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    f32x4 v[U];
    int mrow[U];
    bool ok[U];
    to_stage(mi);
#pragma unroll
    for (int k = 0; k < U; ++k) {
      mrow[k] = unit_row(mi, k);
      ok[k] = n_ok && mrow[k] < p.M;
    }
#pragma unroll
    for (int k = 0; k < U; ++k) v[k] = unit(k);
    if (p.n_dup) {
      // enc1.conv2 over the B images ONCE for all passes of the step (see ConvParams::tbc): per pass the class bias of the
      // unit's pixel (corner / edge / interior), BN + ReLU, the shared image skip, the store and the pool
      const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.scale + nn), sh = *reinterpret_cast<const f32x4 *>(p.shift + nn);
      // (rows are clamped instead of branched over, so that every unit's loads are issued before the first is waited for)
      f32x4 rs[U];
      int cls[U], mc[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        mc[k] = ok[k] ? mrow[k] : m0;
        const int pix = mc[k] - div_hw(mc[k]) * HW, y = div_w(pix), x = pix - y * p.W;
        cls[k] = 3 * (y == 0 ? 0 : (y == p.H - 1 ? 2 : 1)) + (x == 0 ? 0 : (x == p.W - 1 ? 2 : 1));
      }
#pragma unroll
      for (int k = 0; k < U; ++k) rs[k] = p.x3 ? x3_skip(mc[k], nn) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (p.n_dup == 2 && p.dup_stage2 && p.pool_out) {
        // two passes (the CFG pair), room for a second stage behind the partial-tile copies: both passes' values are formed
        // together (their class-bias loads overlap), staged side by side and pooled in ONE sweep -- two barriers per 32-row
        // pass instead of six (per-workgroup timelines put the pass-after-pass form at 27 us per 256 x 64 tile)
        float *stage2 = stage + WK * COPY;
        f32x4 b0[U], b1[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          // (32-bit row arithmetic: launch_conv bounds n_dup * M * cout_p by 2^31; a 64-bit division costs ~100 instructions)
          // (rows below dup_skip have no second pass: their b1 / o1 are formed from a valid row and never stored or pooled)
          const unsigned r0 = (unsigned)mc[k] / (unsigned)p.m_per_tb;
          const unsigned r1 = (unsigned)(p.dup_rows + (mc[k] >= p.dup_skip ? mc[k] - p.dup_skip : mc[k])) / (unsigned)p.m_per_tb;
          b0[k] = *reinterpret_cast<const f32x4 *>(p.tbc + (size_t)r0 * p.tb_stride + cls[k] * p.cout_p + nn);
          b1[k] = *reinterpret_cast<const f32x4 *>(p.tbc + (size_t)r1 * p.tb_stride + cls[k] * p.cout_p + nn);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          f32x4 o0 = (v[k] + b0[k]) * sc + sh, o1 = (v[k] + b1[k]) * sc + sh;
          if (p.relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = fmaxf(o0[e], 0.f); o1[e] = fmaxf(o1[e], 0.f); }
          }
          o0 += rs[k]; o1 += rs[k];
          if (ok[k] && !p.skip_out) {
            *reinterpret_cast<f32x4 *>(p.out + (size_t)mrow[k] * p.cout_p + n) = o0;
            if (mrow[k] >= p.dup_skip) *reinterpret_cast<f32x4 *>(p.out + ((size_t)p.dup_rows + mrow[k] - p.dup_skip) * p.cout_p + n) = o1;
          }
          *reinterpret_cast<f32x4 *>(stage + (row0 + k * (256 / C4)) * P + c4 * 4) = o0;
          *reinterpret_cast<f32x4 *>(stage2 + (row0 + k * (256 / C4)) * P + c4 * 4) = o1;
        }
        pool_pass(mi, p.pool_out, stage, 2, WK * COPY, (size_t)((p.dup_rows - p.dup_skip) >> 2) * p.cout_p);
        continue;
      }
      for (int pass = 0; pass < p.n_dup; ++pass) {
        // dup_skip is a multiple of 256 rows and m0 of the tile height (64 / 128 / 256): a workgroup's rows all lie on one side
        if (pass && m0 < p.dup_skip) break;                              // (block-uniform) single-pass images: no second pass
        f32x4 o[U];
        const int shift = pass ? pass * p.dup_rows - p.dup_skip : 0;     // (dup_skip > 0 only with n_dup == 2)
#pragma unroll
        for (int k = 0; k < U; ++k) {
          o[k] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (ok[k]) {
            const size_t mg = (size_t)shift + mrow[k];
            const unsigned tr = (unsigned)(shift + mrow[k]) / (unsigned)p.m_per_tb;
            const f32x4 b = *reinterpret_cast<const f32x4 *>(p.tbc + (size_t)tr * p.tb_stride + cls[k] * p.cout_p + n);
            o[k] = (v[k] + b) * sc + sh;
            if (p.relu) {
#pragma unroll
              for (int e = 0; e < 4; ++e) o[k][e] = fmaxf(o[k][e], 0.f);
            }
            o[k] += rs[k];
            if (!p.skip_out) *reinterpret_cast<f32x4 *>(p.out + mg * p.cout_p + n) = o[k];
          }
        }
        if (p.pool_out) {
          if (pass) __syncthreads();                   // the previous pass's windows are read
#pragma unroll
          for (int k = 0; k < U; ++k) *reinterpret_cast<f32x4 *>(stage + (row0 + k * (256 / C4)) * P + c4 * 4) = o[k];
          pool_pass(mi, p.pool_out + (size_t)(shift >> 2) * p.cout_p, stage, 1, 0, 0);
        }
      }
      continue;
    }
    if (p.in2) {
      const f32x4 b2 = *reinterpret_cast<const f32x4 *>(p.bias2 + nn);   // BN/ReLU were applied before the skip walk
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] += b2;
    } else {
      const f32x4 sc = *reinterpret_cast<const f32x4 *>(p.scale + nn), sh = *reinterpret_cast<const f32x4 *>(p.shift + nn);
#pragma unroll
      for (int k = 0; k < U; ++k) {
        v[k] = v[k] * sc + sh;
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[k][e] = fmaxf(v[k][e], 0.f);
        }
      }
    }
    if (p.tb) {
      f32x4 t[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        t[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok[k]) t[k] = *reinterpret_cast<const f32x4 *>(p.tb + (size_t)(mrow[k] / p.m_per_tb) * p.tb_stride + n);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] += t[k];
    }
    if (p.add) {
      f32x4 a[U];
#pragma unroll
      for (int k = 0; k < U; ++k) {
        a[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ok[k]) a[k] = *reinterpret_cast<const f32x4 *>(p.add + (size_t)mrow[k] * p.cout_p + n);
      }
#pragma unroll
      for (int k = 0; k < U; ++k) v[k] += a[k];
    }
    if (p.x3) {
#pragma unroll
      for (int k = 0; k < U; ++k)
        if (ok[k]) v[k] += x3_skip(mrow[k], nn);
    }
    if (p.head_out) {
      // final 1x1 head on the rows this workgroup holds completely: C4 consecutive lanes own one row
      f32x4 hw[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        hw[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c < p.head_c && n_ok) {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (n + e < p.head_cin) hw[c][e] = p.head_w[c * p.head_cin + n + e];
        }
      }
#pragma unroll
      for (int k = 0; k < U; ++k) {
        float part[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          part[c] = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) part[c] = fmaf(hw[c][e], v[k][e], part[c]);
#pragma unroll
          for (int off = C4 / 2; off > 0; off >>= 1) part[c] += __shfl_xor(part[c], off, 64);
        }
        if (c4 == 0 && mrow[k] < p.M) {
          f32x4 o = {part[0] + p.head_b[0], 0.f, 0.f, 0.f};
          if (p.head_c > 1) o[1] = part[1] + p.head_b[1];
          if (p.head_c > 2) o[2] = part[2] + p.head_b[2];
          *reinterpret_cast<f32x4 *>(p.head_out + (size_t)mrow[k] * 4) = o;
        }
      }
      continue;                                        // dec1's output itself has no other consumer
    }
#pragma unroll
    for (int k = 0; k < U; ++k)
      if (ok[k]) *reinterpret_cast<f32x4 *>(p.out + (size_t)mrow[k] * p.cout_p + n) = v[k];
    if (p.pool_out) {
#pragma unroll
      for (int k = 0; k < U; ++k) *reinterpret_cast<f32x4 *>(stage + (row0 + k * (256 / C4)) * P + c4 * 4) = v[k];
      pool_pass(mi, p.pool_out, stage, 1, 0, 0);
    }
  }
}

// relu(acc*scale+shift) in registers, between the main K walk and the fused skip walk
template <int MI, int NI>
__device__ inline void conv_midpoint(const ConvParams &p, f32x16 (&acc)[MI][NI], int n0, int wn, int l31) {
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
    const float sc = p.scale[n], sh = p.shift[n];   // arrays are n_p long: always in range
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = acc[mi][ni][r] * sc + sh;
        acc[mi][ni][r] = p.relu ? fmaxf(v, 0.f) : v;
      }
  }
}

}  // namespace dt

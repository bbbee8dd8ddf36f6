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

template <int MI, int NI>
__device__ inline void conv_epilogue(const ConvParams &p, f32x16 (&acc)[MI][NI], int m0, int n0, int wm, int wn,
                                     int half, int l31) {
  // ---- split-K: park the raw partial sums; splitk_epilogue_kernel finishes the layer
  if (p.splits > 1) {
    float *slab = p.slab + (size_t)blockIdx.z * p.M * p.cout_p;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
      if (n >= p.cout_p) continue;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int mb = m0 + wm * (MI * 32) + mi * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m < p.M) slab[(size_t)m * p.cout_p + n] = acc[mi][ni][r];
        }
      }
    }
    return;
  }

  // ---- epilogue: folded BN, ReLU, time bias, residual; one 128-B channel run per (register, half).
  // Everything that depends on the output ROW only (time-bias row, residual / x3 / pool addresses) is worked out once
  // per row, not once per element, and without per-element integer division: a 32-row accumulator tile starts at a
  // wave-uniform row, so the quotient of its first row by m_per_tb (resp. H*W) is divided once and the other rows
  // only compare their remainder (a 32-row run crosses at most one boundary when the divisor is >= 32).
  int ncol[NI];
  float sc[NI], sh[NI], b2[NI];
  float4 w3[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    ncol[ni] = n0 + wn * (NI * 32) + ni * 32 + l31;
    const bool ok = ncol[ni] < p.cout_p;
    const int nn = ok ? ncol[ni] : 0;
    sc[ni] = p.scale[nn]; sh[ni] = p.shift[nn];
    b2[ni] = p.in2 ? p.bias2[nn] : 0.f;
    w3[ni] = p.x3 ? *reinterpret_cast<const float4 *>(p.w3 + 4 * nn) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (!ok) ncol[ni] = -1;
  }
  const int HW = p.H * p.W;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int mt = m0 + wm * (MI * 32) + mi * 32;            // first row of this 32-row tile (wave-uniform)
    const bool tb_fast = p.m_per_tb >= 32, hw_fast = HW >= 32;
    const int tq = mt / p.m_per_tb, tr = mt - tq * p.m_per_tb;
    const int iq = mt / HW, ir = mt - iq * HW;
    // time bias: a 32-row tile sees at most two rows of the table (fast path): both are loaded once per column
    float tb0[NI], tb1[NI];
    const bool tb_two = p.tb && tb_fast;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      tb0[ni] = tb1[ni] = 0.f;
      if (tb_two && ncol[ni] >= 0) {
        const int last = (p.M - 1) / p.m_per_tb;                 // rows past M are never stored; keep the read in range
        tb0[ni] = p.tb[(size_t)(tq < last ? tq : last) * p.tb_stride + ncol[ni]];
        tb1[ni] = p.tb[(size_t)(tq + 1 < last ? tq + 1 : last) * p.tb_stride + ncol[ni]];
      }
    }
    float vv[NI][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int L = 4 * half + (r & 3) + 8 * (r >> 2);       // row inside the tile
      const int m = mt + L;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) vv[ni][r] = 0.f;
      if (m >= p.M) continue;
      const size_t orow = (size_t)m * p.cout_p;
      const float *tbrow = nullptr;
      const bool tb_second = tr + L >= p.m_per_tb;
      if (p.tb && !tb_fast) tbrow = p.tb + (size_t)(m / p.m_per_tb) * p.tb_stride;
      float x0 = 0.f, x1 = 0.f, x2 = 0.f;
      if (p.x3) {
        const float *xr = p.x3 + (size_t)m * p.x3_stride;
        x0 = xr[0];
        if (p.x3_c > 1) x1 = xr[p.x3_step];
        if (p.x3_c > 2) x2 = xr[2 * p.x3_step];
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int n = ncol[ni];
        if (n < 0) continue;
        float v;
        if (p.in2) {
          v = acc[mi][ni][r] + b2[ni];            // BN/ReLU were applied in registers before the skip walk
        } else {
          v = acc[mi][ni][r] * sc[ni] + sh[ni];
          if (p.relu) v = fmaxf(v, 0.f);
        }
        if (tb_two) v += tb_second ? tb1[ni] : tb0[ni];
        else if (tbrow) v += tbrow[n];
        if (p.add) v += p.add[orow + n];
        if (p.x3) {
          float rs = w3[ni].w;
          rs = fmaf(x0, w3[ni].x, rs);
          if (p.x3_c > 1) rs = fmaf(x1, w3[ni].y, rs);
          if (p.x3_c > 2) rs = fmaf(x2, w3[ni].z, rs);
          v += rs;
        }
        p.out[orow + n] = v;
        vv[ni][r] = v;
      }
    }
    if (p.pool_out) {
      // Register r holds tile row (r&3) + 8*(r>>2) + 4*half.  A 32-row tile starts on an even picture row, so for
      // W = 16 the window of even column c is registers {r, r+1, r+8, r+9}, for W = 8 it is {r, r+1, r+4, r+5}.
      const int Wo = p.W >> 1, wsh = p.W == 16 ? 4 : 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r16 = (q & 1) * 2 + (q >> 1) * 4, r8 = (q & 1) * 2 + (q >> 1) * 8;
        const int r = p.W == 16 ? r16 : r8;
        const int L = 4 * half + (r & 3) + 8 * (r >> 2);
        const int m = mt + L;
        if (m >= p.M) continue;
        int b, rem;
        if (hw_fast) { const bool over = ir + L >= HW; b = iq + (over ? 1 : 0); rem = ir + L - (over ? HW : 0); }
        else { b = m / HW; rem = m - b * HW; }
        const int y = rem >> wsh, x = rem & (p.W - 1);        // W is 8 or 16 here
        const size_t mo = (((size_t)b * (p.H >> 1) + (y >> 1)) * Wo + (x >> 1)) * p.cout_p;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          if (ncol[ni] < 0) continue;
          const float mx = p.W == 16 ? fmaxf(fmaxf(vv[ni][r16], vv[ni][r16 + 1]), fmaxf(vv[ni][r16 + 8], vv[ni][r16 + 9]))
                                     : fmaxf(fmaxf(vv[ni][r8], vv[ni][r8 + 1]), fmaxf(vv[ni][r8 + 4], vv[ni][r8 + 5]));
          p.pool_out[mo + ncol[ni]] = mx;
        }
      }
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

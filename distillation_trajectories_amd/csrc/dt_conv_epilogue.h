// Shared epilogue of the implicit-GEMM convolution kernels (fp32-MFMA and split-bf16 variants).
// Accumulator layout (both MFMA families, 32x32 tiles): register r of lane l holds
// D[row (r&3) + 8*(r>>2) + 4*(l>>5)][col l&31]; a register is therefore a 128-B run of consecutive
// channels for two pixel rows -> coalesced NHWC stores.
#pragma once
#include "dt_internal.h"

namespace dt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: plain dwordx4 loads, always promoted to VGPRs

// exact three-way bf16 split of one fp32 value: v == a1 + a2 + a3 (round-to-nearest at each step)
__device__ inline void split3(float v, __bf16 &a1, __bf16 &a2, __bf16 &a3) {
  a1 = (__bf16)v;
  const float r1 = v - (float)a1;
  a2 = (__bf16)r1;
  a3 = (__bf16)(r1 - (float)a2);
}

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

// element index of (row m, channel c, plane 0) in a planes tensor with cq = C/16 chunks per row
__device__ inline size_t plane_index(size_t m, int c, int cq) { return ((m * cq + (c >> 4)) * 3) * 16 + (c & 15); }

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

  // ---- epilogue: folded BN, ReLU, time bias, residual; one 128-B channel run per (register, half)
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (NI * 32) + ni * 32 + l31;
    if (n >= p.cout_p) continue;
    const float sc = p.scale[n], sh = p.shift[n];
    float4 w3 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.x3) w3 = *reinterpret_cast<const float4 *>(p.w3 + 4 * n);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int mb = m0 + wm * (MI * 32) + mi * 32 + 4 * half;
      float vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        vv[r] = 0.f;
        if (m >= p.M) continue;
        float v;
        if (p.in2) {
          v = acc[mi][ni][r] + p.bias2[n];       // BN/ReLU were applied in registers before the skip walk
        } else {
          v = acc[mi][ni][r] * sc + sh;
          if (p.relu) v = fmaxf(v, 0.f);
        }
        if (p.tb) v += p.tb[(size_t)(m / p.m_per_tb) * p.tb_stride + n];
        const size_t o = (size_t)m * p.cout_p + n;
        if (p.add) v += p.add[o];
        if (p.x3) {
          const float *xr = p.x3 + (size_t)m * p.x3_stride;
          float rs = w3.w;
          rs = fmaf(xr[0], w3.x, rs);
          if (p.x3_c > 1) rs = fmaf(xr[p.x3_step], w3.y, rs);
          if (p.x3_c > 2) rs = fmaf(xr[2 * p.x3_step], w3.z, rs);
          v += rs;
        }
        p.out[o] = v;
        vv[r] = v;
        if (p.out_pl) {
          __bf16 *pl = reinterpret_cast<__bf16 *>(p.out_pl) + plane_index(m, n, p.cout_p >> 4);
          __bf16 a1, a2, a3;
          split3(v, a1, a2, a3);
          pl[0] = a1; pl[16] = a2; pl[32] = a3;
        }
      }
      if (p.pool_out) {
        // Register r holds tile row (r&3) + 8*(r>>2) + 4*half.  A 32-row tile starts on an even picture row, so for
        // W = 16 the window of even column c is registers {r, r+1, r+8, r+9}, for W = 8 it is {r, r+1, r+4, r+5}.
        const int HW = p.H * p.W, Wo = p.W >> 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r16 = (q & 1) * 2 + (q >> 1) * 4, r8 = (q & 1) * 2 + (q >> 1) * 8;
          const int r = p.W == 16 ? r16 : r8;
          const int m = mb + (r & 3) + 8 * (r >> 2);
          if (m >= p.M) continue;
          const float mx = p.W == 16 ? fmaxf(fmaxf(vv[r16], vv[r16 + 1]), fmaxf(vv[r16 + 8], vv[r16 + 9]))
                                     : fmaxf(fmaxf(vv[r8], vv[r8 + 1]), fmaxf(vv[r8 + 4], vv[r8 + 5]));
          const int b = m / HW, rem = m - b * HW;
          const int y = rem / p.W, x = rem - y * p.W;
          const size_t mo = ((size_t)b * (p.H >> 1) + (y >> 1)) * Wo + (x >> 1);
          p.pool_out[mo * p.cout_p + n] = mx;
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

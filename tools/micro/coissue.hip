// Microbenchmark: what can a second wave on the same SIMD do while the first streams MFMAs?
// Blocks of 8 waves (2 per SIMD): waves 0-3 run the MFMA loop, waves 4-7 run one of: nothing, VALU FMAs, LDS reads,
// global loads (L2-hot).  Reports the MFMA waves' time and the helper waves' time, alone and together.
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool MFMA_ON, bool HELP_ON>
__global__ __launch_bounds__(512) void k(float *out, const float *src, long long *clk, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = i * 0.5f;
  __syncthreads();
  const long long t0 = clock64();
  float s = 0.f;
  if (wave < 4) {
    if (MFMA_ON) {
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.001f + i); b[i] = (__bf16)(1.0f / (1 + i)); }
      f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
      for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    }
  } else if (HELP_ON) {
    if (MODE == 1) {           // VALU: 4 independent FMA chains, 32 FMAs per iteration
      float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 0.9999f, 0.25f); x2 = fmaf(x2, 1.0002f, 0.125f); x3 = fmaf(x3, 0.9998f, 1.f); }
      }
      s = x0 + x1 + x2 + x3;
    } else if (MODE == 2) {    // LDS: 4 ds_read_b128 per iteration
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      int o = lane * 4;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc += *reinterpret_cast<const f32x4 *>(lds + ((o + j * 256) & 8191)); }
        o = (o + 1024) & 8191;
      }
      s = acc[0] + acc[1] + acc[2] + acc[3];
    } else if (MODE == 3) {    // global (cache-hot) dwordx4 loads: 2 per iteration
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const f32x4 *p = reinterpret_cast<const f32x4 *>(src) + lane;
      for (int i = 0; i < iters; ++i) { acc += p[(i & 63) * 64]; acc += p[((i + 32) & 63) * 64]; }
      s = acc[0] + acc[1] + acc[2] + acc[3];
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, bool M, bool H>
void run(const char *name, float *out, float *src, long long *clk, int iters) {
  k<MODE, M, H><<<256, 512>>>(out, src, clk, iters);
  hipDeviceSynchronize();
  k<MODE, M, H><<<256, 512>>>(out, src, clk, iters);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, clk + 8 * 100, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s mfma wave %8lld clk (%.1f per MFMA)   helper wave %8lld clk (%.2f per iteration)\n", name, h[0], h[0] / (4.0 * iters),
         h[4], (double)h[4] / iters);
}

int main() {
  const int iters = 20000;
  float *out, *src; long long *clk;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, 1 << 20); hipMalloc(&clk, 256 * 8 * 8);
  hipMemset(src, 0, 1 << 20);
  run<1, true, false>("MFMA alone", out, src, clk, iters);
  run<1, false, true>("VALU alone (32 FMA/iter)", out, src, clk, iters);
  run<1, true, true>("MFMA + VALU", out, src, clk, iters);
  run<2, false, true>("LDS alone (4 ds_read_b128/iter)", out, src, clk, iters);
  run<2, true, true>("MFMA + LDS", out, src, clk, iters);
  run<3, false, true>("global alone (2 dwordx4/iter)", out, src, clk, iters);
  run<3, true, true>("MFMA + global", out, src, clk, iters);
  return 0;
}

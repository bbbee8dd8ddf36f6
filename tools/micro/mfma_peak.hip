// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 rate and shader clock on every CU (no memory traffic).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, long long *clk, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f / (1 + i)); }
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  const int iters = 20000;
  for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
    const int blocks = 256 * waves_per_simd;   // 4 waves per block -> one per SIMD
    float *out; long long *clk;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&clk, blocks * 2 * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_loop<<<blocks, 256>>>(out, clk, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_loop<<<blocks, 256>>>(out, clk, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    printf("waves/SIMD %d: %.3f ms, %.0f TF/s bf16 dense; block 0: %lld shader clocks in %lld x 10 ns -> %.0f MHz; %.1f clocks per MFMA per SIMD\n",
           waves_per_simd, ms, flops / ms / 1e9, h[0], h[1], h[0] / (h[1] * 0.01), (double)h[0] / (iters * 4.0 * waves_per_simd));
    hipFree(out); hipFree(clk);
  }
  return 0;
}

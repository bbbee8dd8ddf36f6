// Microbenchmark: what does ONE wave per SIMD pay per v_mfma_f32_32x32x16_bf16 in the shape of the strip kernel's
// tap (24 MFMAs = 4 accumulators x 6 plane products over 6 + 6 fragment registers) when other work is mixed in?
//   0: the 24 MFMAs alone        1: + 12 ds_read_b128 into a second fragment set (conflict-free)
//   2: + 40 integer VALU         3: + one __syncthreads() per block     4: 24 MFMAs on ONE accumulator
//   5: 2 accumulators alternating 6: as 1 but the MFMAs consume the set read in the previous block (true pipeline)
//   7: as 6 + VALU + barrier + 3 global loads and 3 ds_write_b128 (the whole tap skeleton)
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/mfma_block mfma_block.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, const u32x4 *src, long long *clk, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[24 * 1024];      // 48 KB
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24 * 1024 / 8; i += 256) {
    bf16x8 v;
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(((i * 8 + j) * 2654435761u >> 20) * (1.0f / 4096) - 0.5f);
    *reinterpret_cast<bf16x8 *>(lds + i * 8) = v;
  }
  __syncthreads();
  bf16x8 fa[2][2][3], fb[2][2][3];
  const int l31 = lane & 31, half = lane >> 5;
  const int base = (l31 * 16 + ((half ^ ((l31 >> 3) & 1)) << 3));
  for (int s = 0; s < 2; ++s)
    for (int m = 0; m < 2; ++m)
      for (int pl = 0; pl < 3; ++pl) {
        fa[s][m][pl] = *reinterpret_cast<const bf16x8 *>(lds + base + (s * 6 + m * 3 + pl) * 512);
        fb[s][m][pl] = *reinterpret_cast<const bf16x8 *>(lds + base + (12 + s * 6 + m * 3 + pl) * 512);
      }
  f32x16 acc[2][2] = {};
  int addr = base, junk = tid;
  u32x4 rb[3] = {};
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool F = MODE >= 16;
    constexpr bool F_VALU = MODE == 7 || (F && (MODE & 1)), F_BAR = MODE == 7 || (F && (MODE & 2)), F_LD = MODE == 7 || (F && (MODE & (4 | 16)));
    constexpr bool F_WC = F && (MODE & 8), F_WL = MODE == 7 || (F && (MODE & 16));
    if (F_LD) {
      for (int pl = 0; pl < 3; ++pl) rb[pl] = src[(size_t)(it & 63) * 768 + pl * 256 + tid];
    }
    constexpr bool PIPE = MODE == 6 || MODE == 7 || F;
    const int cur = 0, nxt = PIPE ? 1 : 0;
    if (MODE == 1 || MODE == 2 || MODE == 3 || PIPE) {
      // 12 reads into set `nxt` (for PIPE the roles swap every iteration through the register rotation below)
      const int a2 = addr + ((it & 7) << 9);
      for (int m = 0; m < 2; ++m)
        for (int pl = 0; pl < 3; ++pl) {
          fa[1][m][pl] = *reinterpret_cast<const bf16x8 *>(lds + a2 + (m * 3 + pl) * 512);
          fb[1][m][pl] = *reinterpret_cast<const bf16x8 *>(lds + a2 + (6 + m * 3 + pl) * 512);
        }
    }
    if (MODE == 2 || MODE == 3 || F_VALU) {
#pragma unroll
      for (int v = 0; v < 40; ++v) junk = (junk ^ (junk >> 3)) + v;
      asm volatile("" : "+v"(junk));
    }
    if (MODE == 4) {
      f32x16 c = acc[0][0];
      for (int q = 0; q < 24; ++q) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][q & 1][q % 3], fb[0][(q >> 1) & 1][q % 3], c, 0, 0, 0);
      acc[0][0] = c;
    } else if (MODE == 5) {
      f32x16 c = acc[0][0], d = acc[0][1];
      for (int q = 0; q < 12; ++q) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][q & 1][q % 3], fb[0][(q >> 1) & 1][q % 3], c, 0, 0, 0);
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][(q + 1) & 1][q % 3], fb[0][(q >> 1) & 1][(q + 1) % 3], d, 0, 0, 0);
      }
      acc[0][0] = c; acc[0][1] = d;
    } else {
      for (int mi = 0; mi < 2; ++mi)
        for (int ni = 0; ni < 2; ++ni) {
          f32x16 c = acc[mi][ni];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi][2], fb[cur][ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi][1], fb[cur][ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi][0], fb[cur][ni][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi][1], fb[cur][ni][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi][0], fb[cur][ni][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi][0], fb[cur][ni][0], c, 0, 0, 0);
          acc[mi][ni] = c;
        }
    }
    if (PIPE) {                      // the set just read becomes the next block's operand set (register rotation, no copy
      for (int m = 0; m < 2; ++m)    // in the unrolled-by-2 loop the compiler builds from this swap)
        for (int pl = 0; pl < 3; ++pl) {
          const bf16x8 t = fa[0][m][pl]; fa[0][m][pl] = fa[1][m][pl]; fa[1][m][pl] = t;
          const bf16x8 u = fb[0][m][pl]; fb[0][m][pl] = fb[1][m][pl]; fb[1][m][pl] = u;
        }
    } else if (MODE == 1 || MODE == 2 || MODE == 3) {
      asm volatile("" :: "v"(fa[1][0][0]), "v"(fa[1][1][2]), "v"(fb[1][0][1]), "v"(fb[1][1][2]));
    }
    if (F_WL) {
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(lds + 16 * 1024 + pl * 2048 + tid * 8) = rb[pl];
    } else if (F_LD) {
      asm volatile("" :: "v"(rb[0]), "v"(rb[1]), "v"(rb[2]));
    }
    if (F_WC) {
      const u32x4 cst = {(unsigned)tid, 1u, 2u, 3u};
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(lds + 16 * 1024 + pl * 2048 + tid * 8) = cst;
    }
    if (MODE == 3 || F_BAR) __syncthreads();
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = junk * 1e-30f;
  for (int r = 0; r < 16; ++r) s += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char *name, float *out, const u32x4 *src, long long *clk, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, src, clk, 2000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, src, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s blocks/CU %d: %7.3f ms = %6.0f ns per block of 24 (launch time) | clock64: %5.0f per block, %.0f MHz | %5.0f TF/s bf16\n", name, blocks / 256,
         ms, ms * 1e6 / iters / (blocks / 256), (double)h[0] / iters, h[0] / (h[1] * 0.01), (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16 / ms / 1e9);
}

typedef float f32x4v __attribute__((ext_vector_type(4)));
// MODE16: 0 bare 48 x 16x16x32 | 1 + 20 ds_read_b128 (K-concatenated plane pairs: 8 A + 12 B fragments) pipelined | 2 + 12 reads
template <int MODE16>
__global__ __launch_bounds__(256, 2) void k16(float *out, long long *clk, int iters) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[24 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 24 * 1024 / 8; i += 256) {
    bf16x8 v;
    for (int j = 0; j < 8; ++j) v[j] = (__bf16)(((i * 8 + j) * 2654435761u >> 20) * (1.0f / 4096) - 0.5f);
    *reinterpret_cast<bf16x8 *>(lds + i * 8) = v;
  }
  __syncthreads();
  constexpr int NA = 8, NB = 12;
  bf16x8 fa[2][NA], fb[2][NB];
  const int r15 = lane & 15, g = lane >> 4;
  const int base = r15 * 16 + (((g & 1) ^ ((r15 >> 3) & 1)) << 3) + (g >> 1) * 4096;
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < NA; ++i) fa[s][i] = *reinterpret_cast<const bf16x8 *>(lds + base + (s * NA + i) * 256);
    for (int i = 0; i < NB; ++i) fb[s][i] = *reinterpret_cast<const bf16x8 *>(lds + base + 8192 + (s * NB + i) * 256);
  }
  f32x4v acc[4][4] = {};
  const long long t0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    __builtin_amdgcn_sched_barrier(0);
    if (MODE16 >= 1) {
      const int a2 = base + ((it & 7) << 8);
      constexpr int RA = MODE16 == 1 ? NA : 4, RB = MODE16 == 1 ? NB : 8;
      for (int i = 0; i < RA; ++i) fa[1][i] = *reinterpret_cast<const bf16x8 *>(lds + a2 + i * 256);
      for (int i = 0; i < RB; ++i) fb[1][i] = *reinterpret_cast<const bf16x8 *>(lds + a2 + 8192 + i * 256);
    }
    for (int mi = 0; mi < 4; ++mi)
      for (int ni = 0; ni < 4; ++ni) {
        f32x4v c = acc[mi][ni];
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][2 * mi + 1], fb[0][3 * ni + 1], c, 0, 0, 0);   // [a1|a3].[b3|b1]
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][2 * mi], fb[0][3 * ni], c, 0, 0, 0);           // [a1|a2].[b2|b1]
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0][2 * mi], fb[0][3 * ni + 2], c, 0, 0, 0);       // [a1|a2].[b1|b2]
        acc[mi][ni] = c;
      }
    if (MODE16 >= 1) {
      for (int i = 0; i < NA; ++i) { const bf16x8 t = fa[0][i]; fa[0][i] = fa[1][i]; fa[1][i] = t; }
      for (int i = 0; i < NB; ++i) { const bf16x8 t = fb[0][i]; fb[0][i] = fb[1][i]; fb[1][i] = t; }
    }
  }
  const long long t1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 4; ++ni) for (int r = 0; r < 4; ++r) s += acc[mi][ni][r];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE16>
void run16(const char *name, float *out, long long *clk, int blocks) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k16<MODE16><<<blocks, 256>>>(out, clk, 2000);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k16<MODE16><<<blocks, 256>>>(out, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-58s blocks/CU %d: %7.3f ms = %6.0f ns per block of 48 (launch time) | clock64: %5.0f per block, %.0f MHz | %5.0f TF/s bf16\n", name, blocks / 256,
         ms, ms * 1e6 / iters / (blocks / 256), (double)h[0] / iters, h[0] / (h[1] * 0.01), (double)blocks * 4 * iters * 48 * 2.0 * 16 * 16 * 32 / ms / 1e9);
}

int main() {
  float *out; long long *clk; u32x4 *src;
  hipMalloc(&out, 1024 * 256 * sizeof(float));
  hipMalloc(&clk, 1024 * 2 * sizeof(long long));
  hipMalloc(&src, 64 * 768 * sizeof(u32x4));
  hipMemset(src, 0x3c, 64 * 768 * sizeof(u32x4));
  for (int blocks = 256; blocks <= 512; blocks += 256) {
    run16<0>("16x16x32: 48 MFMA, 16 accumulators", out, clk, blocks);
    run16<1>("16x16x32: + 20 ds_read_b128, pipelined", out, clk, blocks);
    run16<2>("16x16x32: + 12 ds_read_b128, pipelined", out, clk, blocks);
    run<0>("0: 24 MFMA, 4 accumulators", out, src, clk, blocks);
    run<6>("6: 32x32x16 pipelined, 12 ds_read_b128", out, src, clk, blocks);
    run16<0>("16x16x32: 48 MFMA, 16 accumulators (again)", out, clk, blocks);
    run<0>("0: 24 MFMA, 4 accumulators (again)", out, src, clk, blocks);
    run<4>("4: 24 MFMA, 1 accumulator", out, src, clk, blocks);
    run<5>("5: 24 MFMA, 2 accumulators", out, src, clk, blocks);
    run<1>("1: + 12 ds_read_b128 (unused)", out, src, clk, blocks);
    run<2>("2: + 12 ds_read_b128 + 40 VALU", out, src, clk, blocks);
    run<3>("3: + reads + VALU + barrier", out, src, clk, blocks);
    run<6>("6: pipelined: MFMAs use the set read one block earlier", out, src, clk, blocks);
    run<7>("7: pipelined + VALU + 3 global loads + 3 ds_write + barrier", out, src, clk, blocks);
    run<16 + 1>("6 + VALU", out, src, clk, blocks);
    run<16 + 2>("6 + barrier", out, src, clk, blocks);
    run<16 + 4>("6 + 3 global loads (waited at the end of the block)", out, src, clk, blocks);
    run<16 + 8>("6 + 3 ds_write_b128 of constants", out, src, clk, blocks);
    run<16 + 8 + 2>("6 + 3 ds_write_b128 of constants + barrier", out, src, clk, blocks);
    run<16 + 16>("6 + 3 global loads written to LDS", out, src, clk, blocks);
    run<16 + 16 + 2>("6 + 3 global loads written to LDS + barrier", out, src, clk, blocks);
    run<16 + 16 + 2 + 1>("6 + loads->LDS + barrier + VALU (= 7)", out, src, clk, blocks);
  }
  return 0;
}

// Trajectory-metric reductions over device-resident trajectories [n][B][E] (HBM-bound).
//
// Reference: analysis/metrics/trajectory_metrics.py:55-231 issues ~6 torch.norm(...).item() calls per
// step (each a device sync when the tensors live on a GPU) and 51 scipy sorts per pair.  Here every
// (pair, step) is one workgroup that streams X_i, X_{i-1}, Y_i, Y_{i-1} once with float4 loads and
// produces the four sums the 25 metrics are built from; partial sums are folded with __shfl_down over
// the 64-lane wavefront and then across the 4 waves through LDS.  Sums are carried in float64 so the
// result is within one fp32 rounding of the exact value (torch's own fp32 reduction is not closer).
#include "dt_internal.h"

namespace dt {

__device__ inline double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// block-wide sum of K doubles per thread; result valid in thread 0
template <int K>
__device__ inline void block_sum(double (&v)[K], double *sm /* [4*K] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    v[k] = wave_sum(v[k]);
    if (lane == 0) sm[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = sm[k] + sm[K + k] + sm[2 * K + k] + sm[3 * K + k];
  }
}

__global__ __launch_bounds__(256) void traj_metrics_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                           int nT, int nS, int B, int E, double *__restrict__ out) {
  __shared__ double sm[16];
  const int b = blockIdx.x, i = blockIdx.y;
  const int n_max = nT > nS ? nT : nS;
  const size_t step = (size_t)B * E;
  const float4 *xi = i < nT ? reinterpret_cast<const float4 *>(X + i * step + (size_t)b * E) : nullptr;
  const float4 *yi = i < nS ? reinterpret_cast<const float4 *>(Y + i * step + (size_t)b * E) : nullptr;
  // i >= 1: previous states; i == 0: the trajectory's own last state (endpoint-to-start distance)
  const int jx = i >= 1 ? i - 1 : nT - 1, jy = i >= 1 ? i - 1 : nS - 1;
  const float4 *xj = reinterpret_cast<const float4 *>(X + jx * step + (size_t)b * E);
  const float4 *yj = reinterpret_cast<const float4 *>(Y + jy * step + (size_t)b * E);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int e = threadIdx.x; e < E / 4; e += 256) {
    float4 a = make_float4(0, 0, 0, 0), c = a, pa = a, pc = a;
    if (xi) { a = xi[e]; pa = xj[e]; }
    if (yi) { c = yi[e]; pc = yj[e]; }
    const float ax[4] = {a.x, a.y, a.z, a.w}, cx[4] = {c.x, c.y, c.z, c.w};
    const float px[4] = {pa.x, pa.y, pa.z, pa.w}, qx[4] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // differences in fp32 exactly as torch forms (X_i - Y_i) before the norm
      const float d = ax[k] - cx[k], dx = ax[k] - px[k], dy = cx[k] - qx[k];
      if (xi && yi) acc[0] += (double)d * (double)d;
      if (xi) acc[1] += (double)dx * (double)dx;
      if (yi) acc[2] += (double)dy * (double)dy;
      if (xi && yi) {
        // i == 0 carries the endpoint term |X_last - Y_last|^2 (each trajectory's own last state)
        const float de = px[k] - qx[k];
        acc[3] += i >= 1 ? (double)dx * (double)dy : (double)de * (double)de;
      }
    }
  }
  block_sum<4>(acc, sm);
  if (threadIdx.x == 0) {
    double *o = out + ((size_t)b * n_max + i) * 4;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
  }
}

int launch_traj_metrics(const float *X, const float *Y, int nT, int nS, int B, int E, double *out, hipStream_t s) {
  if (!X || !Y || !out) return DT_E_NULL;
  if (nT < 1 || nS < 1 || B < 1 || E < 4 || E % 4) return DT_E_SHAPE;
  const int n_max = nT > nS ? nT : nS;
  if (n_max > 65535) return DT_E_SHAPE;
  // algorithmic bytes: one read of both trajectories (2*(T+1)*E*4 per pair)
  ProfileScope prof(KC_METRICS, 0.0, 4.0 * B * E * ((double)nT + nS), s);
  traj_metrics_kernel<<<dim3(B, n_max), 256, 0, s>>>(X, Y, nT, nS, B, E, out);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---------------------------------------------------------------------------------------------
// Same-index statistics of two state sequences (evaluation/metrics.py:118-183 compute_trajectory_divergence,
// analysis/noise_prediction/noise_analysis.py:43-85 calculate_noise_metrics):
// out[b][i] = { sum (x-y)^2, sum |x-y|, sum x*y, sum x^2, sum y^2 } over the E coordinates of state i of pair b.
__global__ __launch_bounds__(256) void pair_stats_kernel(const float *__restrict__ X, const float *__restrict__ Y, int B,
                                                         int E, double *__restrict__ out, int n) {
  __shared__ double sm[20];
  const int b = blockIdx.x, i = blockIdx.y;
  const float4 *x = reinterpret_cast<const float4 *>(X + ((size_t)i * B + b) * E);
  const float4 *y = reinterpret_cast<const float4 *>(Y + ((size_t)i * B + b) * E);
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int e = threadIdx.x; e < E / 4; e += 256) {
    const float4 a = x[e], c = y[e];
    const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = av[k] - cv[k];          // fp32 difference, as torch forms it
      acc[0] += (double)d * (double)d;
      acc[1] += fabs((double)d);
      acc[2] += (double)av[k] * (double)cv[k];
      acc[3] += (double)av[k] * (double)av[k];
      acc[4] += (double)cv[k] * (double)cv[k];
    }
  }
  block_sum<5>(acc, sm);
  if (threadIdx.x == 0) {
    double *o = out + ((size_t)b * n + i) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = acc[k];
  }
}

int launch_pair_stats(const float *X, const float *Y, int n, int B, int E, double *out, hipStream_t s) {
  if (!X || !Y || !out) return DT_E_NULL;
  if (n < 1 || B < 1 || E < 4 || E % 4 || n > 65535) return DT_E_SHAPE;
  ProfileScope prof(KC_METRICS, 0.0, 8.0 * B * E * (double)n, s);
  pair_stats_kernel<<<dim3(B, n), 256, 0, s>>>(X, Y, B, E, out, n);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// per-timestep mean over the B samples of a trajectory tensor [n][B][E] -> [n][E]
// (scripts/analysis/analyze_trajectories.py:467-486 averages sample trajectories before the PCA plots)
__global__ __launch_bounds__(256) void sample_mean_kernel(const float *__restrict__ traj, int B, int E,
                                                          float *__restrict__ out, size_t total) {
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t i = idx / E, e = idx - i * E;
    const float *p = traj + i * (size_t)B * E + e;
    double acc = 0.0;
    for (int b = 0; b < B; ++b) acc += (double)p[(size_t)b * E];
    out[idx] = (float)(acc / (double)B);
  }
}

int launch_sample_mean(const float *traj, int n, int B, int E, float *out, hipStream_t s) {
  if (!traj || !out) return DT_E_NULL;
  if (n < 1 || B < 1 || E < 1) return DT_E_SHAPE;
  const size_t total = (size_t)n * E;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  ProfileScope prof(KC_METRICS, 0.0, 4.0 * total * (B + 1.0), s);
  sample_mean_kernel<<<blocks, 256, 0, s>>>(traj, B, E, out, total);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---------------------------------------------------------------------------------------------
// W1 between the empirical distributions of the coordinates of X_i and Y_i (equal sample sizes):
// mean_k |u_(k) - v_(k)| over the order statistics.  Both samples are bitonic-sorted in LDS
// (padded to a power of two with +inf, which sorts to the tail of BOTH arrays and cancels).
template <int N>
__device__ inline void bitonic_sort(float *a) {
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < N / 2; t += blockDim.x) {
        const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
        const bool up = (lo & k) == 0;
        const float x = a[lo], y = a[hi];
        if ((x > y) == up) { a[lo] = y; a[hi] = x; }
      }
      __syncthreads();
    }
  }
}

template <int N>
__global__ __launch_bounds__(256) void wasserstein_kernel(const float *__restrict__ X, const float *__restrict__ Y,
                                                          int B, int E, const int32_t *__restrict__ index,
                                                          const int32_t *__restrict__ index_row, int n_idx,
                                                          double *__restrict__ out, int n) {
  __shared__ float u[N], v[N];
  __shared__ double sm[4];
  const int b = blockIdx.x, i = blockIdx.y;
  const float *x = X + ((size_t)i * B + b) * E, *y = Y + ((size_t)i * B + b) * E;
  const int cnt = index ? n_idx : E;
  const int32_t *idx = index ? index + ((size_t)(index_row ? index_row[b] : 0) * n + i) * n_idx : nullptr;
  for (int k = threadIdx.x; k < N; k += blockDim.x) {
    float a = __builtin_inff(), c = __builtin_inff();
    if (k < cnt) { const int e = idx ? idx[k] : k; a = x[e]; c = y[e]; }
    u[k] = a; v[k] = c;
  }
  __syncthreads();
  bitonic_sort<N>(u);
  bitonic_sort<N>(v);
  double acc[1] = {0.0};
  for (int k = threadIdx.x; k < cnt; k += blockDim.x) acc[0] += fabs((double)u[k] - (double)v[k]);
  block_sum<1>(acc, sm);
  if (threadIdx.x == 0) out[(size_t)b * n + i] = acc[0] / (double)cnt;
}

int launch_wasserstein(const float *X, const float *Y, int n, int B, int E, const int32_t *index,
                       const int32_t *index_row, int n_idx, double *out, hipStream_t s) {
  if (!X || !Y || !out) return DT_E_NULL;
  const int cnt = index ? n_idx : E;
  if (n < 1 || B < 1 || cnt < 1 || cnt > 4096 || n > 65535) return DT_E_SHAPE;
  dim3 grid(B, n);
  ProfileScope prof(KC_WASSERSTEIN, 0.0, 8.0 * B * n * (double)cnt, s);
  if (cnt <= 1024) wasserstein_kernel<1024><<<grid, 256, 0, s>>>(X, Y, B, E, index, index_row, n_idx, out, n);
  else if (cnt <= 2048) wasserstein_kernel<2048><<<grid, 256, 0, s>>>(X, Y, B, E, index, index_row, n_idx, out, n);
  else wasserstein_kernel<4096><<<grid, 256, 0, s>>>(X, Y, B, E, index, index_row, n_idx, out, n);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---------------------------------------------------------------------------------------------
// One pass over a pair's states for BOTH reductions above (equal lengths, all E <= 4096 coordinates sampled: every
// 16x16 and 32x32 configuration whose Wasserstein term uses all coordinates).  A workgroup owns (pair b, step i): thread t
// holds coordinates t R .. t R + R - 1 of X_i and Y_i in registers (one or more 16-byte loads), forms the four sums
// against X_{i-1}, Y_{i-1} (the same workgroup order makes that second read an L2 hit: consecutive workgroups are
// consecutive steps of one pair, so a state comes from HBM once) and then sorts its two register arrays IN PLACE with a
// bitonic network whose exchanges are, by distance j between partners:
//   j < R        the thread's own registers,
//   j < 64 R     a lane of the same wavefront: __shfl_xor (64-lane cross-lane network, no LDS storage, no barrier),
//   j >= 64 R    another wavefront: through LDS (three of the 55 stages at N = 1024).
// The LDS bitonic sort of wasserstein_kernel runs all 55 stages (110 per pair-step) through LDS behind a barrier each:
// 479 us per 256 x 51 pair-steps against 35 us for the sums; this kernel does both in one launch.
// W1 = mean_k |u_(k) - v_(k)|: both arrays end in the same (ascending) layout, so the sum is thread-local.
template <int R>
__global__ __launch_bounds__(256) void pair_metrics_kernel(const float *__restrict__ X, const float *__restrict__ Y, int n, int B,
                                                           int E, double *__restrict__ out_sums, double *__restrict__ out_w1) {
  constexpr int N = 256 * R;
  __shared__ unsigned ex[2][N];
  __shared__ double sm[20];
  const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const size_t step = (size_t)B * E;
  const int j0 = i >= 1 ? i - 1 : n - 1;               // i == 0: the trajectory's own last state (endpoint terms)
  const float *x = X + i * step + (size_t)b * E, *y = Y + i * step + (size_t)b * E;
  const float *px = X + j0 * step + (size_t)b * E, *py = Y + j0 * step + (size_t)b * E;
  // The sort runs on order-preserving integer keys (sign-magnitude float bits -> unsigned: negative values are complemented,
  // positive ones get the top bit): an exchange is then v_min_u32 + v_max_u32 + one select, three VALU instructions per
  // element, against six for a float compare-and-swap that has to keep NaNs in place -- the kernel is VALU-bound (52 exchange
  // stages x 8 elements per thread).  A NaN coordinate makes the step's term NaN, as in wasserstein_kernel / the reference:
  // it is flagged while loading.
  auto to_key = [](float f) __attribute__((always_inline)) {
    const unsigned bits = __float_as_uint(f);
    return bits ^ ((unsigned)((int)bits >> 31) | 0x80000000u);
  };
  auto from_key = [](unsigned k) __attribute__((always_inline)) { return __uint_as_float(k ^ (((k >> 31) - 1u) | 0x80000000u)); };
  unsigned u[R], v[R];
  double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  bool has_nan = false;
#pragma unroll
  for (int q = 0; q < R / 4; ++q) {
    const int e = tid * R + q * 4;
    float4 a = make_float4(__builtin_inff(), __builtin_inff(), __builtin_inff(), __builtin_inff()), c = a;
    if (e < E) {                                        // E % 4 == 0: a quad is inside or outside as a whole
      a = *reinterpret_cast<const float4 *>(x + e); c = *reinterpret_cast<const float4 *>(y + e);
      const float4 pa = *reinterpret_cast<const float4 *>(px + e), pc = *reinterpret_cast<const float4 *>(py + e);
      const float av[4] = {a.x, a.y, a.z, a.w}, cv[4] = {c.x, c.y, c.z, c.w};
      const float pv[4] = {pa.x, pa.y, pa.z, pa.w}, qv[4] = {pc.x, pc.y, pc.z, pc.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // differences in fp32 exactly as torch forms (X_i - Y_i) before the norm; float64 accumulation (traj_metrics_kernel)
        const float d = av[k] - cv[k], dx = av[k] - pv[k], dy = cv[k] - qv[k], de = pv[k] - qv[k];
        acc[0] += (double)d * (double)d;
        acc[1] += (double)dx * (double)dx;
        acc[2] += (double)dy * (double)dy;
        acc[3] += i >= 1 ? (double)dx * (double)dy : (double)de * (double)de;
        has_nan |= (av[k] != av[k]) || (cv[k] != cv[k]);
      }
    }
    u[q * 4] = to_key(a.x); u[q * 4 + 1] = to_key(a.y); u[q * 4 + 2] = to_key(a.z); u[q * 4 + 3] = to_key(a.w);
    v[q * 4] = to_key(c.x); v[q * 4 + 1] = to_key(c.y); v[q * 4 + 2] = to_key(c.z); v[q * 4 + 3] = to_key(c.w);
  }
  // ---- bitonic sort of element e = tid R + r, ascending; +inf padding sorts to the tail of both arrays
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j >= R; j >>= 1) {             // partners in other threads
      // this thread keeps the smaller key of each pair when it holds the lower index of an ascending pair (or the upper of a
      // descending one); for k >= R the direction bit of an element does not depend on r
      if (j < 64 * R) {
        const int d = j / R;                            // lane distance
        const bool lower = (lane & d) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool keep_min = (((tid * R + r) & k) == 0) == lower;
          const unsigned uo = __shfl_xor(u[r], d, 64), vo = __shfl_xor(v[r], d, 64);
          u[r] = keep_min ? min(u[r], uo) : max(u[r], uo);
          v[r] = keep_min ? min(v[r], vo) : max(v[r], vo);
        }
      } else {
        const int dt = j / R;                           // thread distance (a multiple of 64)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r) { ex[0][r * 256 + tid] = u[r]; ex[1][r * 256 + tid] = v[r]; }
        __syncthreads();
        const bool lower = (tid & dt) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const bool keep_min = (((tid * R + r) & k) == 0) == lower;
          const unsigned uo = ex[0][r * 256 + (tid ^ dt)], vo = ex[1][r * 256 + (tid ^ dt)];
          u[r] = keep_min ? min(u[r], uo) : max(u[r], uo);
          v[r] = keep_min ? min(v[r], vo) : max(v[r], vo);
        }
      }
    }
    // partners inside the thread: j = R/2 .. 1 as COMPILE-TIME constants (a run-time j would index the register arrays
    // dynamically, i.e. move them to scratch memory)
#pragma unroll
    for (int j = R / 2; j > 0; j >>= 1) {
      if (j < k) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if ((r & j) == 0) {
            const bool up = ((tid * R + r) & k) == 0;
            const unsigned ul = min(u[r], u[r | j]), uh = max(u[r], u[r | j]), vl = min(v[r], v[r | j]), vh = max(v[r], v[r | j]);
            u[r] = up ? ul : uh; u[r | j] = up ? uh : ul;
            v[r] = up ? vl : vh; v[r | j] = up ? vh : vl;
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (tid * R + r < E) acc[4] += fabs((double)from_key(u[r]) - (double)from_key(v[r]));
  if (has_nan) acc[4] = __builtin_nan("");
  block_sum<5>(acc, sm);
  if (tid == 0) {
    double *o = out_sums + ((size_t)b * n + i) * 4;
    o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
    out_w1[(size_t)b * n + i] = acc[4] / (double)E;
  }
}

int launch_pair_metrics(const float *X, const float *Y, int n, int B, int E, double *out_sums, double *out_w1, hipStream_t s) {
  if (!X || !Y || !out_sums || !out_w1) return DT_E_NULL;
  if (n < 1 || B < 1 || E < 4 || E % 4 || E > 4096 || n > 65535 || B > 65535) return DT_E_SHAPE;
  // algorithmic bytes: one read of both trajectories (2 (T+1) E 4 per pair)
  ProfileScope prof(KC_PAIR_METRICS, 0.0, 8.0 * B * E * (double)n, s);
  const dim3 grid(n, B);                                // x = step: consecutive workgroups share a state through L2
  if (E <= 1024) pair_metrics_kernel<4><<<grid, 256, 0, s>>>(X, Y, n, B, E, out_sums, out_w1);
  else if (E <= 2048) pair_metrics_kernel<8><<<grid, 256, 0, s>>>(X, Y, n, B, E, out_sums, out_w1);
  else pair_metrics_kernel<16><<<grid, 256, 0, s>>>(X, Y, n, B, E, out_sums, out_w1);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// ---------------------------------------------------------------------------------------------
// trajectory_metrics.py:239-279: the longer trajectory L (n_long states) is resampled by linear
// interpolation (scipy interp1d, float64) onto the shorter's grid linspace(0,1,n_short); output is
// |L'(t_i) - S_i|_2 per (pair, i) in float64, as numpy computes it on the float64 resampled points.
__device__ inline double linspace01(int j, int n) {
  if (n <= 1) return 0.0;
  return j == n - 1 ? 1.0 : (double)j * (1.0 / (double)(n - 1));
}

__global__ __launch_bounds__(256) void resampled_distance_kernel(const float *__restrict__ L, const float *__restrict__ S,
                                                                 int n_long, int n_short, int B, int E,
                                                                 double *__restrict__ out) {
  __shared__ double sm[4];
  const int b = blockIdx.x, i = blockIdx.y;
  // grids as numpy.linspace(0, 1, n) builds them: j*step with the last point forced to exactly 1
  const double x_new = linspace01(i, n_short);
  // interp1d: hi = searchsorted(x, x_new) clipped to [1, n-1]; lo = hi - 1
  int hi = 1;
  while (hi < n_long - 1 && linspace01(hi, n_long) < x_new) ++hi;
  const int lo = hi - 1;
  const double x_lo = linspace01(lo, n_long), x_hi = linspace01(hi, n_long);
  const float *l0 = L + ((size_t)lo * B + b) * E, *l1 = L + ((size_t)hi * B + b) * E;
  const float *sv = S + ((size_t)i * B + b) * E;
  double acc[1] = {0.0};
  for (int e = threadIdx.x; e < E; e += 256) {
    const double y_lo = (double)l0[e], y_hi = (double)l1[e];
    const double slope = (y_hi - y_lo) / (x_hi - x_lo);
    const double yn = slope * (x_new - x_lo) + y_lo;
    const double d = yn - (double)sv[e];
    acc[0] += d * d;
  }
  block_sum<1>(acc, sm);
  if (threadIdx.x == 0) out[(size_t)b * n_short + i] = sqrt(acc[0]);
}

int launch_resampled_distance(const float *L, const float *S, int n_long, int n_short, int B, int E, double *out,
                              hipStream_t s) {
  if (!L || !S || !out) return DT_E_NULL;
  if (n_long < 2 || n_short < 1 || n_short > n_long || B < 1 || E < 1 || n_short > 65535) return DT_E_SHAPE;
  ProfileScope prof(KC_RESAMPLE, 0.0, 12.0 * B * E * (double)n_short, s);
  resampled_distance_kernel<<<dim3(B, n_short), 256, 0, s>>>(L, S, n_long, n_short, B, E, out);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

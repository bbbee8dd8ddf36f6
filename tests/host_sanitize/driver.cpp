// Host-side sanitizer driver ("ASan host build"): links the library's own translation units compiled with
// -Xarch_host -fsanitize=address,undefined (device code is NOT instrumented: GPU ASan is unavailable on this pool) and walks the
// host code of dt_unet.hip through the C ABI -- weight packing, launch plans (heuristic, pinned, autotuned), the shape registry,
// forward / mixed forward, the sampler loops (plain and hipGraph replay), the in-library profiler, the metric launchers and the
// argument-error paths.  Exit status 0 and "driver ok" on stdout mean no sanitizer report and no unexpected status.
// Built by distillation_trajectories_amd/csrc/build.py (build_sanitizer_driver); run by tests/test_host_sanitize.py on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../include/dt_hip.h"

#define CHECK(expr, want)                                                                      \
  do {                                                                                         \
    const int _st = (expr);                                                                    \
    if (_st != (want)) { fprintf(stderr, "%s:%d: %s -> %d (%s), wanted %d\n", __FILE__, __LINE__, #expr, _st, dt_status_string(_st), (want)); return 1; } \
  } while (0)
#define HIP(expr)                                                                              \
  do {                                                                                         \
    const hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) { fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); return 1; } \
  } while (0)

static std::mt19937 rng(1234);
static std::vector<void *> g_allocs;

static float *dev_random(size_t n, float lo, float hi) {
  std::vector<float> h(n);
  std::uniform_real_distribution<float> d(lo, hi);
  for (float &v : h) v = d(rng);
  float *p = nullptr;
  if (hipMalloc((void **)&p, n * sizeof(float)) != hipSuccess) return nullptr;
  (void)hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  g_allocs.push_back(p);
  return p;
}

int main() {
  int n_dev = 0;
  HIP(hipGetDeviceCount(&n_dev));
  if (n_dev < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
  const int C = 3, D = 24, dims[4] = {24, 40, 40, 40};     // odd chunk counts (24 -> 32 / 2 chunks, 40 -> 48 / 3 chunks) on purpose
  const int cin[8] = {C, dims[0], dims[1], dims[2], dims[3], 2 * dims[3], 2 * dims[2], 2 * dims[1]};
  const int cout[8] = {dims[0], dims[1], dims[2], dims[3], dims[3], dims[2], dims[1], dims[0]};
  std::vector<const float *> bt(8 * DT_BT_COUNT, nullptr), gt(DT_GT_COUNT, nullptr);
  for (int j = 0; j < 8; ++j) {
    const float **t = bt.data() + j * DT_BT_COUNT;
    t[DT_BT_TIME_W] = dev_random((size_t)cout[j] * D, -0.2f, 0.2f); t[DT_BT_TIME_B] = dev_random(cout[j], -0.1f, 0.1f);
    t[DT_BT_CONV1_W] = dev_random((size_t)cout[j] * cin[j] * 9, -0.1f, 0.1f); t[DT_BT_CONV1_B] = dev_random(cout[j], -0.1f, 0.1f);
    t[DT_BT_BN1_G] = dev_random(cout[j], 0.8f, 1.2f); t[DT_BT_BN1_B] = dev_random(cout[j], -0.1f, 0.1f);
    t[DT_BT_BN1_MEAN] = dev_random(cout[j], -0.1f, 0.1f); t[DT_BT_BN1_VAR] = dev_random(cout[j], 0.75f, 1.25f);
    t[DT_BT_CONV2_W] = dev_random((size_t)cout[j] * cout[j] * 9, -0.1f, 0.1f); t[DT_BT_CONV2_B] = dev_random(cout[j], -0.1f, 0.1f);
    t[DT_BT_BN2_G] = dev_random(cout[j], 0.8f, 1.2f); t[DT_BT_BN2_B] = dev_random(cout[j], -0.1f, 0.1f);
    t[DT_BT_BN2_MEAN] = dev_random(cout[j], -0.1f, 0.1f); t[DT_BT_BN2_VAR] = dev_random(cout[j], 0.75f, 1.25f);
    if (cin[j] != cout[j]) { t[DT_BT_RES_W] = dev_random((size_t)cout[j] * cin[j], -0.2f, 0.2f); t[DT_BT_RES_B] = dev_random(cout[j], -0.1f, 0.1f); }
  }
  gt[DT_GT_TIME1_W] = dev_random((size_t)D * D, -0.2f, 0.2f); gt[DT_GT_TIME1_B] = dev_random(D, -0.1f, 0.1f);
  gt[DT_GT_COND0_W] = dev_random(D, -0.5f, 0.5f); gt[DT_GT_COND0_B] = dev_random(D, -0.1f, 0.1f);
  gt[DT_GT_COND2_W] = dev_random((size_t)D * D, -0.2f, 0.2f); gt[DT_GT_COND2_B] = dev_random(D, -0.1f, 0.1f);
  gt[DT_GT_FINAL_W] = dev_random((size_t)C * dims[0], -0.2f, 0.2f); gt[DT_GT_FINAL_B] = dev_random(C, -0.1f, 0.1f);
  gt[DT_GT_FREQS] = dev_random(D / 2, 0.001f, 1.0f);

  hipStream_t s;
  HIP(hipStreamCreate(&s));
  dt_unet_desc desc{C, {dims[0], dims[1], dims[2], dims[3]}, D};
  dt_unet *h = nullptr;
  // ---- argument errors first
  CHECK(dt_unet_create(nullptr, bt.data(), gt.data(), s, &h), DT_E_NULL);
  dt_unet_desc bad = desc; bad.channels = 4;
  CHECK(dt_unet_create(&bad, bt.data(), gt.data(), s, &h), DT_E_SHAPE);       // more than 3 image channels: not supported
  { std::vector<const float *> holes = bt; holes[2 * DT_BT_COUNT + DT_BT_CONV2_W] = nullptr;
    CHECK(dt_unet_create(&desc, holes.data(), gt.data(), s, &h), DT_E_NULL); }
  CHECK(dt_unet_create(&desc, bt.data(), gt.data(), s, &h), DT_OK);
  HIP(hipStreamSynchronize(s));
  const int stride = dt_unet_time_bias_stride(h);
  if (stride <= 0) { fprintf(stderr, "time-bias stride %d\n", stride); return 1; }

  const int H = 16, W = 16, B = 6, E = C * H * W, n_steps = 4;
  // time-bias rows: n_steps x (none, one) for the 2-pass loop, and n_steps x 5 rows for the mixed loop (tb_div = 2)
  std::vector<int32_t> t_rows; std::vector<float> cond; std::vector<uint8_t> present;
  for (int i = 0; i < n_steps; ++i) { for (int p = 0; p < 2; ++p) { t_rows.push_back(40 - 10 * i); cond.push_back((float)p); present.push_back((uint8_t)p); } }
  for (int i = 0; i < n_steps; ++i) for (int r = 0; r < 5; ++r) { t_rows.push_back(40 - 10 * i); cond.push_back(r >= 3 ? 1.f : 0.f); present.push_back(r ? 1 : 0); }
  int32_t *t_dev; float *c_dev; uint8_t *p_dev; float *tb;
  HIP(hipMalloc((void **)&t_dev, t_rows.size() * 4)); HIP(hipMalloc((void **)&c_dev, cond.size() * 4)); HIP(hipMalloc((void **)&p_dev, present.size()));
  HIP(hipMemcpy(t_dev, t_rows.data(), t_rows.size() * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(c_dev, cond.data(), cond.size() * 4, hipMemcpyHostToDevice));
  HIP(hipMemcpy(p_dev, present.data(), present.size(), hipMemcpyHostToDevice));
  HIP(hipMalloc((void **)&tb, t_rows.size() * (size_t)stride * 4));
  CHECK(dt_unet_time_bias(h, t_dev, c_dev, p_dev, (int)t_rows.size(), tb, s), DT_OK);
  CHECK(dt_unet_time_bias(h, nullptr, c_dev, p_dev, 2, tb, s), DT_E_NULL);
  const float *tb_mixed = tb + (size_t)2 * n_steps * stride;

  float *x = dev_random((size_t)B * E, -1.f, 1.f), *eps = dev_random((size_t)2 * B * E, 0.f, 0.f);
  const size_t ws_bytes = dt_unet_workspace_bytes(h, 2 * B, H, W);
  if (!ws_bytes || dt_unet_workspace_bytes(h, 2 * B, 17, W) != 0) { fprintf(stderr, "workspace bytes\n"); return 1; }
  void *ws; HIP(hipMalloc(&ws, ws_bytes));
  CHECK(dt_unet_forward(h, x, B, 2, H, W, tb, B, eps, ws, ws_bytes, s), DT_OK);
  CHECK(dt_unet_forward(h, x, B, 2, H, W, tb, B, eps, ws, ws_bytes / 2, s), DT_E_WORKSPACE);
  CHECK(dt_unet_forward(h, x, B, 2, 24, W, tb, B, eps, ws, ws_bytes, s), DT_E_SHAPE);
  CHECK(dt_unet_forward_mixed(h, x, B, 2, H, W, tb_mixed, 2, eps, ws, ws_bytes, s), DT_OK);            // 2 single-pass + 4 CFG images = 10 rows
  CHECK(dt_unet_forward_mixed(h, x, B, 3, H, W, tb_mixed, 2, eps, ws, ws_bytes, s), DT_E_ARG);         // tb_div must divide the single-pass count
  HIP(hipStreamSynchronize(s));
  // ---- the model qualifies for the fused whole-forward kernel (padded dims 32 / 48): the calls above ran through it; a short
  // sampler loop of each kind does too, then the layered kernels take over for the plan / tuning walk below
  if (dt_unet_fused_active(h, H, W) != 1 || dt_unet_fused_active(h, 32, 32) != 0) { fprintf(stderr, "fused path not active\n"); return 1; }
  CHECK(dt_unet_set_fused(nullptr, 1), DT_E_NULL);
  {
    float *ftraj = dev_random((size_t)3 * B * E, -1.f, 1.f), *fz = dev_random((size_t)2 * B * E, -1.f, 1.f), *fw = dev_random(B, 1.f, 7.f);
    const float fcoef[8] = {0.5f, 0.5f, 0.5f, 0.f, 0.5f, 0.5f, 0.5f, 0.f};
    const int32_t fnoise[2] = {1, 0};
    const int64_t fshift[2] = {0, B};
    CHECK(dt_sample_trajectory(h, DT_RULE_ENGINE, B, 2, H, W, 2, tb, fcoef, fnoise, fz, nullptr, fshift, fw, 1.f, ftraj, nullptr, ws, ws_bytes, s), DT_OK);
    CHECK(dt_sample_trajectory(h, DT_RULE_PSAMPLE, B, 1, H, W, 2, tb, fcoef, fnoise, fz, nullptr, fshift, nullptr, 1.f, ftraj, nullptr, ws, ws_bytes, s), DT_OK);
    CHECK(dt_sample_trajectory_mixed(h, DT_RULE_MANAGER, B, 2, H, W, 2, tb_mixed, 2, fcoef, fnoise, fz, nullptr, fshift, fw, ftraj, ws, ws_bytes, s), DT_OK);
    HIP(hipStreamSynchronize(s));
  }
  CHECK(dt_unet_set_fused(h, 0), DT_OK);
  if (dt_unet_fused_active(h, H, W) != 0) { fprintf(stderr, "fused path still active\n"); return 1; }
  // ---- plans: report, pin, declare, autotune, time one launch
  for (int blk = 0; blk < 8; ++blk)
    for (int slot = 0; slot < 3; ++slot) {
      int bm, bn, sp, pr, tu;
      CHECK(dt_unet_conv_choice(h, 2 * B, H, W, blk, slot, &bm, &bn, &sp, &pr, &tu), DT_OK);
    }
  CHECK(dt_unet_declare_shape(h, 2 * B, H, W, B, 0), DT_OK);
  CHECK(dt_unet_declare_shape(h, 2 * B, H, W, 5, 0), DT_E_ARG);
  CHECK(dt_unet_set_conv_choice(h, 2 * B, H, W, 1, 2, 64, 64, 1, 4, 1), DT_OK);     // K = 32 steps on a 3-chunk layer (zero-padded pack)
  CHECK(dt_unet_set_conv_choice(h, 2 * B, H, W, 2, 1, 64, 64, 1, 5, 0), DT_OK);     // K split across waves on a 3-chunk layer
  CHECK(dt_unet_set_conv_choice(h, 2 * B, H, W, 2, 1, 96, 64, 1, 3, 0), DT_E_ARG);
  CHECK(dt_unet_forward(h, x, B, 2, H, W, tb, B, eps, ws, ws_bytes, s), DT_OK);
  CHECK(dt_unet_autotune(h, 2 * B, H, W, ws, ws_bytes, s), DT_OK);
  float ms = 0.f; double fl = 0.0;
  CHECK(dt_unet_time_conv(h, 2 * B, H, W, 1, 2, 64, 64, 1, 3, 0, 2, ws, ws_bytes, s, &ms, &fl), DT_OK);
  CHECK(dt_unet_set_precision(h, DT_PREC_FP32), DT_OK);
  CHECK(dt_unet_forward(h, x, B, 2, H, W, tb, B, eps, ws, ws_bytes, s), DT_OK);
  CHECK(dt_unet_set_precision(h, 7), DT_E_ARG);
  CHECK(dt_unet_set_precision(h, DT_PREC_AUTO), DT_OK);
  CHECK(dt_unet_set_head_fusion(h, 0), DT_OK);
  CHECK(dt_unet_forward(h, x, B, 2, H, W, tb, B, eps, ws, ws_bytes, s), DT_OK);
  size_t off; int cp, oh, ow;
  CHECK(dt_unet_debug_activation(h, 2 * B, H, W, 7, &off, &cp, &oh, &ow), DT_OK);
  CHECK(dt_unet_set_head_fusion(h, 1), DT_OK);
  // ---- sampler loops: plain, hipGraph replay (captured once, replayed once), mixed
  float *traj = dev_random((size_t)(n_steps + 1) * B * E, -1.f, 1.f), *z = dev_random((size_t)n_steps * B * E, -1.f, 1.f);
  std::vector<float> coef(4 * n_steps, 0.5f);
  std::vector<int32_t> noise(n_steps, 1); noise.back() = 0;
  std::vector<int64_t> shift(n_steps);
  for (int i = 0; i < n_steps; ++i) shift[i] = (int64_t)i * B;
  float *wv = dev_random(B, 1.f, 7.f);

  CHECK(dt_profile_begin(), DT_OK);
  CHECK(dt_sample_trajectory(h, DT_RULE_PSAMPLE, B, 2, H, W, n_steps, tb, coef.data(), noise.data(), z, nullptr, shift.data(), nullptr, 3.f, traj, nullptr, ws, ws_bytes, s), DT_OK);
  HIP(hipStreamSynchronize(s));
  CHECK(dt_profile_end(), DT_OK);
  for (int c = 0; c < dt_profile_class_count(); ++c) { const char *name; long long n; double a, b2, c2; CHECK(dt_profile_read(c, &name, &n, &a, &b2, &c2), DT_OK); }
  setenv("DT_GRAPH", "1", 1);
  for (int rep = 0; rep < 2; ++rep)
    CHECK(dt_sample_trajectory(h, DT_RULE_ENGINE, B, 2, H, W, n_steps, tb, coef.data(), noise.data(), z, nullptr, shift.data(), wv, 1.f, traj, nullptr, ws, ws_bytes, s), DT_OK);
  unsetenv("DT_GRAPH");
  CHECK(dt_sample_trajectory(h, 9, B, 2, H, W, n_steps, tb, coef.data(), noise.data(), z, nullptr, shift.data(), nullptr, 3.f, traj, nullptr, ws, ws_bytes, s), DT_E_ARG);
  CHECK(dt_sample_trajectory_mixed(h, DT_RULE_ENGINE, B, 2, H, W, n_steps, tb_mixed, 2, coef.data(), noise.data(), z, nullptr, shift.data(), wv, traj, ws, ws_bytes, s), DT_OK);
  CHECK(dt_sample_trajectory_mixed(h, DT_RULE_ENGINE, B, 0, H, W, n_steps, tb_mixed, 2, coef.data(), noise.data(), z, nullptr, shift.data(), wv, traj, ws, ws_bytes, s), DT_E_ARG);
  HIP(hipStreamSynchronize(s));
  // ---- metric launchers and the resize kernel
  double *sums; HIP(hipMalloc((void **)&sums, (size_t)B * (n_steps + 1) * 5 * sizeof(double)));
  CHECK(dt_traj_metrics(traj, traj, n_steps + 1, n_steps + 1, B, E, sums, s), DT_OK);
  CHECK(dt_traj_metrics(traj, traj, n_steps + 1, n_steps + 1, B, E + 2, sums, s), DT_E_SHAPE);
  CHECK(dt_traj_wasserstein(traj, traj, n_steps + 1, B, E, nullptr, nullptr, 0, sums, s), DT_OK);
  CHECK(dt_traj_resampled_distance(traj, traj, n_steps + 1, 3, B, E, sums, s), DT_OK);
  CHECK(dt_pair_stats(traj, traj, n_steps + 1, B, E, sums, s), DT_OK);
  CHECK(dt_traj_pair_metrics(traj, traj, n_steps + 1, B, E, sums, sums + (size_t)B * (n_steps + 1) * 4, s), DT_OK);
  CHECK(dt_traj_pair_metrics(traj, traj, n_steps + 1, B, 4100, sums, sums, s), DT_E_SHAPE);
  CHECK(dt_traj_sample_mean(traj, n_steps + 1, B, E, eps, s), DT_OK);
  CHECK(dt_resize_bilinear(x, eps, B * C, H, W, 8, 24, s), DT_OK);
  CHECK(dt_resize_bilinear(x, nullptr, B * C, H, W, 8, 24, s), DT_E_NULL);
  const float cf[4] = {1.f, 0.1f, 0.2f, 0.f};
  CHECK(dt_cfg_update(DT_RULE_MANAGER, x, eps, nullptr, z, nullptr, cf, 1, nullptr, 1.f, traj, B, E, s), DT_OK);
  CHECK(dt_cfg_update(DT_RULE_MANAGER, x, eps, nullptr, z, nullptr, cf, 1, nullptr, 1.f, traj, B, E + 1, s), DT_E_SHAPE);
  CHECK(dt_profile_marker(3, s), DT_OK);
  HIP(hipStreamSynchronize(s));
  dt_unet_destroy(h);
  dt_unet_destroy(nullptr);
  for (void *p : g_allocs) (void)hipFree(p);
  (void)hipFree(t_dev); (void)hipFree(c_dev); (void)hipFree(p_dev); (void)hipFree(tb); (void)hipFree(ws); (void)hipFree(sums);
  HIP(hipStreamDestroy(s));
  printf("driver ok (abi %d)\n", dt_abi_version());
  return 0;
}

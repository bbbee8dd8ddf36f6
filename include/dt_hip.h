/*
 * dt_hip.h -- C ABI of libdt_hip.so, the MI355X (gfx950) implementation of the
 * trajectory hot path of henriChevreux/distillation_trajectories.
 *
 * The reference is pure Python on PyTorch and defines no FFI of its own
 * (SURVEY.md §8b); the boundary below is what a ctypes binding on the reference
 * side would call, one entry point per reference function it replaces.  Rules:
 *   - plain pointers and sizes only; every pointer named *_dev is a BORROWED
 *     device pointer (e.g. torch.Tensor.data_ptr()), fp32 unless stated;
 *   - every call takes the HIP stream to launch on (hipStream_t as void*), is
 *     asynchronous w.r.t. the host and never synchronises;
 *   - returns int: 0 ok, <0 argument/shape error (DT_E_*), >0 a hipError_t;
 *   - never throws, never allocates device memory except dt_unet_create
 *     (freed by dt_unet_destroy); scratch comes from the caller (workspace).
 *
 * Layouts: images are NCHW fp32 exactly as the reference holds them
 * (x[B,C,H,W]); trajectories are step-major [n_steps][B][C*H*W].
 */
#ifndef DT_HIP_H
#define DT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: time-bias rows carry enc1.conv2's class-bias columns (dt_unet_time_bias_stride), launch kind 5 and the 256 x 64 tile
 * in the conv-choice hooks; the exported symbols are those of version 1
 * 3: + dt_unet_forward_mixed / dt_sample_trajectory_mixed (single-pass and CFG images in one batch), dt_unet_declare_shape,
 *    dt_resize_bilinear;
 *    dt_sample_trajectory accepts eps_scratch_dev == NULL
 * 4: + dt_unet_set_fused / dt_unet_fused_active (small models: one launch per forward / per sampler call),
 *    dt_traj_pair_metrics (metric sums + Wasserstein term in one pass) */
#define DT_ABI_VERSION 4

enum {
  DT_OK = 0,
  DT_E_NULL = -1,      /* required pointer is NULL */
  DT_E_SHAPE = -2,     /* unsupported or inconsistent shape */
  DT_E_ARG = -3,       /* bad enum / count */
  DT_E_WORKSPACE = -4  /* workspace too small */
};

/* cond modes of one U-Net pass: how `cond` is supplied to models.py:159 forward */
enum { DT_COND_NONE = 0, DT_COND_ZERO = 1, DT_COND_ONE = 2 };

/* update rules of the three reverse loops */
enum {
  DT_RULE_ENGINE = 0,  /* analysis/trajectory_engine.py:86-110  x = c1*x - c2*eps (+ sigma*z)   */
  DT_RULE_PSAMPLE = 1, /* utils/diffusion.py:149-158            x = sra*(x - k*eps) + z*beta     */
  DT_RULE_MANAGER = 2  /* utils/trajectory_manager.py:167-205   x = (x - 0.1*eps)/sqrt(.9) + s*z */
};

int dt_abi_version(void);
const char *dt_status_string(int status);

/* ------------------------------------------------------------------ U-Net ---
 * Replaces models.py:85-224 DiffusionUNet (eval mode).  Weights arrive as the
 * reference's own state_dict tensors (device pointers, native OIHW / [out,in]
 * layouts); dt_unet_create folds BatchNorm, pads channels to multiples of 16 and
 * re-tiles the convolution weights for the MFMA kernels into memory it owns.
 */
typedef struct dt_unet dt_unet;

/* order of the per-block tensor pointers (8 blocks: enc1..enc4, bottleneck, dec3, dec2, dec1) */
enum {
  DT_BT_TIME_W = 0, DT_BT_TIME_B,
  DT_BT_CONV1_W, DT_BT_CONV1_B, DT_BT_BN1_G, DT_BT_BN1_B, DT_BT_BN1_MEAN, DT_BT_BN1_VAR,
  DT_BT_CONV2_W, DT_BT_CONV2_B, DT_BT_BN2_G, DT_BT_BN2_B, DT_BT_BN2_MEAN, DT_BT_BN2_VAR,
  DT_BT_RES_W, DT_BT_RES_B, /* NULL when in_ch == out_ch (identity skip) */
  DT_BT_COUNT
};
/* order of the global tensor pointers */
enum {
  DT_GT_TIME1_W = 0, DT_GT_TIME1_B, /* time_mlp.1 */
  DT_GT_COND0_W, DT_GT_COND0_B,     /* cond_emb.0 */
  DT_GT_COND2_W, DT_GT_COND2_B,     /* cond_emb.2 */
  DT_GT_FINAL_W, DT_GT_FINAL_B,     /* final 1x1 */
  DT_GT_FREQS,                      /* sinusoid frequencies [max(D/2,1)] (models.py:20-21) */
  DT_GT_COUNT
};

typedef struct {
  int32_t channels;  /* image channels C */
  int32_t dims[4];   /* models.py:110 */
  int32_t temb_dim;  /* models.py:101 */
} dt_unet_desc;

int dt_unet_create(const dt_unet_desc *desc,
                   const float *const *block_tensors_dev, /* [8][DT_BT_COUNT] */
                   const float *const *global_tensors_dev, /* [DT_GT_COUNT] */
                   void *stream, dt_unet **out);
void dt_unet_destroy(dt_unet *h);

/* floats per row of the time-bias table: the padded out-channels of the 8 blocks, followed by the nine class-bias
 * vectors of enc1.conv2 (9 x enc1's padded width: the row's enc1 time bias pushed through conv2's in-picture taps for
 * corner / edge / interior pixels, which is what lets the n_pass passes of a forward share ONE enc1 launch) */
int dt_unet_time_bias_stride(const dt_unet *h);

/* models.py:175-185 + the per-block ReLU(Linear(temb)) of models.py:66-77 (+ the class biases above), evaluated once
 * per (t, cond) row instead of once per sample.  cond_dev[r] is the scalar condition of row r;
 * cond_present_dev[r]==0 means cond=None (either array may be NULL: no cond / all present). */
int dt_unet_time_bias(const dt_unet *h, const int32_t *t_dev, const float *cond_dev,
                      const uint8_t *cond_present_dev, int rows, float *out_dev, void *stream);

size_t dt_unet_workspace_bytes(const dt_unet *h, int batch_total, int H, int W);

/* eps[n_pass*B,C,H,W] = U-Net(x[B,C,H,W]) for n_pass condition rows.  Batch row r uses time-bias
 * row r / tb_div of tb_dev (so tb_div = B shares one (t,cond) per pass; tb_div = 1 is the fully
 * general per-sample form of models.py:159).  x is read once and shared by all passes. */
int dt_unet_forward(const dt_unet *h, const float *x_dev, int B, int n_pass, int H, int W,
                    const float *tb_dev, int tb_div, float *eps_dev,
                    void *workspace_dev, size_t workspace_bytes, void *stream);

/* Mixed batch: the first B_single images take ONE pass (the reference's branch for guidance_scale None / <= 1,
 * analysis/trajectory_engine.py:83), the remaining B - B_single take the two CFG passes (:65-80).  One forward covers
 * 2B - B_single rows laid out [pass 0 of all B images | pass 1 of images B_single .. B-1]; eps has that many rows; batch
 * row r uses time-bias row r / tb_div, where tb_div must divide B_single and 2B - B_single.  Same kernels as
 * dt_unet_forward: the single-pass images ride in the CFG images' launches instead of a launch sequence of their own. */
int dt_unet_forward_mixed(const dt_unet *h, const float *x_dev, int B, int B_single, int H, int W,
                          const float *tb_dev, int tb_div, float *eps_dev,
                          void *workspace_dev, size_t workspace_bytes, void *stream);

/* Optional: measure (HIP events, synchronises the stream) every admissible tile / tap-split of every
 * convolution launch of a forward with batch_total = n_pass*B rows and remember the fastest per layer;
 * later dt_unet_forward / dt_sample_trajectory calls with the same (batch_total, H, W) use them.
 * Results are identical up to fp32 summation order (the split changes the grouping of the tap sum). */
int dt_unet_autotune(dt_unet *h, int batch_total, int H, int W, void *workspace_dev, size_t workspace_bytes,
                     void *stream);
/* report hook: tile (bm x bn), tap split and launch kind (0 fp32 MFMA, 1 split-bf16 split in the consumer,
 * 2 unused (round 1's LDS-DMA variant), 3 split-bf16 "strip" kernel that stages a 3x3 layer's activation tile
 * once per channel chunk for all nine taps, 4 the same with two channel chunks (K = 32) per step, 5 the strip
 * kernel with the step's chunks split across the waves of the workgroup -- 64 x 64 wave tiles on 64 x 64 /
 * 128 x 64 / 64 x 128 workgroup tiles, partial tiles summed in wave order; +8 when the block's 1x1 skip is
 * folded into this conv2 launch) in use for block (0..7), slot (0 skip, 1 conv1, 2 conv2);
 * bm = 0 means the slot has no launch of its own */
int dt_unet_conv_choice(const dt_unet *h, int batch_total, int H, int W, int block, int slot, int *bm, int *bn,
                        int *splits, int *prec, int *tuned);

/* The tuning / plan hooks below are keyed by a forward's ROW count; how the rows split into images (images x passes, or a mixed
 * batch) decides enc1's launches, which run over the images.  The hooks use the split of the last forward with that row count;
 * this declares it beforehand (a plan applied before the first forward of a shape). */
int dt_unet_declare_shape(dt_unet *h, int batch_total, int H, int W, int images, int single_pass_images);

/* tuning / test hook: pin the launch choice of one convolution of a forward shape (the other slots keep
 * their current choice).  bm x bn in {64,128}^2, or 256 x 64 for the strip kernels (prec 3 / 4; not 128 x 128 for
 * prec 5); prec 0..5 as reported by dt_unet_conv_choice; splits in {1,3,9} taps, or 1..8 channel-chunk groups for
 * the strip kernels (prec >= 3; reset to 1 where it does not divide); fuse only for slot 2. */
int dt_unet_set_conv_choice(dt_unet *h, int batch_total, int H, int W, int block, int slot, int bm, int bn,
                            int splits, int prec, int fuse);

/* tuning aid: time ONE convolution launch (block, slot) of a forward shape under an explicit choice
 * (prec: 0 fp32 MFMA, 1 split-bf16, 3 / 4 split-bf16 strip kernel with K = 16 / 32 per step, 5 strip kernel with the
 * K split across the waves); averages `reps` launches with HIP events */
int dt_unet_time_conv(const dt_unet *h, int batch_total, int H, int W, int block, int slot, int bm, int bn,
                      int splits, int prec, int fuse, int reps, void *workspace_dev, size_t workspace_bytes,
                      void *stream, float *ms, double *flops);

/* Convolution arithmetic.  Both variants are fp32-accurate: DT_PREC_FP32 uses the exact fp32 MFMA
 * (v_mfma_f32_32x32x2_f32); DT_PREC_SPLIT_BF16 splits every fp32 operand exactly into three bf16
 * planes and sums the six significant plane products on the bf16 matrix cores (fp32 accumulate; the
 * dropped cross terms are < 2^-25 relative).  DT_PREC_AUTO (default) lets the autotuner pick per layer.
 * (Round 1 also carried an LDS-DMA variant fed by pre-split bf16 "plane twin" tensors; it tied the in-consumer
 * split on the large layers and lost the twin-write time, so it was removed.) */
enum { DT_PREC_FP32 = 0, DT_PREC_SPLIT_BF16 = 1, DT_PREC_AUTO = 2 };
int dt_unet_set_precision(dt_unet *h, int precision);

/* test hook: dec1.conv2 normally evaluates the final 1x1 head in its own epilogue and then never writes dec1's output,
 * and enc1.conv2 writes only its pooled output (nothing else reads enc1's full-resolution tensor); on = 0 materialises
 * both block outputs (separate head launch) so that dt_unet_debug_activation can expose every block */
int dt_unet_set_head_fusion(dt_unet *h, int on);

/* Small models (padded dims[0] <= 32, dims[1..3] <= 64: size factors <= 0.25 of models.py:104-110) at 16 x 16 run their
 * WHOLE forward -- and dt_sample_trajectory(_mixed) their whole loop: forward, CFG mix, update and trajectory store of
 * every timestep -- as ONE kernel launch with all activations in LDS, on the exact fp32 MFMA.  On by default where the
 * model qualifies (DT_NO_FUSED=1 in the environment at create time: off); dt_unet_set_fused(h, 0) and
 * dt_unet_set_head_fusion(h, 0) select the layered kernels for the handle (the per-layer launch choices above only
 * concern those).  dt_unet_fused_active: 1 / 0 for an H x W image. */
int dt_unet_set_fused(dt_unet *h, int on);
int dt_unet_fused_active(const dt_unet *h, int H, int W);

/* test hook: float offset / padded channel count of a block output inside the workspace
 * (which: 0..7 block outputs in DT order), valid after dt_unet_forward with the same shape */
int dt_unet_debug_activation(const dt_unet *h, int batch_total, int H, int W, int which,
                             size_t *offset_floats, int *channels_padded, int *out_h, int *out_w);

/* ------------------------------------------------------- fused CFG + update ---
 * One reverse step for a batch: eps = e_u + w*(e_c - e_u) when eps_c_dev != NULL (else eps = e_u),
 * then the rule's x_{t-1}, written to x_out_dev (the next trajectory slot).  coef[4] by rule:
 *   ENGINE : c1, c2, sigma, unused       (has_noise==0: x_out = x, the t==0 step)
 *   PSAMPLE: sqrt_recip_alpha, 1-sqrt(1-acp), beta, unused
 *   MANAGER: beta(=1-0.9), sqrt(alpha), noise_scale, unused
 * w_dev: per-row guidance scale [B] (NULL: use w_scalar).  z row of batch row r is
 * z_dev + (z_row_dev ? z_row_dev[r] : r) * E.   Arithmetic is un-contracted fp32 in the
 * reference's operation order, so given equal eps the result is bit-identical to torch CPU. */
int dt_cfg_update(int rule, const float *x_dev, const float *eps_u_dev, const float *eps_c_dev,
                  const float *z_dev, const int32_t *z_row_dev, const float coef[4], int has_noise,
                  const float *w_dev, float w_scalar, float *x_out_dev, int B, int E, void *stream);

/* ------------------------------------------------------------ whole loop ---
 * Replaces the loops of utils/diffusion.py:199-208, trajectory_engine.py:61-113 and
 * trajectory_manager.py:100-112: n_steps reverse steps, everything device-resident.
 * traj_dev[(n_steps+1)][B][E]: slot 0 must hold x_T on entry; slot i+1 receives the state after
 * step i.  tb_dev holds n_steps*n_pass time-bias rows in (step, pass) order.  coef_host[n_steps][4],
 * has_noise_host[n_steps], z_shift_host[n_steps] (added to the z row index at that step). */
int dt_sample_trajectory(const dt_unet *h, int rule, int B, int n_pass, int H, int W, int n_steps,
                         const float *tb_dev, const float *coef_host, const int32_t *has_noise_host,
                         const float *z_dev, const int32_t *z_row_dev, const int64_t *z_shift_host,
                         const float *w_dev, float w_scalar, float *traj_dev, float *eps_scratch_dev,
                         void *workspace_dev, size_t workspace_bytes, void *stream);

/* The same loop for a mixed batch (see dt_unet_forward_mixed): images [0, B_single) are advanced with the single-pass
 * prediction (eps = model(x, t)), images [B_single, B) with eps_u + w[b] (eps_c - eps_u).  tb_dev holds
 * n_steps * (2B - B_single) / tb_div rows in (step, row group) order; w_dev[B] (entries below B_single are ignored).
 * This is how one launch sequence serves every guidance scale of compare_trajectories (trajectory_engine.py:142-164). */
int dt_sample_trajectory_mixed(const dt_unet *h, int rule, int B, int B_single, int H, int W, int n_steps,
                               const float *tb_dev, int tb_div, const float *coef_host, const int32_t *has_noise_host,
                               const float *z_dev, const int32_t *z_row_dev, const int64_t *z_shift_host,
                               const float *w_dev, float *traj_dev, void *workspace_dev, size_t workspace_bytes,
                               void *stream);

/* ---------------------------------------------------------------- metrics ---
 * Replaces the reductions of analysis/metrics/trajectory_metrics.py:55-231 and
 * analysis/metrics/time_dependent.py:46-82.  Trajectories are [n][B][E].
 * out_sums_dev[B][n_max][4] (float64), n_max = max(nT, nS):
 *   i >= 1: { |X_i-Y_i|^2, |X_i-X_{i-1}|^2, |Y_i-Y_{i-1}|^2, <X_i-X_{i-1}, Y_i-Y_{i-1}> }
 *   i == 0: { |X_0-Y_0|^2, |X_last-X_0|^2, |Y_last-Y_0|^2, |X_last-Y_last|^2 }  (own last states)
 * terms that need a state beyond a trajectory's own length are 0.  The reference's norms run over
 * the whole [B,C,H,W] list entry: for a batched entry call with B=1 and E=B*C*H*W (same memory). */
int dt_traj_metrics(const float *teacher_dev, const float *student_dev, int nT, int nS, int B, int E,
                    double *out_sums_dev, void *stream);

/* trajectory_metrics.py:295-315: W1 between the (sub-sampled) value distributions of X_i and Y_i,
 * = mean |sort(u) - sort(v)| accumulated in float64.  index_dev: [n_tables][n][n_idx] int32 coordinates
 * to sample (NULL: use all E coordinates, requires E <= 4096); pair b uses table index_row_dev[b]
 * (NULL: table 0) -- the reference's indices depend on the sample seed only.  out_w1_dev[B][n] float64. */
int dt_traj_wasserstein(const float *teacher_dev, const float *student_dev, int n, int B, int E,
                        const int32_t *index_dev, const int32_t *index_row_dev, int n_idx,
                        double *out_w1_dev, void *stream);

/* Both reductions above in ONE pass for two trajectories of equal length n whose Wasserstein term uses all E <= 4096
 * coordinates (every 16 x 16 configuration): out_sums_dev[B][n][4] as dt_traj_metrics, out_w1_dev[B][n] as
 * dt_traj_wasserstein(index NULL).  A workgroup per (pair, step) keeps both states in registers: the sums against the
 * previous states, then a register / cross-lane (__shfl_xor) bitonic sort of both; each state comes from HBM once. */
int dt_traj_pair_metrics(const float *teacher_dev, const float *student_dev, int n, int B, int E,
                         double *out_sums_dev, double *out_w1_dev, void *stream);

/* trajectory_metrics.py:239-279: linear (scipy interp1d, float64) resampling of the longer trajectory
 * onto the shorter's normalised time grid, then |L'(t_i) - S_i|_2 in float64.  out_dist_dev[B][n_short] */
int dt_traj_resampled_distance(const float *long_dev, const float *short_dev, int n_long, int n_short,
                               int B, int E, double *out_dist_dev, void *stream);

/* Bilinear resize with align_corners=True of `planes` = N*C contiguous images [h][w] -> [H][W] (NCHW tensors, any ratio):
 * torch.nn.functional.interpolate(img, size=(H, W), mode='bilinear', align_corners=True) as the reference applies it to a
 * student that runs at another resolution (utils/trajectory_manager.py:153-163,298-305; trajectory_metrics.py:40-52). */
int dt_resize_bilinear(const float *in_dev, float *out_dev, int planes, int h, int w, int H, int W, void *stream);

/* Next-row reductions.
 * dt_pair_stats: evaluation/metrics.py:118-183 (compute_trajectory_divergence) and
 * analysis/noise_prediction/noise_analysis.py:43-85 (calculate_noise_metrics): for state i of pair b,
 * out[b][i] = { sum (x-y)^2, sum |x-y|, sum x*y, sum x^2, sum y^2 } in float64.  X, Y are [n][B][E].
 * dt_traj_sample_mean: scripts/analysis/analyze_trajectories.py:467-486: mean over the B samples of a
 * trajectory tensor [n][B][E] -> out[n][E] (float64 accumulation, fp32 result). */
int dt_pair_stats(const float *x_dev, const float *y_dev, int n, int B, int E, double *out_dev, void *stream);
int dt_traj_sample_mean(const float *traj_dev, int n, int B, int E, float *out_dev, void *stream);

/* -------------------------------------------------------------- profiling ---
 * Optional per-launch timing for the benchmark's roofline line: between dt_profile_begin() and
 * dt_profile_end() every kernel launch of this library is bracketed by two hipEventRecord calls on
 * the launch stream.  dt_profile_read (after the stream has been synchronised) sums, per kernel
 * class, the launches, the event-measured milliseconds and the ALGORITHMIC flops / bytes of those
 * launches (real channel counts, no padding; what the reference's own ops would count). */
/* launches the empty kernel `profile_marker_kernel` on the stream: rocprofv3 kernel traces of a run can then be cut to
 * the region between two markers (profiles/summarize_pmc.py keeps the timed steps and drops the autotuner's trials) */
int dt_profile_marker(int id, void *stream);
int dt_profile_begin(void);
int dt_profile_end(void);
int dt_profile_class_count(void);
int dt_profile_read(int cls, const char **name, long long *launches, double *ms, double *flops, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* DT_HIP_H */

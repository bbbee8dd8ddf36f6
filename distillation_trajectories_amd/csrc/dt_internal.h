// Internal declarations shared by the translation units of libdt_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dt_hip.h"

#define DT_HIP_TRY(expr)                         \
  do {                                           \
    hipError_t _e = (expr);                      \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)
#define DT_LAUNCH_CHECK()                        \
  do {                                           \
    hipError_t _e = hipGetLastError();           \
    if (_e != hipSuccess) return (int)_e;        \
  } while (0)

namespace dt {

// ---- optional per-launch timing with HIP events on the launch stream (bench.py roofline numbers).
// Off by default; when on, every instrumented launch is bracketed by two event records.
enum KernelClass {
  KC_CONV_128x128 = 0, KC_CONV_128x64, KC_CONV_64x128, KC_CONV_64x64,
  KC_CONVB_128x128, KC_CONVB_128x64, KC_CONVB_64x128, KC_CONVB_64x64,
  KC_CONVS_128x128, KC_CONVS_128x64, KC_CONVS_64x128, KC_CONVS_64x64, KC_CONVS_256x64, KC_CONVS_K128x64, KC_CONVS_K64x64, KC_CONVS_K64x128,
  KC_SPLITK_EPILOGUE, KC_FIRST_CONV, KC_POOL, KC_UPCAT, KC_HEAD, KC_HEAD_UP, KC_TIME_BIAS, KC_UPDATE, KC_METRICS,
  KC_WASSERSTEIN, KC_RESAMPLE, KC_FUSED, KC_PAIR_METRICS,
  KC_COUNT
};
struct ProfileScope {
  ProfileScope(int cls, double flops, double bytes, hipStream_t s);
  ~ProfileScope();
  int slot;
  hipStream_t stream;
  void *end_event;
};

// bilinear source coordinates for scale factor 2 with align_corners=True (ATen area_pixel_compute_scale:
// scale = (in-1)/(out-1), 0 when out == 1; src = scale*dst), and the four-tap blend with its operation order fixed by
// explicit fmaf so that every translation unit (whatever its fp-contract setting) produces the same bits
__device__ inline void bilinear_src(int dst, int in, int out, int &i0, int &i1, float &l0, float &l1) {
#pragma clang fp contract(off)   // src must be the ROUNDED product, as ATen forms it: fused into `src - i0` it moves the weights by up to 1 ulp of src (2e-6 at src = 31)
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}
__device__ inline float bilinear_blend(float v00, float v01, float v10, float v11, float wy0, float wy1, float wx0, float wx1) {
  const float top = fmaf(wx1, v01, wx0 * v00), bot = fmaf(wx1, v11, wx0 * v10);
  return fmaf(wy1, bot, wy0 * top);
}

constexpr int kChanPad = 16;   // activation channel granularity (= BK of the conv GEMM)
constexpr int kNPad = 64;      // packed-weight N granularity (smallest BN tile)
constexpr int kBlocks = 8;
constexpr int kChunkPad = 4;   // chunk granularity of the split-bf16 weight packs (see ConvParams::ccw)

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// whether a walk of `cc` chunks in groups of `div` (K chunks per step x split-K slices) stays inside a pack of `ccw` chunks
inline bool chunks_fit(int cc, int ccw, int div) { return div >= 1 && round_up(cc, div) <= ccw; }

// One convolution expressed as an implicit GEMM over NHWC activations:
//   out[m][n] = epilogue( sum_{tap,c} in[pixel(m)+tap][c] * w[tap][c][n] )
// with m = (b, y, x) flattened, zero padding of (ksize-1)/2.
struct ConvParams {
  const float *in;     // [Bt][H][W][cin_p]
  const float *w;      // packed [taps*cin_p/16][n_p][16], 16-B slots XOR-swizzled by (n>>2)&3
  const float *scale;  // [n_p]  folded BN scale (1 for plain convs, 0 on padding)
  const float *shift;  // [n_p]  folded BN shift / bias
  const float *tb;     // time-bias rows (already offset to this block's channels) or nullptr
  const float *add;    // residual [M][cout_p] or nullptr
  float *out;          // [M][cout_p]
  int M, H, W;
  int cin_p, cout_p, n_p;
  int cin_real, cout_real;  // unpadded channel counts (algorithmic flop accounting only)
  int ksize;           // 1 or 3
  int tap_lo, tap_hi;  // taps visited (a 1x1 image only sees the centre tap)
  int relu;
  int tb_stride;       // floats between time-bias rows
  int m_per_tb;        // GEMM rows sharing one time-bias row
  // split-K over taps: grid.z = splits, each z walks (tap_hi-tap_lo)/splits taps and stores its raw
  // accumulators to slab[z][M][cout_p]; splitk_epilogue then sums the slabs in z order (deterministic)
  int splits;
  float *slab;
  int bm, bn;          // tile override (0 = pick by heuristic); bn = 128 needs n_p % 128 == 0
  int prec;            // 0: exact fp32 MFMA (w = fp32 pack), 1: split-bf16 (w = bf16x3 pack), 2: unused,
                       // 3: split-bf16 strip kernel (3x3 only; `splits` then divides the channel chunks, not the taps),
                       // 4: strip kernel with two channel chunks (K = 32) per step
  // enc1's 1x1 skip of the C <= 3 channel image x[imgs][C][H*W] (NCHW, shared by all passes), recomputed in the
  // epilogue instead of being materialised: add = sum_c x3[((m / hw) % imgs) * C * hw + c * hw + m % hw] * w3[n*4+c] + w3[n*4+3]
  const float *x3;
  const float *w3;
  int x3_hw, x3_imgs, x3_c;
  // fused 1x1 skip (splits == 1 only): after the last chunk of the main K walk the accumulators become
  // relu(acc*scale+shift) IN REGISTERS, then the walk continues over in2[M][cin2_p] x w2 (the block's
  // residual_conv) on the same accumulators; the epilogue only adds bias2.  Saves the skip launch and the
  // round trip of its [M][cout_p] output through HBM.
  const float *in2;
  const float *w2;
  const float *bias2;
  int cin2_p, cin2_real;
  // fused 2x2 max pool of the output (encoder conv2, unsplit launches, W in {8, 16}: every pooling window then lies in
  // one lane's accumulator registers): also written, [M/4][cout_p]; nullptr = the separate maxpool_kernel runs
  float *pool_out;
  // decoder blocks, strip kernel only: the block input is the concat [upsampled | encoder skip]; when in_b / in2_b is
  // set, 16-channel chunks >= cc_a of `in` / `in2` are read from that tensor (row stride b_stride floats) instead,
  // so the concat buffer's skip half is never written
  const float *in_b, *in2_b;
  int cc_a, b_stride;
  // final 1x1 head (models.py:221-224) folded into dec1.conv2's epilogue when one workgroup holds every output channel
  // of its rows (n_p == BN): lowres[m][c] = sum_n v[m][n] * head_w[c * head_cin + n] + head_b[c], c < head_c <= 3, stored
  // as float4 per pixel; `out` is then not written at all (the head is dec1's only consumer)
  const float *head_w, *head_b;
  float *head_out;
  int head_c, head_cin;
  // enc1.conv2 when the n_dup passes of a CFG step share the block input (n_dup > 0): the launch covers the B images
  // once, WITHOUT the time bias in its input; per pass the epilogue adds that pass's class-bias row (tbc: nine
  // per-channel vectors -- corner / edge / interior -- per time-bias row, see tap_bias_kernel) and writes out / pool_out
  // at row offset pass * dup_rows (pass * dup_rows / 4 for the pooled tensor)
  const float *tbc;
  int n_dup, dup_rows;
  // mixed batches (dt_sample_trajectory_mixed): the first dup_skip rows of the launch (images that take ONE pass) have no second
  // pass; pass 1 of row m >= dup_skip goes to row dup_rows + m - dup_skip.  A multiple of H * W (whole images).
  int dup_skip;
  int skip_out;        // n_dup mode: do not store `out` (enc1's full-resolution output has no reader but the pool)
  int dup_stage2;      // n_dup == 2: the launch's LDS holds a second epilogue stage (set by the strip launcher)
  // 16-channel weight chunks per tap in the packs behind `w` / `w2`.  The split-bf16 packs are zero-padded to a multiple
  // of kChunkPad chunks so that the kernels that multiply 2 or 4 chunks per step (and split them across waves) also run
  // layers with an odd chunk count (cin_p = 48, 80, 112, 208, 240: size factors 0.3, 0.4, 0.6, 0.8, 0.9): the walk is
  // ceil(chunks / step) steps, the activation loads of a chunk >= cin_p / 16 are replaced by zeros
  int ccw, ccw2;
  int ablate;          // timing experiments only (wrong results): 1 no barrier, 2 no LDS reads, 3 no MFMA, 4 no staging
};

int launch_conv(const ConvParams &p, hipStream_t s);
int launch_conv_bf16x6(const ConvParams &p, int bm, int bn, hipStream_t s);
int launch_conv_strip(const ConvParams &p, int bm, int bn, int prec, hipStream_t s);   // prec 3 / 4 / 5: 3x3 only
int strip_kc(int prec, int bm, int bn);                         // 16-channel chunks per step of a strip arithmetic code
// whether the strip kernel can run a bm x bn tile with kc chunks per step on rows of W pixels (staging reach, LDS)
bool strip_admissible(int W, int bm, int bn, int prec);
// cin_w: padded input channels per tap of the pack (>= cin_p, multiple of 16 * kChunkPad; zeros beyond cin_p)
int launch_pack_conv_bf16x3(const float *w_oihw, void *wp, int cout, int cin, int ksize, int cin_p, int cin_w, int n_p,
                            int split_c, int split_cp, hipStream_t s);
struct ConvChoice { int bm, bn, splits, prec, fuse; };
ConvChoice heuristic_choice(int M, int n_p, int taps);
constexpr int kSplitMaxRows = 32768;   // split-K candidates only below this many GEMM rows (bounds the slab)

int launch_pack_conv(const float *w_oihw, float *wp, int cout, int cin, int ksize, int cin_p, int n_p,
                     int split_c, int split_cp, hipStream_t s);
int launch_fold_bn(const float *conv_b, const float *g, const float *b, const float *mean, const float *var,
                   float *scale, float *shift, int cout, int n_p, hipStream_t s);
int launch_pack_linear_rows(const float *w, const float *b, float *wp, float *bp, int out, int in, int out_p,
                            hipStream_t s);

// enc1.conv1 + BN + ReLU + time bias straight from the NCHW image (direct fp32 convolution, K = 9C <= 27)
int launch_first_conv(const float *x, const float *wf, const float *scale, const float *shift, const float *tb, int tb_stride,
                      int tb_div, float *out, int B, int n_pass, int C, int H, int W, int cout, int cout_p, hipStream_t s);
int launch_pack_first_conv(const float *w_oihw, float *wf, int cout, int C, int cout_p, hipStream_t s);
int launch_pack_tap_major(const float *w_oihw, float *w2t, int cout, int cin, int cp, hipStream_t s);   // [n][ci][3][3] -> [tap][ci][cp]
int launch_maxpool(const float *in, float *out, int Bt, int H, int W, int cp, hipStream_t s);
// skip == nullptr: only the upsampled channels are written (the consumers read the skip half in place)
int launch_upcat(const float *lo, const float *skip, float *out, int Bt, int h, int w, int c1p, int c2p,
                 hipStream_t s);
int launch_head(const float *lo, const float *w, const float *bias, float *lowres, int Bt, int h, int w_, int cp, int C, int c_real,
                hipStream_t s);
int launch_head_upsample(const float *lowres, float *eps, int Bt, int h, int w_, int C, hipStream_t s);
int launch_pack_res3(const float *w, const float *b, float *w3, int cout, int C, int n_p, hipStream_t s);

struct TembWeights {
  const float *freqs;  // [half]
  const float *w1, *b1;      // time_mlp.1  [D][D]
  const float *wc0, *bc0;    // cond_emb.0  [D][1]
  const float *wc2, *bc2;    // cond_emb.2  [D][D]
  const float *wt, *bt;      // concatenated per-block time_mlp rows [tb_cols][D] (zero rows on padding)
  int D, half, tb_stride;    // tb_stride: floats per output row
  int tb_cols;               // projected channels per row (sum of the blocks' padded widths); the rest of a row:
  const float *w2t;          // enc1.conv2's weights as [tap][ci][c0p] (nullptr: no class-bias columns)
  int c0p;                   // enc1's padded width; columns [tb_cols, tb_cols + 9 c0p) hold the class biases
};
int launch_time_bias(const TembWeights &tw, const int32_t *t, const float *cond, const uint8_t *present, int rows,
                     float *out, hipStream_t s);

}  // namespace dt

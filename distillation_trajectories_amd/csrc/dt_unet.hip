// Host side of libdt_hip.so: the U-Net handle (packed weights), the forward launch plan, the
// device-resident reverse-diffusion loop and the extern "C" surface declared in include/dt_hip.h.
//
// Plan of one forward (reference models.py:159-224), NHWC activations with channels padded to 16:
//   x(NCHW) -> enc1.conv1 (direct fp32 conv, shared by the passes)
//   enc1 @H      -> pool -> enc2 @H/2 -> pool -> enc3 @H/4 -> pool -> enc4 @H/8 -> pool -> bottleneck @H/16
//   up+cat(enc4) -> dec3 @H/8 -> up+cat(enc3) -> dec2 @H/4 -> up+cat(enc2) -> dec1 @H/2 -> up + 1x1 head @H
// Each residual block is up to three implicit-GEMM launches (1x1 skip, conv1, conv2) whose epilogues
// carry BN/ReLU/time-bias/residual, so a forward is 8 blocks * (2..3) + 4 pools + 3 upcats + 2 = ~32 launches.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>

#include "dt_fused.h"
#include "dt_internal.h"

namespace dt {
int launch_cfg_update(int rule, const float *x, const float *eu, const float *ec, const float *z,
                      const int32_t *z_row, long long z_shift, const float coef[4], int has_noise, const float *w,
                      float w_scalar, float *out, int B, int E, hipStream_t s);
int launch_cfg_update_lowres(int rule, const float *x, const float *lowres_u, const float *lowres_c, const float *z,
                             const int32_t *z_row, long long z_shift, const float coef[4], int has_noise, const float *w,
                             float w_scalar, float *out, int B, int C, int H, int W, int b_single, hipStream_t s);
int launch_traj_metrics(const float *X, const float *Y, int nT, int nS, int B, int E, double *out, hipStream_t s);
int launch_wasserstein(const float *X, const float *Y, int n, int B, int E, const int32_t *index,
                       const int32_t *index_row, int n_idx, double *out, hipStream_t s);
int launch_resampled_distance(const float *L, const float *S, int n_long, int n_short, int B, int E, double *out,
                              hipStream_t s);
int launch_pair_metrics(const float *X, const float *Y, int n, int B, int E, double *out_sums, double *out_w1, hipStream_t s);
int launch_pair_stats(const float *X, const float *Y, int n, int B, int E, double *out, hipStream_t s);
int launch_sample_mean(const float *traj, int n, int B, int E, float *out, hipStream_t s);
int launch_resize_bilinear(const float *in, float *out, int planes, int h, int w, int H, int W, hipStream_t s);
}  // namespace dt

using namespace dt;

// ------------------------------------------------------------------------------ profiler
namespace {
struct ProfRecord { hipEvent_t a, b; int cls; double flops, bytes; };
struct Profiler {
  std::mutex mu;          // launches may come from several host threads (one per stream)
  std::atomic<bool> on{false};   // read by every launch without the lock
  std::vector<ProfRecord> rec;
  std::vector<hipEvent_t> pool;     // events are recycled between sessions
  size_t used = 0;
  hipEvent_t get() {
    if (used == pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      pool.push_back(e);
    }
    return pool[used++];
  }
} g_prof;
const char *kClassName[KC_COUNT] = {
    "conv_gemm_kernel<128,128>", "conv_gemm_kernel<128,64>", "conv_gemm_kernel<64,128>", "conv_gemm_kernel<64,64>",
    "conv_gemm_bf16x6_kernel<128,128>", "conv_gemm_bf16x6_kernel<128,64>", "conv_gemm_bf16x6_kernel<64,128>",
    "conv_gemm_bf16x6_kernel<64,64>",
    "conv_strip_bf16x6_kernel<128,128>", "conv_strip_bf16x6_kernel<128,64>", "conv_strip_bf16x6_kernel<64,128>",
    "conv_strip_bf16x6_kernel<64,64>", "conv_strip_bf16x6_kernel<256,64>", "conv_strip_bf16x6_kernel<128,64,K2>",
    "conv_strip_bf16x6_kernel<64,64,K4>", "conv_strip_bf16x6_kernel<64,128,K2>", "splitk_epilogue_kernel", "first_conv_kernel", "maxpool_kernel", "upcat_kernel", "head_kernel", "head_upsample_kernel",
    "time_bias_kernel", "cfg_update_kernel", "traj_metrics_kernel", "wasserstein_kernel", "resampled_distance_kernel",
    "unet_fused_kernel", "pair_metrics_kernel"};
}  // namespace

namespace dt {
ProfileScope::ProfileScope(int cls, double flops, double bytes, hipStream_t s) : slot(-1), stream(s), end_event(nullptr) {
  if (!g_prof.on.load(std::memory_order_relaxed)) return;
  hipEvent_t a;
  {
    std::lock_guard<std::mutex> lock(g_prof.mu);
    if (!g_prof.on.load(std::memory_order_relaxed)) return;   // a concurrent dt_profile_end
    a = g_prof.get();
    hipEvent_t b = g_prof.get();
    if (!a || !b) return;
    g_prof.rec.push_back(ProfRecord{a, b, cls, flops, bytes});
    slot = (int)g_prof.rec.size() - 1;
    end_event = b;             // kept here: a concurrent dt_profile_begin may clear `rec` before the destructor runs
  }
  (void)hipEventRecord(a, s);
}
ProfileScope::~ProfileScope() {
  if (slot < 0) return;
  (void)hipEventRecord((hipEvent_t)end_event, stream);
}
}  // namespace dt

struct BlockW {
  int cin, cout;          // real channels
  int cin_p, cout_p, n_p; // padded
  int split_c, split_cp;  // concat split of the input channels (== cin, cin_p when no concat)
  int cin_w, cout_w;      // input channels per tap of the split-bf16 packs of conv1 (and the skip) / conv2: cin_p / cout_p padded to kChunkPad chunks
  bool has_res;
  float *w1, *w2, *wr;    // packed conv weights (fp32 tiles)
  float *w1b, *w2b, *wrb; // the same weights split into three bf16 planes (dt_conv_bf16.hip)
  float *s1, *h1, *s2, *h2, *sr, *hr;  // scale/shift per conv
  float *w3;              // enc1 only: skip weights for the in-epilogue 1x1 (n_p x 4)
  int tb_off;             // channel offset of this block in a time-bias row
};

// (tile, split) choice of the three conv slots (0 = 1x1 skip, 1 = conv1, 2 = conv2) of every block for
// one forward shape, filled in by dt_unet_autotune; absent shapes use heuristic_choice
struct TunedShape {
  int Bt, H, W;
  int imgs, single;   // how the Bt rows split into images (see ShapeInfo): enc1's launches run over the images
  ConvChoice c[kBlocks][3];
};

// an instantiated hipGraph of one dt_sample_trajectory call (every launch of the loop), keyed by all its arguments
struct LoopGraph {
  std::vector<unsigned char> key;
  hipGraphExec_t exec;
  hipGraph_t graph;
};

// how a forward shape (Bt rows) splits into images: recorded by every forward so that the tuning / reporting hooks, which are
// keyed by the row count alone, describe the launches the forward really issues (enc1 runs over the images, not the rows)
struct ShapeInfo { int Bt, H, W, imgs, single; };

struct dt_unet {
  std::vector<TunedShape> tuned;
  mutable std::vector<ShapeInfo> shapes;
  mutable std::mutex shape_mu;
  // replay cache of the sampler loop (not part of the handle's logical state, hence mutable + its own lock: the
  // sampler may be entered from several host threads); dropped whenever the launch plan changes
  mutable std::vector<LoopGraph> graphs;
  mutable std::mutex graph_mu;
  int precision;          // DT_PREC_*: which convolution arithmetic the heuristic / autotuner may use
  bool head_fusion = true;   // dec1.conv2's epilogue evaluates the final 1x1 head (dt_unet_set_head_fusion)
  // enc1 runs ONCE per step for all CFG passes (they share x): conv1 without the time bias over the B images, conv2 over the
  // B images with the passes' time biases entering as class-bias rows in its epilogue (tap_bias_kernel); DT_NO_SHARED_ENC1=1
  // at create time keeps the per-pass formulation (A/B runs, and the form the reference's operation order follows literally)
  bool share_enc1 = true;
  int tb_cols = 0;           // projected channels of a time-bias row; the class-bias columns follow
  dt_unet_desc desc;
  BlockW blk[kBlocks];
  int cp[4];              // padded dims
  int tb_stride;
  float *slab;            // one device allocation holding everything below
  size_t slab_floats;
  TembWeights tw;
  const float *final_w, *final_b;   // borrowed? no: copied into the slab
  // Small models (padded dims <= 32 / 64) at 16 x 16: the whole forward -- and the whole sampler loop -- is ONE launch of
  // unet_fused_kernel (dt_fused.hip) with every activation in LDS.  fused_ok: the packs below exist; fused_on: the switch
  // (dt_unet_set_fused; DT_NO_FUSED=1 at create time starts with it off).
  bool fused_ok = false, fused_on = false;
  int fused_G = 2;
  float *fused_slab = nullptr;      // packed fp32 weights of the fused kernel + its layer table
  FusedOp *fused_ops_dev = nullptr;
  int fused_n_ops = 0;
  FusedLds fused_lds{};
  int fused_par = 0, fused_n_par = 0;   // the parameter block inside the fused slab
};

namespace {

struct Bump {
  size_t off = 0;
  size_t take(size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; }   // 256-B aligned
};

bool use_fused(const dt_unet *u, int H, int W);
void fused_common(const dt_unet *u, FusedArgs &a, int B, int n_pass, int B_single, const float *tb, int tb_div);

// activation buffers of one forward, as float offsets into the workspace
struct Plan {
  size_t h[kBlocks], r[kBlocks], o[kBlocks];   // conv1 out, skip out, block out
  size_t pool[4], cat[3];
  size_t slab, lowres;                         // split-K partial sums; low-resolution head output
  size_t total;
  int H[kBlocks], W[kBlocks];                  // spatial size of each block
};

// level (power-of-two divisor of the image size) of the 8 blocks
const int kDiv[kBlocks] = {1, 2, 4, 8, 16, 8, 4, 2};

Plan make_plan(const dt_unet *u, int Bt, int H, int W) {
  Plan p{};
  Bump b;
  size_t slab = 0;
  for (int j = 0; j < kBlocks; ++j) {
    const int h = H / kDiv[j], w = W / kDiv[j];
    p.H[j] = h; p.W[j] = w;
    const size_t px = (size_t)Bt * h * w;
    if (j >= 1 && j <= 4) p.pool[j - 1] = b.take(px * u->blk[j].cin_p);
    if (j >= 5) p.cat[j - 5] = b.take(px * u->blk[j].cin_p);
    p.h[j] = b.take(px * u->blk[j].cout_p);
    p.r[j] = (u->blk[j].has_res && j > 0) ? b.take(px * u->blk[j].cout_p) : 0;
    p.o[j] = b.take(px * u->blk[j].cout_p);
    if (j > 0 && px <= (size_t)kSplitMaxRows) {
      const size_t need = (size_t)9 * px * u->blk[j].cout_p;    // room for the deepest split
      if (need > slab) slab = need;
    }
  }
  p.slab = b.take(slab);
  p.lowres = b.take((size_t)Bt * (H / 2) * (W / 2) * 4);
  p.total = b.off;
  return p;
}

void drop_graphs(dt_unet *u) {
  std::lock_guard<std::mutex> lock(u->graph_mu);
  for (LoopGraph &g : u->graphs) { (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph); }
  u->graphs.clear();
}

const TunedShape *find_tuned(const dt_unet *u, int Bt, int H, int W, int imgs, int single) {
  for (const TunedShape &t : u->tuned)
    if (t.Bt == Bt && t.H == H && t.W == W && t.imgs == imgs && t.single == single) return &t;
  return nullptr;
}

// Parameters of conv slot `slot` (0 = 1x1 skip, 1 = conv1, 2 = conv2) of block j; returns false when the
// block has no such launch (identity skips, and enc1 whose skip is recomputed in conv2's epilogue).
void shape_images(const dt_unet *u, int Bt, int H, int W, int &imgs, int &single) {
  {
    std::lock_guard<std::mutex> lock(u->shape_mu);
    for (const ShapeInfo &si : u->shapes)
      if (si.Bt == Bt && si.H == H && si.W == W) { imgs = si.imgs; single = si.single; return; }
  }
  imgs = Bt % 2 == 0 ? Bt / 2 : Bt;   // never run yet: the sampler's two-pass CFG shape
  single = 0;
}

void note_shape(const dt_unet *u, int Bt, int H, int W, int imgs, int single) {
  std::lock_guard<std::mutex> lock(u->shape_mu);
  for (ShapeInfo &si : u->shapes)
    if (si.Bt == Bt && si.H == H && si.W == W) { si.imgs = imgs; si.single = single; return; }
  u->shapes.push_back(ShapeInfo{Bt, H, W, imgs, single});
}

// x_imgs images make up the Bt rows: every image once (pass 0), then the images from x_single on a second time (pass 1) when
// Bt > x_imgs.  x_imgs <= 0 (tuning / reporting contexts): what the last forward of this row count used.
bool conv_slot(const dt_unet *u, int j, int slot, const float *in, float *ws, const Plan &pl, int Bt, const float *tb,
               int tb_div, const ConvChoice *choice, ConvParams &p, int x_imgs = 0, int x_single = 0) {
  if (x_imgs <= 0) shape_images(u, Bt, pl.H[0], pl.W[0], x_imgs, x_single);
  const BlockW &k = u->blk[j];
  const int h = pl.H[j], w = pl.W[j];
  const bool dot = h == 1 && w == 1;   // a 1x1 image only ever sees the centre tap of a padded 3x3 kernel
  p = ConvParams{};
  p.M = Bt * h * w; p.H = h; p.W = w;
  p.cin_p = k.cin_p; p.cout_p = k.cout_p; p.n_p = k.n_p;
  p.cin_real = k.cin; p.cout_real = k.cout;
  p.tb_stride = u->tb_stride; p.m_per_tb = h * w * tb_div;
  p.slab = ws + pl.slab;
  p.in = in;
  int taps = 1;
  if (slot == 0) {
    if (!k.has_res || j == 0) return false;
    p.w = k.wr; p.scale = k.sr; p.shift = k.hr; p.out = ws + pl.r[j];
    p.ksize = 1; p.tap_lo = 0; p.tap_hi = 1; p.relu = 0;
  } else if (slot == 1) {
    p.w = k.w1; p.scale = k.s1; p.shift = k.h1; p.tb = tb + k.tb_off; p.out = ws + pl.h[j]; p.relu = 1;
    if (j == 0) {
      return false;   // enc1.conv1 is the direct first-layer kernel (launch_first_conv), not an implicit GEMM
    } else {
      p.ksize = 3; p.tap_lo = dot ? 4 : 0; p.tap_hi = dot ? 5 : 9;
      taps = dot ? 1 : 9;
    }
  } else {
    p.in = ws + pl.h[j]; p.cin_p = k.cout_p; p.cin_real = k.cout;
    p.w = k.w2; p.scale = k.s2; p.shift = k.h2; p.out = ws + pl.o[j]; p.relu = 1;
    p.ksize = 3; p.tap_lo = dot ? 4 : 0; p.tap_hi = dot ? 5 : 9;
    taps = dot ? 1 : 9;
    if (j == 0) {
      // the C-channel skip of enc1 is recomputed in the epilogue from the patches' centre taps (k = 9c+4)
      p.x3 = in; p.w3 = k.w3; p.x3_hw = h * w; p.x3_imgs = x_imgs; p.x3_c = u->desc.channels;   // `in` is the NCHW image
      taps = 1;   // keeps the fused (non-split) epilogue
      if (u->share_enc1 && (Bt + x_single) % x_imgs == 0) {   // one launch over the x_imgs images for all their passes
        p.M = x_imgs * h * w;
        p.n_dup = (Bt + x_single) / x_imgs; p.dup_rows = p.M; p.dup_skip = x_single * h * w; p.tbc = tb + u->tb_cols;
        p.skip_out = u->head_fusion ? 1 : 0;   // decided below once pool_out is known: only the pool reads enc1's output
      }
    } else {
      p.add = k.has_res ? ws + pl.r[j] : in;   // identity skip: cin_p == cout_p
    }
  }
  ConvChoice c = choice ? *choice : heuristic_choice(p.M, p.n_p, taps);
  if (!choice) {   // untuned default per mode; the strip kernel wherever the full 3x3 walk runs (uniformly >= the plain one)
    c.prec = u->precision == DT_PREC_FP32 ? 0 : 1;
    if (c.prec == 1 && p.ksize == 3 && p.tap_hi - p.tap_lo == 9 && strip_admissible(p.W, 64, 64, 3)) {
      c.prec = 3;
      const int cc = p.cin_p >> 4;   // tap groups 3 / 9 become channel-chunk groups 4 / 8 where they divide
      c.splits = c.splits == 9 ? (cc % 8 == 0 ? 8 : (cc % 4 == 0 ? 4 : 1)) : (c.splits == 3 ? (cc % 4 == 0 ? 4 : (cc % 2 == 0 ? 2 : 1)) : 1);
    }
  }
  if (c.prec == 2) c.prec = 1;
  if (c.prec >= 3 && (p.tap_hi - p.tap_lo != 9 || !strip_admissible(p.W, 64, 64, 3))) c.prec = 1;   // full 3x3 walks of rows <= 63 px only
  if (c.prec == 5 && (c.bm > 128 || (c.bm == 128 && c.bn == 128))) c.prec = 4;   // K split across waves: tiles below 128 x 128
  // weight chunks per tap of the packs: the split-bf16 packs carry zero chunks up to a multiple of kChunkPad, so layers with an
  // odd chunk count also run the kernels that multiply 2 / 4 chunks per step
  const int cw_main = round_up(p.cin_p, 16 * kChunkPad) >> 4;
  if (c.bm == 256 && c.prec < 3) c.bm = 128;                        // the 256-row tile exists in the strip kernels only
  if (p.x3 || j == 0 || p.M > kSplitMaxRows) c.splits = 1;          // (enc1: no slab; its conv2 keeps the fused x3 epilogue)
  if (c.prec >= 3 ? !chunks_fit(p.cin_p >> 4, cw_main, c.splits * strip_kc(c.prec, c.bm, c.bn))
                  : (((p.tap_hi - p.tap_lo) * (p.cin_p >> 4)) % c.splits != 0)) c.splits = 1;
  p.bm = c.bm; p.bn = c.bn; p.splits = c.splits; p.prec = c.prec;
  p.ccw = c.prec >= 1 ? cw_main : p.cin_p >> 4;
  // encoder blocks enc1..enc4 feed a 2x2 max pool: folded into this launch's staged epilogue where a 32-row tile holds
  // whole row pairs (W a power of two <= 16); split launches pool in their slab-summing epilogue kernel instead
  if (slot == 2 && j <= 3 && h % 2 == 0 && w % 2 == 0 && (c.splits > 1 || (w <= 16 && (w & (w - 1)) == 0))) p.pool_out = ws + pl.pool[j];
  if (!p.pool_out) p.skip_out = 0;            // a separate pooling launch reads the full-resolution output
  if (c.prec >= 1) p.w = slot == 0 ? k.wrb : (slot == 1 ? k.w1b : k.w2b);
  // dec1.conv2 also evaluates the final 1x1 head when its workgroups hold whole rows (one N tile, no split): the head
  // is dec1's only consumer, so dec1's own output is then never written
  if (slot == 2 && j == kBlocks - 1 && u->head_fusion && c.splits == 1 && p.n_p == c.bn && u->desc.channels <= 3) {
    p.head_w = u->final_w; p.head_b = u->final_b; p.head_out = ws + pl.lowres;
    p.head_c = u->desc.channels; p.head_cin = u->desc.dims[0];
  }
  if (slot == 2 && c.fuse && c.splits == 1 && k.has_res && j > 0) {
    // conv2 with the block's 1x1 skip folded into its K walk (the slot-0 launch is then skipped)
    p.add = nullptr;
    p.in2 = in; p.w2 = c.prec >= 1 ? k.wrb : k.wr; p.bias2 = k.hr; p.cin2_p = k.cin_p; p.cin2_real = k.cin;
    p.ccw2 = c.prec >= 1 ? k.cin_w >> 4 : k.cin_p >> 4;
  }
  return true;
}

// Decoder block j reads the concat [upsampled | skip].  When both of its readers are strip launches (conv1, and the
// 1x1 skip conv folded into conv2), they fetch the skip half straight from the encoder's output and the concat
// buffer only ever holds the upsampled half.
bool concat_in_place(const dt_unet *u, int j, float *ws, const Plan &pl, int Bt, const float *tb, int tb_div,
                     const TunedShape *tuned) {
  if (j < 5 || !u->blk[j].has_res) return false;
  ConvParams c1, c2;
  if (!conv_slot(u, j, 1, ws, ws, pl, Bt, tb, tb_div, tuned ? &tuned->c[j][1] : nullptr, c1)) return false;
  if (!conv_slot(u, j, 2, ws, ws, pl, Bt, tb, tb_div, tuned ? &tuned->c[j][2] : nullptr, c2)) return false;
  return c1.prec >= 3 && c1.prec <= 5 && c2.prec >= 3 && c2.prec <= 5 && c2.in2 != nullptr;
}

int run_block(const dt_unet *u, int j, const float *in, float *ws, const Plan &pl, int Bt, const float *tb, int tb_div,
              const TunedShape *tuned, hipStream_t s, int x_imgs, int x_single) {
  ConvParams p;
  bool fused_skip = false;
  if (u->blk[j].has_res && j > 0) {   // is conv2 going to fold the skip in?
    ConvParams c2;
    conv_slot(u, j, 2, in, ws, pl, Bt, tb, tb_div, tuned ? &tuned->c[j][2] : nullptr, c2);
    fused_skip = c2.in2 != nullptr;
  }
  const bool in_place = concat_in_place(u, j, ws, pl, Bt, tb, tb_div, tuned);
  if (j == 0) {
    const BlockW &k = u->blk[0];
    const bool shared = u->share_enc1 && (Bt + x_single) % x_imgs == 0;   // then the passes' time biases enter in conv2's epilogue
    if (!shared && x_single) return DT_E_ARG;                // mixed batches exist in the shared-enc1 formulation only
    const int st = launch_first_conv(in, k.w1, k.s1, k.h1, shared ? nullptr : tb + k.tb_off, u->tb_stride, tb_div, ws + pl.h[0], x_imgs,
                                     shared ? 1 : Bt / x_imgs, u->desc.channels, pl.H[0], pl.W[0], k.cout, k.cout_p, s);
    if (st) return st;
  }
  for (int slot = 0; slot < 3; ++slot) {
    if (slot == 0 && fused_skip) continue;
    if (!conv_slot(u, j, slot, in, ws, pl, Bt, tb, tb_div, tuned ? &tuned->c[j][slot] : nullptr, p, x_imgs, x_single)) continue;
    if (in_place) {
      const int skip = 8 - j;            // dec3<-enc4(3), dec2<-enc3(2), dec1<-enc2(1)
      p.cc_a = u->blk[j].split_cp >> 4;
      p.b_stride = u->blk[skip].cout_p;
      if (slot == 1) p.in_b = ws + pl.o[skip];
      if (slot == 2) p.in2_b = ws + pl.o[skip];
    }
    const int st = launch_conv(p, s);
    if (st) return st;
  }
  return DT_OK;
}

// B images; the first B_single of them take one pass, the others n_pass (B_single > 0 needs n_pass == 2): rows
// [pass 0 of all B images | pass 1 of images B_single .. B-1]
int forward_impl(const dt_unet *u, const float *x, int B, int n_pass, int H, int W, const float *tb, int tb_div,
                 float *eps, float *ws, size_t ws_bytes, hipStream_t s, int B_single = 0) {
  if (!u || !x || !tb || !ws) return DT_E_NULL;
  if (B < 1 || n_pass < 1 || tb_div < 1 || H < 16 || W < 16 || H % 16 || W % 16) return DT_E_SHAPE;
  if (B_single < 0 || B_single >= B || (B_single && n_pass != 2)) return DT_E_ARG;
  const int Bt = B * n_pass - B_single * (n_pass - 1);
  note_shape(u, Bt, H, W, B, B_single);
  const Plan pl = make_plan(u, Bt, H, W);
  if (pl.total * sizeof(float) > ws_bytes) return DT_E_WORKSPACE;
  if (use_fused(u, H, W) && eps) {           // small model: the whole forward is one launch (dt_fused.hip)
    if (Bt % tb_div) return DT_E_ARG;
    FusedArgs a{};
    fused_common(u, a, B, n_pass, B_single, tb, tb_div);
    a.mode = FUSED_FORWARD; a.x = x; a.eps = eps;
    return launch_unet_fused(a, s);
  }
  const TunedShape *tuned = find_tuned(u, Bt, H, W, B, B_single);
  int st = DT_OK;
  const float *cur = x;                     // enc1 reads the NCHW image itself (first-layer kernel, skip in conv2's epilogue)
  for (int j = 0; j < kBlocks; ++j) {
    if (j >= 1 && j <= 4) {        // encoder: pool the previous block's output (unless its conv2 already did)
      ConvParams prev;
      conv_slot(u, j - 1, 2, ws, ws, pl, Bt, tb, tb_div, tuned ? &tuned->c[j - 1][2] : nullptr, prev);
      if (!prev.pool_out)
      st = launch_maxpool(ws + pl.o[j - 1], ws + pl.pool[j - 1], Bt,
                          pl.H[j - 1], pl.W[j - 1], u->blk[j - 1].cout_p, s);
      if (st) return st;
      cur = ws + pl.pool[j - 1];
    } else if (j >= 5) {           // decoder: upsample previous output, concat the matching encoder output
      const int skip = 8 - j;      // dec3<-enc4(3), dec2<-enc3(2), dec1<-enc2(1)
      const bool in_place = concat_in_place(u, j, ws, pl, Bt, tb, tb_div, tuned);
      st = launch_upcat(ws + pl.o[j - 1], in_place ? nullptr : ws + pl.o[skip], ws + pl.cat[j - 5],
                        Bt, pl.H[j - 1], pl.W[j - 1], u->blk[j - 1].cout_p, u->blk[skip].cout_p, s);
      if (st) return st;
      cur = ws + pl.cat[j - 5];
    }
    st = run_block(u, j, cur, ws, pl, Bt, tb, tb_div, tuned, s, B, B_single);
    if (st) return st;
  }
  {   // head at the low resolution (unless dec1.conv2's epilogue already produced it), then the 3-channel upsample
    ConvParams last;
    conv_slot(u, kBlocks - 1, 2, ws, ws, pl, Bt, tb, tb_div, tuned ? &tuned->c[kBlocks - 1][2] : nullptr, last);
    if (!last.head_out)
      st = launch_head(ws + pl.o[7], u->final_w, u->final_b, ws + pl.lowres, Bt, pl.H[7], pl.W[7], u->blk[7].cout_p,
                       u->desc.channels, u->desc.dims[0], s);
    if (st) return st;
  }
  if (!eps) return DT_OK;                  // the sampler's fused update interpolates the low-resolution output itself
  return launch_head_upsample(ws + pl.lowres, eps, Bt, pl.H[7], pl.W[7], u->desc.channels, s);
}

// float offset of the low-resolution head output [Bt][H/2][W/2][4] inside the workspace
size_t lowres_offset(const dt_unet *u, int Bt, int H, int W) { return make_plan(u, Bt, H, W).lowres; }


// ---- the fused small-model path (dt_fused.hip): packs + layer table at create time, one launch per forward / sampler call
int build_fused(dt_unet *u, const float *const *bt, hipStream_t s) {
  const int C = u->desc.channels, c0p = u->cp[0], c1p = u->cp[1];
  if (u->cp[2] != c1p || u->cp[3] != c1p || u->desc.dims[2] != u->desc.dims[1] || u->desc.dims[3] != u->desc.dims[1]) return DT_OK;
  if (!fused_eligible(C, c0p, c1p)) return DT_OK;
  const int G = u->fused_G;
  if (fused_lds_bytes(G, c0p, c1p) > 160 * 1024) return DT_OK;
  // one slab: the parameter block (per block the folded BN vectors of conv1 / conv2 and the skip conv's bias, cout_p floats each,
  // then enc1's image-skip rows: the kernel copies this block into LDS), the packed weights of conv1 (blocks 1..7), conv2
  // (all) and the 1x1 skip convs (enc1's image skip aside), enc1.conv1 as two K chunks, and the layer table
  Bump bump;
  const int tb_cols = 2 * c0p + 6 * c1p;
  const size_t o_par = bump.take((size_t)5 * tb_cols + 4 * c0p);
  size_t o1[kBlocks] = {}, o2[kBlocks] = {}, orr[kBlocks] = {};
  for (int j = 0; j < kBlocks; ++j) {
    const BlockW &k = u->blk[j];
    if (j > 0) o1[j] = bump.take((size_t)9 * k.cin_p * k.cout_p);
    o2[j] = bump.take((size_t)9 * k.cout_p * k.cout_p);
    if (k.has_res && j > 0) orr[j] = bump.take((size_t)k.cin_p * k.cout_p);
  }
  const size_t o_wf = bump.take((size_t)2 * (c0p / 16) * 256);
  const size_t o_ops = bump.take((sizeof(FusedOp) * kFusedMaxOps + 3) / 4);
  if (bump.off >= (1u << 30)) return DT_OK;                     // offsets are ints
  hipError_t e = hipMalloc((void **)&u->fused_slab, bump.off * sizeof(float));
  if (e != hipSuccess) return (int)e;
  float *F = u->fused_slab;
  FusedModel m{};
  m.c0p = c0p; m.c1p = c1p; m.wf = (int)o_wf; m.w3 = 5 * tb_cols;
  int st = DT_OK;
  auto d2d = [&](size_t off, const float *src, size_t n) {
    if (st == DT_OK) {
      const hipError_t ee = hipMemcpyAsync(F + off, src, n * sizeof(float), hipMemcpyDeviceToDevice, s);
      if (ee != hipSuccess) st = (int)ee;
    }
  };
  st = launch_pack_fused_first(bt[DT_BT_CONV1_W], F + o_wf, u->blk[0].cout, C, c0p / 16, s);
  d2d(o_par + 5 * tb_cols, u->blk[0].w3, (size_t)4 * c0p);
  int pj = 0;                                                   // the block's offset inside the parameter block
  for (int j = 0; j < kBlocks && st == DT_OK; ++j) {
    const BlockW &k = u->blk[j];
    const float *const *t = bt + j * DT_BT_COUNT;
    FusedBlockW &f = m.blk[j];
    const int cp = k.cout_p;
    d2d(o_par + pj, k.s1, cp); d2d(o_par + pj + cp, k.h1, cp); d2d(o_par + pj + 2 * cp, k.s2, cp); d2d(o_par + pj + 3 * cp, k.h2, cp);
    if (k.has_res && j > 0) d2d(o_par + pj + 4 * cp, k.hr, cp);
    f.tb_off = k.tb_off;
    f.c1 = FusedConvW{(int)o1[j], pj, pj + cp};
    f.c2 = FusedConvW{(int)o2[j], pj + 2 * cp, pj + 3 * cp};
    f.cr = FusedConvW{(int)orr[j], 0, pj + 4 * cp};
    pj += 5 * cp;
    if (j > 0) st = launch_pack_fused_conv(t[DT_BT_CONV1_W], F + o1[j], k.cout, k.cin, 9, k.cin_p / 16, k.cout_p / 16, k.split_c, k.split_cp, s);
    if (!st) st = launch_pack_fused_conv(t[DT_BT_CONV2_W], F + o2[j], k.cout, k.cout, 9, k.cout_p / 16, k.cout_p / 16, k.cout, k.cout_p, s);
    if (!st && k.has_res && j > 0)
      st = launch_pack_fused_conv(t[DT_BT_RES_W], F + orr[j], k.cout, k.cin, 1, k.cin_p / 16, k.cout_p / 16, k.split_c, k.split_cp, s);
  }
  if (st == DT_OK && hipMemsetAsync(F + o_par + 4 * c0p, 0, sizeof(float) * c0p, s) != hipSuccess) st = (int)hipGetLastError();   // (enc1 has no skip-conv bias)
  u->fused_par = (int)o_par; u->fused_n_par = 5 * tb_cols + 4 * c0p;
  if (st != DT_OK) return st;
  FusedOp ops[kFusedMaxOps];
  u->fused_n_ops = fused_ops(m, G, ops);
  u->fused_ops_dev = reinterpret_cast<FusedOp *>(F + o_ops);
  // (a synchronous copy: `ops` lives on this stack frame)
  e = hipMemcpy(u->fused_ops_dev, ops, sizeof(FusedOp) * u->fused_n_ops, hipMemcpyHostToDevice);
  if (e != hipSuccess) return (int)e;
  u->fused_lds = fused_lds(G, c0p, c1p);
  u->fused_ok = true;
  u->fused_on = getenv("DT_NO_FUSED") == nullptr;
  return DT_OK;
}

// (head fusion off = the test hook that materialises every block output in the workspace: only the layered path has those)
bool use_fused(const dt_unet *u, int H, int W) { return u->fused_ok && u->fused_on && u->head_fusion && H == 16 && W == 16; }

void fused_common(const dt_unet *u, FusedArgs &a, int B, int n_pass, int B_single, const float *tb, int tb_div) {
  a.ops = u->fused_ops_dev; a.n_ops = u->fused_n_ops; a.fbase = u->fused_slab; a.par = u->fused_par; a.n_par = u->fused_n_par;
  a.head_w = u->final_w; a.head_b = u->final_b;
  a.tb = tb; a.tb_stride = u->tb_stride; a.tb_div = tb_div;
  a.lds = u->fused_lds; a.G = u->fused_G;
  a.C = u->desc.channels; a.c0 = u->desc.dims[0]; a.c0p = u->cp[0];
  a.B = B; a.n_pass = n_pass; a.B_single = B_single;
  a.tb_rows = (B * n_pass - B_single * (n_pass - 1)) / tb_div;
  a.flops_per_row = fused_flops_per_row(a.C, u->desc.dims[0], u->desc.dims[1]);
}
}  // namespace

extern "C" {

int dt_abi_version(void) { return DT_ABI_VERSION; }

const char *dt_status_string(int st) {
  switch (st) {
    case DT_OK: return "ok";
    case DT_E_NULL: return "required pointer is NULL";
    case DT_E_SHAPE: return "unsupported or inconsistent shape";
    case DT_E_ARG: return "bad enum or count";
    case DT_E_WORKSPACE: return "workspace too small";
    default: return st > 0 ? hipGetErrorString((hipError_t)st) : "unknown dt status";
  }
}

int dt_unet_create(const dt_unet_desc *desc, const float *const *bt, const float *const *gt, void *stream,
                   dt_unet **out) {
  if (!desc || !bt || !gt || !out) return DT_E_NULL;
  if (desc->channels < 1 || desc->channels > 3 || desc->temb_dim < 2) return DT_E_SHAPE;
  for (int i = 0; i < 4; ++i)
    if (desc->dims[i] < 1) return DT_E_SHAPE;
  for (int i = 0; i < DT_GT_COUNT; ++i)
    if (!gt[i]) return DT_E_NULL;
  hipStream_t s = (hipStream_t)stream;
  dt_unet *u = new (std::nothrow) dt_unet();
  if (!u) return (int)hipErrorOutOfMemory;
  u->desc = *desc;
  u->precision = DT_PREC_AUTO;
  const int C = desc->channels, D = desc->temb_dim;
  const int *d = desc->dims;
  for (int i = 0; i < 4; ++i) u->cp[i] = round_up(d[i], kChanPad);
  // (cin, cout) and the concat split of the eight blocks -- models.py:138-154
  const int cin[kBlocks] = {C, d[0], d[1], d[2], d[3], d[3] + d[3], d[2] + d[2], d[1] + d[1]};
  const int cout[kBlocks] = {d[0], d[1], d[2], d[3], d[3], d[2], d[1], d[0]};
  const int up_c[kBlocks] = {0, 0, 0, 0, 0, d[3], d[2], d[1]};   // channels coming from the upsampled branch
  Bump bump;
  size_t o_w1[kBlocks], o_w2[kBlocks], o_wr[kBlocks], o_ss[kBlocks];
  size_t o_w1b[kBlocks], o_w2b[kBlocks], o_wrb[kBlocks];   // bf16x3 packs: 6 bytes per weight = 1.5 floats
  int tb = 0;
  for (int j = 0; j < kBlocks; ++j) {
    BlockW &k = u->blk[j];
    k.cin = cin[j]; k.cout = cout[j];
    k.cout_p = round_up(cout[j], kChanPad);
    k.n_p = round_up(cout[j], kNPad);
    if (j >= 5) {
      k.split_c = up_c[j]; k.split_cp = round_up(up_c[j], kChanPad);
      k.cin_p = k.split_cp + round_up(cin[j] - up_c[j], kChanPad);
    } else {
      k.cin_p = round_up(cin[j], kChanPad);                    // (enc1.conv1 is the direct first-layer kernel)
      k.split_c = cin[j]; k.split_cp = k.cin_p;
    }
    k.has_res = cin[j] != cout[j];
    k.cin_w = round_up(k.cin_p, 16 * kChunkPad); k.cout_w = round_up(k.cout_p, 16 * kChunkPad);
    for (int t = 0; t < DT_BT_COUNT; ++t) {
      const bool optional = t == DT_BT_RES_W || t == DT_BT_RES_B;
      if (!bt[j * DT_BT_COUNT + t] && (!optional || k.has_res)) { delete u; return DT_E_NULL; }
    }
    o_w1[j] = bump.take(j == 0 ? (size_t)9 * C * k.cout_p : (size_t)9 * k.cin_p * k.n_p);   // enc1: wf[9C][cout_p]
    o_w2[j] = bump.take((size_t)9 * k.cout_p * k.n_p);
    o_wr[j] = k.has_res ? bump.take(j == 0 ? (size_t)4 * k.n_p : (size_t)k.cin_p * k.n_p) : 0;
    o_ss[j] = bump.take((size_t)6 * k.n_p);
    o_w1b[j] = j == 0 ? 0 : bump.take((size_t)9 * k.cin_w * k.n_p * 3 / 2);
    o_w2b[j] = bump.take((size_t)9 * k.cout_w * k.n_p * 3 / 2);
    o_wrb[j] = (k.has_res && j > 0) ? bump.take((size_t)k.cin_w * k.n_p * 3 / 2) : 0;
    k.tb_off = tb;
    tb += k.cout_p;
  }
  u->share_enc1 = getenv("DT_NO_SHARED_ENC1") == nullptr;
  const int c0p = u->blk[0].cout_p;
  u->tb_cols = tb;
  u->tb_stride = tb + (u->share_enc1 ? 9 * c0p : 0);          // [projected channels | nine class-bias vectors of enc1.conv2]
  const size_t o_w2t = u->share_enc1 ? bump.take((size_t)9 * c0p * c0p) : 0;
  const int half = (D / 2 > 1 ? D / 2 : 1);
  const size_t o_wt = bump.take((size_t)tb * D), o_bt = bump.take(tb);
  const size_t o_w1g = bump.take((size_t)D * D), o_b1g = bump.take(D), o_wc0 = bump.take(D), o_bc0 = bump.take(D);
  const size_t o_wc2 = bump.take((size_t)D * D), o_bc2 = bump.take(D), o_fr = bump.take(half);
  const size_t o_fw = bump.take((size_t)C * d[0]), o_fb = bump.take(C);
  u->slab_floats = bump.off;
  hipError_t e = hipMalloc((void **)&u->slab, u->slab_floats * sizeof(float));
  if (e != hipSuccess) { delete u; return (int)e; }
  float *S = u->slab;
  int st = DT_OK;
  auto copy = [&](size_t off, const float *src, size_t n) {
    if (st == DT_OK) {
      hipError_t ee = hipMemcpyAsync(S + off, src, n * sizeof(float), hipMemcpyDeviceToDevice, s);
      if (ee != hipSuccess) st = (int)ee;
    }
  };
  for (int j = 0; j < kBlocks && st == DT_OK; ++j) {
    BlockW &k = u->blk[j];
    const float *const *t = bt + j * DT_BT_COUNT;
    k.w1 = S + o_w1[j]; k.w2 = S + o_w2[j];
    k.wr = (k.has_res && j > 0) ? S + o_wr[j] : nullptr;
    k.w3 = j == 0 ? S + o_wr[j] : nullptr;
    k.w1b = S + o_w1b[j]; k.w2b = S + o_w2b[j]; k.wrb = (k.has_res && j > 0) ? S + o_wrb[j] : nullptr;
    float *ss = S + o_ss[j];
    k.s1 = ss; k.h1 = ss + k.n_p; k.s2 = ss + 2 * k.n_p; k.h2 = ss + 3 * k.n_p; k.sr = ss + 4 * k.n_p; k.hr = ss + 5 * k.n_p;
    if (j == 0)
      st = launch_pack_first_conv(t[DT_BT_CONV1_W], k.w1, k.cout, C, k.cout_p, s);
    else
      st = launch_pack_conv(t[DT_BT_CONV1_W], k.w1, k.cout, k.cin, 3, k.cin_p, k.n_p, k.split_c, k.split_cp, s);
    if (!st) st = launch_pack_conv(t[DT_BT_CONV2_W], k.w2, k.cout, k.cout, 3, k.cout_p, k.n_p, k.cout, k.cout_p, s);
    if (!st && k.has_res && j > 0)
      st = launch_pack_conv(t[DT_BT_RES_W], k.wr, k.cout, k.cin, 1, k.cin_p, k.n_p, k.split_c, k.split_cp, s);
    if (!st && j > 0) st = launch_pack_conv_bf16x3(t[DT_BT_CONV1_W], k.w1b, k.cout, k.cin, 3, k.cin_p, k.cin_w, k.n_p, k.split_c, k.split_cp, s);
    if (!st) st = launch_pack_conv_bf16x3(t[DT_BT_CONV2_W], k.w2b, k.cout, k.cout, 3, k.cout_p, k.cout_w, k.n_p, k.cout, k.cout_p, s);
    if (!st && k.has_res && j > 0)
      st = launch_pack_conv_bf16x3(t[DT_BT_RES_W], k.wrb, k.cout, k.cin, 1, k.cin_p, k.cin_w, k.n_p, k.split_c, k.split_cp, s);
    if (!st && j == 0) st = launch_pack_res3(t[DT_BT_RES_W], t[DT_BT_RES_B], k.w3, k.cout, C, k.n_p, s);
    if (!st && j == 0 && u->share_enc1) st = launch_pack_tap_major(t[DT_BT_CONV2_W], S + o_w2t, k.cout, k.cout, k.cout_p, s);
    if (!st) st = launch_fold_bn(t[DT_BT_CONV1_B], t[DT_BT_BN1_G], t[DT_BT_BN1_B], t[DT_BT_BN1_MEAN], t[DT_BT_BN1_VAR],
                                 k.s1, k.h1, k.cout, k.n_p, s);
    if (!st) st = launch_fold_bn(t[DT_BT_CONV2_B], t[DT_BT_BN2_G], t[DT_BT_BN2_B], t[DT_BT_BN2_MEAN], t[DT_BT_BN2_VAR],
                                 k.s2, k.h2, k.cout, k.n_p, s);
    if (!st && k.has_res && j > 0)
      st = launch_fold_bn(t[DT_BT_RES_B], nullptr, nullptr, nullptr, nullptr, k.sr, k.hr, k.cout, k.n_p, s);
    if (!st) st = launch_pack_linear_rows(t[DT_BT_TIME_W], t[DT_BT_TIME_B], S + o_wt + (size_t)k.tb_off * D,
                                          S + o_bt + k.tb_off, k.cout, D, k.cout_p, s);
  }
  copy(o_w1g, gt[DT_GT_TIME1_W], (size_t)D * D); copy(o_b1g, gt[DT_GT_TIME1_B], D);
  copy(o_wc0, gt[DT_GT_COND0_W], D); copy(o_bc0, gt[DT_GT_COND0_B], D);
  copy(o_wc2, gt[DT_GT_COND2_W], (size_t)D * D); copy(o_bc2, gt[DT_GT_COND2_B], D);
  copy(o_fr, gt[DT_GT_FREQS], half);
  copy(o_fw, gt[DT_GT_FINAL_W], (size_t)C * d[0]); copy(o_fb, gt[DT_GT_FINAL_B], C);
  if (st != DT_OK) { (void)hipFree(u->slab); delete u; return st; }
  u->tw = TembWeights{S + o_fr, S + o_w1g, S + o_b1g, S + o_wc0, S + o_bc0, S + o_wc2, S + o_bc2, S + o_wt, S + o_bt,
                      D, half, u->tb_stride, tb, u->share_enc1 ? S + o_w2t : nullptr, c0p};
  u->final_w = S + o_fw; u->final_b = S + o_fb;
  st = build_fused(u, bt, s);
  if (st != DT_OK) { (void)hipFree(u->slab); if (u->fused_slab) (void)hipFree(u->fused_slab); delete u; return st; }
  *out = u;
  return DT_OK;
}

void dt_unet_destroy(dt_unet *h) {
  if (!h) return;
  drop_graphs(h);
  if (h->slab) (void)hipFree(h->slab);
  if (h->fused_slab) (void)hipFree(h->fused_slab);
  delete h;
}

int dt_unet_time_bias_stride(const dt_unet *h) { return h ? h->tb_stride : DT_E_NULL; }

int dt_unet_time_bias(const dt_unet *h, const int32_t *t, const float *cond, const uint8_t *present, int rows,
                      float *out, void *stream) {
  if (!h || !t || !out) return DT_E_NULL;
  if (rows < 0) return DT_E_ARG;
  return launch_time_bias(h->tw, t, cond, present, rows, out, (hipStream_t)stream);
}

size_t dt_unet_workspace_bytes(const dt_unet *h, int batch_total, int H, int W) {
  if (!h || batch_total < 1 || H < 16 || W < 16 || H % 16 || W % 16) return 0;
  return make_plan(h, batch_total, H, W).total * sizeof(float);
}

int dt_unet_forward(const dt_unet *h, const float *x, int B, int n_pass, int H, int W, const float *tb, int tb_div,
                    float *eps, void *ws, size_t ws_bytes, void *stream) {
  if (!eps) return DT_E_NULL;
  return forward_impl(h, x, B, n_pass, H, W, tb, tb_div, eps, (float *)ws, ws_bytes, (hipStream_t)stream);
}

// Times every admissible (tile, tap split) of every conv launch of one forward shape with HIP events on
// `stream` (synchronises; call it outside hot loops and graph captures) and records the fastest.
// Activations in the workspace are whatever the last forward left there (run one first: all-zero or
// garbage operands would let the chip clock differently from real data).
int dt_unet_autotune(dt_unet *h, int batch_total, int H, int W, void *workspace, size_t ws_bytes, void *stream) {
  if (!h || !workspace) return DT_E_NULL;
  if (batch_total < 1 || H < 16 || W < 16 || H % 16 || W % 16) return DT_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const Plan pl = make_plan(h, batch_total, H, W);
  if (pl.total * sizeof(float) > ws_bytes) return DT_E_WORKSPACE;
  float *ws = (float *)workspace;
  hipEvent_t e0, e1;
  DT_HIP_TRY(hipEventCreate(&e0));
  DT_HIP_TRY(hipEventCreate(&e1));
  TunedShape t{};
  t.Bt = batch_total; t.H = H; t.W = W;
  shape_images(h, batch_total, H, W, t.imgs, t.single);   // the split of the last forward with this row count
  const float *tb = h->slab;   // any readable floats: only timing matters here
  int st = DT_OK;
  {   // The clocks of an idle GPU take tens of milliseconds of load to settle (the same launch measures 15 % slower
      // cold): run one mid-sized layer for a while first so that the first candidates are not timed against a ramp.
    ConvParams wp;
    if (conv_slot(h, 1, 1, ws + pl.pool[0], ws, pl, batch_total, tb, batch_total, nullptr, wp))
      for (int rep = 0; rep < 200 && st == DT_OK; ++rep) st = launch_conv(wp, s);
    if (hipStreamSynchronize(s) != hipSuccess) st = (int)hipGetLastError();
  }
  for (int j = 0; j < kBlocks && st == DT_OK; ++j) {
    const float *in = j == 0 ? ws + pl.h[0] : (j <= 4 ? ws + pl.pool[j - 1] : ws + pl.cat[j - 5]);   // (enc1: stands in for the image)
    float skip_ms = 0.f;
    for (int slot = 0; slot < 3 && st == DT_OK; ++slot) {
      ConvParams p;
      if (!conv_slot(h, j, slot, in, ws, pl, batch_total, tb, batch_total, nullptr, p)) continue;
      {   // candidates start from the UNFUSED launch (the heuristic may have folded the skip in)
        const ConvChoice unfused{p.bm, p.bn, p.splits, p.prec, 0};
        conv_slot(h, j, slot, in, ws, pl, batch_total, tb, batch_total, &unfused, p);
      }
      const bool can_split = j > 0 && !p.x3 && p.M <= kSplitMaxRows;
      const bool walk9 = p.tap_hi - p.tap_lo == 9;
      ConvChoice best{p.bm, p.bn, p.splits, p.prec, 0};
      float best_ms = 1e30f;
      const bool can_fuse = slot == 2 && j > 0 && h->blk[j].has_res;
      const BlockW &kw = h->blk[j];
      const bool full3x3 = p.ksize == 3 && p.tap_hi - p.tap_lo == 9;
      // average milliseconds of `reps` back-to-back launches of one candidate (one warm launch first): in the sampler
      // launches queue behind each other, so a candidate is not charged the idle-queue launch latency per kernel
      // (which would bias against split launches = two kernels); < 0: not admissible here
      auto measure = [&](const ConvParams &q, int reps) -> float {
        int lst = launch_conv(q, s);
        if (lst == DT_E_SHAPE || lst == DT_E_ARG) return -1.f;
        if (lst != DT_OK) { st = lst; return -1.f; }
        (void)hipEventRecord(e0, s);
        for (int rep = 0; rep < reps && lst == DT_OK; ++rep) lst = launch_conv(q, s);
        (void)hipEventRecord(e1, s);
        if (hipEventSynchronize(e1) != hipSuccess) lst = (int)hipGetLastError();
        if (lst != DT_OK) { st = lst; return -1.f; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / reps;
      };
      struct Cand { ConvChoice c; ConvParams q; float cost; };
      std::vector<Cand> cands;
      static const float split_margin = getenv("DT_TUNE_SPLIT_MARGIN") ? (float)atof(getenv("DT_TUNE_SPLIT_MARGIN")) : 1.05f;
      // a fused conv2 also saves the separate skip launch measured for slot 0; launches are timed on an otherwise idle
      // GPU, where extra workgroups are free, whereas in the sampler other streams fill idle CUs: a split (slab traffic
      // + one more launch) must win by a margin to be taken
      auto cost_of = [&](float ms, const ConvChoice &c) {
        return (ms + (c.fuse ? 0.f : (can_fuse ? skip_ms : 0.f))) * (c.splits > 1 ? split_margin : 1.0f);
      };
      // Kernel families worth timing (the search is paid once per model and shape, so it is pruned to what the
      // per-layer tables show can win): exact-fp32 mode -> the fp32-MFMA kernel only; otherwise the strip kernels
      // for full 3x3 walks and the plain split-bf16 kernel for everything else.
      for (int prec = 0; prec <= 5 && st == DT_OK; ++prec) {
        if (h->precision == DT_PREC_FP32 ? prec != 0 : prec == 0) continue;
        if (prec == 2) continue;
        static const int skip_mask = getenv("DT_TUNE_SKIP_PREC") ? atoi(getenv("DT_TUNE_SKIP_PREC")) : 0;   // experiments: bit p drops family p
        if (skip_mask & (1 << prec)) continue;
        const bool strip_ok = full3x3 && strip_admissible(p.W, 64, 64, 3);   // some strip tile fits this row width
        if (prec >= 3 && !strip_ok) continue;
        if (prec == 1 && strip_ok) continue;                                  // the plain kernel competes where the strip one cannot run
      for (int bm = 64; bm <= 256; bm *= 2)
        for (int bn = 64; bn <= 128; bn += 64) {
          if (p.n_p % bn) continue;
          if (bm == 256 && (prec < 3 || bn != 64)) continue;          // the 4 x 1 wave layout: strip kernels, 64-column tiles
          if (prec == 5 && (bm > 128 || (bm == 128 && bn == 128))) continue;   // K split across the waves: tiles below 128 x 128
          // dec1.conv2 with one N tile also evaluates the head (saves the head launch and dec1's output round trip)
          if (j == kBlocks - 1 && slot == 2 && h->head_fusion && h->desc.channels <= 3 && p.n_p <= 128 && bn != p.n_p) continue;
          // 3x3 walks split by taps 1/3/9; the strip kernel and single-tap layers by channel chunks 1/2/4/8
          const long long tiles = (long long)((p.M + bm - 1) / bm) * (p.n_p / bn);
          for (int sp = 1; sp <= (can_split ? 9 : 1); sp *= ((prec >= 3 || !walk9) ? 2 : 3))
          for (int fuse = 0; fuse <= ((can_fuse && sp == 1) ? 1 : 0); ++fuse) {
            if (prec >= 3 ? !chunks_fit(p.cin_p >> 4, round_up(p.cin_p, 16 * kChunkPad) >> 4, sp * strip_kc(prec, bm, bn))
                          : (!walk9 && (p.cin_p >> 4) % sp)) continue;
            if (prec >= 3 && !strip_admissible(p.W, bm, bn, prec)) continue;   // LDS footprint of this tile
            if (sp > 1 && tiles * (sp / 2) >= 1024) continue;          // already >= 4 workgroups per CU without this split
            ConvParams q = p;
            q.bm = bm; q.bn = bn; q.splits = sp; q.prec = prec;
            q.w = prec ? (slot == 0 ? kw.wrb : (slot == 1 ? kw.w1b : kw.w2b)) : (slot == 0 ? kw.wr : (slot == 1 ? kw.w1 : kw.w2));
            q.ccw = prec ? round_up(p.cin_p, 16 * kChunkPad) >> 4 : p.cin_p >> 4;
            if (fuse) {
              q.add = nullptr; q.in2 = in; q.w2 = prec ? kw.wrb : kw.wr; q.bias2 = kw.hr; q.cin2_p = kw.cin_p;
              q.cin2_real = kw.cin; q.ccw2 = prec ? kw.cin_w >> 4 : kw.cin_p >> 4;
            }
            const float ms = measure(q, 3);
            if (ms < 0.f) continue;
            const ConvChoice c{bm, bn, sp, prec, fuse};
            cands.push_back(Cand{c, q, cost_of(ms, c)});
          }
        }
      }
      // second look at the three cheapest: longer, interleaved runs (clock drift and timing noise otherwise flip
      // choices between near-equal candidates from run to run); ties within 1 % go to the later (larger-tile) candidate
      if (!cands.empty() && st == DT_OK) {
        std::vector<size_t> top;
        for (size_t i = 0; i < cands.size(); ++i) top.push_back(i);
        std::stable_sort(top.begin(), top.end(), [&](size_t a, size_t b) { return cands[a].cost < cands[b].cost; });
        if (top.size() > 3) top.resize(3);
        std::sort(top.begin(), top.end());
        std::vector<float> fin(top.size(), 1e30f);
        for (int round = 0; round < 2 && st == DT_OK; ++round)
          for (size_t k = 0; k < top.size() && st == DT_OK; ++k) {
            const float ms = measure(cands[top[k]].q, 10);
            if (ms >= 0.f && ms < fin[k]) fin[k] = ms;
          }
        for (size_t k = 0; k < top.size(); ++k) {
          const float cost = cost_of(fin[k], cands[top[k]].c);
          if (cost < best_ms * 1.01f) { best_ms = cost < best_ms ? cost : best_ms; best = cands[top[k]].c; }
        }
      }
      t.c[j][slot] = best;
      if (slot == 0) skip_ms = best_ms;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (st != DT_OK) return st;
  for (TunedShape &old : h->tuned)
    if (old.Bt == t.Bt && old.H == t.H && old.W == t.W && old.imgs == t.imgs && old.single == t.single) { old = t; drop_graphs(h); return DT_OK; }
  h->tuned.push_back(t);
  drop_graphs(h);
  return DT_OK;
}

/* tuning / profiling aid: average milliseconds (HIP events, synchronises) of ONE convolution launch of a
 * forward shape under an explicit (tile, split, arithmetic, fuse) choice, plus its algorithmic FLOPs */
int dt_unet_time_conv(const dt_unet *h, int batch_total, int H, int W, int block, int slot, int bm, int bn, int splits,
                      int prec, int fuse, int reps, void *workspace, size_t ws_bytes, void *stream, float *ms,
                      double *flops) {
  if (!h || !workspace || !ms || !flops) return DT_E_NULL;
  if (block < 0 || block >= kBlocks || slot < 0 || slot > 2 || reps < 1) return DT_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const Plan pl = make_plan(h, batch_total, H, W);
  if (pl.total * sizeof(float) > ws_bytes) return DT_E_WORKSPACE;
  float *ws = (float *)workspace;
  const float *in = block == 0 ? ws + pl.h[0] : (block <= 4 ? ws + pl.pool[block - 1] : ws + pl.cat[block - 5]);
  const ConvChoice c{bm, bn, splits, prec, fuse};
  ConvParams p;
  if (!conv_slot(h, block, slot, in, ws, pl, batch_total, h->slab, batch_total, &c, p)) { *ms = 0.f; *flops = 0.0; return DT_OK; }
  *flops = 2.0 * p.M * (double)p.cout_real * ((double)p.cin_real * p.ksize * p.ksize + (p.in2 ? p.cin2_real : 0));
#ifdef DT_TOOLS
  if (const char *ab = getenv("DT_ABLATE")) p.ablate = atoi(ab);   // timing experiments only (tools build: build.py --tools)
#endif
  hipEvent_t e0, e1;
  DT_HIP_TRY(hipEventCreate(&e0));
  DT_HIP_TRY(hipEventCreate(&e1));
  int st = launch_conv(p, s);   // warm
  (void)hipEventRecord(e0, s);
  for (int r = 0; r < reps && st == DT_OK; ++r) st = launch_conv(p, s);
  (void)hipEventRecord(e1, s);
  if (hipEventSynchronize(e1) != hipSuccess) st = (int)hipGetLastError();
  float t = 0.f;
  (void)hipEventElapsedTime(&t, e0, e1);
  *ms = t / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return st;
}

int dt_unet_set_precision(dt_unet *h, int precision) {
  if (!h) return DT_E_NULL;
  if (precision < DT_PREC_FP32 || precision > DT_PREC_AUTO) return DT_E_ARG;
  h->tuned.clear();                               // choices are per arithmetic mode: back to the heuristic plan
  drop_graphs(h);
  h->precision = precision;
  return DT_OK;
}

int dt_unet_set_head_fusion(dt_unet *h, int on) {
  if (!h) return DT_E_NULL;
  h->head_fusion = on != 0;
  drop_graphs(h);
  return DT_OK;
}

int dt_unet_set_fused(dt_unet *h, int on) {
  if (!h) return DT_E_NULL;
  h->fused_on = on != 0 && h->fused_ok;
  drop_graphs(h);
  return DT_OK;
}

int dt_unet_fused_active(const dt_unet *h, int H, int W) {
  if (!h) return DT_E_NULL;
  return use_fused(h, H, W) ? 1 : 0;
}

/* test / report hook: the (bm, bn, splits) in use for block j, slot (0 skip, 1 conv1, 2 conv2) at a shape */
int dt_unet_conv_choice(const dt_unet *h, int batch_total, int H, int W, int block, int slot, int *bm, int *bn,
                        int *splits, int *prec, int *tuned) {
  if (!h || !bm || !bn || !splits || !prec || !tuned) return DT_E_NULL;
  if (block < 0 || block >= kBlocks || slot < 0 || slot > 2 || H % 16 || W % 16 || batch_total < 1) return DT_E_ARG;
  const Plan pl = make_plan(h, batch_total, H, W);
  int imgs = 0, single = 0;
  shape_images(h, batch_total, H, W, imgs, single);
  const TunedShape *t = find_tuned(h, batch_total, H, W, imgs, single);
  ConvParams p;
  float dummy = 0.f;
  if (!conv_slot(h, block, slot, &dummy, &dummy, pl, batch_total, &dummy, 1, t ? &t->c[block][slot] : nullptr, p)) {
    *bm = *bn = *splits = *prec = 0; *tuned = 0;
    return DT_OK;
  }
  if (!p.bm || !p.bn) { const ConvChoice c = heuristic_choice(p.M, p.n_p, 1); p.bm = c.bm; p.bn = c.bn; }
  *bm = p.bm; *bn = p.bn; *splits = p.splits; *prec = p.prec + (p.in2 ? 8 : 0); *tuned = t != nullptr;
  if (slot == 0) {   // folded into conv2?
    ConvParams c2;
    conv_slot(h, block, 2, &dummy, &dummy, pl, batch_total, &dummy, 1, t ? &t->c[block][2] : nullptr, c2);
    if (c2.in2) *bm = *bn = *splits = 0;
  }
  return DT_OK;
}

int dt_unet_declare_shape(dt_unet *h, int batch_total, int H, int W, int images, int single) {
  if (!h) return DT_E_NULL;
  if (batch_total < 1 || H < 16 || W < 16 || H % 16 || W % 16 || images < 1 || single < 0 || single >= images) return DT_E_ARG;
  if (batch_total != images && batch_total != 2 * images - single && (single || batch_total % images)) return DT_E_ARG;
  note_shape(h, batch_total, H, W, images, single);
  return DT_OK;
}

int dt_unet_set_conv_choice(dt_unet *h, int batch_total, int H, int W, int block, int slot, int bm, int bn, int splits,
                            int prec, int fuse) {
  if (!h) return DT_E_NULL;
  if (block < 0 || block >= kBlocks || slot < 0 || slot > 2 || H < 16 || W < 16 || H % 16 || W % 16 || batch_total < 1)
    return DT_E_ARG;
  if ((bm != 64 && bm != 128 && !(bm == 256 && bn == 64 && prec >= 3)) || (bn != 64 && bn != 128)) return DT_E_ARG;
  if (splits < 1 || splits > 9 || prec < 0 || prec > 5 || (fuse && (slot != 2 || splits != 1))) return DT_E_ARG;
  if (prec == 5 && (bm > 128 || (bm == 128 && bn == 128))) return DT_E_ARG;
  if (splits == 5 || splits == 6 || splits == 7) return DT_E_ARG;
  if (prec == 2) return DT_E_ARG;
  if (h->blk[block].n_p % bn) return DT_E_ARG;
  const Plan pl = make_plan(h, batch_total, H, W);
  TunedShape *t = nullptr;
  int imgs = 0, single = 0;
  shape_images(h, batch_total, H, W, imgs, single);
  for (TunedShape &old : h->tuned)
    if (old.Bt == batch_total && old.H == H && old.W == W && old.imgs == imgs && old.single == single) t = &old;
  if (!t) {   // start from what an untuned forward would launch
    TunedShape fresh{};
    fresh.Bt = batch_total; fresh.H = H; fresh.W = W; fresh.imgs = imgs; fresh.single = single;
    float dummy = 0.f;
    for (int j = 0; j < kBlocks; ++j)
      for (int sl = 0; sl < 3; ++sl) {
        ConvParams p;
        if (!conv_slot(h, j, sl, &dummy, &dummy, pl, batch_total, &dummy, 1, nullptr, p)) continue;
        if (!p.bm || !p.bn) { const ConvChoice c = heuristic_choice(p.M, p.n_p, 1); p.bm = c.bm; p.bn = c.bn; }
        fresh.c[j][sl] = ConvChoice{p.bm, p.bn, p.splits, p.prec, p.in2 ? 1 : 0};
      }
    h->tuned.push_back(fresh);
    t = &h->tuned.back();
  }
  t->c[block][slot] = ConvChoice{bm, bn, splits, prec, fuse};
  drop_graphs(h);
  return DT_OK;
}

int dt_unet_debug_activation(const dt_unet *h, int batch_total, int H, int W, int which, size_t *off, int *cp,
                             int *oh, int *ow) {
  if (!h || !off || !cp || !oh || !ow) return DT_E_NULL;
  if (which < 0 || which > kBlocks || batch_total < 1 || H % 16 || W % 16) return DT_E_ARG;
  const Plan pl = make_plan(h, batch_total, H, W);
  if (which == kBlocks) { *off = pl.slab; *cp = 0; *oh = 0; *ow = 0; return DT_OK; }   // the split-K slab (diagnostics)
  *off = pl.o[which]; *cp = h->blk[which].cout_p; *oh = pl.H[which]; *ow = pl.W[which];
  return DT_OK;
}

int dt_cfg_update(int rule, const float *x, const float *eu, const float *ec, const float *z, const int32_t *z_row,
                  const float coef[4], int has_noise, const float *w, float w_scalar, float *out, int B, int E,
                  void *stream) {
  return launch_cfg_update(rule, x, eu, ec, z, z_row, 0, coef, has_noise, w, w_scalar, out, B, E, (hipStream_t)stream);
}

// tb_div rows of a step's forward share one time-bias row (0: one row per pass, i.e. tb_div = B); B_single: see forward_impl
static int sample_loop(const dt_unet *h, int rule, int B, int n_pass, int H, int W, int n_steps, const float *tb,
                       const float *coef, const int32_t *has_noise, const float *z, const int32_t *z_row,
                       const int64_t *z_shift, const float *w, float w_scalar, float *traj, float *eps_scratch,
                       void *ws, size_t ws_bytes, hipStream_t s, int B_single = 0, int tb_div = 0) {
  const int E = h->desc.channels * H * W;
  const size_t slot = (size_t)B * E;
  const int Bt = B * n_pass - B_single * (n_pass - 1);
  if (!tb_div) tb_div = B;
  if (Bt % tb_div) return DT_E_ARG;
  const int tb_rows = Bt / tb_div;         // time-bias rows per step
  if (make_plan(h, Bt, H, W).total * sizeof(float) > ws_bytes) return DT_E_WORKSPACE;
  if (use_fused(h, H, W)) {
    // small model: forward + CFG mix + update of up to kFusedMaxSteps timesteps per launch, images resident in LDS
    for (int i0 = 0; i0 < n_steps; i0 += kFusedMaxSteps) {
      const int n = n_steps - i0 < kFusedMaxSteps ? n_steps - i0 : kFusedMaxSteps;
      FusedArgs a{};
      fused_common(h, a, B, n_pass, B_single, tb + (size_t)i0 * tb_rows * h->tb_stride, tb_div);
      a.mode = FUSED_LOOP; a.rule = rule; a.n_steps = n;
      a.traj = traj + (size_t)i0 * slot; a.z = z; a.z_row = z_row; a.wg = w; a.w_scalar = w_scalar;
      for (int i = 0; i < n; ++i) {
        a.coef[i][0] = coef[4 * (i0 + i)]; a.coef[i][1] = coef[4 * (i0 + i) + 1]; a.coef[i][2] = coef[4 * (i0 + i) + 2];
        a.z_shift[i] = z_shift ? (long long)z_shift[i0 + i] : 0;
        if (has_noise[i0 + i]) a.noise_mask |= 1ull << i;
        if (has_noise[i0 + i] && !z) return DT_E_NULL;
      }
      const int st = launch_unet_fused(a, s);
      if (st) return st;
    }
    return DT_OK;
  }
  const float *lowres = (const float *)ws + lowres_offset(h, Bt, H, W);
  (void)eps_scratch;                       // kept in the ABI: callers still hand in the scratch the unfused path used
  for (int i = 0; i < n_steps; ++i) {
    const float *x = traj + (size_t)i * slot;
    float *xn = traj + (size_t)(i + 1) * slot;
    const bool dead = rule == DT_RULE_ENGINE && !has_noise[i];   // t == 0: the prediction is never used
    if (!dead) {      // the forward stops at the low-resolution head output; the update interpolates it (no eps tensor)
      int st = forward_impl(h, x, B, n_pass, H, W, tb + (size_t)i * tb_rows * h->tb_stride, tb_div, nullptr, (float *)ws, ws_bytes, s, B_single);
      if (st) return st;
    }
    // second-pass rows start at row B and belong to images B_single .. B-1: indexed by image through a pointer shifted back
    int st = launch_cfg_update_lowres(rule, x, lowres, n_pass == 2 ? lowres + (size_t)(B - B_single) * (H / 2) * (W / 2) * 4 : nullptr, z, z_row,
                                      z_shift ? (long long)z_shift[i] : 0, coef + 4 * i, has_noise[i], w, w_scalar, xn, B,
                                      h->desc.channels, H, W, B_single, s);
    if (st) return st;
  }
  return DT_OK;
}

/* The loop is a fixed chain of (n_steps x ~45) launches whose arguments depend only on this call's arguments, so a
 * repeated call (same buffers, same schedule) can replay an instantiated hipGraph instead of re-issuing each launch.
 * Opt-in with DT_GRAPH=1: measured on MI355X / ROCm 7.2 the replay is time-neutral (batch 8: 31.5 vs 31.5 ms per
 * 100 forwards, batch 256: 94.9 vs 94.9 ms/step) -- the loop is bound by the GPU-side latency of its dependent
 * kernels (~6 us each at batch 8), not by host launch cost, which the host threads already hide. */
int dt_sample_trajectory(const dt_unet *h, int rule, int B, int n_pass, int H, int W, int n_steps, const float *tb,
                         const float *coef, const int32_t *has_noise, const float *z, const int32_t *z_row,
                         const int64_t *z_shift, const float *w, float w_scalar, float *traj, float *eps_scratch,
                         void *ws, size_t ws_bytes, void *stream) {
  if (!h || !tb || !coef || !has_noise || !traj || !ws) return DT_E_NULL;   // (eps_scratch is unused since the fused update: NULL is fine)
  if (n_pass < 1 || n_pass > 2 || n_steps < 0 || rule < 0 || rule > DT_RULE_MANAGER) return DT_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  const char *genv = getenv("DT_GRAPH");
  const bool use_graph = genv && atoi(genv) != 0;
  if (!use_graph || g_prof.on.load() || n_steps < 4)
    return sample_loop(h, rule, B, n_pass, H, W, n_steps, tb, coef, has_noise, z, z_row, z_shift, w, w_scalar, traj,
                       eps_scratch, ws, ws_bytes, s);
  // ---- key: every argument by value (the small host arrays by content)
  std::vector<unsigned char> key;
  auto put = [&key](const void *p, size_t n) { const unsigned char *c = (const unsigned char *)p; key.insert(key.end(), c, c + n); };
  const int ints[7] = {rule, B, n_pass, H, W, n_steps, z_shift ? 1 : 0};
  const void *ptrs[8] = {tb, z, z_row, w, traj, eps_scratch, ws, (const void *)s};
  put(ints, sizeof(ints)); put(ptrs, sizeof(ptrs)); put(&w_scalar, sizeof(w_scalar)); put(&ws_bytes, sizeof(ws_bytes));
  put(coef, sizeof(float) * 4 * n_steps); put(has_noise, sizeof(int32_t) * n_steps);
  if (z_shift) put(z_shift, sizeof(int64_t) * n_steps);
  const dt_unet *hm = h;
  std::lock_guard<std::mutex> lock(hm->graph_mu);
  for (const LoopGraph &g : hm->graphs)
    if (g.key == key) return (int)hipGraphLaunch(g.exec, s);
  // ---- first call with these arguments: capture the loop, instantiate, keep (at most 8 per handle)
  hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) {   // stream cannot be captured (e.g. the legacy default stream): plain launches
    (void)hipGetLastError();
    return sample_loop(h, rule, B, n_pass, H, W, n_steps, tb, coef, has_noise, z, z_row, z_shift, w, w_scalar, traj,
                       eps_scratch, ws, ws_bytes, s);
  }
  const int st = sample_loop(h, rule, B, n_pass, H, W, n_steps, tb, coef, has_noise, z, z_row, z_shift, w, w_scalar, traj,
                             eps_scratch, ws, ws_bytes, s);
  LoopGraph g{};
  e = hipStreamEndCapture(s, &g.graph);
  if (st != DT_OK) { if (e == hipSuccess && g.graph) (void)hipGraphDestroy(g.graph); return st; }
  if (e != hipSuccess) return (int)e;
  e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g.graph); return (int)e; }
  g.key = std::move(key);
  if (hm->graphs.size() >= 8) { (void)hipGraphExecDestroy(hm->graphs.front().exec); (void)hipGraphDestroy(hm->graphs.front().graph); hm->graphs.erase(hm->graphs.begin()); }
  hm->graphs.push_back(std::move(g));
  if (getenv("DT_GRAPH_DEBUG")) fprintf(stderr, "[dt_hip] captured sampler loop: %d steps, batch %d -> graph #%zu\n", n_steps, B, hm->graphs.size());
  return (int)hipGraphLaunch(hm->graphs.back().exec, s);
}

int dt_sample_trajectory_mixed(const dt_unet *h, int rule, int B, int B_single, int H, int W, int n_steps, const float *tb,
                               int tb_div, const float *coef, const int32_t *has_noise, const float *z, const int32_t *z_row,
                               const int64_t *z_shift, const float *w, float *traj, void *ws, size_t ws_bytes, void *stream) {
  if (!h || !tb || !coef || !has_noise || !traj || !ws || !w) return DT_E_NULL;
  if (n_steps < 0 || rule < 0 || rule > DT_RULE_MANAGER || B < 2 || B_single < 1 || B_single >= B || tb_div < 1) return DT_E_ARG;
  if ((2 * B - B_single) % tb_div || B_single % tb_div || !h->share_enc1) return DT_E_ARG;   // a time-bias row never straddles the single / CFG boundary
  return sample_loop(h, rule, B, 2, H, W, n_steps, tb, coef, has_noise, z, z_row, z_shift, w, 1.f, traj, nullptr, ws, ws_bytes,
                     (hipStream_t)stream, B_single, tb_div);
}

int dt_unet_forward_mixed(const dt_unet *h, const float *x, int B, int B_single, int H, int W, const float *tb, int tb_div,
                          float *eps, void *ws, size_t ws_bytes, void *stream) {
  if (!eps || !h) return DT_E_NULL;
  if (B_single < 1 || B_single >= B || tb_div < 1 || (2 * B - B_single) % tb_div || B_single % tb_div || !h->share_enc1) return DT_E_ARG;
  return forward_impl(h, x, B, 2, H, W, tb, tb_div, eps, (float *)ws, ws_bytes, (hipStream_t)stream, B_single);
}

// an empty kernel with a recognisable name: lets an external profiler (rocprofv3 traces) bracket a region of the stream
__global__ void profile_marker_kernel(int) {}

int dt_profile_marker(int id, void *stream) {
  profile_marker_kernel<<<1, 64, 0, (hipStream_t)stream>>>(id);
  DT_LAUNCH_CHECK();
  return DT_OK;
}

int dt_profile_begin(void) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.rec.clear();
  g_prof.used = 0;
  g_prof.on = true;
  return DT_OK;
}

int dt_profile_end(void) {
  std::lock_guard<std::mutex> lock(g_prof.mu);
  g_prof.on = false;
  return DT_OK;
}

int dt_profile_class_count(void) { return KC_COUNT; }

int dt_profile_read(int cls, const char **name, long long *launches, double *ms, double *flops, double *bytes) {
  if (cls < 0 || cls >= KC_COUNT || !launches || !ms || !flops || !bytes) return DT_E_ARG;
  if (name) *name = kClassName[cls];
  *launches = 0; *ms = 0; *flops = 0; *bytes = 0;
  std::lock_guard<std::mutex> lock(g_prof.mu);
  for (const ProfRecord &r : g_prof.rec) {
    if (r.cls != cls) continue;
    DT_HIP_TRY(hipEventSynchronize(r.b));
    float t = 0.f;
    DT_HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    *launches += 1; *ms += t; *flops += r.flops; *bytes += r.bytes;
  }
  return DT_OK;
}

int dt_traj_metrics(const float *X, const float *Y, int nT, int nS, int B, int E, double *out, void *stream) {
  return launch_traj_metrics(X, Y, nT, nS, B, E, out, (hipStream_t)stream);
}

int dt_traj_wasserstein(const float *X, const float *Y, int n, int B, int E, const int32_t *index,
                        const int32_t *index_row, int n_idx, double *out, void *stream) {
  return launch_wasserstein(X, Y, n, B, E, index, index_row, n_idx, out, (hipStream_t)stream);
}

int dt_traj_pair_metrics(const float *X, const float *Y, int n, int B, int E, double *out_sums, double *out_w1, void *stream) {
  return launch_pair_metrics(X, Y, n, B, E, out_sums, out_w1, (hipStream_t)stream);
}

int dt_pair_stats(const float *X, const float *Y, int n, int B, int E, double *out, void *stream) {
  return launch_pair_stats(X, Y, n, B, E, out, (hipStream_t)stream);
}

int dt_traj_sample_mean(const float *traj, int n, int B, int E, float *out, void *stream) {
  return launch_sample_mean(traj, n, B, E, out, (hipStream_t)stream);
}

int dt_resize_bilinear(const float *in, float *out, int planes, int h, int w, int H, int W, void *stream) {
  return launch_resize_bilinear(in, out, planes, h, w, H, W, (hipStream_t)stream);
}

int dt_traj_resampled_distance(const float *L, const float *S, int n_long, int n_short, int B, int E, double *out,
                               void *stream) {
  return launch_resampled_distance(L, S, n_long, n_short, B, E, out, (hipStream_t)stream);
}

}  // extern "C"

// Declarations of the whole-forward kernel for small U-Nets (dt_fused.hip), shared with the host side in dt_unet.hip.
#pragma once
#include "dt_internal.h"

namespace dt {

constexpr int kFusedMaxSteps = 64;     // timesteps per launch (their coefficients travel in the kernel arguments)
constexpr int kFusedMaxOps = 24;

enum { FUSED_OP_FIRST = 0, FUSED_OP_CONV, FUSED_OP_UP, FUSED_OP_HEAD };
// what follows relu(bn(conv)) in a convolution's epilogue
enum { FUSED_E_TIMEBIAS = 0, FUSED_E_IDENTITY, FUSED_E_SKIPCONV, FUSED_E_IMAGE };
enum { FUSED_FORWARD = 0, FUSED_LOOP = 1 };

// Everything the kernel reads about the model sits in ONE device allocation (the handle's fused slab) and is named by its
// float offset from that base: a pointer fetched from a table is a generic pointer to the compiler (flat loads, which wait
// for every outstanding memory operation), an offset from a kernel-argument pointer is a global load.
struct FusedConvW { int w, scale, shift; };       // packed weights, folded BN (the bias of a plain conv in shift)
struct FusedBlockW { FusedConvW c1, c2, cr; int tb_off; };
struct FusedModel {
  FusedBlockW blk[kBlocks];
  int wf;                 // enc1.conv1 as [9C][c0p] (launch_pack_first_conv)
  int w3;                 // enc1's image skip [c0p][4] (launch_pack_res3)
  int c0p, c1p;
};

// float offsets of the LDS buffers of one workgroup (fused_lds)
struct FusedLds { int x, low, zero, tbo, head, scr, par, tb, tb_cols, b, c, e, d, total; };

// one entry of the layer table; LDS positions are float offsets, strides are floats per pixel
struct FusedOp {
  int w, wr;                           // CONV / FIRST: packed weights; SKIPCONV: the 1x1 skip conv's packed weights    (slab offsets)
  int scale, shift, br;                // folded BN, the skip conv's bias; IMAGE: wr = the [c0p][4] image-skip rows    (offsets into the
                                       // parameter block, which the kernel keeps in LDS)
  int kind, pb, emode, wsh;            // pb: pixel tiles per unit (4 / 2 / 1); wsh: log2 of the level's picture width
  int kc, ot, tap_lo, tap_hi;          // 16-channel chunks of the input, 16-channel tiles of the output, taps walked
  int ks;                              // K split: the (tap, chunk) steps of a unit are cut into ks slices on ks waves (pb == 1 only)
  int inA, inB, cs_in, cca;            // input: chunks < cca from inA, the rest from inB (both with pixel stride cs_in)
  int kcr, rA, rB, rcs, rcca;          // SKIPCONV: the same for the skip conv's input; IDENTITY: rA / rcs = the block input
  int tb_off;                          // TIMEBIAS / FIRST: the block's column offset in a time-bias row
  int out, cs, pool;                   // output buffer (-1: none), its stride, pooled output buffer (-1: none; stride cs)
                                       // UP: inA -> out, cs channels; HEAD: inA with stride cs
};

struct FusedArgs {
  const FusedOp *ops;                  // device memory
  const float *fbase;                  // the fused slab: FusedOp's weight offsets count from here
  int par, n_par;                      // the parameter block inside the slab (float offset, length): copied to LDS at kernel start
  const float *head_w, *head_b;        // final 1x1: [C][c0], [C]
  const float *tb;                     // time-bias rows of the launch's first step
  const float *x;                      // FORWARD: images [B][C][16][16]
  float *eps;                          // FORWARD: [rows][C][16][16]
  float *traj;                         // LOOP: trajectory slot of the launch's first step; slot i + 1 receives step i
  const float *z;                      // LOOP: noise rows
  const int32_t *z_row;
  const float *wg;                     // per-image guidance scale (nullptr: w_scalar)
  long long z_shift[kFusedMaxSteps];
  float coef[kFusedMaxSteps][3];
  unsigned long long noise_mask;       // bit i: step i adds noise
  double flops_per_row;                // profiler only
  unsigned long long *trace;           // tools build: s_memtime stamps of workgroup 0's first step, one per layer (else nullptr)
  FusedLds lds;
  float w_scalar;
  int n_ops, mode, rule, n_steps;
  int G, C, c0, c0p;
  int B, B_single, n_pass;             // rows = [pass 0 of B images | pass 1 of images B_single ..]
  int tb_stride, tb_div, tb_rows;      // row r uses time-bias row r / tb_div; tb_rows rows per step
};

bool fused_eligible(int C, int c0p, int c1p);
FusedLds fused_lds(int G, int c0p, int c1p);
size_t fused_lds_bytes(int G, int c0p, int c1p);
int fused_ops(const FusedModel &m, int G, FusedOp *ops);      // fills at most kFusedMaxOps entries, returns the count
double fused_flops_per_row(int C, int c0, int c1);
int launch_pack_fused_first(const float *w_oihw, float *dst, int cout, int C, int ot, hipStream_t s);
int launch_pack_fused_conv(const float *w_oihw, float *dst, int cout, int cin, int taps, int kc, int ot, int split_c, int split_cp,
                           hipStream_t s);
int launch_unet_fused(const FusedArgs &a, hipStream_t s);

}  // namespace dt

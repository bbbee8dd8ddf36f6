// Fused classifier-free-guidance mix + reverse-diffusion update + trajectory store.
//
// One streaming pass replaces the ~8 elementwise ATen launches per step of the reference:
//   eps = e_u + w*(e_c - e_u)                         utils/diffusion.py:126, trajectory_engine.py:80
//   ENGINE : x' = c1*x - c2*eps ; x' += sigma*z       trajectory_engine.py:104-110
//   PSAMPLE: x' = sra*(x - k*eps) + z*beta            utils/diffusion.py:149-158
//   MANAGER: x' = (x - b*eps)/sqrt(a) ; x' += s*z     utils/trajectory_manager.py:196-203
// x' is written straight into the next trajectory slot (which is the next step's input), so the
// algorithmic HBM traffic is read x + read z + write x' = 3*E*4 B per sample-step, plus the eps
// read(s).  All arithmetic is plain fp32 operators with fma contraction switched off, in the
// reference's operation order: equal inputs give bit-identical outputs to torch CPU.
// (HIP's __fmul_rn/__fadd_rn are header inlines compiled with contraction on, so they are not used.)
#include "dt_update_math.h"

#pragma clang fp contract(off)

namespace dt {

// LOWRES builds: eu / ec are the low-resolution head outputs lowres[b][h][w][4] of the two passes and the prediction
// is their bilinear x2 upsampling (models.py:221, align_corners=True), evaluated here instead of by a separate
// head_upsample launch + an eps round trip through HBM; lh, lw = low-resolution size, hw = H*W of the image.
struct UpdateArgs {
  const float *x, *eu, *ec, *z;
  const int32_t *z_row;
  const float *w;
  float *out;
  StepCoef k;          // the step's three coefficients (dt_update_math.h)
  float w_scalar;
  long long z_shift;   // added to the z row index (rows of E floats)
  int has_noise, B, E4;
  int lh, lw, hw;
  int b_single;        // rows [0, b_single) take the first prediction as it is (single-pass images of a mixed batch)
};

// eps of element (b, c, y, x..x+3) from a low-resolution head output (same arithmetic as head_upsample_kernel)
__device__ inline float4 eps_from_lowres(const float *lowres, int b, int e, const UpdateArgs &a) {
  const int W = 2 * a.lw, H = 2 * a.lh;
  const int idx = e * 4, c = idx / a.hw, pix = idx - c * a.hw;
  const int y = pix / W, x0 = pix - y * W;
  int y0, y1;
  float wy0, wy1;
  bilinear_src(y, a.lh, H, y0, y1, wy0, wy1);
  const float *base = lowres + (size_t)b * a.lh * a.lw * 4 + c;
  float r[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int xa, xb;
    float wx0, wx1;
    bilinear_src(x0 + j, a.lw, W, xa, xb, wx0, wx1);
    const float v00 = base[((size_t)y0 * a.lw + xa) * 4], v01 = base[((size_t)y0 * a.lw + xb) * 4];
    const float v10 = base[((size_t)y1 * a.lw + xa) * 4], v11 = base[((size_t)y1 * a.lw + xb) * 4];
    r[j] = bilinear_blend(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
  }
  return make_float4(r[0], r[1], r[2], r[3]);
}

template <int RULE, bool LOWRES = false>
__global__ __launch_bounds__(256) void cfg_update_kernel(const UpdateArgs a) {
  const size_t total = (size_t)a.B * a.E4;
  const float4 *x4 = reinterpret_cast<const float4 *>(a.x);
  const float4 *eu4 = reinterpret_cast<const float4 *>(a.eu);
  const float4 *ec4 = reinterpret_cast<const float4 *>(a.ec);
  const float4 *z4 = reinterpret_cast<const float4 *>(a.z);
  float4 *o4 = reinterpret_cast<float4 *>(a.out);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = i / a.E4, e = i - (size_t)b * a.E4;
    const float4 xv = x4[i];
    if (RULE == DT_RULE_ENGINE && !a.has_noise) { o4[i] = xv; continue; }   // t == 0: x is recorded unchanged
    float4 ev = LOWRES ? eps_from_lowres(a.eu, b, e, a) : eu4[i];
    if (a.ec && b >= a.b_single) {
      const float4 cv = LOWRES ? eps_from_lowres(a.ec, b, e, a) : ec4[i];
      const float w = a.w ? a.w[b] : a.w_scalar;
      ev.x = ev.x + w * (cv.x - ev.x);
      ev.y = ev.y + w * (cv.y - ev.y);
      ev.z = ev.z + w * (cv.z - ev.z);
      ev.w = ev.w + w * (cv.w - ev.w);
    }
    float4 zv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool noise = a.has_noise != 0;
    if (noise) {
      const long long zr = (a.z_row ? (long long)a.z_row[b] : (long long)b) + a.z_shift;
      zv = z4[(size_t)zr * a.E4 + e];
    }
    float4 o;
    if (RULE != DT_RULE_PSAMPLE && !noise) {
      // MANAGER at t == 0 never reaches the update in the reference; keep the deterministic part
      o.x = step1<RULE>(xv.x, ev.x, 0.f, a.k, false); o.y = step1<RULE>(xv.y, ev.y, 0.f, a.k, false);
      o.z = step1<RULE>(xv.z, ev.z, 0.f, a.k, false); o.w = step1<RULE>(xv.w, ev.w, 0.f, a.k, false);
    } else {
      o.x = step1<RULE>(xv.x, ev.x, zv.x, a.k, noise); o.y = step1<RULE>(xv.y, ev.y, zv.y, a.k, noise);
      o.z = step1<RULE>(xv.z, ev.z, zv.z, a.k, noise); o.w = step1<RULE>(xv.w, ev.w, zv.w, a.k, noise);
    }
    o4[i] = o;
  }
}

int launch_cfg_update(int rule, const float *x, const float *eu, const float *ec, const float *z,
                      const int32_t *z_row, long long z_shift, const float coef[4], int has_noise, const float *w,
                      float w_scalar, float *out, int B, int E, hipStream_t s) {
  if (!x || !eu || !out || !coef) return DT_E_NULL;
  if (has_noise && !z) return DT_E_NULL;
  if (B <= 0 || E <= 0 || E % 4) return DT_E_SHAPE;
  UpdateArgs a{x, eu, ec, z, z_row, w, out, {coef[0], coef[1], coef[2]}, w_scalar, z_shift, has_noise, B, E / 4, 0, 0, 0, 0};
  const size_t total = (size_t)B * (E / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  // algorithmic bytes: read x + read z + write x' = 3*E*4 per sample-step (+ eps reads)
  ProfileScope prof(KC_UPDATE, 0.0, 4.0 * B * E * (2.0 + (has_noise ? 1.0 : 0.0) + (ec ? 2.0 : 1.0)), s);
  switch (rule) {
    case DT_RULE_ENGINE: cfg_update_kernel<DT_RULE_ENGINE><<<blocks, 256, 0, s>>>(a); break;
    case DT_RULE_PSAMPLE: cfg_update_kernel<DT_RULE_PSAMPLE><<<blocks, 256, 0, s>>>(a); break;
    case DT_RULE_MANAGER: cfg_update_kernel<DT_RULE_MANAGER><<<blocks, 256, 0, s>>>(a); break;
    default: return DT_E_ARG;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

// the same step with the prediction taken from the low-resolution head outputs of the two passes (lowres_u, lowres_c:
// [B][H/2][W/2][4]; lowres_c == nullptr: one pass).  b_single > 0 (mixed batch): images [0, b_single) have no second pass
// and lowres_c is indexed from image b_single on (the caller passes the second pass's first row shifted back by b_single images).
int launch_cfg_update_lowres(int rule, const float *x, const float *lowres_u, const float *lowres_c, const float *z,
                             const int32_t *z_row, long long z_shift, const float coef[4], int has_noise, const float *w,
                             float w_scalar, float *out, int B, int C, int H, int W, int b_single, hipStream_t s) {
  if (!x || !lowres_u || !out || !coef) return DT_E_NULL;
  if (has_noise && !z) return DT_E_NULL;
  const int E = C * H * W;
  if (B <= 0 || E <= 0 || W % 4 || H % 2 || W % 2) return DT_E_SHAPE;
  if (b_single < 0 || b_single > B) return DT_E_ARG;
  UpdateArgs a{x, lowres_u, lowres_c, z, z_row, w, out, {coef[0], coef[1], coef[2]}, w_scalar, z_shift, has_noise, B, E / 4,
               H / 2, W / 2, H * W, b_single};
  const size_t total = (size_t)B * (E / 4);
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  ProfileScope prof(KC_UPDATE, 0.0, 4.0 * B * E * (2.0 + (has_noise ? 1.0 : 0.0)) + 4.0 * B * H * W * (lowres_c ? 2.0 : 1.0), s);
  switch (rule) {
    case DT_RULE_ENGINE: cfg_update_kernel<DT_RULE_ENGINE, true><<<blocks, 256, 0, s>>>(a); break;
    case DT_RULE_PSAMPLE: cfg_update_kernel<DT_RULE_PSAMPLE, true><<<blocks, 256, 0, s>>>(a); break;
    case DT_RULE_MANAGER: cfg_update_kernel<DT_RULE_MANAGER, true><<<blocks, 256, 0, s>>>(a); break;
    default: return DT_E_ARG;
  }
  DT_LAUNCH_CHECK();
  return DT_OK;
}

}  // namespace dt

// Arithmetic of one reverse-diffusion update, shared by cfg_update_kernel (dt_update.hip) and the fused small-model
// kernel (dt_fused.hip).  Every function switches fp contraction OFF in its own body: the operations are the
// reference's, in its order, each rounded to fp32 on its own, so equal inputs give bit-identical outputs to torch CPU
// whatever the contraction setting of the translation unit that includes this header.
//   ENGINE : x' = c1*x - c2*eps ; x' += sigma*z       trajectory_engine.py:104-110
//   PSAMPLE: x' = sra*(x - k*eps) + z*beta            utils/diffusion.py:149-158
//   MANAGER: x' = (x - b*eps)/sqrt(a) ; x' += s*z     utils/trajectory_manager.py:196-203
#pragma once
#include "dt_internal.h"

namespace dt {

// coefficients of a step by rule: ENGINE (c1, c2, sigma), PSAMPLE (sqrt_recip_alpha, 1 - sqrt(1 - acp) factor, beta),
// MANAGER (beta, sqrt(alpha), noise scale)
struct StepCoef { float c0, c1, c2; };

template <int RULE>
__device__ inline float step1(float x, float eps, float z, const StepCoef &a, bool noise) {
#pragma clang fp contract(off)
  if (RULE == DT_RULE_ENGINE) {
    const float v = a.c0 * x - a.c1 * eps;
    return v + a.c2 * z;
  } else if (RULE == DT_RULE_PSAMPLE) {
    const float v = a.c0 * (x - a.c1 * eps);
    return v + (noise ? z * a.c2 : 0.f * a.c2);
  } else {
    const float v = (x - a.c0 * eps) / a.c1;       // IEEE division (hipcc's default for fp32 '/')
    return v + a.c2 * z;
  }
}

// the rule dispatched at run time (the fused kernel carries the rule as an argument)
__device__ inline float step1_rt(int rule, float x, float eps, float z, const StepCoef &a, bool noise) {
  if (rule == DT_RULE_ENGINE) return step1<DT_RULE_ENGINE>(x, eps, z, a, noise);
  if (rule == DT_RULE_PSAMPLE) return step1<DT_RULE_PSAMPLE>(x, eps, z, a, noise);
  return step1<DT_RULE_MANAGER>(x, eps, z, a, noise);
}

// eps = e_u + w (e_c - e_u)                            utils/diffusion.py:126, trajectory_engine.py:80
__device__ inline float cfg_mix(float eu, float ec, float w) {
#pragma clang fp contract(off)
  return eu + w * (ec - eu);
}

// Prediction at pixel (y, x) of channel c from a low-resolution head output lowres[lh][lw][4] of ONE image (models.py:221:
// bilinear x2, align_corners=True; same arithmetic as head_upsample_kernel).  `lowres` may point into global memory or LDS.
__device__ inline float eps_at(const float *lowres, int lh, int lw, int c, int y, int x) {
  int y0, y1, xa, xb;
  float wy0, wy1, wx0, wx1;
  bilinear_src(y, lh, 2 * lh, y0, y1, wy0, wy1);
  bilinear_src(x, lw, 2 * lw, xa, xb, wx0, wx1);
  const float *base = lowres + c;
  const float v00 = base[(y0 * lw + xa) * 4], v01 = base[(y0 * lw + xb) * 4];
  const float v10 = base[(y1 * lw + xa) * 4], v11 = base[(y1 * lw + xb) * 4];
  return bilinear_blend(v00, v01, v10, v11, wy0, wy1, wx0, wx1);
}

}  // namespace dt

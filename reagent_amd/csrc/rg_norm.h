// rg_norm.h — per-element dense feature normalization shared by rg_normalize_dense and the
// normalize-on-gather epilogue of rg_replay_gather.
// Preprocessor per-type bodies, reagent/preprocessing/preprocessor.py:197-525.
#pragma once
#include <rg_platform.h>
#include "../../include/reagent_hip.h"

namespace rg {


constexpr float kMaxFeature = 11.513f;  // preprocessing/normalization.py:34-35
constexpr float kEps = 1e-6f;           // preprocessing/normalization.py:36

__device__ __forceinline__ float quantile_value(float x, const float* q, int n) {
  // _preprocess_QUANTILE, preprocessor.py:434-505 (same arithmetic, scalar form)
  float qmax = q[0], qmin = q[0];
  for (int j = 1; j < n; ++j) {
    qmax = fmaxf(qmax, q[j]);
    qmin = fminf(qmin, q[j]);
  }
  const float set_to_max = (x >= qmax) ? 1.f : 0.f;
  const float set_to_min = (x <= qmin) ? 1.f : 0.f;
  const float interpolate = ((set_to_min + set_to_max) < 0.01f) ? 1.f : 0.f;
  float left = -3.4028235e38f, right = 3.4028235e38f, count_ge = 0.f;
  for (int j = 0; j < n; ++j) {
    const bool ge = x >= q[j];
    left = fmaxf(left, ge ? q[j] : -1e20f);
    right = fminf(right, ge ? 1e20f : q[j]);
    count_ge += ge ? 1.f : 0.f;
  }
  const float left_start = count_ge - 1.f;
  const float val = (left_start + (x - left) / ((right + kEps) - left)) / (float)(n - 1);
  return set_to_max + interpolate * val;
}


// one output column: value -> normalized value * presence, clamped like Preprocessor.forward :156-168
__device__ __forceinline__ float normalize_value(const rg_norm_col c, float v, float p, const float* quantiles) {
  float r;
  bool clamp = true;
  switch (c.op) {
    case RG_NORM_BINARY: r = 1.f - ((v == 0.f) ? 1.f : 0.f); break;
    case RG_NORM_PROBABILITY: {
      const float cl = fminf(fmaxf(v, 1e-5f), (float)(1.0 - 1e-5));
      r = -1.f * logf((1.f / cl) - 1.f);
      break;
    }
    case RG_NORM_CONTINUOUS: r = (v - c.p0) / c.p1; break;
    case RG_NORM_BOXCOX: {
      const float t = (powf(fmaxf(v + c.p0, 1e-6f), c.p1) - 1.f) / c.p1;
      r = (t - c.p2) / c.p3;
      break;
    }
    case RG_NORM_ENUM: r = (v == c.p0) ? 1.f : 0.f; clamp = false; break;
    case RG_NORM_QUANTILE: r = quantile_value(v, quantiles + (int)c.p0, (int)c.p1); break;
    case RG_NORM_CONTINUOUS_ACTION: {
      const float t = (v - c.p0) * c.p1 + c.p2;
      r = fminf(fmaxf(t, -1.f + kEps), 1.f - kEps);
      break;
    }
    case RG_NORM_CLIP_LOG: r = logf(fmaxf(v, kEps)); break;
    case RG_NORM_DO_NOT_PREPROCESS: r = v; clamp = false; break;
    default: r = v; break;  // DISCRETE_ACTION: identity (but clamped like the reference)
  }
  r = r * p;
  if (clamp) r = fminf(fmaxf(r, -kMaxFeature), kMaxFeature);
  return r;
}

}  // namespace rg

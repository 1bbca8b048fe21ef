// rg_optim.h — the per-element optimizer arithmetic, shared by the stand-alone kernels (optim.hip)
// and the fused update (mlp_fused.hip) so that both produce the same bits.  Floating-point
// contraction is switched off inside: whether `a + b * c` becomes an FMA would otherwise depend on
// the surrounding code, and the two call sites differed in the last place on the MI355X.
#pragma once
#include <rg_platform.h>

namespace rg {

// torch/optim/adam.py::_single_tensor_adam, same operation order:
//   g      = grad * grad_scale (+ wd * p)
//   m      = m + (g - m) * (1 - beta1)                      (lerp_)
//   v      = v * beta2 + ((1 - beta2) * g) * g              (mul_ + addcmul_)
//   denom  = sqrt(v) / bias_correction2_sqrt + eps
//   p      = p + (-step_size * m) / denom                   (addcdiv_)
struct AdamCoef {
  float w1, beta2, w2, eps, wd, neg_step_size, bc2_sqrt, grad_scale;
};

__device__ __forceinline__ float adam_element(const AdamCoef& c, float pi, float gi, float& mi, float& vi) {
#pragma clang fp contract(off)
  if (c.grad_scale != 1.f) gi = gi * c.grad_scale;
  if (c.wd != 0.f) gi = gi + c.wd * pi;
  const float dm = gi - mi;
  mi = mi + c.w1 * dm;
  vi = vi * c.beta2;
  const float g2 = (c.w2 * gi) * gi;
  vi = vi + g2;
  const float denom = sqrtf(vi) / c.bc2_sqrt + c.eps;
  const float num = c.neg_step_size * mi;
  return pi + num / denom;
}

// Device-resident Adam schedule (graph-safe steps): the two step-dependent coefficients are looked up in HBM
// instead of arriving as launch arguments, so a captured HIP graph replays the RIGHT step every time.
//   sched[0] = steps applied so far (the launch applies step sched[0] + 1; rg_sched_tick adds 1 afterwards)
//   sched[1] = lr      sched[2] = n (table entries)      sched[3] reserved
//   sched[4 + 2*(t-1)], sched[5 + 2*(t-1)] = 1 - beta1^t, sqrt(1 - beta2^t) for t = 1..n, computed by the host
//   in double exactly as for the scalar entry points; steps past n use entry n (the host builds the table up to
//   the step where both have reached their limit 1.0).
// The divisions below are IEEE double operations, so the coefficients equal the host-computed ones bit for bit.
// pre_ticked: the step has already been counted in sched[0] (by the sampler launch of a replayed step)
__device__ __forceinline__ void sched_lookup(const double* __restrict__ sched, double& lr, double& bc1, double& bc2_sqrt,
                                             int pre_ticked = 0) {
  long t = (long)sched[0] + (pre_ticked ? 0 : 1);
  const long n = (long)sched[2];
  if (t > n) t = n;
  lr = sched[1];
  bc1 = sched[4 + 2 * (t - 1)];
  bc2_sqrt = sched[5 + 2 * (t - 1)];
}
__device__ __forceinline__ AdamCoef sched_coef(AdamCoef c, const double* __restrict__ sched, int pre_ticked = 0) {
  if (sched) {
    double lr, bc1, bc2s;
    sched_lookup(sched, lr, bc1, bc2s, pre_ticked);
    c.neg_step_size = (float)(-(lr / bc1));
    c.bc2_sqrt = (float)bc2s;
  }
  return c;
}

// reagent/optimizer/soft_update.py:60-70: target = tau * source + (1 - tau) * target
__device__ __forceinline__ float soft_update_element(float tau, float one_minus_tau, float src, float tgt) {
#pragma clang fp contract(off)
  const float a = tau * src;
  const float b = one_minus_tau * tgt;
  return a + b;
}

}  // namespace rg

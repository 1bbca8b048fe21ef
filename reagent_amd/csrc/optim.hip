// optim.hip — fused Adam and soft target update over flat fp32 parameter slabs (HBM-bound).
#include "rg_optim.h"
#include "../../include/reagent_hip.h"

namespace rg {

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, AdamCoef c0, const double* __restrict__ sched) {
  const AdamCoef c = sched_coef(c0, sched);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float mi = m[i], vi = v[i];
    p[i] = adam_element(c, p[i], g[i], mi, vi);
    m[i] = mi;
    v[i] = vi;
  }
}

__global__ void sched_tick_kernel(double* sched) {
  if (blockIdx.x == 0 && threadIdx.x == 0) sched[0] = sched[0] + 1.0;
}

struct TickMany {
  double* s[RG_MAX_TICKS];
};
__global__ void sched_tick_many_kernel(TickMany t, int n) {
  const int i = threadIdx.x;
  if (blockIdx.x == 0 && i < n) t.s[i][0] = t.s[i][0] + 1.0;
}

__global__ void soft_update_kernel(float* __restrict__ tgt, const float* __restrict__ src, long n, float tau,
                                   float one_minus_tau) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    tgt[i] = soft_update_element(tau, one_minus_tau, src[i], tgt[i]);
}

static unsigned grid_for(long n) {
  long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace rg

using namespace rg;

extern "C" {

static int adam_launch(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr,
                       double beta1, double beta2, double eps, double weight_decay, double bias_correction1,
                       double bias_correction2_sqrt, double grad_scale, const double* sched, rg_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || bias_correction1 == 0.0) return RG_EINVAL;
  if (n == 0) return RG_OK;
  const double step_size = lr / bias_correction1;
  const AdamCoef c = {(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                      (float)(-step_size), (float)bias_correction2_sqrt, (float)grad_scale};
  RG_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (long)n,
            c, sched);
  return (int)hipGetLastError();
}

int rg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 double lr, double beta1, double beta2, double eps, double weight_decay,
                 double bias_correction1, double bias_correction2_sqrt, double grad_scale,
                 rg_stream_t stream) {
  return adam_launch(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bias_correction1,
                     bias_correction2_sqrt, grad_scale, nullptr, stream);
}

int rg_adam_step_sched(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double beta1,
                       double beta2, double eps, double weight_decay, double grad_scale, const double* sched,
                       rg_stream_t stream) {
  if (!sched) return RG_EINVAL;
  return adam_launch(param, grad, exp_avg, exp_avg_sq, n, 0.0, beta1, beta2, eps, weight_decay, 1.0, 1.0, grad_scale,
                     sched, stream);
}

int rg_sched_tick(double* sched, rg_stream_t stream) {
  if (!sched) return RG_EINVAL;
  RG_LAUNCH(sched_tick_kernel, dim3(1), dim3(64), (hipStream_t)stream, sched);
  return (int)hipGetLastError();
}

int rg_sched_tick_many(double* const* scheds, int n, rg_stream_t stream) {
  if (!scheds || n < 0 || n > RG_MAX_TICKS) return RG_EINVAL;
  if (n == 0) return RG_OK;
  TickMany t;
  for (int i = 0; i < RG_MAX_TICKS; ++i) {
    t.s[i] = scheds[i < n ? i : 0];
    if (i < n && !scheds[i]) return RG_EINVAL;
    for (int j = 0; j < i && i < n; ++j)
      if (scheds[j] == scheds[i]) return RG_EINVAL;  // one tick per schedule and launch
  }
  RG_LAUNCH(sched_tick_many_kernel, dim3(1), dim3(64), (hipStream_t)stream, t, n);
  return (int)hipGetLastError();
}

int rg_soft_update(float* target, const float* source, int64_t n, double tau, rg_stream_t stream) {
  if (!target || !source || n < 0 || tau < 0.0 || tau > 1.0) return RG_EINVAL;
  if (n == 0) return RG_OK;
  RG_LAUNCH(soft_update_kernel, dim3(grid_for(n)), dim3(256), (hipStream_t)stream, target, source,
            (long)n, (float)tau, (float)(1.0 - tau));
  return (int)hipGetLastError();
}

/* 2: rg_mlp_desc grew (x3, panels, rowmap, grouped output layer), _sched and grouped-QR entry points
 * 3: rg_mlp_desc.dx_only (was reserved) and save = 2; layer norm, dueling, policy input, SAC KLD entry points
 * 4: rg_mlp_desc.x2_dtype (the two input panels may have different element types) */
int rg_abi_version(void) { return 11; }

const char* rg_strerror(int code) {
  switch (code) {
    case RG_OK: return "ok";
    case RG_EINVAL: return "invalid argument";
    case RG_EALIGN: return "misaligned pointer or leading dimension";
    case RG_EUNSUPPORTED: return "unsupported dtype/precision";
    case RG_EWORKSPACE: return "workspace too small";
    default: return code > 0 ? "HIP runtime error (code is hipError_t)" : "unknown error";
  }
}

}  // extern "C"

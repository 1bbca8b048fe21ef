// sac.hip — SAC-specific elementwise kernels: the tanh-squashed Gaussian policy head (forward and
// backward), the critic / actor / temperature loss heads.  All are per-transition VALU work on
// [B, action_dim] or [B] arrays (HBM-bound, tiny next to the FC stacks).
#include <rg_platform.h>
#include "rg_optim.h"
#include "../../include/reagent_hip.h"

namespace rg {

constexpr float LOG_PROB_MIN = -2.f, LOG_PROB_MAX = 2.f;  // reagent/models/actor.py:18-19
constexpr float ACT_EPS = 1e-6f;                          // actor.py:160 self.eps
constexpr float LOG_SQRT_2PI = 0.91893853320467274178f;   // actor.py:159 self.const

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// GaussianFullyConnectedActor.forward (actor.py:215-231) given the FC output loc_scale [B, 2A]:
//   loc, scale_log = split; scale_log.clamp(-2, 2)
//   raw = loc + r * exp(scale_log); action = clamp(tanh(raw), +-(1 - eps))
//   log_prob = get_log_prob(state, action) (actor.py:233-261), which re-derives r' = (atanh(a) - loc)/sigma
// One thread per (b, d) computes the action and its log-prob term; the A terms of a row are summed
// by a row-owner thread in index order.
// Element-parallel form: one thread per (b, d) — the transcendental chain (exp, tanh, atanh, log) of
// the A elements of a row runs on A lanes instead of one after the other on a single lane, which left
// a 65 536-row batch at one wave per SIMD (C4: 116 us per launch) — the per-element terms go through
// LDS and the row owner adds them in index order, i.e. the same order as a sequential loop.
constexpr int GH_THREADS = 256;

__device__ __forceinline__ float gaussian_lp_term(float loc, float s, float a) {
  const float r2 = (atanhf(a) - loc) / expf(s);
  const float normal = -(r2 * r2) / 2.f - s - LOG_SQRT_2PI;  // _normal_log_prob :166-181
  const float squash = logf(1.f - a * a + ACT_EPS);          // _squash_correction :183-188
  return normal - squash;
}

// rows_per_wg = GH_THREADS / A rows per workgroup (A <= GH_THREADS); thread t -> (row t / A, d = t % A)
__global__ void gaussian_head_fwd_kernel(const float* __restrict__ ls, long ldls,
                                         const float* __restrict__ noise, int batch, int A,
                                         float* __restrict__ action, long lda,
                                         float* __restrict__ log_prob, float* __restrict__ squashed_mean) {
  __shared__ float terms[GH_THREADS];
  const int rows_per_wg = GH_THREADS / A;
  const int t = threadIdx.x, rl = t / A, d = t - rl * A;
  const int b = blockIdx.x * rows_per_wg + rl;
  const bool live = rl < rows_per_wg && b < batch;
  if (live) {
    const float loc = ls[(long)b * ldls + d];
    const float s = clampf(ls[(long)b * ldls + A + d], LOG_PROB_MIN, LOG_PROB_MAX);
    const float sigma = expf(s);
    const float raw = loc + noise[(long)b * A + d] * sigma;
    const float a = clampf(tanhf(raw), -1.f + ACT_EPS, 1.f - ACT_EPS);
    action[(long)b * lda + d] = a;
    if (squashed_mean) squashed_mean[(long)b * A + d] = clampf(tanhf(loc), -1.f + ACT_EPS, 1.f - ACT_EPS);
    const float r2 = (atanhf(a) - loc) / sigma;
    const float normal = -(r2 * r2) / 2.f - s - LOG_SQRT_2PI;
    const float squash = logf(1.f - a * a + ACT_EPS);
    terms[t] = normal - squash;
  }
  __syncthreads();
  if (live && d == 0 && log_prob) {
    float lp = 0.f;
    for (int k = 0; k < A; ++k) lp += terms[t + k];
    log_prob[b] = lp;
  }
}

// get_log_prob(state, squashed_action) for a GIVEN action (actor.py:233-261)
__global__ void gaussian_log_prob_kernel(const float* __restrict__ ls, long ldls,
                                         const float* __restrict__ action, long lda, int batch, int A,
                                         float* __restrict__ log_prob) {
  __shared__ float terms[GH_THREADS];
  const int rows_per_wg = GH_THREADS / A;
  const int t = threadIdx.x, rl = t / A, d = t - rl * A;
  const int b = blockIdx.x * rows_per_wg + rl;
  const bool live = rl < rows_per_wg && b < batch;
  if (live) {
    const float loc = ls[(long)b * ldls + d];
    const float s = clampf(ls[(long)b * ldls + A + d], LOG_PROB_MIN, LOG_PROB_MAX);
    terms[t] = gaussian_lp_term(loc, s, action[(long)b * lda + d]);
  }
  __syncthreads();
  if (live && d == 0) {
    float lp = 0.f;
    for (int k = 0; k < A; ++k) lp += terms[t + k];
    log_prob[b] = lp;
  }
}

// one thread per row: action widths beyond a workgroup
__global__ void gaussian_head_fwd_rows_kernel(const float* __restrict__ ls, long ldls,
                                              const float* __restrict__ noise, int batch, int A,
                                              float* __restrict__ action, long lda,
                                              float* __restrict__ log_prob, float* __restrict__ squashed_mean) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  float lp = 0.f;
  for (int d = 0; d < A; ++d) {
    const float loc = ls[(long)b * ldls + d];
    const float s = clampf(ls[(long)b * ldls + A + d], LOG_PROB_MIN, LOG_PROB_MAX);
    const float sigma = expf(s);
    const float raw = loc + noise[(long)b * A + d] * sigma;
    const float a = clampf(tanhf(raw), -1.f + ACT_EPS, 1.f - ACT_EPS);
    action[(long)b * lda + d] = a;
    if (squashed_mean) squashed_mean[(long)b * A + d] = clampf(tanhf(loc), -1.f + ACT_EPS, 1.f - ACT_EPS);
    const float r2 = (atanhf(a) - loc) / sigma;
    lp += (-(r2 * r2) / 2.f - s - LOG_SQRT_2PI) - logf(1.f - a * a + ACT_EPS);
  }
  if (log_prob) log_prob[b] = lp;
}

__global__ void gaussian_log_prob_rows_kernel(const float* __restrict__ ls, long ldls,
                                              const float* __restrict__ action, long lda, int batch, int A,
                                              float* __restrict__ log_prob) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  float lp = 0.f;
  for (int d = 0; d < A; ++d) {
    const float loc = ls[(long)b * ldls + d];
    const float s = clampf(ls[(long)b * ldls + A + d], LOG_PROB_MIN, LOG_PROB_MAX);
    lp += gaussian_lp_term(loc, s, action[(long)b * lda + d]);
  }
  log_prob[b] = lp;
}

// backward of the head: given g_a [B, A] = d loss / d action (may be null) and g_lp [B] = d loss /
// d log_prob, produce d loss / d loc_scale [B, 2A].  Both uses of (loc, scale_log) in the reference
// (sampling and the second FC evaluation inside get_log_prob) are the same values, so their
// gradients add.
// kld_coef (nullable) [2A] = (c0, c1) of the action-embedding KLD term (rg_sac_kld): its gradient is c0_d + c1_d x with
// x the action (kld_on_mean == 0) or the squashed mean clamp(tanh(loc)) (kld_on_mean != 0).
__global__ void gaussian_head_bwd_kernel(const float* __restrict__ ls, long ldls,
                                         const float* __restrict__ noise, const float* __restrict__ g_a,
                                         long ldga, const float* __restrict__ g_lp, int batch, int A,
                                         float* __restrict__ d_ls, long lddls, const float* __restrict__ kld_coef,
                                         int kld_on_mean) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)batch * A) return;
  const int b = (int)(i / A), d = (int)(i % A);
  const float loc = ls[(long)b * ldls + d];
  const float sl = ls[(long)b * ldls + A + d];
  const float s = clampf(sl, LOG_PROB_MIN, LOG_PROB_MAX);
  const float ds_dsl = (sl >= LOG_PROB_MIN && sl <= LOG_PROB_MAX) ? 1.f : 0.f;
  const float sigma = expf(s);
  const float r = noise[(long)b * A + d];
  const float raw = loc + r * sigma;
  const float t = tanhf(raw);
  const float a = clampf(t, -1.f + ACT_EPS, 1.f - ACT_EPS);
  const float da_dt = (t >= -1.f + ACT_EPS && t <= 1.f - ACT_EPS) ? 1.f : 0.f;
  const float da_draw = da_dt * (1.f - t * t);
  const float u = atanhf(a);
  const float r2 = (u - loc) / sigma;
  // partials of lp_d = -r2^2/2 - s - c - log(1 - a^2 + eps)
  const float dlp_da = -r2 * (1.f / (1.f - a * a)) / sigma + 2.f * a / (1.f - a * a + ACT_EPS);
  const float dlp_dloc = r2 / sigma;
  const float dlp_ds = r2 * r2 - 1.f;
  const float glp = g_lp ? g_lp[b] : 0.f;
  float ga = g_a ? g_a[(long)b * ldga + d] : 0.f;
  float d_loc_kld = 0.f;
  if (kld_coef) {
    const float c0 = kld_coef[d], c1 = kld_coef[A + d];
    if (kld_on_mean) {
      const float tm = tanhf(loc);
      const float inside = (tm >= -1.f + ACT_EPS && tm <= 1.f - ACT_EPS) ? 1.f : 0.f;
      d_loc_kld = (c0 + c1 * clampf(tm, -1.f + ACT_EPS, 1.f - ACT_EPS)) * inside * (1.f - tm * tm);
    } else {
      ga += c0 + c1 * a;
    }
  }
  const float tot_a = ga + glp * dlp_da;  // everything that reaches `action`
  const float d_loc = glp * dlp_dloc + tot_a * da_draw + d_loc_kld;
  const float d_s = glp * dlp_ds + tot_a * da_draw * r * sigma;
  d_ls[(long)b * lddls + d] = d_loc;
  d_ls[(long)b * lddls + A + d] = d_s * ds_dsl;
}

constexpr int SAC_THREADS = 256;
__device__ __forceinline__ float block_sum(float v, float* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += shfl_xor(v, off);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  __syncthreads();
  return s;
}

// critic segment (sac_trainer.py:217-248):
//   v' = min(q1_t, q2_t) - alpha * clamp(log_prob', -2, 2); y = r + gamma * v' * not_done (r if gamma == 0)
//   q_i loss = mse(q_i, y);  dq_i = 2 (q_i - y) / B
__global__ void sac_critic_head_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                       const float* __restrict__ q1t, const float* __restrict__ q2t,
                                       const float* __restrict__ lp_next, const float* __restrict__ reward,
                                       const float* __restrict__ not_terminal, float gamma,
                                       const double* __restrict__ alpha, int batch,
                                       float* __restrict__ target, float* __restrict__ dq1,
                                       float* __restrict__ dq2, float* __restrict__ loss1_part,
                                       float* __restrict__ loss2_part) {
  __shared__ float scratch[4];
  const int b = blockIdx.x * SAC_THREADS + threadIdx.x;
  float l1 = 0.f, l2 = 0.f;
  if (b < batch) {
    float v = q1t[b];
    if (q2t) v = fminf(v, q2t[b]);
    v = (float)((double)v - alpha[0] * (double)clampf(lp_next[b], LOG_PROB_MIN, LOG_PROB_MAX));
    const float y = gamma > 0.f ? reward[b] + gamma * v * not_terminal[b] : reward[b];
    target[b] = y;
    const float d1 = q1[b] - y;
    l1 = d1 * d1;
    dq1[b] = 2.f * d1 / (float)batch;
    if (q2) {
      const float d2 = q2[b] - y;
      l2 = d2 * d2;
      dq2[b] = 2.f * d2 / (float)batch;
    }
  }
  const float s1 = block_sum(l1, scratch);
  const float s2 = block_sum(l2, scratch);
  if (threadIdx.x == 0) {
    loss1_part[blockIdx.x] = s1;
    if (loss2_part) loss2_part[blockIdx.x] = s2;
  }
}

// actor segment (sac_trainer.py:254-280): loss = mean(alpha * clamp(log_prob, -2, 2) - min(q1a, q2a))
// -> g_lp [B], dq1a / dq2a [B] (torch.minimum's backward: the smaller input takes the gradient,
// ties are split), loss partials.  Also the temperature segment's input (:311-320):
// ent_part = partial sums of (clamp(log_prob) + target_entropy).
// Variants (sac_trainer.py:262-279): backprop_log_prob == 0 detaches log_prob in the actor loss (g_lp = 0);
// crr_mode != 0 replaces the loss by -(clamp(log_prob) * w), w = CRRWeightFn(min(q1a, q2a) - V(s)) (:23-47):
// 1 = indicator (advantage >= p0), 2 = exp(advantage / p0) clamped to [0, crr_clamp] when crr_clamp > 0; the
// critics then receive no gradient.
__global__ void sac_actor_head_kernel(const float* __restrict__ lp, const float* __restrict__ q1a,
                                      const float* __restrict__ q2a, const double* __restrict__ alpha,
                                      float target_entropy, int batch, const float* __restrict__ v_cur, int crr_mode,
                                      float crr_p0, float crr_clamp, int backprop_log_prob, float* __restrict__ g_lp,
                                      float* __restrict__ dq1a, float* __restrict__ dq2a,
                                      float* __restrict__ loss_part, float* __restrict__ ent_part) {
  __shared__ float scratch[4];
  const int b = blockIdx.x * SAC_THREADS + threadIdx.x;
  float l = 0.f, e = 0.f;
  if (b < batch) {
    const float raw = lp[b];
    const float c = clampf(raw, LOG_PROB_MIN, LOG_PROB_MAX);
    const float inside = (raw >= LOG_PROB_MIN && raw <= LOG_PROB_MAX && backprop_log_prob) ? 1.f : 0.f;
    const float a1 = q1a[b];
    float mq = a1, w1 = 1.f, w2 = 0.f;
    if (q2a) {
      const float a2 = q2a[b];
      mq = fminf(a1, a2);
      if (a1 == a2) { w1 = 0.5f; w2 = 0.5f; }
      else if (a2 < a1) { w1 = 0.f; w2 = 1.f; }
    }
    const double al = alpha[0];
    const float invb = 1.f / (float)batch;
    if (crr_mode) {
      const float adv = mq - v_cur[b];
      float w;
      if (crr_mode == 1) {
        w = adv >= crr_p0 ? 1.f : 0.f;
      } else {
        w = expf(adv / crr_p0);
        if (crr_clamp > 0.f) w = fminf(fmaxf(w, 0.f), crr_clamp);
      }
      l = -(c * w);
      g_lp[b] = -w * inside * invb;
      dq1a[b] = 0.f;
      if (dq2a) dq2a[b] = 0.f;
    } else {
      l = (float)(al * (double)c - (double)mq);
      g_lp[b] = (float)al * inside * invb;
      dq1a[b] = -w1 * invb;
      if (dq2a) dq2a[b] = -w2 * invb;
    }
    e = c + target_entropy;
  }
  const float sl = block_sum(l, scratch);
  const float se = block_sum(e, scratch);
  if (threadIdx.x == 0) {
    loss_part[blockIdx.x] = sl;
    ent_part[blockIdx.x] = se;
  }
}

// temperature segment: alpha_loss = -mean(log_alpha * (clamp(log_prob) + target_entropy)) so
// d/d log_alpha = -mean(...).  One block: ordered sum of the partials, writes the fp64 gradient and
// the loss value.
__global__ void sac_alpha_grad_kernel(const float* __restrict__ ent_part, int nparts, int batch,
                                      const double* __restrict__ log_alpha, double* __restrict__ grad,
                                      double* __restrict__ alpha_loss) {
  __shared__ float scratch[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += SAC_THREADS) acc += ent_part[i];
  const float s = block_sum(acc, scratch);
  if (threadIdx.x == 0) {
    const double m = (double)s / (double)batch;
    grad[0] = -m;
    if (alpha_loss) alpha_loss[0] = -(log_alpha[0] * m);
  }
}

// torch.optim.Adam single-tensor arithmetic in fp64 (log_alpha is a float64 parameter,
// sac_trainer.py:124-126) + alpha = exp(log_alpha) for the next step (:322)
__global__ void adam_f64_kernel(double* __restrict__ p, const double* __restrict__ g, double* __restrict__ m,
                                double* __restrict__ v, int n, double lr, double beta1, double beta2,
                                double eps, double bc1, double bc2_sqrt, double* __restrict__ exp_out,
                                const double* __restrict__ sched) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (sched) sched_lookup(sched, lr, bc1, bc2_sqrt);
  const double gi = g[i];
  double mi = m[i], vi = v[i];
  mi = mi + (1.0 - beta1) * (gi - mi);
  vi = vi * beta2 + ((1.0 - beta2) * gi) * gi;
  const double denom = sqrt(vi) / bc2_sqrt + eps;
  const double pi = p[i] + (-(lr / bc1) * mi) / denom;
  p[i] = pi;
  m[i] = mi;
  v[i] = vi;
  if (exp_out) exp_out[i] = exp(pi);
}

// generic scalar MSE head (critics): loss = mean((q - y)^2), dq = 2 (q - y) / B
__global__ void add_cols_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                int batch, int cols, float* __restrict__ out, long ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)batch * cols) return;
  const long r = i / cols;
  const int c = (int)(i % cols);
  out[r * ldo + c] = a[r * lda + c] + (b ? b[r * ldb + c] : 0.f);
}

// TD3 target-policy smoothing (reagent/training/td3_trainer.py:141-146):
//   a' = clamp(actor_target(s') + clamp(noise * noise_variance, clip_lo, clip_hi), lo, hi)
// written straight into the action columns of the critics' input matrix
__global__ void td3_target_action_kernel(const float* __restrict__ next_actor, long ld_a,
                                         const float* __restrict__ noise, float noise_variance, float clip_lo,
                                         float clip_hi, float lo, float hi, float* __restrict__ out, long ld_out, int batch, int A) {
  const long total = (long)batch * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / A;
    const int a = (int)(i % A);
    float n = noise[i] * noise_variance;
    n = fminf(fmaxf(n, clip_lo), clip_hi);
    const float v = next_actor[b * ld_a + a] + n;
    out[b * ld_out + a] = fminf(fmaxf(v, lo), hi);
  }
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_gaussian_head_forward(const float* loc_scale, int64_t ldls, const float* noise, int batch,
                             int action_dim, float* action, int64_t lda, float* log_prob,
                             float* squashed_mean, rg_stream_t stream) {
  if (!loc_scale || !noise || !action || batch <= 0 || action_dim <= 0) return RG_EINVAL;
  if (action_dim <= GH_THREADS) {
    const int rows = GH_THREADS / action_dim;
    RG_LAUNCH(gaussian_head_fwd_kernel, dim3((batch + rows - 1) / rows), dim3(GH_THREADS), (hipStream_t)stream,
              loc_scale, (long)ldls, noise, batch, action_dim, action, (long)lda, log_prob, squashed_mean);
  } else {
    RG_LAUNCH(gaussian_head_fwd_rows_kernel, dim3((batch + 255) / 256), dim3(256), (hipStream_t)stream, loc_scale,
              (long)ldls, noise, batch, action_dim, action, (long)lda, log_prob, squashed_mean);
  }
  return (int)hipGetLastError();
}

int rg_gaussian_log_prob(const float* loc_scale, int64_t ldls, const float* action, int64_t lda, int batch,
                         int action_dim, float* log_prob, rg_stream_t stream) {
  if (!loc_scale || !action || !log_prob || batch <= 0 || action_dim <= 0) return RG_EINVAL;
  if (action_dim <= GH_THREADS) {
    const int rows = GH_THREADS / action_dim;
    RG_LAUNCH(gaussian_log_prob_kernel, dim3((batch + rows - 1) / rows), dim3(GH_THREADS), (hipStream_t)stream,
              loc_scale, (long)ldls, action, (long)lda, batch, action_dim, log_prob);
  } else {
    RG_LAUNCH(gaussian_log_prob_rows_kernel, dim3((batch + 255) / 256), dim3(256), (hipStream_t)stream, loc_scale,
              (long)ldls, action, (long)lda, batch, action_dim, log_prob);
  }
  return (int)hipGetLastError();
}

int rg_gaussian_head_backward_kld(const float* loc_scale, int64_t ldls, const float* noise, const float* g_action,
                                  int64_t ldga, const float* g_log_prob, int batch, int action_dim,
                                  float* d_loc_scale, int64_t lddls, const float* kld_coef, int kld_on_mean,
                                  rg_stream_t stream) {
  if (!loc_scale || !noise || !d_loc_scale || batch <= 0 || action_dim <= 0) return RG_EINVAL;
  const long n = (long)batch * action_dim;
  RG_LAUNCH(gaussian_head_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)stream,
            loc_scale, (long)ldls, noise, g_action, (long)ldga, g_log_prob, batch, action_dim, d_loc_scale,
            (long)lddls, kld_coef, kld_on_mean);
  return (int)hipGetLastError();
}

int rg_gaussian_head_backward(const float* loc_scale, int64_t ldls, const float* noise, const float* g_action,
                              int64_t ldga, const float* g_log_prob, int batch, int action_dim,
                              float* d_loc_scale, int64_t lddls, rg_stream_t stream) {
  return rg_gaussian_head_backward_kld(loc_scale, ldls, noise, g_action, ldga, g_log_prob, batch, action_dim,
                                       d_loc_scale, lddls, nullptr, 0, stream);
}

// ---- action-embedding KLD term of the actor loss (sac_trainer.py:282-306) ---------------------------------
// per action dimension d over the batch: m = mean(x), v = var(x) (unbiased), x = the sampled action or the squashed
// mean; kld = 0.5 * sum_d ((v + (m - mu)^2) / s2 - 1 + log s2 - log v), loss += weight * kld.
// d (weight * kld) / d x[b, d] = c0_d + c1_d x[b, d] with
//   c1 = weight * (1 / s2 - 1 / v) / (B - 1),   c0 = weight * (m - mu) / (B * s2) - c1 * m
// One workgroup per action dimension: two passes (mean, then centred squares) with fp64 partials combined in a
// fixed order; the per-dimension terms are summed by one thread of the finishing launch.
constexpr int KLD_THREADS = 256;
__device__ __forceinline__ double block_sum_f64(double v, double* scratch) {
  // wave reduction through shuffles of the two halves
  for (int off = 32; off >= 1; off >>= 1) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = shfl_xor((int)bits, off), hi = shfl_xor((int)(bits >> 32), off);
    v += __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
  }
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  const double s = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  __syncthreads();
  return s;
}

__global__ void sac_kld_cols_kernel(const float* __restrict__ x, long ldx, int squash, int batch, int A,
                                    const float* __restrict__ emb_mean, const float* __restrict__ emb_var, float weight,
                                    float* __restrict__ coef, float* __restrict__ kld_terms) {
  __shared__ double scratch[4];
  const int d = blockIdx.x;
  auto value = [&](int b) {
    const float raw = x[(long)b * ldx + d];
    return squash ? clampf(tanhf(raw), -1.f + ACT_EPS, 1.f - ACT_EPS) : raw;
  };
  double s = 0.0;
  for (int b = threadIdx.x; b < batch; b += KLD_THREADS) s += (double)value(b);
  const double m = block_sum_f64(s, scratch) / (double)batch;
  double q = 0.0;
  for (int b = threadIdx.x; b < batch; b += KLD_THREADS) {
    const double c = (double)value(b) - m;
    q += c * c;
  }
  const double v = block_sum_f64(q, scratch) / (double)(batch - 1);
  if (threadIdx.x == 0) {
    const double mu = emb_mean[d], s2 = emb_var[d];
    const double c1 = (double)weight * (1.0 / s2 - 1.0 / v) / (double)(batch - 1);
    const double c0 = (double)weight * (m - mu) / ((double)batch * s2) - c1 * m;
    coef[d] = (float)c0;
    coef[A + d] = (float)c1;
    kld_terms[d] = (float)(0.5 * ((v + (m - mu) * (m - mu)) / s2 - 1.0 + log(s2) - log(v)));
  }
}

__global__ void sac_kld_finish_kernel(const float* __restrict__ kld_terms, int A, float weight, float* __restrict__ kld,
                                      float* __restrict__ loss_inout) {
  if (threadIdx.x || blockIdx.x) return;
  float s = 0.f;
  for (int d = 0; d < A; ++d) s += kld_terms[d];
  if (kld) kld[0] = s;
  if (loss_inout) loss_inout[0] += weight * s;
}

int rg_sac_kld(const float* x, int64_t ldx, int squash, int batch, int action_dim, const float* emb_mean,
               const float* emb_var, double weight, float* coef, float* kld_terms, float* kld, float* loss_inout,
               rg_stream_t stream) {
  if (!x || !emb_mean || !emb_var || !coef || !kld_terms || batch < 2 || action_dim <= 0) return RG_EINVAL;
  RG_LAUNCH(sac_kld_cols_kernel, dim3(action_dim), dim3(KLD_THREADS), (hipStream_t)stream, x, (long)ldx, squash, batch,
            action_dim, emb_mean, emb_var, (float)weight, coef, kld_terms);
  RG_LAUNCH(sac_kld_finish_kernel, dim3(1), dim3(64), (hipStream_t)stream, (const float*)kld_terms, action_dim,
            (float)weight, kld, loss_inout);
  return (int)hipGetLastError();
}

int rg_sac_partials(int batch) { return (batch + SAC_THREADS - 1) / SAC_THREADS; }

int rg_sac_critic_head(const float* q1, const float* q2, const float* q1_target, const float* q2_target,
                       const float* log_prob_next, const float* reward, const float* not_terminal, double gamma,
                       const double* alpha, int batch, float* target, float* dq1, float* dq2,
                       float* loss1_partials, float* loss2_partials, rg_stream_t stream) {
  if (!q1 || !q1_target || !log_prob_next || !reward || !not_terminal || !alpha || !target || !dq1 ||
      !loss1_partials || batch <= 0)
    return RG_EINVAL;
  if (q2 && (!dq2 || !loss2_partials)) return RG_EINVAL;
  RG_LAUNCH(sac_critic_head_kernel, dim3(rg_sac_partials(batch)), dim3(SAC_THREADS), (hipStream_t)stream, q1,
            q2, q1_target, q2_target, log_prob_next, reward, not_terminal, (float)gamma, alpha, batch, target,
            dq1, dq2, loss1_partials, loss2_partials);
  return (int)hipGetLastError();
}

int rg_sac_actor_head(const float* log_prob, const float* q1_actor, const float* q2_actor, const double* alpha,
                      double target_entropy, int batch, const float* v_cur, int crr_mode, double crr_p0, double crr_clamp,
                      int backprop_log_prob, float* g_log_prob, float* dq1_actor, float* dq2_actor,
                      float* loss_partials, float* entropy_partials, rg_stream_t stream) {
  if (!log_prob || !q1_actor || !alpha || !g_log_prob || !dq1_actor || !loss_partials || !entropy_partials ||
      batch <= 0)
    return RG_EINVAL;
  if (q2_actor && !dq2_actor) return RG_EINVAL;
  if (crr_mode < 0 || crr_mode > 2 || (crr_mode && (!v_cur || (crr_mode == 2 && crr_p0 <= 0.0)))) return RG_EINVAL;
  RG_LAUNCH(sac_actor_head_kernel, dim3(rg_sac_partials(batch)), dim3(SAC_THREADS), (hipStream_t)stream,
            log_prob, q1_actor, q2_actor, alpha, (float)target_entropy, batch, v_cur, crr_mode, (float)crr_p0,
            (float)crr_clamp, backprop_log_prob, g_log_prob, dq1_actor, dq2_actor, loss_partials, entropy_partials);
  return (int)hipGetLastError();
}

int rg_sac_alpha_grad(const float* entropy_partials, int batch, const double* log_alpha, double* grad,
                      double* alpha_loss, rg_stream_t stream) {
  if (!entropy_partials || !log_alpha || !grad || batch <= 0) return RG_EINVAL;
  RG_LAUNCH(sac_alpha_grad_kernel, dim3(1), dim3(SAC_THREADS), (hipStream_t)stream, entropy_partials,
            rg_sac_partials(batch), batch, log_alpha, grad, alpha_loss);
  return (int)hipGetLastError();
}

int rg_adam_step_f64(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n, double lr,
                     double beta1, double beta2, double eps, double bias_correction1,
                     double bias_correction2_sqrt, double* exp_param_out, rg_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || bias_correction1 == 0.0) return RG_EINVAL;
  RG_LAUNCH(adam_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)stream, param, grad,
            exp_avg, exp_avg_sq, (int)n, lr, beta1, beta2, eps, bias_correction1, bias_correction2_sqrt,
            exp_param_out, (const double*)nullptr);
  return (int)hipGetLastError();
}

int rg_adam_step_f64_sched(double* param, const double* grad, double* exp_avg, double* exp_avg_sq, int64_t n,
                           double beta1, double beta2, double eps, const double* sched, double* exp_param_out,
                           rg_stream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || !sched) return RG_EINVAL;
  RG_LAUNCH(adam_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)stream, param, grad,
            exp_avg, exp_avg_sq, (int)n, 0.0, beta1, beta2, eps, 1.0, 1.0, exp_param_out, sched);
  return (int)hipGetLastError();
}

int rg_add_cols(const float* a, int64_t lda, const float* b, int64_t ldb, int batch, int cols, float* out,
                int64_t ldo, rg_stream_t stream) {
  if (!a || !out || batch <= 0 || cols <= 0) return RG_EINVAL;
  const long n = (long)batch * cols;
  RG_LAUNCH(add_cols_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)stream, a, (long)lda, b,
            (long)ldb, batch, cols, out, (long)ldo);
  return (int)hipGetLastError();
}

int rg_td3_target_action(const float* next_actor, int64_t ld_a, const float* noise, double noise_variance,
                         double noise_clip_lo, double noise_clip_hi, double lo, double hi, float* out,
                         int64_t ld_out, int batch, int action_dim, rg_stream_t stream) {
  if (!next_actor || !noise || !out || batch <= 0 || action_dim <= 0) return RG_EINVAL;
  const long total = (long)batch * action_dim;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  RG_LAUNCH(td3_target_action_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, next_actor, (long)ld_a,
            noise, (float)noise_variance, (float)noise_clip_lo, (float)noise_clip_hi, (float)lo, (float)hi, out, (long)ld_out, batch,
            action_dim);
  return (int)hipGetLastError();
}

}  // extern "C"

// crr.hip — loss heads of the discrete CRR step.
// Replaces reagent/training/discrete_crr_trainer.py:191-206 (compute_target_q_values), :208-212
// (compute_td_loss, both critics) and :214-285 (compute_actor_loss).  One thread per transition;
// |A| is small (action names), every row is a handful of registers.
#include <rg_platform.h>
#include "../../include/reagent_hip.h"
#include "rg_reduce.h"

namespace rg {

constexpr int CRR_THREADS = 256;

// first maximal entry of a row (torch.argmax)
__device__ __forceinline__ int row_argmax(const float* __restrict__ row, int A) {
  int best = 0;
  float v = row[0];
  for (int a = 1; a < A; ++a)
    if (row[a] > v) {
      v = row[a];
      best = a;
    }
  return best;
}

// log-sum-exp of a row of logits: pyd.Categorical(logits=...) normalises with it
// (torch/distributions/categorical.py: logits - logits.logsumexp(-1))
__device__ __forceinline__ float row_logsumexp(const float* __restrict__ row, int A) {
  float mx = row[0];
  for (int a = 1; a < A; ++a) mx = fmaxf(mx, row[a]);
  float s = 0.f;
  for (int a = 0; a < A; ++a) s += expf(row[a] - mx);
  return mx + logf(s);
}

// target = r (+ boost) + gamma * not_terminal * min_k sum_a Qk_target(s', a) * pi(a | s');
// loss_k = mean (sum_a Qk(s, a) * action - target)^2 ; dQk = action * 2 (q - target) / B
__global__ void crr_critic_head_kernel(const float* __restrict__ q1, const float* __restrict__ q2,
                                       const float* __restrict__ q1_next_t, const float* __restrict__ q2_next_t,
                                       const float* __restrict__ next_logits, const float* __restrict__ action,
                                       const float* __restrict__ reward, const float* __restrict__ reward_boosts,
                                       const float* __restrict__ not_terminal, float gamma, int batch, int A,
                                       float* __restrict__ target_out, float* __restrict__ dq1,
                                       float* __restrict__ dq2, float* __restrict__ partials1,
                                       float* __restrict__ partials2) {
  __shared__ float scratch[4];
  const int b = blockIdx.x * CRR_THREADS + threadIdx.x;
  float l1 = 0.f, l2 = 0.f;
  if (b < batch) {
    const long o = (long)b * A;
    const float lse = row_logsumexp(next_logits + o, A);
    float v1 = 0.f, v2 = 0.f;
    for (int a = 0; a < A; ++a) {
      const float p = expf(next_logits[o + a] - lse);  // next_dist.probs
      v1 += q1_next_t[o + a] * p;
      if (q2_next_t) v2 += q2_next_t[o + a] * p;
    }
    const float next_v = q2_next_t ? fminf(v1, v2) : v1;
    float r = reward[b];
    if (reward_boosts) {  // boost_rewards, dqn_trainer_base.py:216-241
      float boost = 0.f;
      for (int a = 0; a < A; ++a) boost += action[o + a] * reward_boosts[a];
      r += boost;
    }
    const float target = r + gamma * next_v * not_terminal[b];
    if (target_out) target_out[b] = target;
    const float inv = 2.f / (float)batch;
    float qa1 = 0.f, qa2 = 0.f;
    for (int a = 0; a < A; ++a) {
      qa1 += q1[o + a] * action[o + a];
      if (q2) qa2 += q2[o + a] * action[o + a];
    }
    const float d1 = qa1 - target, d2 = qa2 - target;
    l1 = d1 * d1;
    for (int a = 0; a < A; ++a) dq1[o + a] = action[o + a] * d1 * inv;
    if (q2) {
      l2 = d2 * d2;
      for (int a = 0; a < A; ++a) dq2[o + a] = action[o + a] * d2 * inv;
    }
  }
  const float s1 = block_sum_256(l1, scratch);
  const float s2 = block_sum_256(l2, scratch);
  if (threadIdx.x == 0) {
    partials1[blockIdx.x] = s1;
    if (partials2) partials2[blockIdx.x] = s2;
  }
}

// compute_actor_loss (:214-285).  With z the actor's scores, p = softmax(z), i the logged action:
//   weight  = clamp(exp((Q(s,i) - sum_a Q(s,a) p_a) / beta), 0, max_weight)          (detached)
//   plain   = mean(-log p_i * weight)
//   entropy = mean(clip(p_i / pi_b, 1e-4, clip_limit) * log p_i)      (entropy_coeff > 0 only)
//   loss    = plain + entropy_coeff * entropy
// d loss / d z_a = [ (-weight + c * ratio) * (1[a=i] - p_a) + c * inside * log p_i * p_i (1[a=i] - p_a) / pi_b ] / B
// with c = entropy_coeff and `inside` = the clip passes the gradient (min <= x <= max, torch.clamp).
__global__ void crr_actor_head_kernel(const float* __restrict__ q, const float* __restrict__ logits,
                                      const float* __restrict__ action, const float* __restrict__ logged_prob,
                                      float inv_beta, float max_weight, float entropy_coeff, float clip_limit,
                                      int batch, int A, float* __restrict__ dlogits,
                                      float* __restrict__ plain_partials, float* __restrict__ entropy_partials) {
  __shared__ float scratch[4];
  const int b = blockIdx.x * CRR_THREADS + threadIdx.x;
  float plain = 0.f, ent = 0.f;
  if (b < batch) {
    const long o = (long)b * A;
    const float lse = row_logsumexp(logits + o, A);
    float values = 0.f;
    for (int a = 0; a < A; ++a) values += q[o + a] * expf(logits[o + a] - lse);
    float adv = 0.f, pi_t = 0.f;
    for (int a = 0; a < A; ++a) {
      adv += (q[o + a] - values) * action[o + a];
      pi_t += expf(logits[o + a] - lse) * action[o + a];
    }
    const float weight = fminf(fmaxf(expf(inv_beta * adv), 0.f), max_weight);
    const int i = row_argmax(action + o, A);
    const float log_pi = logits[o + i] - lse;
    plain = -log_pi * weight;
    float g_logp = -weight, g_pit = 0.f;  // d/d log p_i and d/d pi_t of the per-row loss
    if (entropy_coeff > 0.f) {
      const float pi_b = logged_prob[b];
      const float raw = pi_t / pi_b;
      const float ratio = fminf(fmaxf(raw, 1e-4f), clip_limit);
      ent = ratio * log_pi;
      g_logp += entropy_coeff * ratio;
      if (raw >= 1e-4f && raw <= clip_limit) g_pit = entropy_coeff * log_pi / pi_b;
    }
    const float inv = 1.f / (float)batch;
    for (int a = 0; a < A; ++a) {
      const float p = expf(logits[o + a] - lse);
      // d log p_i / d z_a = 1[a=i] - p_a ;  d pi_t / d z_a = p_a (action_a - pi_t)
      dlogits[o + a] = (g_logp * ((a == i ? 1.f : 0.f) - p) + g_pit * p * (action[o + a] - pi_t)) * inv;
    }
  }
  const float sp = block_sum_256(plain, scratch);
  const float se = block_sum_256(ent, scratch);
  if (threadIdx.x == 0) {
    plain_partials[blockIdx.x] = sp;
    if (entropy_partials) entropy_partials[blockIdx.x] = se;
  }
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_crr_partials(int batch) { return (batch + CRR_THREADS - 1) / CRR_THREADS; }

int rg_crr_critic_head(const float* q1, const float* q2, const float* q1_next_target, const float* q2_next_target,
                       const float* next_logits, const float* action, const float* reward,
                       const float* reward_boosts, const float* not_terminal, double gamma, int batch,
                       int num_actions, float* target_out, float* dq1, float* dq2, float* partials1,
                       float* partials2, rg_stream_t stream) {
  if (!q1 || !q1_next_target || !next_logits || !action || !reward || !not_terminal || !dq1 || !partials1 ||
      batch <= 0 || num_actions <= 0)
    return RG_EINVAL;
  if ((q2 != nullptr) != (q2_next_target != nullptr) || (q2 && (!dq2 || !partials2))) return RG_EINVAL;
  RG_LAUNCH(crr_critic_head_kernel, dim3(rg_crr_partials(batch)), dim3(CRR_THREADS), (hipStream_t)stream, q1, q2,
            q1_next_target, q2_next_target, next_logits, action, reward, reward_boosts, not_terminal, (float)gamma,
            batch, num_actions, target_out, dq1, dq2, partials1, partials2);
  return (int)hipGetLastError();
}

int rg_crr_actor_head(const float* q, const float* logits, const float* action, const float* logged_prob,
                      double beta, double max_weight, double entropy_coeff, double clip_limit, int batch,
                      int num_actions, float* dlogits, float* plain_partials, float* entropy_partials,
                      rg_stream_t stream) {
  if (!q || !logits || !action || !dlogits || !plain_partials || batch <= 0 || num_actions <= 0 || beta == 0.0)
    return RG_EINVAL;
  if (entropy_coeff > 0.0 && (!logged_prob || !entropy_partials)) return RG_EINVAL;
  // (1 / self.beta) is a Python float the reference multiplies an fp32 tensor by (:237)
  RG_LAUNCH(crr_actor_head_kernel, dim3(rg_crr_partials(batch)), dim3(CRR_THREADS), (hipStream_t)stream, q, logits,
            action, logged_prob, (float)(1.0 / beta), (float)max_weight, (float)entropy_coeff, (float)clip_limit,
            batch, num_actions, dlogits, plain_partials, entropy_partials);
  return (int)hipGetLastError();
}

}  // extern "C"

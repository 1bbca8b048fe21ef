// heads.hip — loss heads that sit between the last FC forward and the first FC backward.
#include <rg_platform.h>
#include "../../include/reagent_hip.h"
#include "rg_reduce.h"

namespace rg {

constexpr int HEAD_THREADS = 256;

// (Finishing the mean loss inside this launch — the workgroup that takes the last ticket sums the
// partials after a device-scope fence — was measured on MI355X: the kernel went from 13.4 to 44 us.
// A release fence at agent scope writes the XCD's L2 back, here with the whole dq matrix dirty in it;
// the separate 6 us rg_reduce_sum launch is the cheaper way to cross the XCDs.)
// One thread per transition.  Masked max / arg-max over |A| with first-index tie-break
// (torch.max semantics), double-Q gather, TD target, MSE / Huber value and d loss / d q.
// dqn_trainer_base.py:33-77 + dqn_trainer.py:201-238.
__global__ void dqn_head_kernel(const float* __restrict__ q, const float* __restrict__ qn_online,
                                const float* __restrict__ qn_target, const float* __restrict__ action,
                                const float* __restrict__ next_mask, const float* __restrict__ reward,
                                const float* __restrict__ reward_boosts,
                                const float* __restrict__ not_terminal, float gamma,
                                const float* __restrict__ gamma_exponent, int batch, int A, int double_q,
                                int loss_type, float* __restrict__ dq, float* __restrict__ loss_partials,
                                float* __restrict__ next_q_out, int64_t* __restrict__ next_idx_out,
                                float* __restrict__ q_sel_out) {
  __shared__ float scratch[4];
  const int b = blockIdx.x * HEAD_THREADS + threadIdx.x;
  float loss = 0.f;
  // rows of 4k floats on 16-byte boundaries are read / written as float4 (a quarter of the memory
  // instructions; same arithmetic in the same order, so results do not depend on the path)
  const bool v4 = (A & 3) == 0 && ((((uintptr_t)q | (uintptr_t)qn_target | (uintptr_t)action | (uintptr_t)next_mask |
                                     (uintptr_t)dq | (uintptr_t)(double_q ? qn_online : qn_target)) & 15) == 0);
  if (b < batch) {
    const long o = (long)b * A;
    float best = 0.f, best_t = 0.f;
    int best_i = 0;
    float rb = 0.f, qs = 0.f;
    if (v4) {
      for (int a0 = 0; a0 < A; a0 += 4) {
        const f32x4 m4 = *(const f32x4*)(next_mask + o + a0);
        const f32x4 qt4 = *(const f32x4*)(qn_target + o + a0);
        const f32x4 qo4 = double_q ? *(const f32x4*)(qn_online + o + a0) : qt4;
        const f32x4 ac4 = *(const f32x4*)(action + o + a0);
        const f32x4 q4 = *(const f32x4*)(q + o + a0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pen = -1e9f * (1.f - m4[e]);
          const float qo = qo4[e] + pen, qt = qt4[e] + pen;
          const float key = double_q ? qo : qt;
          if (a0 + e == 0 || key > best) {
            best = key;
            best_t = qt;
            best_i = a0 + e;
          }
          if (reward_boosts) rb += ac4[e] * reward_boosts[a0 + e];
          qs += q4[e] * ac4[e];
        }
      }
    } else {
      for (int a = 0; a < A; ++a) {
        const float pen = -1e9f * (1.f - next_mask[o + a]);  // ACTION_NOT_POSSIBLE_VAL * (1 - mask)
        const float qo = (double_q ? qn_online[o + a] : qn_target[o + a]) + pen;
        const float qt = qn_target[o + a] + pen;
        const float key = double_q ? qo : qt;
        if (a == 0 || key > best) {
          best = key;
          best_t = qt;
          best_i = a;
        }
      }
      // boost_rewards (dqn_trainer_base.py:216-241)
      if (reward_boosts)
        for (int a = 0; a < A; ++a) rb += action[o + a] * reward_boosts[a];
      for (int a = 0; a < A; ++a) qs += q[o + a] * action[o + a];
    }
    const float next_q = best_t;
    // compute_discount_tensor (dqn_trainer.py:166-177)
    const float rew = reward[b] + rb;
    const float disc = gamma_exponent ? powf(gamma, gamma_exponent[b]) : gamma;
    const float target = rew + disc * (next_q * not_terminal[b]);
    const float d = qs - target;
    float g;
    if (loss_type == RG_LOSS_HUBER) {  // F.smooth_l1_loss, beta = 1
      const float ad = fabsf(d);
      loss = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
      g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
    } else {  // F.mse_loss
      loss = d * d;
      g = 2.f * d;
    }
    g /= (float)batch;
    if (v4) {
      for (int a0 = 0; a0 < A; a0 += 4) {
        const f32x4 ac4 = *(const f32x4*)(action + o + a0);
        *(f32x4*)(dq + o + a0) = f32x4{g * ac4[0], g * ac4[1], g * ac4[2], g * ac4[3]};
      }
    } else {
      for (int a = 0; a < A; ++a) dq[o + a] = g * action[o + a];
    }
    if (next_q_out) next_q_out[b] = next_q;
    if (next_idx_out) next_idx_out[b] = best_i;
    if (q_sel_out) q_sel_out[b] = qs;
  }
  const float s = block_sum_256(loss, scratch);
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = s;
}

// The same head with G = A / 4 lanes per transition (A = 4, 8, 16): every lane holds one float4 of each
// row, so the five [B, A] operands are read with fully coalesced 16-byte requests and the launch has
// G times the waves in flight (one thread per row leaves a 65 536-row batch at one wave per SIMD,
// which is latency-bound).  The row reductions run over the G lanes of a row with xor shuffles:
//   * arg-max: larger key wins, equal keys -> lower index (torch.max's first-index rule);
//   * sum_a q * action and sum_a action * boost: a one-hot row has one non-zero product, so the order
//     of the adds cannot change the result.
// A workgroup still covers 256 transitions (256 * G threads), so loss_partials keeps its length.
template <int G>
__global__ void __launch_bounds__(256 * G) dqn_head_lanes_kernel(
    const float* __restrict__ q, const float* __restrict__ qn_online, const float* __restrict__ qn_target,
    const float* __restrict__ action, const float* __restrict__ next_mask, const float* __restrict__ reward,
    const float* __restrict__ reward_boosts, const float* __restrict__ not_terminal, float gamma,
    const float* __restrict__ gamma_exponent, int batch, int double_q, int loss_type, float* __restrict__ dq,
    float* __restrict__ loss_partials, float* __restrict__ next_q_out, int64_t* __restrict__ next_idx_out,
    float* __restrict__ q_sel_out) {
  constexpr int A = 4 * G, WAVES = 4 * G;
  __shared__ float scratch[WAVES];
  const int t = threadIdx.x, sub = t % G;
  const int b_raw = blockIdx.x * 256 + t / G;
  // rows past the end recompute the last row and store nothing: every lane takes part in the shuffles
  const bool live = b_raw < batch;
  const int b = live ? b_raw : batch - 1;
  float loss = 0.f;
  {
    const long o = (long)b * A + sub * 4;
    const f32x4 m4 = *(const f32x4*)(next_mask + o);
    const f32x4 qt4 = *(const f32x4*)(qn_target + o);
    const f32x4 qo4 = double_q ? *(const f32x4*)(qn_online + o) : qt4;
    const f32x4 ac4 = *(const f32x4*)(action + o);
    const f32x4 q4 = *(const f32x4*)(q + o);
    float best = 0.f, best_t = 0.f, rb = 0.f, qs = 0.f;
    int best_i = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float pen = -1e9f * (1.f - m4[e]);  // ACTION_NOT_POSSIBLE_VAL * (1 - mask)
      const float qo = qo4[e] + pen, qt = qt4[e] + pen;
      const float key = double_q ? qo : qt;
      if (e == 0 || key > best) {
        best = key;
        best_t = qt;
        best_i = sub * 4 + e;
      }
      if (reward_boosts) rb += ac4[e] * reward_boosts[sub * 4 + e];
      qs += q4[e] * ac4[e];
    }
#pragma unroll
    for (int off = 1; off < G; off <<= 1) {
      const float ok = shfl_xor(best, off), ot = shfl_xor(best_t, off);
      const int oi = shfl_xor(best_i, off);
      if (ok > best || (ok == best && oi < best_i)) {
        best = ok;
        best_t = ot;
        best_i = oi;
      }
      rb += shfl_xor(rb, off);
      qs += shfl_xor(qs, off);
    }
    const float rew = reward[b] + rb;
    const float disc = gamma_exponent ? powf(gamma, gamma_exponent[b]) : gamma;
    const float target = rew + disc * (best_t * not_terminal[b]);
    const float d = qs - target;
    float g, row_loss;
    if (loss_type == RG_LOSS_HUBER) {
      const float ad = fabsf(d);
      row_loss = ad < 1.f ? 0.5f * d * d : ad - 0.5f;
      g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
    } else {
      row_loss = d * d;
      g = 2.f * d;
    }
    g /= (float)batch;
    if (live) *(f32x4*)(dq + o) = f32x4{g * ac4[0], g * ac4[1], g * ac4[2], g * ac4[3]};
    if (live && sub == 0) {
      loss = row_loss;
      if (next_q_out) next_q_out[b] = best_t;
      if (next_idx_out) next_idx_out[b] = best_i;
      if (q_sel_out) q_sel_out[b] = qs;
    }
  }
  // workgroup sum in a fixed order: wave shuffles, then the wave sums added in order
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) loss += shfl_xor(loss, off);
  if ((t & 63) == 0) scratch[t >> 6] = loss;
  __syncthreads();
  if (t == 0) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) s += scratch[w];
    loss_partials[blockIdx.x] = s;
  }
}

// CPE heads of the DQN step (reagent/training/dqn_trainer_base.py:338-452, _calculate_cpes): one
// thread per transition, M metrics x A actions.
//   propensities = masked_softmax(all_next_action_scores, next_mask, temperature)
//                  (core/torch_utils.py:62-73: x/T, -(1-mask)*1e20, -rowmax, exp, *mask, /sum, NaN->0)
//   reward net   : mse(reward_est[b, i*A + a_b], real[b, i]),                  a_b = argmax(action[b])
//   CPE q-net    : loss(q_cpe[b, i*A + a_b],
//                       real[b, i] + discount_b * not_done_b * sum_a tgt_next[b, i*A + a] * prop[a])
//   real[b, 0] = reward[b] (unboosted), real[b, i >= 1] = extra_metrics[b, i-1]
// Writes both gradients w.r.t. the network outputs (zero except the logged action's column of
// every metric) and per-workgroup loss partials (means are over B*M elements).
__global__ void cpe_head_kernel(const float* __restrict__ reward_est, const float* __restrict__ q_cpe,
                                const float* __restrict__ q_cpe_tgt_next, const float* __restrict__ next_scores,
                                const float* __restrict__ next_mask, const float* __restrict__ action,
                                const float* __restrict__ reward, const float* __restrict__ extra_metrics,
                                const float* __restrict__ not_terminal, float gamma,
                                const float* __restrict__ gamma_exponent, float temperature, int batch, int A,
                                int M, int loss_type, float* __restrict__ d_reward_est, float* __restrict__ d_q_cpe,
                                float* __restrict__ reward_partials, float* __restrict__ cpe_partials,
                                float* __restrict__ propensities_out) {
  __shared__ float scratch[4];
  const int b = blockIdx.x * HEAD_THREADS + threadIdx.x;
  float loss_r = 0.f, loss_c = 0.f;
  if (b < batch) {
    const long o = (long)b * A, om = (long)b * M * A;
    // logged action = first maximal entry of the action row (torch.argmax)
    int a_log = 0;
    float a_best = action[o];
    for (int a = 1; a < A; ++a)
      if (action[o + a] > a_best) {
        a_best = action[o + a];
        a_log = a;
      }
    // masked softmax, pass 1: row max of x/T - (1-mask)*1e20 ; pass 2: sum of exp * mask
    float mx = 0.f;
    for (int a = 0; a < A; ++a) {
      const float v = next_scores[o + a] / temperature - ((1.0f - next_mask[o + a]) * 1e20f);
      if (a == 0 || v > mx) mx = v;
    }
    float den = 0.f;
    for (int a = 0; a < A; ++a) {
      const float v = next_scores[o + a] / temperature - ((1.0f - next_mask[o + a]) * 1e20f);
      den += expf(v - mx) * next_mask[o + a];
    }
    if (propensities_out)
      for (int a = 0; a < A; ++a) {
        const float v = next_scores[o + a] / temperature - ((1.0f - next_mask[o + a]) * 1e20f);
        const float p = expf(v - mx) * next_mask[o + a] / den;
        propensities_out[o + a] = p != p ? 0.f : p;
      }
    const float disc = gamma_exponent ? powf(gamma, gamma_exponent[b]) : gamma;
    const float nd = not_terminal[b];
    const float inv = 1.f / ((float)batch * (float)M);
    for (int i = 0; i < M; ++i) {
      const float real = i == 0 ? reward[b] : extra_metrics[(long)b * (M - 1) + (i - 1)];
      for (int a = 0; a < A; ++a) {
        d_reward_est[om + (long)i * A + a] = 0.f;
        d_q_cpe[om + (long)i * A + a] = 0.f;
      }
      const long at = om + (long)i * A + a_log;
      const float dr = reward_est[at] - real;  // F.mse_loss
      loss_r += dr * dr;
      d_reward_est[at] = 2.f * dr * inv;
      float nextq = 0.f;
      for (int a = 0; a < A; ++a) {
        const float v = next_scores[o + a] / temperature - ((1.0f - next_mask[o + a]) * 1e20f);
        float p = expf(v - mx) * next_mask[o + a] / den;
        p = p != p ? 0.f : p;
        nextq += q_cpe_tgt_next[om + (long)i * A + a] * p;
      }
      nextq *= nd;
      const float target = real + disc * nextq;
      const float d = q_cpe[at] - target;
      float g;
      if (loss_type == RG_LOSS_HUBER) {
        const float ad = fabsf(d);
        loss_c += ad < 1.f ? 0.5f * d * d : ad - 0.5f;
        g = ad < 1.f ? d : (d > 0.f ? 1.f : -1.f);
      } else {
        loss_c += d * d;
        g = 2.f * d;
      }
      d_q_cpe[at] = g * inv;
    }
  }
  const float sr = block_sum_256(loss_r, scratch);
  const float sc = block_sum_256(loss_c, scratch);
  if (threadIdx.x == 0) {
    reward_partials[blockIdx.x] = sr;
    cpe_partials[blockIdx.x] = sc;
  }
}

// QR-DQN head (reagent/training/qrdqn_trainer.py:108-160): one workgroup per transition.
//   next atoms : target(next_state)[b, a*, :] with a* = argmax_a mean_atoms(online or target)(+mask)
//                (maxq) or sum_a target[b,a,:] * next_action[b,a] (SARSA)
//   target_Q   = r + (gamma * not_done) * next atoms                                  [N]
//   current    = sum_a q[b,a,:] * action[b,a]                                         [N]
//   loss       = mean_{i,b,j} huber(T_i - C_j) * |tau_j - 1[T_i - C_j < 0]|   (the (N,B,N) tensor of
//                the reference is never materialised: the N x N pairs live in registers / LDS)
//   dq[b,a,j]  = action[b,a] * d loss / d C_j
constexpr int QR_MAX_ATOMS = 1024;
constexpr int QR_MAX_ACTIONS = 256;

__global__ void qr_head_kernel(const float* __restrict__ q, const float* __restrict__ qn_online,
                               const float* __restrict__ qn_target, const float* __restrict__ action,
                               const float* __restrict__ next_mask, const float* __restrict__ reward,
                               const float* __restrict__ reward_boosts,
                               const float* __restrict__ not_terminal, float gamma,
                               const float* __restrict__ gamma_exponent,
                               const float* __restrict__ quantiles, int batch, int A, int N, int maxq,
                               float* __restrict__ dq, float* __restrict__ loss_partials,
                               float* __restrict__ all_q) {
  __shared__ float T[QR_MAX_ATOMS];
  __shared__ float C[QR_MAX_ATOMS];
  __shared__ float G[QR_MAX_ATOMS];
  __shared__ float means[QR_MAX_ACTIONS];
  __shared__ float scratch[4];
  __shared__ int a_star;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = (long)b * A * N;
  const float* act_row = action + (long)b * A;
  const float* mask_row = next_mask + (long)b * A;
  // mean over atoms per action (next-state selection values and, for logging, current q)
  for (int a = wave; a < A; a += HEAD_THREADS / 64) {
    const float* sel = (qn_online ? qn_online : qn_target) + row + (long)a * N;
    const float* cur = q + row + (long)a * N;
    float s = 0.f, c = 0.f;
    for (int j = lane; j < N; j += 64) {
      s += sel[j];
      c += cur[j];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      s += shfl_xor(s, off);
      c += shfl_xor(c, off);
    }
    if (lane == 0) {
      means[a] = s / (float)N;
      if (all_q) all_q[(long)b * A + a] = c / (float)N;
    }
  }
  __syncthreads();
  if (tid == 0 && maxq) {
    int best = 0;
    float bv = 0.f;
    for (int a = 0; a < A; ++a) {
      const float v = means[a] + -1e9f * (1.f - mask_row[a]);
      if (a == 0 || v > bv) {
        bv = v;
        best = a;
      }
    }
    a_star = best;
  }
  __syncthreads();
  float rb = 0.f;
  if (reward_boosts)
    for (int a = 0; a < A; ++a) rb += act_row[a] * reward_boosts[a];
  const float rew = reward[b] + rb;
  const float disc = gamma_exponent ? powf(gamma, gamma_exponent[b]) : gamma;
  const float dn = disc * not_terminal[b];
  for (int j = tid; j < N; j += HEAD_THREADS) {
    float nq;
    if (maxq) {
      nq = qn_target[row + (long)a_star * N + j];
    } else {
      nq = 0.f;
      for (int a = 0; a < A; ++a) nq += qn_target[row + (long)a * N + j] * mask_row[a];
    }
    T[j] = rew + dn * nq;
    float c = 0.f;
    for (int a = 0; a < A; ++a) c += q[row + (long)a * N + j] * act_row[a];
    C[j] = c;
  }
  __syncthreads();
  const float inv = 1.f / ((float)N * (float)batch * (float)N);
  float loss = 0.f;
  for (int j = tid; j < N; j += HEAD_THREADS) {
    const float cj = C[j], tau = quantiles[j];
    float l = 0.f, g = 0.f;
    for (int i = 0; i < N; ++i) {
      const float td = T[i] - cj;
      const float ad = fabsf(td);
      const float w = fabsf(tau - (td < 0.f ? 1.f : 0.f));
      l += (ad < 1.f ? 0.5f * td * td : ad - 0.5f) * w;
      g += (ad < 1.f ? td : (td > 0.f ? 1.f : -1.f)) * w;
    }
    loss += l;
    G[j] = -g * inv;
  }
  __syncthreads();
  for (int k = tid; k < A * N; k += HEAD_THREADS) dq[row + k] = act_row[k / N] * G[k % N];
  const float s = block_sum_256(loss, scratch);
  if (tid == 0) loss_partials[b] = s * inv;
}

// C51 head (reagent/training/c51_trainer.py:98-187, models/categorical_dqn.py:36-38): one workgroup
// per transition; logits viewed (B, A, N).
//   dist(x)[a, j]  = exp(log_softmax(x[a, :])[j])
//   next dist      : target dist of a* = argmax_a (sum_j dist_sel[a, j] * support[j] + mask penalty),
//                    sel = online net (double Q) or target net; SARSA: sum_a target dist[a] * next_action[a]
//   projection     : tq_j = clamp(r + disc * not_terminal * support_j, qmin, qmax); b_j = (tq_j - qmin) / dz;
//                    l = floor, u = ceil with the reference's l == b == u fix-ups; m[l] += p_j (u - b),
//                    m[u] += p_j (b - l)  — the two scatter_adds, applied in j order like torch's on the CPU
//   loss           = -sum_j m[j] * sum_a action[a] * log_softmax(q[a, :])[j]          (mean over the batch)
//   dq[a, k]       = action[a] * (softmax(q[a, :])[k] * sum_j m[j] - m[k]) / B
constexpr int C51_MAX_ATOMS = 1024;
constexpr int C51_MAX_ACTIONS = 256;

__global__ void c51_head_kernel(const float* __restrict__ q, const float* __restrict__ qn_online,
                                const float* __restrict__ qn_target, const float* __restrict__ action,
                                const float* __restrict__ next_mask, const float* __restrict__ reward,
                                const float* __restrict__ reward_boosts, const float* __restrict__ not_terminal,
                                float gamma, const float* __restrict__ gamma_exponent,
                                const float* __restrict__ support, float qmin, float qmax, float scale_support,
                                int batch, int A, int N, int maxq, float* __restrict__ dq,
                                float* __restrict__ loss_partials, float* __restrict__ all_q) {
  __shared__ float P[C51_MAX_ATOMS];   // next distribution
  __shared__ float Mm[C51_MAX_ATOMS];  // projected target distribution m
  __shared__ float LD[C51_MAX_ATOMS];  // sum_a action[a] * log_dist(state)[a, :]
  __shared__ float W0[C51_MAX_ATOMS], W1[C51_MAX_ATOMS];  // a source atom's contributions to bins lo / up
  __shared__ short LO[C51_MAX_ATOMS], UP[C51_MAX_ATOMS];
  __shared__ float t_max[C51_MAX_ACTIONS], t_lse[C51_MAX_ACTIONS];  // target net, per action
  __shared__ float c_max[C51_MAX_ACTIONS], c_lse[C51_MAX_ACTIONS];  // online net on `state`
  __shared__ float sel_q[C51_MAX_ACTIONS];                          // next-state expected values
  __shared__ float scratch[4];
  __shared__ int a_star;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = (long)b * A * N;
  const float* act_row = action + (long)b * A;
  const float* mask_row = next_mask + (long)b * A;
  // per action: log-softmax statistics (max, log sum exp) of the three logit rows and E[support]
  auto row_stats = [&](const float* x, float& mx, float& lse) {
    float m = -3.4e38f;
    for (int j = lane; j < N; j += 64) m = fmaxf(m, x[j]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, shfl_xor(m, off));
    float se = 0.f;
    for (int j = lane; j < N; j += 64) se += expf(x[j] - m);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) se += shfl_xor(se, off);
    mx = m;
    lse = logf(se);
  };
  auto expectation = [&](const float* x, float mx, float lse) {
    float e = 0.f;
    for (int j = lane; j < N; j += 64) e += expf((x[j] - mx) - lse) * support[j];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) e += shfl_xor(e, off);
    return e;
  };
  auto wave_max = [&](float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, shfl_xor(v, off));
    return v;
  };
  auto wave_sum = [&](float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += shfl_xor(v, off);
    return v;
  };
  if (N <= 64) {
    // one atom per lane: the three logit rows of an action are loaded once, all of a wave's actions up
    // front (the generic loop below re-reads each row for max / sum-exp / expectation, one dependent
    // L2 round trip after the other — with 65 536 short workgroups that latency was the kernel's time).
    // Same operations in the same order per lane and the same xor-shuffle reductions: same bits.
    constexpr int PER_WAVE = C51_MAX_ACTIONS / (HEAD_THREADS / 64) > 8 ? 8 : C51_MAX_ACTIONS / (HEAD_THREADS / 64);
    for (int a0 = wave; a0 < A; a0 += (HEAD_THREADS / 64) * PER_WAVE) {
      float vt[PER_WAVE], vc[PER_WAVE], vo[PER_WAVE];
      const bool in = lane < N;
      const float sup = in ? support[lane] : 0.f;
#pragma unroll
      for (int u = 0; u < PER_WAVE; ++u) {
        const int a = a0 + u * (HEAD_THREADS / 64);
        vt[u] = vc[u] = vo[u] = 0.f;
        if (a < A) {  // wave-uniform
          const long o = row + (long)a * N + (in ? lane : 0);
          vt[u] = qn_target[o];
          vc[u] = q[o];
          if (qn_online) vo[u] = qn_online[o];
        }
      }
#pragma unroll
      for (int u = 0; u < PER_WAVE; ++u) {
        const int a = a0 + u * (HEAD_THREADS / 64);
        if (a >= A) break;  // wave-uniform
        const float tm = wave_max(in ? vt[u] : -3.4e38f), tl = logf(wave_sum(in ? expf(vt[u] - tm) : 0.f));
        const float cm = wave_max(in ? vc[u] : -3.4e38f), cl = logf(wave_sum(in ? expf(vc[u] - cm) : 0.f));
        float sq;
        if (qn_online) {
          const float om = wave_max(in ? vo[u] : -3.4e38f), ol = logf(wave_sum(in ? expf(vo[u] - om) : 0.f));
          sq = wave_sum(in ? expf((vo[u] - om) - ol) * sup : 0.f);
        } else {
          sq = wave_sum(in ? expf((vt[u] - tm) - tl) * sup : 0.f);
        }
        const float cq = all_q ? wave_sum(in ? expf((vc[u] - cm) - cl) * sup : 0.f) : 0.f;
        if (lane == 0) {
          t_max[a] = tm; t_lse[a] = tl; c_max[a] = cm; c_lse[a] = cl; sel_q[a] = sq;
          if (all_q) all_q[(long)b * A + a] = cq;
        }
      }
    }
  } else
  for (int a = wave; a < A; a += HEAD_THREADS / 64) {
    float tm, tl, cm, cl;
    row_stats(qn_target + row + (long)a * N, tm, tl);
    row_stats(q + row + (long)a * N, cm, cl);
    float sq;
    if (qn_online) {
      float om, ol;
      row_stats(qn_online + row + (long)a * N, om, ol);
      sq = expectation(qn_online + row + (long)a * N, om, ol);
    } else {
      sq = expectation(qn_target + row + (long)a * N, tm, tl);
    }
    const float cq = all_q ? expectation(q + row + (long)a * N, cm, cl) : 0.f;
    if (lane == 0) {
      t_max[a] = tm; t_lse[a] = tl; c_max[a] = cm; c_lse[a] = cl; sel_q[a] = sq;
      if (all_q) all_q[(long)b * A + a] = cq;
    }
  }
  __syncthreads();
  if (tid == 0 && maxq) {  // argmax_with_mask (:203-211): first maximal index
    int best = 0;
    float bv = 0.f;
    for (int a = 0; a < A; ++a) {
      const float v = sel_q[a] + -1e9f * (1.f - mask_row[a]);
      if (a == 0 || v > bv) {
        bv = v;
        best = a;
      }
    }
    a_star = best;
  }
  __syncthreads();
  for (int j = tid; j < N; j += HEAD_THREADS) {
    float p = 0.f, ld = 0.f;
    if (maxq) {
      const int a = a_star;
      p = expf((qn_target[row + (long)a * N + j] - t_max[a]) - t_lse[a]);
    } else {
      for (int a = 0; a < A; ++a) {
        const float w = mask_row[a];  // next_action one-hot
        if (w != 0.f) p += expf((qn_target[row + (long)a * N + j] - t_max[a]) - t_lse[a]) * w;
      }
    }
    for (int a = 0; a < A; ++a) {
      const float w = act_row[a];
      if (w != 0.f) ld += ((q[row + (long)a * N + j] - c_max[a]) - c_lse[a]) * w;
    }
    P[j] = p;
    LD[j] = ld;
    Mm[j] = 0.f;
  }
  __syncthreads();
  // The categorical projection (:136-155): m.scatter_add_(lo, p * (u - b)) then m.scatter_add_(up, p * (b - l)).
  // One thread walking the 2N adds in order was the serial tail of this kernel (C51 step: 1.6 of 5.1 ms
  // in this head).  Same arithmetic in parallel: every source atom's (lo, up, weights) is computed once,
  // then target bin k collects its own adds — first the `lo` pass in source order, then the `up` pass —
  // which is exactly the order in which the sequential scatter_adds reach that bin.
  {
    float rb = 0.f;
    if (reward_boosts)
      for (int a = 0; a < A; ++a) rb += act_row[a] * reward_boosts[a];
    const float rew = reward[b] + rb;
    const float disc = gamma_exponent ? powf(gamma, gamma_exponent[b]) : gamma;
    const float dn = disc * not_terminal[b];
    for (int j = tid; j < N; j += HEAD_THREADS) {
      float tq = rew + dn * support[j];
      tq = fminf(fmaxf(tq, qmin), qmax);
      const float bpos = (tq - qmin) / scale_support;
      int lo = (int)floorf(bpos), up = (int)ceilf(bpos);
      if (up > 0 && lo == up) lo -= 1;
      if (lo < N - 1 && lo == up) up += 1;
      LO[j] = lo;
      UP[j] = up;
      W0[j] = P[j] * ((float)up - bpos);
      W1[j] = P[j] * (bpos - (float)lo);
    }
  }
  __syncthreads();
  for (int k = tid; k < N; k += HEAD_THREADS) {
    float m = 0.f;
    for (int j = 0; j < N; ++j)
      if (LO[j] == k) m += W0[j];
    for (int j = 0; j < N; ++j)
      if (UP[j] == k) m += W1[j];
    Mm[k] = m;
  }
  __syncthreads();
  float lsum = 0.f, msum = 0.f;
  for (int j = tid; j < N; j += HEAD_THREADS) {
    lsum += Mm[j] * LD[j];
    msum += Mm[j];
  }
  const float tot_l = block_sum_256(lsum, scratch);
  const float tot_m = block_sum_256(msum, scratch);
  const float inv_b = 1.f / (float)batch;
  for (int i = tid; i < A * N; i += HEAD_THREADS) {
    const int a = i / N, k = i % N;
    const float w = act_row[a];
    float g = 0.f;
    if (w != 0.f) g = w * (expf((q[row + i] - c_max[a]) - c_lse[a]) * tot_m - Mm[k]) * inv_b;
    dq[row + i] = g;
  }
  if (tid == 0) loss_partials[b] = -tot_l * inv_b;
}

// get_valid_actions_from_imitator (reagent/training/imitator_training.py:12-25) applied to a mask:
//   p = softmax(imitator logits); keep[a] = (p[a] / max_a p) >= drop_threshold; mask[b, a] *= keep[a]
// (batch-constrained q-learning, dqn_trainer.py:209-215).  One thread per transition, the reference's
// operation order (exp(x - max) / sum, then the division by the row maximum of the probabilities).
__global__ void bcq_filter_kernel(const float* __restrict__ logits, int batch, int A, float drop_threshold,
                                  float* __restrict__ mask) {
  const int b = blockIdx.x * HEAD_THREADS + threadIdx.x;
  if (b >= batch) return;
  const long o = (long)b * A;
  float mx = logits[o];
  for (int a = 1; a < A; ++a) mx = fmaxf(mx, logits[o + a]);
  float den = 0.f;
  for (int a = 0; a < A; ++a) den += expf(logits[o + a] - mx);
  float pmax = 0.f;
  for (int a = 0; a < A; ++a) pmax = fmaxf(pmax, expf(logits[o + a] - mx) / den);
  for (int a = 0; a < A; ++a) {
    const float keep = ((expf(logits[o + a] - mx) / den) / pmax >= drop_threshold) ? 1.f : 0.f;
    mask[o + a] *= keep;
  }
}

__global__ void reduce_sum_kernel(const float* __restrict__ in, int n, float scale,
                                  float* __restrict__ out) {
  __shared__ float scratch[4];
  static_assert(HEAD_THREADS == 256, "strided_sum_256");
  const float s = block_sum_256(strided_sum_256(in, n, threadIdx.x), scratch);
  if (threadIdx.x == 0) out[0] = s * scale;
}

}  // namespace rg

namespace rg {
// Dueling aggregation (reagent/models/dueling_q_network.py:96-107), value [B, N], raw advantage [B, A*N] viewed
// (B, A, N) (N = 1 without atoms):  q[b,a,n] = value[b,n] + adv[b,a,n] - mean over (a, n) of adv[b].
// One wave per row: lanes stride the A*N entries, the mean is a wave reduction in fixed order.
__global__ void dueling_combine_kernel(const float* __restrict__ value, long ldv, const float* __restrict__ adv, long lda,
                                       int batch, int A, int N, float* __restrict__ q, long ldq) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= batch) return;
  const int M = A * N;
  float s = 0.f;
  for (int i = lane; i < M; i += 64) s += adv[(long)b * lda + i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor(s, off);
  const float mean = s / (float)M;
  for (int i = lane; i < M; i += 64) q[(long)b * ldq + i] = value[(long)b * ldv + i % N] + (adv[(long)b * lda + i] - mean);
}
// its adjoint: dadv[b,a,n] = dq[b,a,n] - mean over (a, n) of dq[b];  dvalue[b,n] = sum over a of dq[b,a,n]
__global__ void dueling_split_kernel(const float* __restrict__ dq, long lddq, int batch, int A, int N,
                                     float* __restrict__ dadv, long ldda, float* __restrict__ dvalue, long lddv) {
  const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= batch) return;
  const int M = A * N;
  float s = 0.f;
  for (int i = lane; i < M; i += 64) s += dq[(long)b * lddq + i];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) s += shfl_xor(s, off);
  const float mean = s / (float)M;
  for (int i = lane; i < M; i += 64) dadv[(long)b * ldda + i] = dq[(long)b * lddq + i] - mean;
  for (int n = lane; n < N; n += 64) {
    float t = 0.f;
    for (int a = 0; a < A; ++a) t += dq[(long)b * lddq + (long)a * N + n];
    dvalue[(long)b * lddv + n] = t;
  }
}
}  // namespace rg

using namespace rg;

extern "C" {

int rg_dueling_combine(const float* value, int64_t ldv, const float* advantage, int64_t lda, int batch, int num_actions,
                       int num_atoms, float* q, int64_t ldq, rg_stream_t stream) {
  if (!value || !advantage || !q || batch <= 0 || num_actions <= 0 || num_atoms <= 0) return RG_EINVAL;
  RG_LAUNCH(dueling_combine_kernel, dim3((batch + 3) / 4), dim3(256), (hipStream_t)stream, value, (long)ldv, advantage,
            (long)lda, batch, num_actions, num_atoms, q, (long)ldq);
  return (int)hipGetLastError();
}

int rg_dueling_split(const float* dq, int64_t lddq, int batch, int num_actions, int num_atoms, float* dadvantage,
                     int64_t ldda, float* dvalue, int64_t lddv, rg_stream_t stream) {
  if (!dq || !dadvantage || !dvalue || batch <= 0 || num_actions <= 0 || num_atoms <= 0) return RG_EINVAL;
  RG_LAUNCH(dueling_split_kernel, dim3((batch + 3) / 4), dim3(256), (hipStream_t)stream, dq, (long)lddq, batch, num_actions,
            num_atoms, dadvantage, (long)ldda, dvalue, (long)lddv);
  return (int)hipGetLastError();
}

int rg_dqn_head_partials(int batch) { return (batch + HEAD_THREADS - 1) / HEAD_THREADS; }

int rg_dqn_head(const float* q, const float* qn_online, const float* qn_target, const float* action,
                const float* next_mask, const float* reward, const float* reward_boosts,
                const float* not_terminal, double gamma, const float* gamma_exponent, int batch,
                int num_actions, int double_q, int loss_type, float* dq, float* loss_partials,
                float* next_q, int64_t* next_idx, float* q_sel, rg_stream_t stream) {
  if (!q || !qn_online || !qn_target || !action || !next_mask || !reward || !not_terminal || !dq ||
      !loss_partials || batch <= 0 || num_actions <= 0)
    return RG_EINVAL;
  if (loss_type != RG_LOSS_MSE && loss_type != RG_LOSS_HUBER) return RG_EINVAL;
  const bool aligned = ((((uintptr_t)q | (uintptr_t)qn_target | (uintptr_t)action | (uintptr_t)next_mask |
                          (uintptr_t)dq | (uintptr_t)(double_q ? qn_online : qn_target)) & 15) == 0);
  const dim3 grid(rg_dqn_head_partials(batch));
#define RG_HEAD_LANES(G)                                                                                       \
  RG_LAUNCH(dqn_head_lanes_kernel<G>, grid, dim3(256 * G), (hipStream_t)stream, q, qn_online, qn_target, action, \
            next_mask, reward, reward_boosts, not_terminal, (float)gamma, gamma_exponent, batch, double_q,     \
            loss_type, dq, loss_partials, next_q, next_idx, q_sel)
  if (aligned && num_actions == 16) RG_HEAD_LANES(4);
  else if (aligned && num_actions == 8) RG_HEAD_LANES(2);
  else if (aligned && num_actions == 4) RG_HEAD_LANES(1);
  else
    RG_LAUNCH(dqn_head_kernel, grid, dim3(HEAD_THREADS), (hipStream_t)stream, q, qn_online, qn_target, action,
              next_mask, reward, reward_boosts, not_terminal, (float)gamma, gamma_exponent, batch, num_actions,
              double_q, loss_type, dq, loss_partials, next_q, next_idx, q_sel);
#undef RG_HEAD_LANES
  return (int)hipGetLastError();
}

int rg_cpe_head(const float* reward_est, const float* q_cpe, const float* q_cpe_tgt_next, const float* next_scores,
                const float* next_mask, const float* action, const float* reward, const float* extra_metrics,
                const float* not_terminal, double gamma, const float* gamma_exponent, double temperature, int batch,
                int num_actions, int num_metrics, int loss_type, float* d_reward_est, float* d_q_cpe,
                float* reward_partials, float* cpe_partials, float* propensities_out, rg_stream_t stream) {
  if (!reward_est || !q_cpe || !q_cpe_tgt_next || !next_scores || !next_mask || !action || !reward || !not_terminal ||
      !d_reward_est || !d_q_cpe || !reward_partials || !cpe_partials || batch <= 0 || num_actions <= 0 ||
      num_metrics <= 0 || (num_metrics > 1 && !extra_metrics))
    return RG_EINVAL;
  if (loss_type != RG_LOSS_MSE && loss_type != RG_LOSS_HUBER) return RG_EINVAL;
  RG_LAUNCH(cpe_head_kernel, dim3(rg_dqn_head_partials(batch)), dim3(HEAD_THREADS), (hipStream_t)stream, reward_est,
            q_cpe, q_cpe_tgt_next, next_scores, next_mask, action, reward, extra_metrics, not_terminal, (float)gamma,
            gamma_exponent, (float)temperature, batch, num_actions, num_metrics, loss_type, d_reward_est, d_q_cpe,
            reward_partials, cpe_partials, propensities_out);
  return (int)hipGetLastError();
}

int rg_qr_head(const float* q, const float* qn_online, const float* qn_target, const float* action,
               const float* next_mask, const float* reward, const float* reward_boosts,
               const float* not_terminal, double gamma, const float* gamma_exponent,
               const float* quantiles, int batch, int num_actions, int num_atoms, int maxq, float* dq,
               float* loss_partials, float* all_q, rg_stream_t stream) {
  if (!q || !qn_target || !action || !next_mask || !reward || !not_terminal || !quantiles || !dq ||
      !loss_partials || batch <= 0 || num_actions <= 0 || num_atoms <= 0)
    return RG_EINVAL;
  if (num_atoms > QR_MAX_ATOMS || num_actions > QR_MAX_ACTIONS) return RG_EUNSUPPORTED;
  RG_LAUNCH(qr_head_kernel, dim3(batch), dim3(HEAD_THREADS), (hipStream_t)stream, q, qn_online, qn_target,
            action, next_mask, reward, reward_boosts, not_terminal, (float)gamma, gamma_exponent, quantiles,
            batch, num_actions, num_atoms, maxq, dq, loss_partials, all_q);
  return (int)hipGetLastError();
}

int rg_c51_head(const float* q, const float* qn_online, const float* qn_target, const float* action,
                const float* next_mask, const float* reward, const float* reward_boosts, const float* not_terminal,
                double gamma, const float* gamma_exponent, const float* support, double qmin, double qmax,
                int batch, int num_actions, int num_atoms, int maxq, float* dq, float* loss_partials, float* all_q,
                rg_stream_t stream) {
  if (!q || !qn_target || !action || !next_mask || !reward || !not_terminal || !support || !dq || !loss_partials ||
      batch <= 0 || num_actions <= 0 || num_atoms <= 1)
    return RG_EINVAL;
  if (num_atoms > C51_MAX_ATOMS || num_actions > C51_MAX_ACTIONS) return RG_EUNSUPPORTED;
  // c51_trainer.py:72: scale_support = (qmax - qmin) / (num_atoms - 1.0), a Python float the reference
  // divides an fp32 tensor by
  const float scale = (float)((qmax - qmin) / ((double)num_atoms - 1.0));
  RG_LAUNCH(c51_head_kernel, dim3(batch), dim3(HEAD_THREADS), (hipStream_t)stream, q, qn_online, qn_target, action,
            next_mask, reward, reward_boosts, not_terminal, (float)gamma, gamma_exponent, support, (float)qmin,
            (float)qmax, scale, batch, num_actions, num_atoms, maxq, dq, loss_partials, all_q);
  return (int)hipGetLastError();
}

int rg_bcq_filter(const float* imitator_logits, int batch, int num_actions, double drop_threshold, float* mask,
                  rg_stream_t stream) {
  if (!imitator_logits || !mask || batch < 0 || num_actions <= 0) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  RG_LAUNCH(bcq_filter_kernel, dim3((batch + HEAD_THREADS - 1) / HEAD_THREADS), dim3(HEAD_THREADS),
            (hipStream_t)stream, imitator_logits, batch, num_actions, (float)drop_threshold, mask);
  return (int)hipGetLastError();
}

int rg_reduce_sum(const float* in, int n, float scale, float* out, rg_stream_t stream) {
  if (!in || !out || n < 0) return RG_EINVAL;
  RG_LAUNCH(reduce_sum_kernel, dim3(1), dim3(HEAD_THREADS), (hipStream_t)stream, in, n, scale, out);
  return (int)hipGetLastError();
}

}  // extern "C"

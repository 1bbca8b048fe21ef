// qr_grouped.hip — QR-DQN's wide output layer as a GROUPED layer: the logits never reach HBM.
//
// Reference: reagent/training/qrdqn_trainer.py:108-160.  The Q-network ends in a [A * N, H] linear layer
// (A actions x N quantiles; BASELINE C3: 16 x 200 = 3200 outputs over H = 512) and the step touches its
// 65536 x 3200 output three times (3 x 839 MB of fp32 logits written, read back by the loss, 839 MB of gradient).
// What the loss actually consumes per transition is
//   * the MEAN over the N quantiles of every action of the next state (to pick a* = arg max, :125-135),
//   * the N quantiles of ONE action of the target network (a*, :137-141) and
//   * the N quantiles of ONE action of the online network (the logged action, :143-146),
// and d loss / d logits is zero outside the logged action's N columns.  So:
//   * mean_n(h . W[a, n] + b[a, n]) = h . mean_n W[a, n] + mean_n b[a, n]: the per-action means are ONE A-wide
//     linear layer (rg_wide_head_mean builds its weights) — an ordinary narrow output layer of the fused stack;
//   * the rows of the batch are sorted by the action whose quantiles are needed, every action's rows padded to
//     whole 128-row tiles ("grouped space", built on the device — no host round trip), and the fused trunk runs
//     in that row order (rg_mlp_desc.rowmap).  A tile then needs ONE action's [N, H] slice of the wide layer:
//     forward (rg_group_head_forward), input gradient (rg_group_head_dgrad) and weight gradient
//     (rg_group_head_wgrad) are 1/A of the dense work and touch [B, N] instead of [B, A * N].
// The quantile-Huber loss itself (rg_qr_compact_head) is the N x N pair loop of rg_qr_head on those compact rows.
#include "rg_mlp_frag.h"

namespace rg {

constexpr int GH_THREADS = 512, GH_NW = 8;

// ---- weights of the grouped layer -------------------------------------------------------------------------
// wf[g]: B fragments of W_g [Ng, K] (forward), wb[g]: B fragments of W_g^T [K, Ng] (input gradient)
__global__ void group_stage_kernel(const float* __restrict__ w, int G, int Ng, int K, bf16_t* __restrict__ wf,
                                   bf16_t* __restrict__ wb, long per_f, long per_b) {
  const long per = per_f > per_b ? per_f : per_b;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per * G) return;
  const int g = (int)(i / per);
  stage_weight_elem(w + (long)g * Ng * K, Ng, K, wf ? wf + g * per_f : nullptr, wb ? wb + g * per_b : nullptr, i % per);
}

// wbar[g][k] = mean_n w[(g * Ng + n) * K + k], bbar[g] = mean_n b[g * Ng + n]  (fp32; 8 row strides per column
// summed separately and combined in fixed order)
__global__ void wide_mean_kernel(const float* __restrict__ w, const float* __restrict__ b, int G, int Ng, int K,
                                 float* __restrict__ wbar, float* __restrict__ bbar) {
  __shared__ float red[8][33];
  const int g = blockIdx.y, c = threadIdx.x & 31, rg = threadIdx.x >> 5, k = blockIdx.x * 32 + c;
  float s = 0.f;
  if (k < K) {
    const float* p = w + (long)g * Ng * K + k;
    for (int n = rg; n < Ng; n += 8) s += p[(long)n * K];
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && k < K) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][c];
    wbar[(long)g * K + k] = t / (float)Ng;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float t = 0.f;
    if (b)
      for (int n = 0; n < Ng; ++n) t += b[g * Ng + n];
    bbar[g] = t / (float)Ng;
  }
}

// key[b] = arg max_a (q[b, a] - 1e9 (1 - mask[b, a]))  (maxq; qrdqn_trainer.py:210-214, first maximum wins) or the
// position of the 1 in the one-hot row mask[b, :] (SARSA: mask = next_action), A if the row is all zero
__global__ void select_action_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ mask, int batch,
                                     int A, int maxq, int* __restrict__ key) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const float* m = mask + (long)b * A;
  int best = A;
  if (maxq) {
    float bv = 0.f;
    for (int a = 0; a < A; ++a) {
      const float v = q[(long)b * ldq + a] + -1e9f * (1.f - m[a]);
      if (a == 0 || v > bv) {
        bv = v;
        best = a;
      }
    }
  } else {
    for (int a = A - 1; a >= 0; --a)
      if (m[a] != 0.f) best = a;
  }
  key[b] = best;
}

// ---- forward of the grouped layer -------------------------------------------------------------------------
// z[dst(r), n] = sum_k h[r, k] W_g[n, k] + b_g[n] for the rows r of one 128-row tile (group g = tile_key[tile]);
// h arrives in C-fragment order (the saved input of the stack's last layer), dst(r) = rowmap[r] (scatter back
// to batch order) or r (stay in grouped space).
struct GroupFwdArgs {
  const bf16_t* h_frag;
  const int* rowmap;
  const int* tile_key;
  const bf16_t* wf;
  const float* bias;
  long per_f;
  int Ng, K, scatter;
  float* z;
  long ldz;
};

template <int PITCH>
__global__ void RG_LAUNCH_BOUNDS(GH_THREADS, 1) group_head_fwd_kernel(GroupFwdArgs a) {
  RG_DYN_LDS(smem);
  bf16_t* act = (bf16_t*)smem;
  const int g = a.tile_key[blockIdx.x];
  if (g < 0) return;  // an empty tail tile of the grouped space
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  constexpr int pitch = PITCH;
  const int NT = a.K / 32, KC = a.K / 16;
  // fragment order -> row-major LDS tile (a lane holds 8 rows of one column)
  for (int f = wave; f < 4 * NT * 2; f += GH_NW) {
    const int h = f & 1, nt = (f >> 1) % NT, mbl = (f >> 1) / NT;
    const u16x8 v = *(const u16x8*)(a.h_frag + frag_offset((long)blockIdx.x * 4 + mbl, nt, NT, h, lane));
#pragma unroll
    for (int e = 0; e < 8; ++e) act[(mbl * 32 + frag_row(h, e, lg)) * pitch + nt * 32 + lr] = v[e];
  }
  __syncthreads();
  const int NTo = (a.Ng + 31) / 32;
  const bf16_t* wf = a.wf + (long)g * a.per_f;
  const float* bias = a.bias ? a.bias + (long)g * a.Ng : nullptr;
  for (int t = wave; t < 4 * NTo; t += GH_NW) {
    const int tm = t & 3, nt = t >> 2;
    const f32x16 acc = tile_kloop(act, pitch, KC, wf, tm, nt, lane);
    const int col = nt * 32 + lr;
    if (col < a.Ng) {
      const float b = bias ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = blockIdx.x * 128 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        const int src = a.rowmap[row];
        if (src >= 0) a.z[(long)(a.scatter ? src : row) * a.ldz + col] = acc[r] + b;
      }
    }
  }
}

// ---- quantile-Huber loss on compact rows (qrdqn_trainer.py:137-160, huber :217-218) ---------------------------
// One workgroup per row r of the grouped space (b = rowmap[r]; padding rows write zeros):
//   T_i = reward[b] (+ boost of the logged action) + gamma^e[b] * not_terminal[b] * zt[b, i]
//   C_j = z[r, j]
//   loss = mean over (i, b, j) of huber(T_i - C_j) * |tau_j - 1{T_i - C_j < 0}|;  dz[r, j] = d loss / d C_j
constexpr int QC_MAX_ATOMS = 1024;
struct CompactHeadArgs {
  const float* z;
  const float* zt;
  long ldz, ldzt;
  const int* rowmap;
  const int* tile_key;
  const float* reward;
  const float* reward_boosts;
  const float* not_terminal;
  const float* gamma_exponent;
  const float* quantiles;
  float gamma;
  int batch, N;
  float* dz;
  long lddz;
  float* loss_partials;
};

__device__ __forceinline__ float block_sum_256_(float v, float* scratch) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += shfl_xor(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

__global__ void qr_compact_head_kernel(CompactHeadArgs a) {
  __shared__ __attribute__((aligned(16))) float T[QC_MAX_ATOMS];
  __shared__ float C[QC_MAX_ATOMS];
  __shared__ float scratch[4];
  const int r = blockIdx.x, tid = threadIdx.x, N = a.N;
  const int b = a.rowmap[r];
  float* dz = a.dz + (long)r * a.lddz;
  if (b < 0) {
    for (int j = tid; j < a.lddz; j += 256) dz[j] = 0.f;
    if (tid == 0) a.loss_partials[r] = 0.f;
    return;
  }
  const int g = a.tile_key[r >> 7];
  const float rew = a.reward[b] + (a.reward_boosts ? a.reward_boosts[g] : 0.f);
  const float disc = a.gamma_exponent ? powf(a.gamma, a.gamma_exponent[b]) : a.gamma;
  const float dn = disc * a.not_terminal[b];
  for (int j = tid; j < N; j += 256) {
    T[j] = rew + dn * a.zt[(long)b * a.ldzt + j];
    C[j] = a.z[(long)r * a.ldz + j];
  }
  __syncthreads();
  const float inv = 1.f / ((float)N * (float)a.batch * (float)N);
  float loss = 0.f;
  // per pair, with td = T_i - C_j and c = clamp(td, -1, 1):
  //   huber(td) = c * (td - c / 2)   (= td^2 / 2 inside [-1, 1], |td| - 1/2 outside),   huber'(td) = c,
  //   weight |tau_j - 1{td < 0}| = td < 0 ? 1 - tau_j : tau_j
  // eight VALU operations per pair; the N x N pairs of the batch are the whole cost of this kernel
  const int N4 = N & ~3;
  for (int j = tid; j < a.lddz; j += 256) {
    float gsum = 0.f;
    if (j < N) {
      const float cj = C[j], tau = a.quantiles[j], omt = 1.f - tau;
      float l = 0.f;
      int i = 0;
      for (; i < N4; i += 4) {
        const f32x4 t4 = *(const f32x4*)&T[i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float td = t4[u] - cj;
          const float c = fminf(fmaxf(td, -1.f), 1.f);
          const float w = td < 0.f ? omt : tau;
          l = fmaf(c * fmaf(-0.5f, c, td), w, l);
          gsum = fmaf(c, w, gsum);
        }
      }
      for (; i < N; ++i) {
        const float td = T[i] - cj;
        const float c = fminf(fmaxf(td, -1.f), 1.f);
        const float w = td < 0.f ? omt : tau;
        l = fmaf(c * fmaf(-0.5f, c, td), w, l);
        gsum = fmaf(c, w, gsum);
      }
      loss += l;
    }
    dz[j] = -gsum * inv;
  }
  const float s = block_sum_256_(loss, scratch);
  if (tid == 0) a.loss_partials[r] = s * inv;
}

// tile_sums[t] = sum of loss_partials[128 t .. 128 t + 127] (row order), so that the final deterministic single-
// workgroup sum runs over n_tiles values instead of one per row
__global__ void tile_sum_kernel(const float* __restrict__ v, float* __restrict__ out) {
  __shared__ float red[2];
  float x = v[(long)blockIdx.x * 128 + threadIdx.x];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) x += shfl_xor(x, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = red[0] + red[1];
}

// ---- input gradient of the grouped layer --------------------------------------------------------------------
// dh[r, k] = sum_n dz[r, n] W_g[n, k]; written as d loss / d (pre-activation of the stack's last hidden layer):
// dz3[r, k] = dh[r, k] * act'(h[r, k]) (ReLU family: h > 0 read from the saved fragments).  Also emits dz in
// C-fragment order (the weight-gradient operand) and the per-tile column sums of dz (bias gradient partials).
struct GroupDgradArgs {
  const float* dz;
  long lddz;
  const int* tile_key;
  const bf16_t* wb;
  long per_b;
  const bf16_t* h_frag;
  int Ng, K, leaky;
  float* dz3;
  long lddz3;
  bf16_t* dzw_frag;
  float* db_part;  // [tiles][NgP]
};

template <int PITCH>
__global__ void RG_LAUNCH_BOUNDS(GH_THREADS, 1) group_head_dgrad_kernel(GroupDgradArgs a) {
  RG_DYN_LDS(smem);
  bf16_t* act = (bf16_t*)smem;
  const int g = a.tile_key[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  constexpr int pitch = PITCH;
  const int NgP = (a.Ng + 31) / 32 * 32, NTz = NgP / 32, NT = a.K / 32;
  const int row_base = blockIdx.x * 128;
  if (g < 0) {  // empty tail tile: the trunk backward and the weight gradients still read these rows
    for (int i = tid; i < 128 * a.K; i += GH_THREADS) a.dz3[(long)(row_base + i / a.K) * a.lddz3 + i % a.K] = 0.f;
    for (int i = tid; i < 128 * NgP / 8; i += GH_THREADS)
      *(u16x8*)(a.dzw_frag + ((long)blockIdx.x * 4 * NTz * 2 * 64) * 8 + (long)i * 8) = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    if (a.db_part && tid < NgP) a.db_part[(long)blockIdx.x * NgP + tid] = 0.f;
    return;
  }
  load_tile_to_lds<float, GH_THREADS>(act, pitch, a.dz, a.lddz, row_base, row_base + 128, a.Ng, NgP, tid);
  __syncthreads();
  emit_frags_from_lds(act, pitch, NTz, a.dzw_frag, blockIdx.x * 4, wave, GH_NW, lane);
  if (a.db_part && tid < NgP) {
    float s = 0.f;
    for (int r = 0; r < 128; ++r) s += bf16_to_f32(act[r * pitch + tid]);
    a.db_part[(long)blockIdx.x * NgP + tid] = s;
  }
  const int KC = (a.Ng + 15) / 16;
  const bf16_t* wb = a.wb + (long)g * a.per_b;
  for (int t = wave; t < 4 * NT; t += GH_NW) {
    const int tm = t & 3, nt = t >> 2;
    const f32x16 acc = tile_kloop(act, pitch, KC, wb, tm, nt, lane);
    const int col = nt * 32 + lr;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const u16x8 hf = *(const u16x8*)(a.h_frag + frag_offset((long)blockIdx.x * 4 + tm, nt, NT, h, lane));
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int row = row_base + tm * 32 + frag_row(h, e, lg);
        const float gr = bf16_to_f32(hf[e]) > 0.f ? 1.f : (a.leaky ? 0.01f : 0.f);
        a.dz3[(long)row * a.lddz3 + col] = acc[8 * h + e] * gr;
      }
    }
  }
}

// db[g * Ng + n] = sum over the tiles of group g of db_part[tile][n]
__global__ void group_bias_reduce_kernel(const float* __restrict__ db_part, const int* __restrict__ tile_begin, int Ng,
                                         int NgP, float* __restrict__ db) {
  const int g = blockIdx.x;
  for (int n = threadIdx.x; n < Ng; n += blockDim.x) {
    float s = 0.f;
    for (int t = tile_begin[g]; t < tile_begin[g + 1]; ++t) s += db_part[(long)t * NgP + n];
    db[g * Ng + n] = s;
  }
}

static int group_pitch(int K) { return K <= 256 ? 264 : 520; }

}  // namespace rg

using namespace rg;

extern "C" {

size_t rg_group_wfrag_elems(int group_rows, int in_features, int transposed) {
  return transposed ? wfrag_elems(in_features, group_rows) : wfrag_elems(group_rows, in_features);
}

int rg_group_weights_stage(const float* w, int n_groups, int group_rows, int in_features, void* wfrag_fwd,
                           void* wfrag_bwd, rg_stream_t stream) {
  if (!w || n_groups <= 0 || group_rows <= 0 || in_features <= 0 || (!wfrag_fwd && !wfrag_bwd)) return RG_EINVAL;
  const long per_f = (long)wfrag_elems(group_rows, in_features), per_b = (long)wfrag_elems(in_features, group_rows);
  const long per = per_f > per_b ? per_f : per_b;
  const long total = per * n_groups;
  RG_LAUNCH(group_stage_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (hipStream_t)stream, w, n_groups,
            group_rows, in_features, (bf16_t*)wfrag_fwd, (bf16_t*)wfrag_bwd, per_f, per_b);
  return (int)hipGetLastError();
}

int rg_wide_head_mean(const float* w, const float* b, int n_groups, int group_rows, int in_features, float* wbar,
                      float* bbar, rg_stream_t stream) {
  if (!w || !wbar || !bbar || n_groups <= 0 || group_rows <= 0 || in_features <= 0) return RG_EINVAL;
  RG_LAUNCH(wide_mean_kernel, dim3((in_features + 31) / 32, n_groups), dim3(256), (hipStream_t)stream, w, b, n_groups,
            group_rows, in_features, wbar, bbar);
  return (int)hipGetLastError();
}

int rg_qr_select_action(const float* q, int64_t ldq, const float* mask, int batch, int num_actions, int maxq, int32_t* key,
                        rg_stream_t stream) {
  if (!mask || !key || batch <= 0 || num_actions <= 0 || (maxq && !q)) return RG_EINVAL;
  RG_LAUNCH(select_action_kernel, dim3((batch + 255) / 256), dim3(256), (hipStream_t)stream, q, (long)ldq, mask, batch,
            num_actions, maxq, key);
  return (int)hipGetLastError();
}

int rg_group_head_forward(const void* h_frag, const int32_t* rowmap, const int32_t* tile_key, int n_tiles,
                          const void* wfrag_fwd, const float* bias, int group_rows, int in_features, int scatter, float* z,
                          int64_t ldz, rg_stream_t stream) {
  if (!h_frag || !rowmap || !tile_key || !wfrag_fwd || !z || n_tiles <= 0 || group_rows <= 0) return RG_EINVAL;
  if (in_features != 256 && in_features != 512) return RG_EUNSUPPORTED;
  GroupFwdArgs a;
  a.h_frag = (const bf16_t*)h_frag; a.rowmap = rowmap; a.tile_key = tile_key; a.wf = (const bf16_t*)wfrag_fwd; a.bias = bias;
  a.per_f = (long)wfrag_elems(group_rows, in_features); a.Ng = group_rows; a.K = in_features; a.scatter = scatter; a.z = z;
  a.ldz = ldz;
  const int pitch = group_pitch(in_features);
  const size_t lds = (size_t)128 * pitch * sizeof(bf16_t);
  if (pitch == 264) {
    RG_ALLOW_LDS(group_head_fwd_kernel<264>, lds);
    RG_LAUNCH_DYN(group_head_fwd_kernel<264>, dim3(n_tiles), dim3(GH_THREADS), lds, (hipStream_t)stream, a);
  } else {
    RG_ALLOW_LDS(group_head_fwd_kernel<520>, lds);
    RG_LAUNCH_DYN(group_head_fwd_kernel<520>, dim3(n_tiles), dim3(GH_THREADS), lds, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

int rg_qr_compact_head(const float* z, int64_t ldz, const float* zt, int64_t ldzt, const int32_t* rowmap,
                       const int32_t* tile_key, int padded_rows, const float* reward, const float* reward_boosts,
                       const float* not_terminal, double gamma, const float* gamma_exponent, const float* quantiles,
                       int batch, int num_atoms, float* dz, int64_t lddz, float* loss_partials, float* tile_losses,
                       rg_stream_t stream) {
  if (!z || !zt || !rowmap || !tile_key || !reward || !not_terminal || !quantiles || !dz || !loss_partials ||
      padded_rows <= 0 || (padded_rows % 128) != 0 || batch <= 0 || num_atoms <= 0)
    return RG_EINVAL;
  if (num_atoms > QC_MAX_ATOMS || lddz < num_atoms) return RG_EUNSUPPORTED;
  CompactHeadArgs a;
  a.z = z; a.zt = zt; a.ldz = ldz; a.ldzt = ldzt; a.rowmap = rowmap; a.tile_key = tile_key; a.reward = reward;
  a.reward_boosts = reward_boosts; a.not_terminal = not_terminal; a.gamma_exponent = gamma_exponent; a.quantiles = quantiles;
  a.gamma = (float)gamma; a.batch = batch; a.N = num_atoms; a.dz = dz; a.lddz = lddz; a.loss_partials = loss_partials;
  RG_LAUNCH(qr_compact_head_kernel, dim3(padded_rows), dim3(256), (hipStream_t)stream, a);
  int rc = (int)hipGetLastError();
  if (rc || !tile_losses) return rc;
  RG_LAUNCH(tile_sum_kernel, dim3(padded_rows / 128), dim3(128), (hipStream_t)stream, (const float*)loss_partials, tile_losses);
  return (int)hipGetLastError();
}

int rg_group_head_dgrad(const float* dz, int64_t lddz, const int32_t* tile_key, const int32_t* tile_begin, int n_tiles,
                        int n_groups, const void* wfrag_bwd, const void* h_frag, int group_rows, int in_features,
                        int leaky_relu, float* dz3, int64_t lddz3, void* dzw_frag, float* db_partials, float* db,
                        rg_stream_t stream) {
  if (!dz || !tile_key || !wfrag_bwd || !h_frag || !dz3 || !dzw_frag || n_tiles <= 0 || group_rows <= 0) return RG_EINVAL;
  if (in_features != 256 && in_features != 512) return RG_EUNSUPPORTED;
  if (db && (!db_partials || !tile_begin)) return RG_EINVAL;
  const int NgP = (group_rows + 31) / 32 * 32;
  if (NgP + 8 > group_pitch(in_features)) return RG_EUNSUPPORTED;
  GroupDgradArgs a;
  a.dz = dz; a.lddz = lddz; a.tile_key = tile_key; a.wb = (const bf16_t*)wfrag_bwd;
  a.per_b = (long)wfrag_elems(in_features, group_rows); a.h_frag = (const bf16_t*)h_frag; a.Ng = group_rows; a.K = in_features;
  a.leaky = leaky_relu; a.dz3 = dz3; a.lddz3 = lddz3; a.dzw_frag = (bf16_t*)dzw_frag; a.db_part = db ? db_partials : nullptr;
  const int pitch = group_pitch(in_features);
  const size_t lds = (size_t)128 * pitch * sizeof(bf16_t);
  if (pitch == 264) {
    RG_ALLOW_LDS(group_head_dgrad_kernel<264>, lds);
    RG_LAUNCH_DYN(group_head_dgrad_kernel<264>, dim3(n_tiles), dim3(GH_THREADS), lds, (hipStream_t)stream, a);
  } else {
    RG_ALLOW_LDS(group_head_dgrad_kernel<520>, lds);
    RG_LAUNCH_DYN(group_head_dgrad_kernel<520>, dim3(n_tiles), dim3(GH_THREADS), lds, (hipStream_t)stream, a);
  }
  int rc = (int)hipGetLastError();
  if (rc || !db) return rc;
  RG_LAUNCH(group_bias_reduce_kernel, dim3(n_groups), dim3(256), (hipStream_t)stream, (const float*)db_partials, tile_begin,
            group_rows, NgP, db);
  return (int)hipGetLastError();
}

}  // extern "C"

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
//   * the rows of the batch are sorted by the action whose quantiles are needed ("grouped space", built on the
//     device — no host round trip), and the fused trunk runs in that row order (rg_mlp_desc.rowmap).  A run of rows
//     then needs ONE action's [N, H] slice of the wide layer: it becomes the fused stack's OUTPUT layer with
//     per-group weights (rg_mlp_desc.tile_key / row_begin: forward and input gradient inside rg_mlp_forward_fused /
//     rg_mlp_backward_fused), its weight gradient is rg_group_head_wgrad — 1/A of the dense work, [B, N] instead of
//     [B, A * N] bytes.  Rounds 2-3 padded every action's rows to whole 128-row tiles (B / 128 + ~A / 2 tiles: 520 for
//     C3 — two rounds of the 256 CUs plus a sliver, i.e. THREE rounds per launch); round 4 packs the groups densely
//     (B / 128 tiles exactly) and a tile that holds rows of several groups runs the layer once per group (next_segment).
// The quantile-Huber loss itself (rg_qr_compact_head) runs on those compact rows — in O(N log N) per row, see there.
#include "rg_mlp_frag.h"

namespace rg {

// ---- weights of the grouped layer -------------------------------------------------------------------------
// wf[g]: B fragments of W_g [Ng, K] (forward), wb[g]: B fragments of W_g^T [K, Ng] (input gradient)
// x3 (split-bf16): a group's set is [hi plane | lo plane], lo = bf16(w - hi) — stage_weight_elem's layout per group
__global__ void group_stage_kernel(const float* __restrict__ w, int G, int Ng, int K, bf16_t* __restrict__ wf,
                                   bf16_t* __restrict__ wb, long per_f, long per_b, int x3) {
  const long per = per_f > per_b ? per_f : per_b;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per * G) return;
  const int g = (int)(i / per);
  const int planes = x3 ? 2 : 1;
  stage_weight_elem(w + (long)g * Ng * K, Ng, K, wf ? wf + g * per_f * planes : nullptr, wb ? wb + g * per_b * planes : nullptr,
                    i % per, x3);
}

// wbar[g][k] = mean_n w[(g * Ng + n) * K + k], bbar[g] = mean_n b[g * Ng + n]  (fp32; 8 row strides per column
// summed separately and combined in fixed order)
// wfrag (nullable): the forward B fragments of the [G, K] mean layer, written with the means (what a separate
// rg_stage_weights_frag of wbar would write to the slots of rows < G; the padding rows were zeroed by the first staging)
// x3: the fragments' lo plane (wfrag_elems(G, K) elements behind the hi plane) gets bf16(mean - hi)
__global__ void wide_mean_kernel(const float* __restrict__ w, const float* __restrict__ b, int G, int Ng, int K,
                                 float* __restrict__ wbar, float* __restrict__ bbar, bf16_t* __restrict__ wfrag, int x3) {
  __shared__ float red[8][33];
  const int g = blockIdx.y, c = threadIdx.x & 31, rg = threadIdx.x >> 5, k = blockIdx.x * 32 + c;
  float s = 0.f;
  if (k < K) {
    // four rows in flight per thread (one dependent load after the other made this launch 19 us for 6.6 MB)
    const float* p = w + (long)g * Ng * K + k;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int n = rg;
    for (; n + 24 < Ng; n += 32) {
      s0 += p[(long)n * K];
      s1 += p[(long)(n + 8) * K];
      s2 += p[(long)(n + 16) * K];
      s3 += p[(long)(n + 24) * K];
    }
    for (; n < Ng; n += 8) s0 += p[(long)n * K];
    s = (s0 + s1) + (s2 + s3);
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && k < K) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) t += red[i][c];
    const float mean = t / (float)Ng;
    wbar[(long)g * K + k] = mean;
    if (wfrag) {
      const int KCf = (K + 15) / 16;
      const long j = ((((long)(g >> 5) * KCf + (k >> 4)) * 64) + ((g & 31) + 32 * ((k & 15) >> 3))) * 8 + (k & 7);
      const bf16_t hi = f32_to_bf16(mean);
      wfrag[j] = hi;
      if (x3) wfrag[(long)((G + 31) / 32) * KCf * 512 + j] = f32_to_bf16(mean - bf16_to_f32(hi));
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {  // the group's bias mean: one wave, lane-strided partial sums
    float t = 0.f;
    if (b)
      for (int n = threadIdx.x; n < Ng; n += 64) t += b[g * Ng + n];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) t += shfl_xor(t, off);
    if (threadIdx.x == 0) bbar[g] = t / (float)Ng;
  }
}

// key[b] = arg max_a (q[b, a] - 1e9 (1 - mask[b, a]))  (maxq; qrdqn_trainer.py:210-214, first maximum wins) or the
// position of the 1 in the one-hot row mask[b, :] (SARSA: mask = next_action), A if the row is all zero
// (the row's values are requested eight at a time: one load per loop turn was a chain of A dependent L2 / HBM round trips per
// thread — 40-60 us for the counting launch of a C3 step when it shares the chip with a forward)
__device__ __forceinline__ int select_action_row(const float* __restrict__ q, long ldq, const float* __restrict__ mask,
                                                 int b, int A, int maxq) {
  const float* m = mask + (long)b * A;
  const float* qr = q ? q + (long)b * ldq : nullptr;  // (only read for maxq)
  int best = A;
  if (maxq) {
    float bv = 0.f;
    int a = 0;
    for (; a + 8 <= A; a += 8) {
      float mv[8], qv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        mv[i] = m[a + i];
        qv[i] = qr[a + i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float v = qv[i] + -1e9f * (1.f - mv[i]);
        if (a + i == 0 || v > bv) {
          bv = v;
          best = a + i;
        }
      }
    }
    for (; a < A; ++a) {
      const float v = qr[a] + -1e9f * (1.f - m[a]);
      if (a == 0 || v > bv) {
        bv = v;
        best = a;
      }
    }
  } else {  // the FIRST non-zero entry
    int a = 0;
    for (; a + 8 <= A && best == A; a += 8) {
      float mv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) mv[i] = m[a + i];
#pragma unroll
      for (int i = 7; i >= 0; --i)
        if (mv[i] != 0.f) best = a + i;
    }
    for (; a < A && best == A; ++a)
      if (m[a] != 0.f) best = a;
  }
  return best;
}

__global__ void select_action_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ mask, int batch,
                                     int A, int maxq, int* __restrict__ key) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  key[b] = select_action_row(q, ldq, mask, b, A, maxq);
}

// ---- the grouped row space: a stable counting sort by key ------------------------------------------------------
// rowmap [128 * n_tiles]: batch row of every grouped row (-1: padding); row_begin [G + 1]: group g owns the grouped rows
// [row_begin[g], row_begin[g + 1]) — its valid rows first, then (dense == 0 only) padding to the next multiple of 128;
// tile_key [n_tiles]: the first group with rows in each tile (-1: none).  Keys >= G mean "no group": dropped.  Rows keep
// their batch order inside a group (rank = rows of the same key in earlier 256-row blocks + earlier rows of the own
// block), so the layout — and every sum taken over it — is deterministic.  Two launches, no host round trip.
constexpr int GR_BLOCK = 256, GR_MAX_KEYS = 130;

// sel_mask != null: the key is the row's selected action (rg_qr_select_action's rule, evaluated here and written to `key`)
__global__ void group_count_kernel(int* __restrict__ key, int batch, int G, int* __restrict__ block_hist,
                                   int* __restrict__ rowmap, int padded_rows, const float* __restrict__ sel_q, long sel_ldq,
                                   const float* __restrict__ sel_mask, int sel_maxq) {
  __shared__ int hist[GR_MAX_KEYS];
  const int tid = threadIdx.x, b = blockIdx.x * GR_BLOCK + tid;
  for (int i = tid; i <= G; i += GR_BLOCK) hist[i] = 0;
  __syncthreads();
  if (b < batch) {
    int k;
    if (sel_mask) {
      k = select_action_row(sel_q, sel_ldq, sel_mask, b, G, sel_maxq);
      key[b] = k;
    } else {
      k = key[b];
    }
    atomicAdd(&hist[k < G ? (k < 0 ? G : k) : G], 1);
  }
  for (int j = b; j < padded_rows; j += gridDim.x * GR_BLOCK) rowmap[j] = -1;
  __syncthreads();
  for (int i = tid; i <= G; i += GR_BLOCK) block_hist[blockIdx.x * (G + 1) + i] = hist[i];
}

// Every block derives what it needs from the raw per-block histogram itself — the rows of its keys in earlier blocks
// (its base ranks) and the groups' totals (row_begin) — 256 x (G + 1) integers, L2-resident: the single-workgroup scan
// launch between count and scatter is gone (it was the launch that waited longest for a CU, up to 90 us, while the other
// stream's forward held them all).  Block 0 also publishes row_begin and tile_key.  Integer sums: order-independent.
// Round 4: this launch took 70-110 us of a C3 step's critical path (kernel trace, `profiles/scripts/gpu_timeline.sh c3`) — a
// chain of G dependent L2 round trips per thread (one histogram row per thread, a load and two LDS atomics per group) and a
// rank loop of up to 255 dependent LDS reads.  Now the histogram is summed with thread = (group, slice of the blocks), four
// independent coalesced loads in flight and ONE pair of LDS atomics per thread, and a row's rank inside its block comes from
// G wave ballots (rows of the same key in lower lanes) plus the lower waves' counts.
__global__ void group_scatter_kernel(const int* __restrict__ key, int batch, int G, const int* __restrict__ block_hist,
                                     int n_blocks, int n_tiles, int dense, int* __restrict__ row_begin,
                                     int* __restrict__ tile_key, int* __restrict__ rowmap) {
  __shared__ int total[GR_MAX_KEYS], base[GR_MAX_KEYS], tb[GR_MAX_KEYS];
  __shared__ int wcount[GR_BLOCK / 64][GR_MAX_KEYS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, me = blockIdx.x, b = me * GR_BLOCK + tid;
  for (int g = tid; g <= G; g += GR_BLOCK) total[g] = base[g] = 0;
  int k = b < batch ? key[b] : G;
  if (k < 0 || k > G) k = G;
  __syncthreads();
  {  // totals and base ranks: thread = (group g, slice s) sums the blocks s, s + S, ... (G < GR_BLOCK: GR_MAX_KEYS)
    const int S = GR_BLOCK / G, g = tid % G, s = tid / G;
    if (s < S) {
      int t0 = 0, t1 = 0, t2 = 0, t3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
      int blk = s;
      for (; blk + 3 * S < n_blocks; blk += 4 * S) {
        const int c0 = block_hist[(long)blk * (G + 1) + g], c1 = block_hist[(long)(blk + S) * (G + 1) + g];
        const int c2 = block_hist[(long)(blk + 2 * S) * (G + 1) + g], c3 = block_hist[(long)(blk + 3 * S) * (G + 1) + g];
        t0 += c0; t1 += c1; t2 += c2; t3 += c3;
        b0 += blk < me ? c0 : 0; b1 += blk + S < me ? c1 : 0; b2 += blk + 2 * S < me ? c2 : 0; b3 += blk + 3 * S < me ? c3 : 0;
      }
      for (; blk < n_blocks; blk += S) {
        const int c = block_hist[(long)blk * (G + 1) + g];
        t0 += c;
        b0 += blk < me ? c : 0;
      }
      const int t = (t0 + t1) + (t2 + t3), bs = (b0 + b1) + (b2 + b3);
      if (t) atomicAdd(&total[g], t);
      if (bs) atomicAdd(&base[g], bs);
    }
  }
  // rank inside the block: rows of the same key in lower lanes of the wave (ballots), then the lower waves' counts
  int in_wave = 0;
  for (int g = 0; g < G; ++g) {
    const unsigned long long m = wave_ballot(k == g);
    if (k == g) in_wave = __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wv][g] = __builtin_popcountll(m);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int g = 0; g < G; ++g) {
      tb[g] = run;
      run += dense ? total[g] : (total[g] + 127) / 128 * 128;
    }
    tb[G] = run;
  }
  __syncthreads();
  if (me == 0) {
    if (tid <= G) row_begin[tid] = tb[tid];
    for (int t = tid; t < n_tiles; t += GR_BLOCK) {
      int g = -1;
      for (int q = G - 1; q >= 0; --q)  // the first group whose (non-empty) range reaches into the tile
        if (tb[q + 1] > tb[q] && tb[q + 1] > t * 128 && tb[q] < t * 128 + 128) g = q;
      tile_key[t] = g;
    }
  }
  if (k >= G) return;
  int rank = base[k] + in_wave;
  for (int w = 0; w < wv; ++w) rank += wcount[w][k];
  rowmap[tb[k] + rank] = b;
}

// ---- quantile-Huber loss on compact rows (qrdqn_trainer.py:137-160, huber :217-218) ---------------------------
// One workgroup per row r of the grouped space (b = rowmap[r]; padding rows write zeros):
//   T_i = reward[b] (+ boost of the logged action) + gamma^e[b] * not_terminal[b] * zt[b, i]
//   C_j = z[r, j]
//   loss = mean over (i, b, j) of huber(T_i - C_j) * |tau_j - 1{T_i - C_j < 0}|;  dz[r, j] = d loss / d C_j
struct CompactHeadArgs {
  const float* z;
  const float* zt;
  long ldz, ldzt;
  const int* rowmap;
  const int* row_key;  // [batch]: the group (logged action) of a batch row — only read for the reward boost
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

// One workgroup of 256 threads per row, and NOT the N x N pair loop: for a fixed C_j the pairs fall into four
// ranges of T_i - C_j — (-inf, -1], (-1, 0), [0, 1), [1, inf) — on each of which huber(td) * weight is a polynomial
// in T_i of degree <= 2 with coefficients that depend on C_j and tau_j only:
//     td <= -1 : (1 - tau) (C - 1/2 - T)        -1 < td < 0 : (1 - tau) (T - C)^2 / 2
//     0 <= td < 1 : tau (T - C)^2 / 2            td >= 1     : tau (T - C - 1/2)
// So the row's T is sorted once (bitonic, in LDS), prefix sums of T and T^2 are taken over the sorted order (fp64:
// the range sums are differences of prefixes), and every j needs three binary searches and a handful of operations:
// O(N log N) per row instead of O(N^2) — 2.6e9 pairs per 65536-row batch at N = 200 made the pair loop the largest
// kernel of the C3 step (0.46 ms, VALU-bound).  Values agree with the pair loop to fp32 rounding (the sum is taken
// in a different order; at td = +-1 and td = 0 the neighbouring polynomials and their derivatives coincide, so the
// side a boundary element is counted on does not matter).
// One WAVE per row (four rows per 256-thread workgroup) and no workgroup barrier anywhere: a lane holds four of the
// row's (padded) 256 targets, the bitonic network exchanges across lanes by shuffles and inside a lane in registers,
// the sorted targets and their prefix sums go to the wave's own slice of LDS (the binary searches need random access).
constexpr int QC_MAX_N = 256, QC_WAVES = 4;

// (94 registers = five waves per SIMD.  Forcing six / seven with amdgpu_waves_per_eu — 80 registers + 8 spilled, 72 + 18 —
// measured 134-135 / 148-150 us against 130 on one box, `profiles/scripts/gpu_batch19.sh`: not kept.)
__global__ void RG_LAUNCH_BOUNDS(QC_WAVES * 64, 1) qr_compact_head_kernel(CompactHeadArgs a) {
  // Ts[1 + 256 w + e] = the e-th smallest target of wave w's row ([0] unused).  A bisection keeps its count c as a BYTE offset
  // 1024 w + 4 c, so that the candidate count c + step is `offset | 4 * step` (disjoint bits) and at once the offset of
  // T[c + step - 1] in the array: v_or, ds_read, v_cmp, v_cndmask per step (round 4; the shift-add that formed the address from
  // a count was a fourth VALU operation in each of the 96 steps of a row's twelve bisections — measured: no change in the
  // kernel's time, which is bound by a row's latency chain at five waves per SIMD as much as by issue)
  __shared__ float Ts[1 + QC_WAVES * QC_MAX_N];
  __shared__ double P1[QC_WAVES][QC_MAX_N + 1], P2[QC_WAVES][QC_MAX_N + 1];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, N = a.N;
  const int r = blockIdx.x * QC_WAVES + wv;
  // the row's own quantiles C_j, j = lane + 64 q: they depend on r alone, so they are requested FIRST and travel under the
  // rowmap -> reward / targets chain and the sort (round 4: requested where the bisections start, their HBM round trip was
  // exposed there — the kernel runs five waves per SIMD and is bound by a row's ~15k cycles of latency as much as by issue)
  constexpr int QJ = QC_MAX_N / 64;
  float cq[QJ];
#pragma unroll
  for (int q = 0; q < QJ; ++q) {
    const int j = lane + 64 * q;
    cq[q] = j < N ? a.z[(long)r * a.ldz + j] : 0.f;
  }
  const int b = a.rowmap[r];
  float* dz = a.dz + (long)r * a.lddz;
  if (b < 0) {  // wave-uniform
    for (int j = lane; j < a.lddz; j += 64) dz[j] = 0.f;
    if (lane == 0) a.loss_partials[r] = 0.f;
    return;
  }
  const float rew = a.reward[b] + (a.reward_boosts ? a.reward_boosts[a.row_key[b]] : 0.f);
  const float disc = a.gamma_exponent ? powf(a.gamma, a.gamma_exponent[b]) : a.gamma;
  const float dn = disc * a.not_terminal[b];
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = lane * 4 + k;
    v[k] = e < N ? rew + dn * a.zt[(long)b * a.ldzt + e] : __builtin_inff();
  }
  // bitonic sort, ascending, element index e = 4 * lane + k.  Fully unrolled (36 stages); a compare-exchange is ONE v_med3
  // with the direction as its third operand (-inf keeps the smaller, +inf the larger: the inputs are NaN-free), and the
  // partner comes by DPP / ds_swizzle for lane distances below 32 (round 4: min + max + select and a ds_bpermute per element
  // and stage were the largest share of the kernel's ~2000 VALU operations per row).
  const float NEG = -__builtin_inff(), POS = __builtin_inff();
  static_for<1, 9>([&](auto kk_c) __attribute__((always_inline)) {
    constexpr int K = 1 << decltype(kk_c)::value;  // 2 .. 256
    static_for<0, decltype(kk_c)::value>([&](auto jj_c) __attribute__((always_inline)) {
      constexpr int j = K >> (1 + decltype(jj_c)::value);  // K/2 .. 1
      if constexpr (j >= 4) {
        constexpr int lm = j >> 2;
        // ascending block (e & K == 0) and lower partner keep the minimum; K >= 8 here, so the direction is the lane's alone
        const bool keep_min = ((lane & lm) == 0) == (((lane * 4) & K) == 0);
        const float dir = keep_min ? NEG : POS;
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = med3(v[k], shfl_xor_c<lm>(v[k]), dir);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int q = k ^ j;
          if (q > k) {
            const bool up = ((lane * 4 + k) & K) == 0;
            const float a_ = v[k], b_ = v[q];
            v[k] = med3(a_, b_, up ? NEG : POS);
            v[q] = med3(a_, b_, up ? POS : NEG);
          }
        }
      }
    });
  });
  // exclusive prefix sums of T and T^2 in sorted order (fp64; the +inf padding counts as 0)
  double t1[4], t2[4], s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double t = (lane * 4 + k) < N ? (double)v[k] : 0.0;
    t1[k] = s1;  // exclusive within the lane
    t2[k] = s2;
    s1 += t;
    s2 += t * t;
  }
  // inclusive scan of the lane totals across the wave by DPP moves (round 4; before: six steps of four ds_bpermute each, and
  // through LDS with a wave hand-off per step before that)
  const double i1 = wave_inclusive_sum_f64(s1), i2 = wave_inclusive_sum_f64(s2);
  const double base1 = i1 - s1, base2 = i2 - s2;  // totals of the lanes before this one
  const double tot1 = read_lane_f64<63>(i1), tot2 = read_lane_f64<63>(i2);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = lane * 4 + k;
    Ts[1 + wv * QC_MAX_N + e] = v[k];
    P1[wv][e] = base1 + t1[k];
    P2[wv][e] = base2 + t2[k];
  }
  if (lane == 63) {
    P1[wv][QC_MAX_N] = tot1;
    P2[wv][QC_MAX_N] = tot2;
  }
  wave_lds_sync();
  const char* Tb = (const char*)&Ts[0];
  const int slice = wv * QC_MAX_N * 4;  // byte offset of this wave's slice
  const double* Q1 = P1[wv];
  const double* Q2 = P2[wv];
  const float inv = 1.f / ((float)N * (float)a.batch * (float)N);
  float loss = 0.f;
  // The lane's (up to four) quantiles j = lane + 64 q are searched TOGETHER: for each, the number of sorted targets
  // <= C - 1, < C and < C + 1.  Branch-free bisection over the padded 256 entries (the +inf padding never counts), the
  // twelve chains advancing in lock step, so that each of the eight rounds has twelve independent LDS reads in flight
  // instead of one (three searches after each other per quantile left the wave waiting on a chain of 24 dependent
  // reads; the counts, hence the results, are the same).
  int c1[QJ], c2[QJ], c3[QJ];  // byte offsets: slice + 4 * count
#pragma unroll
  for (int q = 0; q < QJ; ++q) c1[q] = c2[q] = c3[q] = slice;
#pragma unroll
  for (int step = QC_MAX_N / 2; step >= 1; step >>= 1) {
#pragma unroll
    for (int q = 0; q < QJ; ++q) {
      const int n1 = c1[q] | (4 * step), n2 = c2[q] | (4 * step), n3 = c3[q] | (4 * step);  // count + step, as an address
      const float t1 = *(const float*)(Tb + n1), t2 = *(const float*)(Tb + n2), t3 = *(const float*)(Tb + n3);
      c1[q] = t1 <= cq[q] - 1.f ? n1 : c1[q];
      c2[q] = t2 < cq[q] ? n2 : c2[q];
      c3[q] = t3 < cq[q] + 1.f ? n3 : c3[q];
    }
  }
#pragma unroll
  for (int q = 0; q < QJ; ++q) {  // back to counts; the bisection stops at 255: the last entry (a target only when N == 256)
    c1[q] = (c1[q] - slice) >> 2;
    c2[q] = (c2[q] - slice) >> 2;
    c3[q] = (c3[q] - slice) >> 2;
    const float last = *(const float*)(Tb + slice + 4 * QC_MAX_N);
    c1[q] += (c1[q] == QC_MAX_N - 1 && last <= cq[q] - 1.f) ? 1 : 0;
    c2[q] += (c2[q] == QC_MAX_N - 1 && last < cq[q]) ? 1 : 0;
    c3[q] += (c3[q] == QC_MAX_N - 1 && last < cq[q] + 1.f) ? 1 : 0;
  }
#pragma unroll
  for (int q = 0; q < QJ; ++q) {
    const int j = lane + 64 * q;
    if (j >= a.lddz) break;
    float gsum = 0.f;
    if (j < N) {
      const float c = cq[q], tau = a.quantiles[j];
      const int p1 = c1[q], p2 = c2[q], p3 = c3[q];
      const double cd = (double)c, omt = 1.0 - (double)tau, td_ = (double)tau;
      const double n1 = p1, n2 = p2 - p1, n3 = p3 - p2, n4 = N - p3;
      const double s1_1 = Q1[p1], s1_2 = Q1[p2] - Q1[p1], s1_3 = Q1[p3] - Q1[p2], s1_4 = Q1[N] - Q1[p3];
      const double s2_2 = Q2[p2] - Q2[p1], s2_3 = Q2[p3] - Q2[p2];
      const double l = omt * (n1 * (cd - 0.5) - s1_1) + omt * 0.5 * (s2_2 - 2.0 * cd * s1_2 + n2 * cd * cd) +
                       td_ * 0.5 * (s2_3 - 2.0 * cd * s1_3 + n3 * cd * cd) + td_ * (s1_4 - n4 * (cd + 0.5));
      // sum_i huber'(T_i - C) * weight:  -1, (T - C), (T - C), +1 on the four ranges
      const double gs = -omt * n1 + omt * (s1_2 - n2 * cd) + td_ * (s1_3 - n3 * cd) + td_ * n4;
      loss += (float)l;
      gsum = (float)gs;
    }
    dz[j] = -gsum * inv;
  }
  for (int j = QC_MAX_N + lane; j < a.lddz; j += 64) dz[j] = 0.f;  // row padding beyond the 256 searched columns
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) loss += shfl_xor(loss, off);
  if (lane == 0) a.loss_partials[r] = loss * inv;
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

// db[g * Ng + n] = sum over the segments of group g of db_part[unit + g][n] (rg_mlp_frag.h: grouped_bias_reduce_launch)
__global__ void group_bias_reduce_kernel(const float* __restrict__ db_part, const int* __restrict__ row_begin, int Ng,
                                         int NgP, float* __restrict__ db, int unit_rows) {
  const int g = blockIdx.x;
  const int r0 = row_begin[g], r1 = row_begin[g + 1];
  const int t0 = r0 / unit_rows + g, t1 = r1 > r0 ? (r1 + unit_rows - 1) / unit_rows + g : t0;
  for (int n = threadIdx.x; n < Ng; n += blockDim.x) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four tiles in flight (a ~30-deep chain of dependent loads was 15 us)
    int t = t0;
    for (; t + 3 < t1; t += 4) {
      s0 += db_part[(long)t * NgP + n];
      s1 += db_part[(long)(t + 1) * NgP + n];
      s2 += db_part[(long)(t + 2) * NgP + n];
      s3 += db_part[(long)(t + 3) * NgP + n];
    }
    for (; t < t1; ++t) s0 += db_part[(long)t * NgP + n];
    db[g * Ng + n] = (s0 + s1) + (s2 + s3);
  }
}

void grouped_bias_reduce_launch(const float* db_part, const int* row_begin, int n_groups, int Ng, float* db, int unit_rows,
                                hipStream_t stream) {
  RG_LAUNCH(group_bias_reduce_kernel, dim3(n_groups), dim3(256), stream, db_part, row_begin, Ng, Ng, db, unit_rows);
}


}  // namespace rg

using namespace rg;

extern "C" {

size_t rg_group_wfrag_elems(int group_rows, int in_features, int transposed) {
  return transposed ? wfrag_elems(in_features, group_rows) : wfrag_elems(group_rows, in_features);
}

int rg_group_weights_stage(const float* w, int n_groups, int group_rows, int in_features, int x3, void* wfrag_fwd,
                           void* wfrag_bwd, rg_stream_t stream) {
  if (!w || n_groups <= 0 || group_rows <= 0 || in_features <= 0 || (!wfrag_fwd && !wfrag_bwd)) return RG_EINVAL;
  const long per_f = (long)wfrag_elems(group_rows, in_features), per_b = (long)wfrag_elems(in_features, group_rows);
  const long per = per_f > per_b ? per_f : per_b;
  const long total = per * n_groups;
  RG_LAUNCH(group_stage_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (hipStream_t)stream, w, n_groups,
            group_rows, in_features, (bf16_t*)wfrag_fwd, (bf16_t*)wfrag_bwd, per_f, per_b, x3 ? 1 : 0);
  return (int)hipGetLastError();
}

int rg_wide_head_mean(const float* w, const float* b, int n_groups, int group_rows, int in_features, float* wbar,
                      float* bbar, rg_stream_t stream) {
  if (!w || !wbar || !bbar || n_groups <= 0 || group_rows <= 0 || in_features <= 0) return RG_EINVAL;
  RG_LAUNCH(wide_mean_kernel, dim3((in_features + 31) / 32, n_groups), dim3(256), (hipStream_t)stream, w, b, n_groups,
            group_rows, in_features, wbar, bbar, (bf16_t*)nullptr, 0);
  return (int)hipGetLastError();
}

int rg_wide_head_mean_staged(const float* w, const float* b, int n_groups, int group_rows, int in_features, float* wbar,
                             float* bbar, void* wfrag_fwd, int x3, rg_stream_t stream) {
  if (!w || !wbar || !bbar || !wfrag_fwd || n_groups <= 0 || group_rows <= 0 || in_features <= 0) return RG_EINVAL;
  RG_LAUNCH(wide_mean_kernel, dim3((in_features + 31) / 32, n_groups), dim3(256), (hipStream_t)stream, w, b, n_groups,
            group_rows, in_features, wbar, bbar, (bf16_t*)wfrag_fwd, x3 ? 1 : 0);
  return (int)hipGetLastError();
}

size_t rg_group_rows_workspace_bytes(int batch, int n_groups) {
  return (size_t)((batch + GR_BLOCK - 1) / GR_BLOCK) * (n_groups + 1) * sizeof(int);
}

int rg_group_rows(const int32_t* key, int batch, int n_groups, int n_tiles, int dense, int32_t* rowmap, int32_t* tile_key,
                  int32_t* row_begin, void* workspace, size_t workspace_bytes, rg_stream_t stream) {
  if (!key || !rowmap || !tile_key || !row_begin || batch <= 0 || n_groups <= 0 || n_groups + 1 > GR_MAX_KEYS ||
      n_tiles < (batch + 127) / 128 + (dense ? 0 : n_groups))
    return RG_EINVAL;
  if (!workspace || workspace_bytes < rg_group_rows_workspace_bytes(batch, n_groups)) return RG_EWORKSPACE;
  const int nblk = (batch + GR_BLOCK - 1) / GR_BLOCK;
  int* hist = (int*)workspace;
  RG_LAUNCH(group_count_kernel, dim3(nblk), dim3(GR_BLOCK), (hipStream_t)stream, (int*)key, batch, n_groups, hist, rowmap,
            n_tiles * 128, (const float*)nullptr, 0L, (const float*)nullptr, 0);
  RG_LAUNCH(group_scatter_kernel, dim3(nblk), dim3(GR_BLOCK), (hipStream_t)stream, key, batch, n_groups, (const int*)hist,
            nblk, n_tiles, dense ? 1 : 0, row_begin, tile_key, rowmap);
  return (int)hipGetLastError();
}

int rg_qr_select_group_rows(const float* q, int64_t ldq, const float* mask, int batch, int num_actions, int maxq,
                            int32_t* key, int n_tiles, int dense, int32_t* rowmap, int32_t* tile_key, int32_t* row_begin,
                            void* workspace, size_t workspace_bytes, rg_stream_t stream) {
  const int n_groups = num_actions;
  if (!mask || !key || !rowmap || !tile_key || !row_begin || batch <= 0 || n_groups <= 0 || n_groups + 1 > GR_MAX_KEYS ||
      n_tiles < (batch + 127) / 128 + (dense ? 0 : n_groups) || (maxq && !q))
    return RG_EINVAL;
  if (!workspace || workspace_bytes < rg_group_rows_workspace_bytes(batch, n_groups)) return RG_EWORKSPACE;
  const int nblk = (batch + GR_BLOCK - 1) / GR_BLOCK;
  int* hist = (int*)workspace;
  RG_LAUNCH(group_count_kernel, dim3(nblk), dim3(GR_BLOCK), (hipStream_t)stream, key, batch, n_groups, hist, rowmap,
            n_tiles * 128, q, (long)ldq, mask, maxq);
  RG_LAUNCH(group_scatter_kernel, dim3(nblk), dim3(GR_BLOCK), (hipStream_t)stream, (const int*)key, batch, n_groups,
            (const int*)hist, nblk, n_tiles, dense ? 1 : 0, row_begin, tile_key, rowmap);
  return (int)hipGetLastError();
}

int rg_qr_select_action(const float* q, int64_t ldq, const float* mask, int batch, int num_actions, int maxq, int32_t* key,
                        rg_stream_t stream) {
  if (!mask || !key || batch <= 0 || num_actions <= 0 || (maxq && !q)) return RG_EINVAL;
  RG_LAUNCH(select_action_kernel, dim3((batch + 255) / 256), dim3(256), (hipStream_t)stream, q, (long)ldq, mask, batch,
            num_actions, maxq, key);
  return (int)hipGetLastError();
}

int rg_qr_compact_head(const float* z, int64_t ldz, const float* zt, int64_t ldzt, const int32_t* rowmap,
                       const int32_t* row_key, int padded_rows, const float* reward, const float* reward_boosts,
                       const float* not_terminal, double gamma, const float* gamma_exponent, const float* quantiles,
                       int batch, int num_atoms, float* dz, int64_t lddz, float* loss_partials, float* tile_losses,
                       rg_stream_t stream) {
  if (!z || !zt || !rowmap || (reward_boosts && !row_key) || !reward || !not_terminal || !quantiles || !dz || !loss_partials ||
      padded_rows <= 0 || (padded_rows % 128) != 0 || batch <= 0 || num_atoms <= 0)
    return RG_EINVAL;
  if (num_atoms > QC_MAX_N || lddz < num_atoms) return RG_EUNSUPPORTED;
  CompactHeadArgs a;
  a.z = z; a.zt = zt; a.ldz = ldz; a.ldzt = ldzt; a.rowmap = rowmap; a.row_key = row_key; a.reward = reward;
  a.reward_boosts = reward_boosts; a.not_terminal = not_terminal; a.gamma_exponent = gamma_exponent; a.quantiles = quantiles;
  a.gamma = (float)gamma; a.batch = batch; a.N = num_atoms; a.dz = dz; a.lddz = lddz; a.loss_partials = loss_partials;
  RG_LAUNCH(qr_compact_head_kernel, dim3(padded_rows / QC_WAVES), dim3(QC_WAVES * 64), (hipStream_t)stream, a);
  int rc = (int)hipGetLastError();
  if (rc || !tile_losses) return rc;
  RG_LAUNCH(tile_sum_kernel, dim3(padded_rows / 128), dim3(128), (hipStream_t)stream, (const float*)loss_partials, tile_losses);
  return (int)hipGetLastError();
}

}  // extern "C"

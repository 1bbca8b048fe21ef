// replay.hip — device-resident circular replay buffer sampling (HBM-bound byte movers).
// Replaces the ~17 advanced-indexing ops of ReplayBuffer.sample_transition_batch,
// reagent/replay_memory/circular_replay_buffer.py:614-706 (+ :741-774).  Bit-exact by
// construction: rows are copied, never recomputed; the n-step reward uses the reference's
// operation order (r * gamma^k * mask, summed k = 0..h-1).
#include "rg_norm.h"

namespace rg {

// steps / next index / terminal / n-step reward for every sampled index
__global__ void replay_nstep_kernel(const int64_t* __restrict__ indices,
                                    const uint8_t* __restrict__ terminal,
                                    const float* __restrict__ reward,
                                    const float* __restrict__ decays, int64_t capacity, int horizon,
                                    int batch, int64_t* __restrict__ steps,
                                    int64_t* __restrict__ next_indices,
                                    uint8_t* __restrict__ out_terminal,
                                    float* __restrict__ out_reward) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int64_t idx = indices[b];
  // _get_steps (:759-774): first terminal inside the window, the last slot counts as terminal
  int st = horizon;
  for (int k = 0; k < horizon; ++k) {
    if (terminal[(idx + k) % capacity]) {
      st = k + 1;
      break;
    }
  }
  if (steps) steps[b] = st;
  if (next_indices) next_indices[b] = (idx + st) % capacity;
  if (out_terminal) out_terminal[b] = terminal[(idx + st - 1) % capacity] ? 1 : 0;
  if (out_reward) {
    // _reduce_multi_step_reward (:741-747): (reward * decays * masks).sum(dim=1)
    float acc = 0.f;
    for (int k = 0; k < horizon; ++k) {
      const float m = (k < st) ? 1.f : 0.f;
      acc += (reward[(idx + k) % capacity] * decays[k]) * m;
    }
    out_reward[b] = acc;
  }
}

struct GatherTable {
  rg_gather_col c[RG_MAX_GATHER_COLS];
};

// stack == 1: each sampled row is a contiguous run of row_bytes; copy it with the widest
// aligned unit (16 B when the row pitch allows, else the element size).  One workgroup moves
// ROWS_PER_WG rows of one column; consecutive lanes take consecutive 16-B pieces of a row so a
// 512-B observation row is one fully coalesced half-wave request.
#ifndef RG_GATHER_REGCOLS
#define RG_GATHER_REGCOLS 1  // replay_dqn_batch_kernel: column descriptors as a structure of arrays in LDS (16-byte, conflict-free reads)
#endif
constexpr int GATHER_ROWS_PER_WG = 64;
constexpr int GATHER_MAX_LDS_COLS = 512;  // descriptors of one column staged in LDS (12 KB)

__global__ void replay_gather_kernel(GatherTable t, int64_t capacity, int batch) {
  // a reference: the fields are read from the kernel-argument segment with scalar loads (a by-value
  // copy of a dynamically indexed element became a private array that LLVM promoted to 64 KB of LDS)
  const rg_gather_col& col = t.c[blockIdx.y];
  const int row0 = blockIdx.x * GATHER_ROWS_PER_WG;
  const int nrows = (batch - row0 < GATHER_ROWS_PER_WG) ? batch - row0 : GATHER_ROWS_PER_WG;
  const long row_bytes = (long)col.row_elems * col.elem_bytes;
  const char* src = (const char*)col.src;
  char* dst = (char*)col.dst;
  const bool vec16 = (row_bytes % 16 == 0) && ((((uintptr_t)src) & 15) == 0) &&
                     ((((uintptr_t)dst) & 15) == 0);
  // the workgroup's sampled indices, fetched once and coalesced: without this every 16-byte piece
  // pays an index load and then a dependent row load (two serial memory latencies per piece)
  __shared__ int64_t s_idx[GATHER_ROWS_PER_WG];
  if ((int)threadIdx.x < nrows) s_idx[threadIdx.x] = col.indices[row0 + threadIdx.x];
  __syncthreads();
  constexpr int GU = 4;  // pieces per thread in flight (all loads issued before the first store)
  if (col.norm && ((col.row_elems & 3) == 0) && col.row_elems <= GATHER_MAX_LDS_COLS &&
      ((((uintptr_t)src) & 15) == 0) && ((((uintptr_t)dst) & 7) == 0)) {
    // normalize-on-gather, aligned rows: 4 fp32 features per lane, op-code table applied in
    // registers, optional bf16 output (the normalized fp32 matrix never exists in HBM).
    // The descriptors are staged in LDS once per workgroup and every thread has ONE 16-byte piece in
    // flight: the parallelism comes from occupancy (few registers), not from unrolling.  Measured at
    // C2 (MI355X, same box): this loop 49 us; one 4-feature slot per thread with its descriptors in
    // registers and four pieces in flight 55.5 us (100 VGPRs, 5 waves per SIMD).
    __shared__ rg_norm_col s_nc[GATHER_MAX_LDS_COLS];
    const rg_norm_col* nc = (const rg_norm_col*)col.norm;
    const int epr = col.row_elems, cpr = epr / 4;
    for (int j = threadIdx.x; j < epr; j += blockDim.x) s_nc[j] = nc[j];
    __syncthreads();
    const int total = nrows * cpr;
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
      const int r = it / cpr, ch = it - r * cpr;
      const f32x4 raw = stream_load((const f32x4*)((const float*)src + s_idx[r] * epr + ch * 4));  // a sampled row is read once
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = normalize_value(s_nc[ch * 4 + e], raw[e], 1.f, col.norm_quantiles);
      if (col.out_dtype == RG_DT_BF16) {
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)((bf16_t*)dst + (long)(row0 + r) * epr + ch * 4) = o;
      } else {
        *(f32x4*)((float*)dst + (long)(row0 + r) * epr + ch * 4) = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  } else if (col.norm) {
    // normalize-on-gather: 4 fp32 features per lane, op-code table applied in registers, optional
    // bf16 output (the network-ready layout: the normalized fp32 matrix never exists in HBM)
    const rg_norm_col* nc = (const rg_norm_col*)col.norm;
    const int epr = col.row_elems;
    const int cpr = (epr + 3) / 4;
    const int total = nrows * cpr;
    const bool v4 = ((epr & 3) == 0) && ((((uintptr_t)src) & 15) == 0);
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
      const int r = it / cpr, ch = it % cpr;
      const int64_t idx = s_idx[r];
      const float* sp = (const float*)src + idx * epr + ch * 4;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (v4) {
        const f32x4 t = *(const f32x4*)sp;
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
      } else {
        for (int e = 0; e < 4; ++e)
          if (ch * 4 + e < epr) v[e] = sp[e];
      }
      for (int e = 0; e < 4; ++e)
        if (ch * 4 + e < epr) v[e] = normalize_value(nc[ch * 4 + e], v[e], 1.f, col.norm_quantiles);
      if (col.out_dtype == RG_DT_BF16) {
        bf16_t* dp = (bf16_t*)dst + (long)(row0 + r) * epr + ch * 4;
        if (v4) {
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]);
          o.y = pack_bf16x2(v[2], v[3]);
          *(uint2*)dp = o;
        } else {
          for (int e = 0; e < 4; ++e)
            if (ch * 4 + e < epr) dp[e] = f32_to_bf16(v[e]);
        }
      } else {
        float* dp = (float*)dst + (long)(row0 + r) * epr + ch * 4;
        for (int e = 0; e < 4; ++e)
          if (ch * 4 + e < epr) dp[e] = v[e];
      }
    }
  } else if (vec16) {
    const int cpr = (int)(row_bytes / 16);
    const int total = nrows * cpr;
    for (int it0 = threadIdx.x; it0 < total; it0 += blockDim.x * GU) {
      f32x4 raw[GU];  // 16-byte carrier (a native vector: HIP's uint4 struct would not leave memory)
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int it = it0 + u * blockDim.x < total ? it0 + u * (int)blockDim.x : total - 1;
        raw[u] = stream_load((const f32x4*)(src + s_idx[it / cpr] * row_bytes + (long)(it % cpr) * 16));
      }
#pragma unroll
      for (int u = 0; u < GU; ++u) {
        const int it = it0 + u * blockDim.x;
        if (it >= total) continue;
        *(f32x4*)(dst + (long)(row0 + it / cpr) * row_bytes + (long)(it % cpr) * 16) = raw[u];
      }
    }
  } else {
    const int eb = col.elem_bytes;
    const int epr = col.row_elems;
    const int total = nrows * epr;
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
      const int r = it / epr, e = it % epr;
      const int64_t idx = s_idx[r];
      const char* s = src + idx * row_bytes + (long)e * eb;
      char* d = dst + (long)(row0 + r) * row_bytes + (long)e * eb;
      if (eb == 8) *(uint64_t*)d = *(const uint64_t*)s;
      else if (eb == 4) *(uint32_t*)d = *(const uint32_t*)s;
      else if (eb == 2) *(uint16_t*)d = *(const uint16_t*)s;
      else *d = *s;
    }
  }
}

// stack > 1 (_get_stack_for_indices :749-757 + DenseMetadata.sample_to_output :133-141):
// dst[b, e, s] = src[(idx[b] - (stack-1) + s) mod capacity, e]
__global__ void replay_gather_stack_kernel(GatherTable t, int64_t capacity, int stack, int batch) {
  const rg_gather_col col = t.c[blockIdx.y];
  const int eb = col.elem_bytes, epr = col.row_elems;
  const long per_row = (long)epr * stack;
  const long total = (long)batch * per_row;
  const char* src = (const char*)col.src;
  char* dst = (char*)col.dst;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < total;
       it += (long)gridDim.x * blockDim.x) {
    const long b = it / per_row;
    const int rem = (int)(it % per_row);
    const int e = rem / stack, s = rem % stack;
    int64_t r = (col.indices[b] - (stack - 1) + s) % capacity;
    if (r < 0) r += capacity;
    const char* sp = src + (r * epr + e) * (long)eb;
    char* dp = dst + it * (long)eb;
    if (eb == 8) *(uint64_t*)dp = *(const uint64_t*)sp;
    else if (eb == 4) *(uint32_t*)dp = *(const uint32_t*)sp;
    else if (eb == 2) *(uint16_t*)dp = *(const uint16_t*)sp;
    else *dp = *sp;
  }
}

// DiscreteDqnInputMaker.__call__ / one_hot_actions,
// reagent/gym/preprocessors/trainer_preprocessor.py:72-97,118-158, in one pass:
//   action      = one_hot(action)                      (fp32)
//   next_action = one_hot(next_action), zero rows where terminal
//   not_terminal = 1 - terminal ;  action_probability = exp(log_prob)
__global__ void make_dqn_input_kernel(const int64_t* __restrict__ action,
                                      const int64_t* __restrict__ next_action,
                                      const uint8_t* __restrict__ terminal,
                                      const float* __restrict__ log_prob, int batch, int A,
                                      float* __restrict__ action_1h, float* __restrict__ next_action_1h,
                                      float* __restrict__ not_terminal,
                                      float* __restrict__ action_probability) {
  const long total = (long)batch * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / A;
    const int a = (int)(i % A);
    const bool term = terminal[b] != 0;
    action_1h[i] = (action[b] == a) ? 1.f : 0.f;
    next_action_1h[i] = (!term && next_action[b] == a) ? 1.f : 0.f;
    if (a == 0) {
      not_terminal[b] = 1.0f - (term ? 1.f : 0.f);
      if (action_probability) action_probability[b] = expf(log_prob[b]);
    }
  }
}

// ---- sparse replay elements (IDListMetadata / IDScoreListMetadata, circular_replay_buffer.py:144-274) -------------
// A feature's lists live in padded slots ids [capacity, W] (+ scores [capacity, W]) with lens [capacity] beside them.
// sample_to_output of a batch = (offsets [B] = exclusive prefix sums of the sampled rows' lengths, the rows' ids
// (and scores) back to back).  Two launches: the scan (one workgroup, carry over 1024-row chunks: the offsets are
// sequential by definition) writes offsets and the total; the copy moves one row per wave.
__global__ void ragged_offsets_kernel(const int* __restrict__ lens, const int64_t* __restrict__ idx, int batch,
                                      int* __restrict__ offsets, int* __restrict__ total) {
  __shared__ int part[1024];
  __shared__ int carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < batch; base += 1024) {
    const int b = base + t;
    const int v = b < batch ? lens[idx[b]] : 0;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan of the chunk
      const int add = t >= off ? part[t - off] : 0;
      __syncthreads();
      part[t] += add;
      __syncthreads();
    }
    if (b < batch) offsets[b] = carry + part[t] - v;
    __syncthreads();
    if (t == 1023) carry += part[1023];
    __syncthreads();
  }
  if (t == 0) total[0] = carry;
}

__global__ void ragged_copy_kernel(const int64_t* __restrict__ ids, const float* __restrict__ scores, int W,
                                   const int* __restrict__ lens, const int64_t* __restrict__ idx, const int* __restrict__ offsets,
                                   int batch, int64_t* __restrict__ ids_out, float* __restrict__ scores_out) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = lane_id();
  if (b >= batch) return;
  const int64_t row = idx[b];
  const int n = lens[row], o = offsets[b];
  for (int j = lane; j < n; j += 64) {
    ids_out[o + j] = ids[row * W + j];
    if (scores) scores_out[o + j] = scores[row * W + j];
  }
}

// PolicyNetworkInputMaker (trainer_preprocessor.py:161-227, dense path) in one launch:
//   action, next_action = rescale_actions(., new = training range, prev = the environment's range)
//                       = ((a - prev_min) / (prev_max - prev_min)) * (new_max - new_min) + new_min   (training/utils.py:13-29),
//   next_action rows of terminal transitions are zero, not_terminal = 1 - terminal, action_probability = exp(log_prob).
// ranges [4][A] = prev_min, prev_max, new_min, new_max per action dimension.  The arithmetic keeps the reference's
// operation order with every product and sum rounded on its own (no fused multiply-add), so the fp32 results are torch's.
__global__ void make_policy_input_kernel(const float* __restrict__ action, long lda, const float* __restrict__ next_action,
                                         long ldna, const uint8_t* __restrict__ terminal, const float* __restrict__ log_prob,
                                         const float* __restrict__ ranges, int batch, int A, float* __restrict__ action_out,
                                         float* __restrict__ next_action_out, float* __restrict__ not_terminal,
                                         float* __restrict__ action_probability) {
  const long total = (long)batch * A;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / A;
    const int d = (int)(i % A);
    const float lo = ranges[d], hi = ranges[A + d], tl = ranges[2 * A + d], th = ranges[3 * A + d];
    const float prev_range = __fsub_rn(hi, lo), new_range = __fsub_rn(th, tl);
    const float nt = 1.0f - (terminal[b] ? 1.f : 0.f);
    const float a = __fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(action[b * lda + d], lo), prev_range), new_range), tl);
    const float n = __fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(next_action[b * ldna + d], lo), prev_range), new_range), tl);
    action_out[i] = a;
    next_action_out[i] = terminal[b] ? 0.f : n;  // zeros_like + assignment of the non-terminal rows (:186-196)
    if (d == 0) {
      not_terminal[b] = nt;
      if (action_probability) action_probability[b] = expf(log_prob[b]);
    }
  }
}

// ---- sampled indices -> rlt.DiscreteDqnInput in one launch --------------------------------------
// ReplayBuffer.sample_transition_batch (:614-706, stack_size 1) + DiscreteDqnInputMaker
// (reagent/gym/preprocessors/trainer_preprocessor.py:100-158) [+ Preprocessor.forward on both state
// matrices, all features present]: the n-step bookkeeping of replay_nstep_kernel is recomputed by each
// of the three workgroups that serve a block of 64 transitions (a handful of byte loads), so the
// next-state rows can be fetched in the launch that finds them, and the one-hot / not_terminal /
// exp(log_prob) work of make_dqn_input_kernel reads the store directly.  Same arithmetic, operation
// for operation, as the three kernels it replaces (tests compare them bit for bit).
struct ReplayBatchArgs {
  rg_replay_view v;
  rg_dqn_batch_out o;
};

// cursor != null (rg_replay_dqn_batch_pooled, replayed HIP graphs): `indices` is a POOL of index rows [pool rows][batch]
// and this launch samples row cursor[0]; pre_tick != null: the launch also counts the step in the device-resident Adam
// schedule (sched[0] += 1: nobody reads it between this launch and the step's update, which is then told that the count
// already includes it — rg_mlp_update_desc.sched_pre_ticked).  The update launch advances the cursor in turn
// (rg_mlp_update_desc.post_tick): a replayed step needs neither an index copy nor a tick launch of its own.
__global__ void replay_dqn_batch_kernel(ReplayBatchArgs a, const int64_t* __restrict__ indices, int batch,
                                        const rg_norm_col* __restrict__ cols, const float* __restrict__ quantiles,
                                        const int64_t* __restrict__ cursor, double* __restrict__ pre_tick) {
  if (cursor) indices += cursor[0] * (long)batch;
  if (pre_tick && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) pre_tick[0] = pre_tick[0] + 1.0;
  __shared__ int64_t s_src[GATHER_ROWS_PER_WG];  // the row this piece reads: idx (state), next idx (next_state)
  __shared__ int64_t s_nxt[GATHER_ROWS_PER_WG];
  __shared__ int s_steps[GATHER_ROWS_PER_WG];
  __shared__ unsigned char s_term[GATHER_ROWS_PER_WG];
  __shared__ int s_act[GATHER_ROWS_PER_WG], s_nact[GATHER_ROWS_PER_WG];
  __shared__ __attribute__((aligned(16))) rg_norm_col s_nc[GATHER_MAX_LDS_COLS];
  const rg_replay_view& v = a.v;
  const rg_dqn_batch_out& o = a.o;
  const int row0 = blockIdx.x * GATHER_ROWS_PER_WG;
  const int nrows = (batch - row0 < GATHER_ROWS_PER_WG) ? batch - row0 : GATHER_ROWS_PER_WG;
  // (Round 3, same-box A/B at C2, 39-40 us per launch in every form: one workgroup reading BOTH rows of a transition — two
  // loads in flight per thread, the next-state row the store neighbour of the state row — and two independent row pieces
  // in flight per thread measured the same as this one-piece-per-thread loop: with eight waves per SIMD the launch is not
  // short of requests in flight, it runs at what random 512-byte rows get from the HBM.)
  const int piece = blockIdx.y;  // 0 = state, 1 = next_state, 2 = everything else
  const int F = v.n_features, A = v.n_actions, H = v.update_horizon;
  const int64_t C = v.capacity;
  // (idx + k) % C without the 64-bit division: idx < C and k <= H <= C
  auto wrap = [&](int64_t t) { return t >= C ? t - C : t; };
  if ((int)threadIdx.x < nrows) {
    const int64_t idx = indices[row0 + threadIdx.x];
    int st = H;  // _get_steps (:759-774); with H == 1 the window is one slot whatever it holds
    if (H > 1)
      for (int k = 0; k < H; ++k)
        if (v.terminal[wrap(idx + k)]) {
          st = k + 1;
          break;
        }
    const int64_t nidx = wrap(idx + st);
    s_src[threadIdx.x] = piece == 1 ? nidx : idx;
    s_nxt[threadIdx.x] = nidx;
    s_steps[threadIdx.x] = st;
    if (piece == 2) {  // the row pieces never read these
      s_term[threadIdx.x] = v.terminal[wrap(idx + st - 1)] ? 1 : 0;
      const int64_t a0 = v.action[idx], a1 = v.action[nidx];  // fetched once per row, not once per (row, action)
      s_act[threadIdx.x] = (a0 >= 0 && a0 < A) ? (int)a0 : -1;
      s_nact[threadIdx.x] = (a1 >= 0 && a1 < A) ? (int)a1 : -1;
    }
  }
  // Round 5: the descriptors of the four columns of a 16-byte chunk come out of LDS as three 16-byte reads of a
  // structure-of-arrays image (op | p0 | p1; p2 / p3 only for the two ops that have them) instead of twelve 4-byte reads of
  // 24-byte records at a lane stride of 96 bytes — an 8-way bank conflict each: PMC (profiles/r04_pmc) showed the LDS pipe
  // busy for half of the launch, 74 % of those cycles conflicts.  Same arithmetic, same bits.
  const int cpr = F >> 2;  // F % 4 == 0 (checked by the host)
#if RG_GATHER_REGCOLS
  int* s_op = (int*)s_nc;                       // [GATHER_MAX_LDS_COLS] each, inside the same 12 KB
  float* s_p0 = (float*)s_nc + GATHER_MAX_LDS_COLS;
  float* s_p1 = s_p0 + GATHER_MAX_LDS_COLS;
  float* s_p2 = s_p1 + GATHER_MAX_LDS_COLS;
  float* s_p3 = s_p2 + GATHER_MAX_LDS_COLS;
  if (piece < 2 && cols)
    for (int j = threadIdx.x; j < F; j += blockDim.x) {
      const rg_norm_col c = cols[j];
      s_op[j] = c.op; s_p0[j] = c.p0; s_p1[j] = c.p1; s_p2[j] = c.p2; s_p3[j] = c.p3;
    }
  __syncthreads();
  if (piece < 2) {
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    void* dst = piece == 0 ? o.state : o.next_state;
    const int total = nrows * cpr;
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
      const int r = it / cpr, ch = it - r * cpr;
      const f32x4 raw = stream_load((const f32x4*)(v.observation + s_src[r] * F + ch * 4));  // (read once: streaming, see below)
      float w[4] = {raw[0], raw[1], raw[2], raw[3]};
      if (cols) {
        const i32x4 op = *(const i32x4*)(s_op + ch * 4);
        const f32x4 p0 = *(const f32x4*)(s_p0 + ch * 4), p1 = *(const f32x4*)(s_p1 + ch * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rg_norm_col c;
          c.op = op[e]; c.in_col = 0; c.p0 = p0[e]; c.p1 = p1[e]; c.p2 = 0.f; c.p3 = 0.f;
          if (c.op == RG_NORM_BOXCOX || c.op == RG_NORM_CONTINUOUS_ACTION) {
            c.p2 = s_p2[ch * 4 + e];
            c.p3 = s_p3[ch * 4 + e];
          }
          w[e] = normalize_value(c, w[e], 1.f, quantiles);
        }
      }
      const long at = (long)(row0 + r) * F + ch * 4;
      if (o.state_dtype == RG_DT_BF16) {
        uint2 pk;
        pk.x = pack_bf16x2(w[0], w[1]);
        pk.y = pack_bf16x2(w[2], w[3]);
        *(uint2*)((bf16_t*)dst + at) = pk;
      } else {
        *(f32x4*)((float*)dst + at) = f32x4{w[0], w[1], w[2], w[3]};
      }
    }
    return;
  }
#else
  if (piece < 2 && cols)
    for (int j = threadIdx.x; j < F; j += blockDim.x) s_nc[j] = cols[j];
  __syncthreads();
#endif
  if (piece < 2) {
    void* dst = piece == 0 ? o.state : o.next_state;
    const int total = nrows * cpr;
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
      const int r = it / cpr, ch = it - r * cpr;
      // a sampled row is read once: streaming load, so that 67 MB of replay rows per batch do not push the networks'
      // weights out of L2 (same-box A/B with the streaming reduce loads: C2 step -2.8 %)
      const f32x4 raw = stream_load((const f32x4*)(v.observation + s_src[r] * F + ch * 4));
      float w[4] = {raw[0], raw[1], raw[2], raw[3]};
      if (cols) {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = normalize_value(s_nc[ch * 4 + e], w[e], 1.f, quantiles);
      }
      const long at = (long)(row0 + r) * F + ch * 4;
      if (o.state_dtype == RG_DT_BF16) {
        uint2 pk;
        pk.x = pack_bf16x2(w[0], w[1]);
        pk.y = pack_bf16x2(w[2], w[3]);
        *(uint2*)((bf16_t*)dst + at) = pk;
      } else {
        *(f32x4*)((float*)dst + at) = f32x4{w[0], w[1], w[2], w[3]};
      }
    }
    return;
  }
  if ((int)threadIdx.x < nrows) {
    const int b = row0 + threadIdx.x, st = s_steps[threadIdx.x];
    const int64_t idx = s_src[threadIdx.x];
    // _reduce_multi_step_reward (:741-747): (reward * decays * masks).sum(dim=1), in that order
    float acc = 0.f;
    for (int k = 0; k < H; ++k) {
      const float m = (k < st) ? 1.f : 0.f;
      acc += (v.reward[wrap(idx + k)] * v.decays[k]) * m;
    }
    o.reward[b] = acc;
    o.not_terminal[b] = 1.0f - (s_term[threadIdx.x] ? 1.f : 0.f);
    if (o.step) o.step[b] = (float)st;
    if (o.time_diff) o.time_diff[b] = 1.f;
    if (o.action_probability) o.action_probability[b] = v.log_prob ? expf(v.log_prob[idx]) : 1.f;
    if (o.mdp_id) o.mdp_id[b] = v.mdp_id ? v.mdp_id[idx] : 0;
    if (o.sequence_number) o.sequence_number[b] = v.sequence_number ? v.sequence_number[idx] : 0;
  }
  const int total = nrows * A;
  for (int it = threadIdx.x; it < total; it += blockDim.x) {
    const int r = it / A, k = it - r * A;
    const int64_t idx = s_src[r], nidx = s_nxt[r];
    const long at = (long)(row0 + r) * A + k;
    o.action[at] = (s_act[r] == k) ? 1.f : 0.f;
    o.next_action[at] = (!s_term[r] && s_nact[r] == k) ? 1.f : 0.f;
    if (o.possible_actions_mask)
      o.possible_actions_mask[at] = v.possible_actions_mask ? v.possible_actions_mask[idx * A + k] : 1.f;
    o.possible_next_actions_mask[at] = v.possible_actions_mask ? v.possible_actions_mask[nidx * A + k] : 1.f;
  }
}

// ---- sampled indices -> rlt.PolicyNetworkInput in one launch (ABI 11) -------------------------------
// The continuous-action twin of replay_dqn_batch_kernel (BASELINE C4's sampler: rounds 1-5 took rg_replay_nstep +
// rg_replay_gather + rg_make_policy_input, 72 us in three launches of which 17 us are launch-bound tails): workgroup
// (block of 64 transitions, piece) with piece 0 = state rows, 1 = next_state rows (both normalized on the way, bf16 or fp32
// out), 2 = the n-step reward, not_terminal, exp(log_prob) and the two rescaled action rows.  Every workgroup recomputes the
// n-step bookkeeping of its 64 transitions (a handful of byte loads).  Same arithmetic, operation for operation, as
// replay_nstep_kernel / replay_gather_kernel's normalize-on-gather branch / make_policy_input_kernel: tests compare bit for bit.
struct PolicyBatchArgs {
  rg_policy_replay_view v;
  rg_policy_batch_out o;
};

__global__ void replay_policy_batch_kernel(PolicyBatchArgs a, const int64_t* __restrict__ indices, int batch,
                                           const rg_norm_col* __restrict__ cols, const float* __restrict__ quantiles) {
  __shared__ int64_t s_src[GATHER_ROWS_PER_WG];  // the row this piece reads: idx (state), next idx (next_state)
  __shared__ int64_t s_nxt[GATHER_ROWS_PER_WG];
  __shared__ int s_steps[GATHER_ROWS_PER_WG];
  __shared__ unsigned char s_term[GATHER_ROWS_PER_WG];
  __shared__ __attribute__((aligned(16))) rg_norm_col s_nc[GATHER_MAX_LDS_COLS];
  const rg_policy_replay_view& v = a.v;
  const rg_policy_batch_out& o = a.o;
  const int row0 = blockIdx.x * GATHER_ROWS_PER_WG;
  const int nrows = (batch - row0 < GATHER_ROWS_PER_WG) ? batch - row0 : GATHER_ROWS_PER_WG;
  const int piece = blockIdx.y;  // 0 = state, 1 = next_state, 2 = everything else
  const int F = v.n_features, A = v.action_dim, H = v.update_horizon;
  const int64_t C = v.capacity;
  auto wrap = [&](int64_t t) { return t >= C ? t - C : t; };  // (idx + k) % C: idx < C and k <= H <= C
  if ((int)threadIdx.x < nrows) {
    const int64_t idx = indices[row0 + threadIdx.x];
    int st = H;  // _get_steps (:759-774); with H == 1 the window is one slot whatever it holds
    if (H > 1)
      for (int k = 0; k < H; ++k)
        if (v.terminal[wrap(idx + k)]) {
          st = k + 1;
          break;
        }
    const int64_t nidx = wrap(idx + st);
    s_src[threadIdx.x] = piece == 1 ? nidx : idx;
    s_nxt[threadIdx.x] = nidx;
    s_steps[threadIdx.x] = st;
    if (piece == 2) s_term[threadIdx.x] = v.terminal[wrap(idx + st - 1)] ? 1 : 0;
  }
  const int cpr = F >> 2;  // F % 4 == 0 (checked by the host)
  int* s_op = (int*)s_nc;  // the descriptors as a structure of arrays (replay_dqn_batch_kernel: conflict-free 16-byte reads)
  float* s_p0 = (float*)s_nc + GATHER_MAX_LDS_COLS;
  float* s_p1 = s_p0 + GATHER_MAX_LDS_COLS;
  float* s_p2 = s_p1 + GATHER_MAX_LDS_COLS;
  float* s_p3 = s_p2 + GATHER_MAX_LDS_COLS;
  if (piece < 2 && cols)
    for (int j = threadIdx.x; j < F; j += blockDim.x) {
      const rg_norm_col c = cols[j];
      s_op[j] = c.op; s_p0[j] = c.p0; s_p1[j] = c.p1; s_p2[j] = c.p2; s_p3[j] = c.p3;
    }
  __syncthreads();
  if (piece < 2) {
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    void* dst = piece == 0 ? o.state : o.next_state;
    const int total = nrows * cpr;
    for (int it = threadIdx.x; it < total; it += blockDim.x) {
      const int r = it / cpr, ch = it - r * cpr;
      const f32x4 raw = stream_load((const f32x4*)(v.observation + s_src[r] * F + ch * 4));  // a sampled row is read once
      float w[4] = {raw[0], raw[1], raw[2], raw[3]};
      if (cols) {
        const i32x4 op = *(const i32x4*)(s_op + ch * 4);
        const f32x4 p0 = *(const f32x4*)(s_p0 + ch * 4), p1 = *(const f32x4*)(s_p1 + ch * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rg_norm_col c;
          c.op = op[e]; c.in_col = 0; c.p0 = p0[e]; c.p1 = p1[e]; c.p2 = 0.f; c.p3 = 0.f;
          if (c.op == RG_NORM_BOXCOX || c.op == RG_NORM_CONTINUOUS_ACTION) {
            c.p2 = s_p2[ch * 4 + e];
            c.p3 = s_p3[ch * 4 + e];
          }
          w[e] = normalize_value(c, w[e], 1.f, quantiles);
        }
      }
      const long at = (long)(row0 + r) * F + ch * 4;
      if (o.state_dtype == RG_DT_BF16) {
        uint2 pk;
        pk.x = pack_bf16x2(w[0], w[1]);
        pk.y = pack_bf16x2(w[2], w[3]);
        *(uint2*)((bf16_t*)dst + at) = pk;
      } else {
        *(f32x4*)((float*)dst + at) = f32x4{w[0], w[1], w[2], w[3]};
      }
    }
    return;
  }
  if ((int)threadIdx.x < nrows) {
    const int b = row0 + threadIdx.x, st = s_steps[threadIdx.x];
    const int64_t idx = s_src[threadIdx.x];
    // _reduce_multi_step_reward (:741-747): (reward * decays * masks).sum(dim=1), in that order
    float acc = 0.f;
    for (int k = 0; k < H; ++k) {
      const float m = (k < st) ? 1.f : 0.f;
      acc += (v.reward[wrap(idx + k)] * v.decays[k]) * m;
    }
    o.reward[b] = acc;
    o.not_terminal[b] = 1.0f - (s_term[threadIdx.x] ? 1.f : 0.f);
    if (o.action_probability) o.action_probability[b] = v.log_prob ? expf(v.log_prob[idx]) : 1.f;
  }
  // rescale_actions (training/utils.py:13-29) of the logged action and of the action stored at the next index; every product
  // and sum rounded on its own, as in make_policy_input_kernel
  const int total = nrows * A;
  for (int it = threadIdx.x; it < total; it += blockDim.x) {
    const int r = it / A, d = it - r * A;
    const float lo = v.ranges[d], hi = v.ranges[A + d], tl = v.ranges[2 * A + d], th = v.ranges[3 * A + d];
    const float prev_range = __fsub_rn(hi, lo), new_range = __fsub_rn(th, tl);
    const float av = v.action[s_src[r] * A + d], nv = v.action[s_nxt[r] * A + d];
    const float x = __fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(av, lo), prev_range), new_range), tl);
    const float n = __fadd_rn(__fmul_rn(__fdiv_rn(__fsub_rn(nv, lo), prev_range), new_range), tl);
    const long at = (long)(row0 + r) * A + d;
    o.action[at] = x;
    o.next_action[at] = s_term[r] ? 0.f : n;
  }
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_replay_nstep(const int64_t* indices, const uint8_t* terminal, const float* reward,
                    const float* decays, int64_t capacity, int update_horizon, int batch,
                    int64_t* steps, int64_t* next_indices, uint8_t* out_terminal, float* out_reward,
                    rg_stream_t stream) {
  if (!indices || !terminal || capacity <= 0 || update_horizon <= 0 || batch < 0) return RG_EINVAL;
  if (out_reward && (!reward || !decays)) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  RG_LAUNCH(replay_nstep_kernel, dim3((batch + 255) / 256), dim3(256), (hipStream_t)stream, indices,
            terminal, reward, decays, capacity, update_horizon, batch, steps, next_indices,
            out_terminal, out_reward);
  return (int)hipGetLastError();
}

int rg_replay_gather(const rg_gather_col* cols, int ncols, int64_t capacity, int stack, int batch,
                     rg_stream_t stream) {
  if (!cols || ncols <= 0 || ncols > RG_MAX_GATHER_COLS || capacity <= 0 || stack <= 0 || batch < 0)
    return RG_EINVAL;
  if (batch == 0) return RG_OK;
  GatherTable t;
  for (int i = 0; i < ncols; ++i) {
    t.c[i] = cols[i];
    const int eb = cols[i].elem_bytes;
    if (!cols[i].src || !cols[i].dst || !cols[i].indices || cols[i].row_elems <= 0) return RG_EINVAL;
    if (eb != 1 && eb != 2 && eb != 4 && eb != 8) return RG_EINVAL;
    if (cols[i].norm && (eb != 4 || stack != 1)) return RG_EUNSUPPORTED;
    if (cols[i].norm && cols[i].out_dtype != RG_DT_F32 && cols[i].out_dtype != RG_DT_BF16) return RG_EINVAL;
  }
  for (int i = ncols; i < RG_MAX_GATHER_COLS; ++i) t.c[i] = cols[0];
  if (stack == 1) {
    RG_LAUNCH(replay_gather_kernel,
              dim3((batch + GATHER_ROWS_PER_WG - 1) / GATHER_ROWS_PER_WG, ncols), dim3(256),
              (hipStream_t)stream, t, capacity, batch);
  } else {
    int gx = (batch + 3) / 4;
    if (gx > 4096) gx = 4096;
    RG_LAUNCH(replay_gather_stack_kernel, dim3(gx, ncols), dim3(256), (hipStream_t)stream, t,
              capacity, stack, batch);
  }
  return (int)hipGetLastError();
}

static int replay_dqn_batch_launch(const rg_replay_view* view, const int64_t* indices, int batch, const rg_norm_col* cols,
                                   const float* quantiles, const rg_dqn_batch_out* out, const int64_t* cursor,
                                   double* pre_tick, rg_stream_t stream) {
  if (!view || !out || batch < 0) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  const rg_replay_view& v = *view;
  const rg_dqn_batch_out& o = *out;
  if (!indices || !v.observation || !v.action || !v.reward || !v.terminal || !v.decays || v.capacity <= 0 ||
      v.n_features <= 0 || v.n_actions <= 0 || v.update_horizon <= 0)
    return RG_EINVAL;
  if (!o.state || !o.next_state || !o.action || !o.next_action || !o.reward || !o.not_terminal ||
      !o.possible_next_actions_mask)
    return RG_EINVAL;
  if (o.state_dtype != RG_DT_F32 && o.state_dtype != RG_DT_BF16) return RG_EINVAL;
  if (!cols && o.state_dtype != RG_DT_F32) return RG_EINVAL;  // a raw copy stays fp32
  if ((v.n_features & 3) || v.n_features > GATHER_MAX_LDS_COLS || (((uintptr_t)v.observation) & 15) ||
      (((uintptr_t)o.state) & 15) || (((uintptr_t)o.next_state) & 15))
    return RG_EUNSUPPORTED;  // callers fall back to rg_replay_nstep + rg_replay_gather + rg_make_dqn_input
  ReplayBatchArgs a{v, o};
  RG_LAUNCH(replay_dqn_batch_kernel, dim3((batch + GATHER_ROWS_PER_WG - 1) / GATHER_ROWS_PER_WG, 3), dim3(256),
            (hipStream_t)stream, a, indices, batch, cols, quantiles, cursor, pre_tick);
  return (int)hipGetLastError();
}

int rg_replay_dqn_batch(const rg_replay_view* view, const int64_t* indices, int batch, const rg_norm_col* cols,
                        const float* quantiles, const rg_dqn_batch_out* out, rg_stream_t stream) {
  return replay_dqn_batch_launch(view, indices, batch, cols, quantiles, out, nullptr, nullptr, stream);
}

int rg_replay_dqn_batch_pooled(const rg_replay_view* view, const int64_t* index_pool, const int64_t* cursor,
                               double* pre_tick_sched, int batch, const rg_norm_col* cols, const float* quantiles,
                               const rg_dqn_batch_out* out, rg_stream_t stream) {
  if (!cursor) return RG_EINVAL;
  return replay_dqn_batch_launch(view, index_pool, batch, cols, quantiles, out, cursor, pre_tick_sched, stream);
}

int rg_replay_policy_batch(const rg_policy_replay_view* view, const int64_t* indices, int batch, const rg_norm_col* cols,
                           const float* quantiles, const rg_policy_batch_out* out, rg_stream_t stream) {
  if (!view || !out || batch < 0) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  const rg_policy_replay_view& v = *view;
  const rg_policy_batch_out& o = *out;
  if (!indices || !v.observation || !v.action || !v.reward || !v.terminal || !v.decays || !v.ranges || v.capacity <= 0 ||
      v.n_features <= 0 || v.action_dim <= 0 || v.update_horizon <= 0)
    return RG_EINVAL;
  if (!o.state || !o.next_state || !o.action || !o.next_action || !o.reward || !o.not_terminal) return RG_EINVAL;
  if (o.state_dtype != RG_DT_F32 && o.state_dtype != RG_DT_BF16) return RG_EINVAL;
  if (!cols && o.state_dtype != RG_DT_F32) return RG_EINVAL;  // a raw copy stays fp32
  if ((v.n_features & 3) || v.n_features > GATHER_MAX_LDS_COLS || (((uintptr_t)v.observation) & 15) ||
      (((uintptr_t)o.state) & 15) || (((uintptr_t)o.next_state) & 15))
    return RG_EUNSUPPORTED;  // callers fall back to rg_replay_nstep + rg_replay_gather + rg_make_policy_input
  PolicyBatchArgs a{v, o};
  RG_LAUNCH(replay_policy_batch_kernel, dim3((batch + GATHER_ROWS_PER_WG - 1) / GATHER_ROWS_PER_WG, 3), dim3(256),
            (hipStream_t)stream, a, indices, batch, cols, quantiles);
  return (int)hipGetLastError();
}

int rg_make_dqn_input(const int64_t* action, const int64_t* next_action, const uint8_t* terminal,
                      const float* log_prob, int batch, int num_actions, float* action_onehot,
                      float* next_action_onehot, float* not_terminal, float* action_probability,
                      rg_stream_t stream) {
  if (!action || !next_action || !terminal || !action_onehot || !next_action_onehot || !not_terminal ||
      batch < 0 || num_actions <= 0 || (action_probability && !log_prob))
    return RG_EINVAL;
  if (batch == 0) return RG_OK;
  long blocks = ((long)batch * num_actions + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  RG_LAUNCH(make_dqn_input_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, action,
            next_action, terminal, log_prob, batch, num_actions, action_onehot, next_action_onehot,
            not_terminal, action_probability);
  return (int)hipGetLastError();
}

int rg_ragged_offsets(const int32_t* lens, const int64_t* indices, int batch, int32_t* offsets, int32_t* total,
                      rg_stream_t stream) {
  if (!total || batch < 0 || (batch > 0 && (!lens || !indices || !offsets))) return RG_EINVAL;
  RG_LAUNCH(ragged_offsets_kernel, dim3(1), dim3(1024), (hipStream_t)stream, lens, indices, batch, offsets, total);
  return (int)hipGetLastError();
}

int rg_ragged_copy(const int64_t* ids, const float* scores, int width, const int32_t* lens, const int64_t* indices,
                   const int32_t* offsets, int batch, int64_t* ids_out, float* scores_out, rg_stream_t stream) {
  if (batch == 0) return RG_OK;
  if (!ids || !lens || !indices || !offsets || !ids_out || width <= 0 || batch < 0 || (scores && !scores_out)) return RG_EINVAL;
  RG_LAUNCH(ragged_copy_kernel, dim3((batch + 3) / 4), dim3(256), (hipStream_t)stream, ids, scores, width, lens, indices,
            offsets, batch, ids_out, scores_out);
  return (int)hipGetLastError();
}

int rg_make_policy_input(const float* action, int64_t lda, const float* next_action, int64_t ldna, const uint8_t* terminal,
                         const float* log_prob, const float* ranges, int batch, int action_dim, float* action_out,
                         float* next_action_out, float* not_terminal, float* action_probability, rg_stream_t stream) {
  if (batch == 0) return RG_OK;  // (an empty batch has no buffers to check)
  if (!action || !next_action || !terminal || !ranges || !action_out || !next_action_out || !not_terminal ||
      (action_probability && !log_prob) || batch < 0 || action_dim <= 0)
    return RG_EINVAL;
  long blocks = ((long)batch * action_dim + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  RG_LAUNCH(make_policy_input_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, action, (long)lda,
            next_action, (long)ldna, terminal, log_prob, ranges, batch, action_dim, action_out, next_action_out,
            not_terminal, action_probability);
  return (int)hipGetLastError();
}

}  // extern "C"

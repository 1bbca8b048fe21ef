// table.hip — training batches straight from an offline table resident in HBM.
// The table is the post-timeline dataset with the column schema of select_relevant_columns
// (reagent/data/oss_data_fetcher.py:293-336), one device array per column.  One launch turns a batch
// of row indices into the fields of rlt.DiscreteDqnInput, i.e. the row fetch of the data loader plus
// DiscreteDqnBatchPreprocessor.forward (reagent/preprocessing/batch_preprocessor.py:35-66):
//   Preprocessor.forward on state / next_state (presence multiply, per-type ops, +-11.513 clamp;
//   preprocessor.py:115-170) while the row streams through, one-hot action / next_action
//   (next_action == num_actions means "none": an all-zero row), not_terminal = max of
//   possible_next_actions_mask, and the pass-through columns.
// HBM-bound byte work: per transition 2 F (4 + 1) bytes read, 2 F' (4 or 2) written, ~60 B of
// scalars and 4 A of masks.  The normalised fp32 matrix never exists when bf16 output is selected.
#include "rg_norm.h"

namespace rg {

constexpr int TABLE_ROWS_PER_WG = 64;
constexpr int TABLE_THREADS = 256;

struct TableArgs {
  rg_dqn_table t;
  rg_dqn_batch_out o;
};

__global__ void table_dqn_batch_kernel(TableArgs a, const int64_t* __restrict__ indices, int batch,
                                       const rg_norm_col* __restrict__ cols, int n_out,
                                       const float* __restrict__ quantiles) {
  RG_DYN_LDS(smem);
  int64_t* s_idx = (int64_t*)smem;                                   // [TABLE_ROWS_PER_WG]
  rg_norm_col* s_cols = (rg_norm_col*)(smem + TABLE_ROWS_PER_WG * 8);  // [n_out] (feature pieces only)
  const int row0 = blockIdx.x * TABLE_ROWS_PER_WG;
  const int nrows = (batch - row0 < TABLE_ROWS_PER_WG) ? batch - row0 : TABLE_ROWS_PER_WG;
  const int piece = blockIdx.y;  // 0 = state, 1 = next_state, 2 = everything else
  int* s_act = (int*)(smem + TABLE_ROWS_PER_WG * 8 + (size_t)n_out * sizeof(rg_norm_col));  // [2][TABLE_ROWS_PER_WG]
  if ((int)threadIdx.x < nrows) {
    const int64_t i = indices[row0 + threadIdx.x];
    s_idx[threadIdx.x] = i;
    if (piece == 2) {  // fetched once per row, not once per (row, action)
      s_act[threadIdx.x] = (int)a.t.action[i];
      s_act[TABLE_ROWS_PER_WG + threadIdx.x] = (int)a.t.next_action[i];
    }
  }
  if (piece < 2)
    for (int j = threadIdx.x; j < n_out; j += TABLE_THREADS) s_cols[j] = cols[j];
  __syncthreads();
  const rg_dqn_table& t = a.t;
  const rg_dqn_batch_out& o = a.o;
  const int F = t.n_features, A = t.n_actions;
  if (piece < 2) {
    const float* x = piece == 0 ? t.state_features : t.next_state_features;
    const uint8_t* pres = piece == 0 ? t.state_features_presence : t.next_state_features_presence;
    void* dst = piece == 0 ? o.state : o.next_state;
    if ((n_out & 3) == 0 && (F & 3) == 0 && ((((uintptr_t)x) & 15) == 0) && ((((uintptr_t)dst) & 15) == 0) &&
        (!pres || ((((uintptr_t)pres) & 3) == 0))) {
      // four output columns per lane: where they read four consecutive, 4-aligned input features (every
      // type but ENUM keeps the 1:1 order) the lane moves 16 B of values + 4 B of presence per request
      const int cpr = n_out >> 2, total = nrows * cpr;
      for (int it = threadIdx.x; it < total; it += TABLE_THREADS) {
        const int r = it / cpr, ch = it - r * cpr;
        rg_norm_col d[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = s_cols[ch * 4 + e];
        const long base = s_idx[r] * F;
        float raw[4], p[4];
        if ((d[0].in_col & 3) == 0 && d[1].in_col == d[0].in_col + 1 && d[2].in_col == d[0].in_col + 2 &&
            d[3].in_col == d[0].in_col + 3) {
          const f32x4 t = *(const f32x4*)(x + base + d[0].in_col);
          const unsigned pb = pres ? *(const unsigned*)(pres + base + d[0].in_col) : 0x01010101u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            raw[e] = t[e];
            p[e] = (float)((pb >> (8 * e)) & 0xffu);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            raw[e] = x[base + d[e].in_col];
            p[e] = pres ? (float)pres[base + d[e].in_col] : 1.f;
          }
        }
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = normalize_value(d[e], raw[e], p[e], quantiles);
        const long at = (long)(row0 + r) * n_out + ch * 4;
        if (o.state_dtype == RG_DT_BF16) {
          uint2 w;
          w.x = pack_bf16x2(v[0], v[1]);
          w.y = pack_bf16x2(v[2], v[3]);
          *(uint2*)((bf16_t*)dst + at) = w;
        } else {
          *(f32x4*)((float*)dst + at) = f32x4{v[0], v[1], v[2], v[3]};
        }
      }
      return;
    }
    const int total = nrows * n_out;
    for (int it = threadIdx.x; it < total; it += TABLE_THREADS) {
      const int r = it / n_out, j = it - r * n_out;
      const rg_norm_col c = s_cols[j];
      const long src = s_idx[r] * F + c.in_col;
      const float p = pres ? (float)pres[src] : 1.f;
      const float v = normalize_value(c, x[src], p, quantiles);
      const long at = (long)(row0 + r) * n_out + j;
      if (o.state_dtype == RG_DT_BF16)
        ((bf16_t*)dst)[at] = f32_to_bf16(v);
      else
        ((float*)dst)[at] = v;
    }
    return;
  }
  if ((int)threadIdx.x < nrows) {
    const int b = row0 + threadIdx.x;
    const int64_t i = s_idx[threadIdx.x];
    o.reward[b] = t.reward[i];
    if (o.time_diff) o.time_diff[b] = t.time_diff ? (float)t.time_diff[i] : 1.f;
    if (o.step) o.step[b] = t.step ? (float)t.step[i] : 1.f;
    if (o.action_probability) o.action_probability[b] = t.action_probability ? t.action_probability[i] : 1.f;
    if (o.mdp_id) o.mdp_id[b] = t.mdp_id ? t.mdp_id[i] : 0;
    if (o.sequence_number) o.sequence_number[b] = t.sequence_number ? t.sequence_number[i] : 0;
    // not terminal iff at least one possible next action (batch_preprocessor.py:43-44)
    uint8_t any = 0;
    for (int k = 0; k < A; ++k) any = any > t.possible_next_actions_mask[i * A + k] ? any : t.possible_next_actions_mask[i * A + k];
    o.not_terminal[b] = (float)any;
  }
  const int total = nrows * A;
  for (int it = threadIdx.x; it < total; it += TABLE_THREADS) {
    const int r = it / A, k = it - r * A;
    const int64_t i = s_idx[r];
    const long at = (long)(row0 + r) * A + k;
    o.action[at] = s_act[r] == k ? 1.f : 0.f;                           // F.one_hot(action, A)
    o.next_action[at] = s_act[TABLE_ROWS_PER_WG + r] == k ? 1.f : 0.f;  // F.one_hot(next_action, A + 1)[:, :A]
    if (o.possible_actions_mask)
      o.possible_actions_mask[at] = t.possible_actions_mask ? (float)t.possible_actions_mask[i * A + k] : 1.f;
    o.possible_next_actions_mask[at] = (float)t.possible_next_actions_mask[i * A + k];
  }
}

// F.one_hot raises on an index outside [0, classes): report it instead of emitting a silent zero row
__global__ void table_check_actions_kernel(const int64_t* __restrict__ action, const int64_t* __restrict__ next_action,
                                           const int64_t* __restrict__ indices, int batch, int64_t n_rows, int A,
                                           int* __restrict__ bad) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int64_t i = indices[b];
  if (i < 0 || i >= n_rows) {
    atomicMax(bad, 2);
    return;
  }
  if (action[i] < 0 || action[i] >= A || next_action[i] < 0 || next_action[i] > A) atomicMax(bad, 1);
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_table_dqn_batch(const rg_dqn_table* table, const int64_t* indices, int batch, const rg_norm_col* cols,
                       int n_out, const float* quantiles, const rg_dqn_batch_out* out, rg_stream_t stream) {
  if (!table || !out || batch < 0 || n_out <= 0) return RG_EINVAL;
  if (batch == 0) return RG_OK;  // empty outputs have no storage to point at
  if (!indices || !cols) return RG_EINVAL;
  const rg_dqn_table& t = *table;
  const rg_dqn_batch_out& o = *out;
  if (!t.state_features || !t.next_state_features || !t.action || !t.next_action || !t.reward ||
      !t.possible_next_actions_mask || t.n_rows <= 0 || t.n_features <= 0 || t.n_actions <= 0)
    return RG_EINVAL;
  if (!o.state || !o.next_state || !o.action || !o.next_action || !o.reward || !o.not_terminal ||
      !o.possible_next_actions_mask)
    return RG_EINVAL;
  if (o.state_dtype != RG_DT_F32 && o.state_dtype != RG_DT_BF16) return RG_EINVAL;
  const size_t lds = (size_t)TABLE_ROWS_PER_WG * 8 + (size_t)n_out * sizeof(rg_norm_col) + 2 * TABLE_ROWS_PER_WG * sizeof(int);
  if (lds > 64 * 1024) return RG_EUNSUPPORTED;  // > 2700 output features
  TableArgs a{t, o};
  const dim3 grid((batch + TABLE_ROWS_PER_WG - 1) / TABLE_ROWS_PER_WG, 3);
  RG_LAUNCH_DYN(table_dqn_batch_kernel, grid, dim3(TABLE_THREADS), lds, (hipStream_t)stream, a, indices, batch, cols,
                n_out, quantiles);
  return (int)hipGetLastError();
}

int rg_table_check_actions(const rg_dqn_table* table, const int64_t* indices, int batch, int* bad_flag,
                           rg_stream_t stream) {
  if (!table || !indices || !bad_flag || batch < 0 || !table->action || !table->next_action) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  RG_LAUNCH(table_check_actions_kernel, dim3((batch + 255) / 256), dim3(256), (hipStream_t)stream, table->action,
            table->next_action, indices, batch, table->n_rows, table->n_actions, bad_flag);
  return (int)hipGetLastError();
}

}  // extern "C"

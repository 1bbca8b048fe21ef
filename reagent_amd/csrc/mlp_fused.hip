// mlp_fused.hip — the bf16 throughput path of a FullyConnected stack: whole-network forward and
// backward kernels that keep a 128-row activation tile resident in LDS across all layers, plus
// the weight-gradient kernel that consumes activations saved in MFMA-fragment order.
//
// Why (measured on MI355X, profiles/r01_run1): one GEMM launch per layer is bound by the HBM
// round trip of the [batch, 512] activation (67 MB bf16) between layers and by 2-byte epilogue
// stores, not by MFMA.  Here
//   * a workgroup (8 waves, 512 threads) owns 128 rows; the activation tile lives in LDS
//     (128 x (width+8) bf16 = 133 KB of the CU's 160 KB) and is rewritten in place layer by layer;
//   * each wave owns 32*TN output columns of a hidden layer (all 128 rows): its weight operand is
//     private, so weights are pre-staged in HBM in B-fragment order and stream L2 -> VGPR as
//     perfectly coalesced 1 KB wave loads, no LDS and no barrier on the weight path;
//   * what backward needs is stored straight from the accumulators in "C-fragment order"
//     (lane = column, 8 rows per lane per half-tile; 1 KB coalesced stores).  Both wgrad operands
//     (dZ and X) are produced in that order, and because an MFMA reduction may visit the
//     reduction index in any order as long as A and B agree, wgrad consumes them as MFMA A/B
//     fragments directly: no transposes anywhere, no 2-byte stores.
//
// Replaces: FullyConnectedNetwork.forward (reagent/models/fully_connected_network.py:157-163)
// and its autograd backward for stacks whose hidden layers share one width in {256, 512}.

#include "rg_mlp_frag.h"
#include "rg_reduce.h"
#include <cstdio>
// workgroups per layer of the grouped weight-gradient launch (splits = this / tiles).
// Round 3, same-box A/B in the C2 step (wgrad + reduce, us): this uniform 128 per layer 132-134; workgroups shared out in
// proportion to the operand bytes a layer's tiles stream — 256 in all (one round on the chip, 52 MB of partials
// instead of 84) 143-144, 384 in all 157, 512 in all 165; the tiles that share an operand half walking their split's
// blocks 2 / 4 / 8 blocks apart (so that the second reader finds the line in L2 instead of joining the in-flight miss)
// 143-145 / 162-164 / 172-173: the lockstep second reader is the cheap one.  Not kept.
#ifndef RG_OUT_LDS
#define RG_OUT_LDS 1  // a thin output layer's weight fragments resident in LDS (tile_kloop_ldsb)
#endif
#ifndef RG_BWD_SIGNS_EARLY
#define RG_BWD_SIGNS_EARLY 1  // backward: the first step's sign planes are requested before the dout tile (0 = before its main loop)
#endif
#ifndef RG_GROUPED_STAGE_OUT
#define RG_GROUPED_STAGE_OUT 1  // a wide grouped output leaves through the (dead) activation tile as whole rows
#endif
#ifndef RG_GROUPED_RING
#define RG_GROUPED_RING 8   // weight chunks in flight per wave in that path's K loop
#endif
#ifndef RG_WGRAD_TARGET
#define RG_WGRAD_TARGET 128
#endif
#ifndef RG_FWD_SWAP
#define RG_FWD_SWAP 0  // non-saving forwards on transposed accumulator tiles (mlp_fwd_swap_kernel).  Round 4, same box: the two
// non-saving forwards of a C2 step 169 -> 177 us, C4's 100 -> 106 — the 8-byte LDS writes collide two ways on the tile's row
// pitch and the 16 bias values per tile are re-requested per row tile; bit-identical, slower: not the default.
#endif
#ifndef RG_WGRAD_BF16_PART
#define RG_WGRAD_BF16_PART 1  // bf16 stack launch: split partials as bf16 tiles in accumulator order (WgradFragArgs.part_mode)
#endif
#ifndef RG_WGRAD_PART_NT
#define RG_WGRAD_PART_NT 0   // bf16 partial tiles: bit 0 = non-temporal stores (wgrad), bit 1 = non-temporal loads (reduce).  Round 5,
#endif                       // same box: 0 / 1 / 2 / 3 all within 1 us of each other; plain stores and loads are the default
#ifndef RG_WGRAD_UNEVEN
#define RG_WGRAD_UNEVEN 125  // stack launch: uneven splits of the multi-tile layers (rg_mlp_wgrad_fused); value = cost of a single-tile
#endif                       // workgroup's block in percent of a multi-tile one's; 0 = every split of a layer the same length
#ifndef RG_REDUCE_FLY
#define RG_REDUCE_FLY 16      // split-reduce: fp32 partials requested per thread before the first add (multiple of 4; 4 = rounds 1-4)
#endif
#ifndef RG_REDUCE_FLY_BF16
#define RG_REDUCE_FLY_BF16 8  // the same for the 32-byte bf16 tile records (even)
#endif
#ifndef RG_WGRAD_PIPE
#define RG_WGRAD_PIPE 1  // weight gradient: LDS fragment reads one half ahead of the MFMAs (wgrad_shape_core)
#endif
#ifndef RG_WGRAD_PAIR
#define RG_WGRAD_PAIR 0  // weight gradient, 256 x 256 tiles: two 32-row blocks per barrier (measured +5 %: not the default)
#endif

namespace rg {

// NW waves per workgroup, each owning 32*TN columns of a hidden layer (hidden width = 32*TN*NW).
// NW = 4 (one wave per SIMD, up to 512 registers each): 16 accumulator tiles per wave and a weight
// ring deep enough to cover the L2 latency from a single wave.  NW = 8: two waves per SIMD.
template <int NW> struct MlpCfg {
  static constexpr int THREADS = NW * 64;
#ifdef RG_FUSED_RING
  static constexpr int RING = RG_FUSED_RING;
#else
  static constexpr int RING = NW == 4 ? 8 : 2;
#endif
};

// Grouped forward, the LAST segment of a tile (mlp_fwd_fused_body): wave w sums column tile w of the group's [N, K] layer for all four
// row tiles in the software-pipelined main loop (one weight stream per wave, ring of 8 chunks), the 128 x N outputs are staged
// in the activation tile — dead once every wave has left its K loop — as 128 x P floats and leave as whole rows, a wave per row.
// (Its own function, with its own lane / wave derivations: written out inside the segment loop the 512-wide kernel spilled 45
// registers; called through a non-inlined function 70.)
template <int NW>
__device__ __forceinline__ void grouped_whole_tile_out(bf16_t* act, int pitch, int KC, const bf16_t* wf_out, const float* b_out,
                                                                 int N, int NTo, int out_act, int lo, int hi, int row_base,
                                                                 const int* scatter, int batch, float* out32, long ldo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 31, lg = lane >> 5;
  const int P = NTo * 32 + 4, np = N >> 2;
  f32x16 acc4[4][1];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc4[tm][0][r] = 0.f;
  const int col = wave * 32 + lr;
  float b = 0.f;
  if (wave < NTo) {
    if (b_out && col < N) b = b_out[col];  // (requested before the K loop)
    wide_mainloop<1, RG_GROUPED_RING>(act, pitch, KC, wf_out + (long)wave * KC * 512, 0, acc4, lane, k_rotation(blockIdx.x, wave, KC));
  }
  RG_STAMP(16);
  __syncthreads();  // every wave is done reading the layer input
  RG_STAMP(17);
  float* stage = (float*)act;
  if (wave < NTo) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc4[tm][0][r] + b;
      act_apply_n(v, out_act);
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * P + col] = v[r];
    }
  }
  __syncthreads();
  RG_STAMP(18);
  for (int rel = lo + wave; rel < hi; rel += NW) {  // a wave per row: np <= 64 16-byte pieces
    int row = row_base + rel;
    if (scatter) row = scatter[row];  // back to batch order; padding rows (-1) are dropped
    if (row >= 0 && (scatter || row < batch) && lane < np)
      stream_store(*(const f32x4*)(stage + rel * P + lane * 4), (f32x4*)(out32 + (long)row * ldo + lane * 4));
  }
  RG_STAMP(19);
}

// PITCH (LDS row pitch in elements) is a template constant so that every LDS offset of the
// epilogue stores folds into an instruction immediate instead of a vector add per store.
// (Persistent workgroups — one per CU walking tile, tile + 256 with the next tile's input rows requested
// during the output layer — were measured: 78.8-79.3 us against 80.1 us per launch in
// profiles/microbench/fwd_phases, but 92 against 85 us inside the training step, where the static
// tile assignment loses the dispatcher's load balancing; the kernel stays one tile per workgroup.)
// GROUPED: the launch of a stack whose output layer takes per-tile weights (rg_mlp_desc.tile_key, qr_grouped.hip) is its
// own instantiation — the ordinary kernel does not carry its code paths (or their registers).
// SWAP (round 4): the non-saving launch of the plain kernel computes its hidden layers with transposed accumulator tiles
// (rg_mlp_frag.h: wide_mainloop<.., SWAP>, fwd_hidden_pack_swapped) — same values bit for bit, a cheaper epilogue.
template <int TN, int NW, int PITCH, bool GROUPED, bool SWAP = false>
__device__ __forceinline__ void mlp_fwd_fused_body(const MlpArgs& a) {
  constexpr int THREADS = MlpCfg<NW>::THREADS, RING = MlpCfg<NW>::RING;
  RG_DYN_LDS(smem);
  bf16_t* act = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  const int tile = GROUPED ? grouped_tile(blockIdx.x, (a.batch + FB_BM - 1) / FB_BM) : (int)blockIdx.x;
  if (GROUPED && tile * FB_BM >= round_up(a.batch, FB_BM)) return;  // padding blocks (workgroup-uniform)
  const int row_base = tile * FB_BM;
  constexpr int pitch = PITCH;
  const int k0p = round_up(a.dims[0], 32);
  RG_STAMP(0);
  if (a.rowmap) {  // grouped space: rows gathered through the map
    if (a.x_is_f32)
      load_tile_rows_mapped<float, THREADS>(act, pitch, (const float*)a.x, a.ldx, a.rowmap, row_base, a.dims[0], k0p, tid);
    else
      load_tile_rows_mapped<bf16_t, THREADS>(act, pitch, (const bf16_t*)a.x, a.ldx, a.rowmap, row_base, a.dims[0], k0p, tid);
  } else if (a.x2) {  // two panels (state | action): columns [0, x_split) from x, the rest from x2
    const int n2 = a.dims[0] - a.x_split;
    if (a.x_is_f32)  // each panel in its own element type (bf16 state rows from the sampler next to fp32 actions)
      load_tile_to_lds<float, THREADS>(act, pitch, (const float*)a.x, a.ldx, row_base, a.batch, a.x_split, a.x_split, tid);
    else
      load_tile_to_lds<bf16_t, THREADS>(act, pitch, (const bf16_t*)a.x, a.ldx, row_base, a.batch, a.x_split, a.x_split, tid);
    if (a.x2_is_f32)
      load_tile_to_lds<float, THREADS>(act + a.x_split, pitch, (const float*)a.x2, a.ldx2, row_base, a.batch, n2, k0p - a.x_split, tid);
    else
      load_tile_to_lds<bf16_t, THREADS>(act + a.x_split, pitch, (const bf16_t*)a.x2, a.ldx2, row_base, a.batch, n2, k0p - a.x_split, tid);
  } else if (a.x_is_f32)
    load_tile_to_lds<float, THREADS>(act, pitch, (const float*)a.x, a.ldx, row_base, a.batch, a.dims[0], k0p, tid);
  else
    load_tile_to_lds<bf16_t, THREADS>(act, pitch, (const bf16_t*)a.x, a.ldx, row_base, a.batch, a.dims[0], k0p, tid);
  __syncthreads();
  RG_STAMP(1);
  if (a.save == 1 && a.act_frag[0]) emit_frags_from_lds(act, pitch, k0p / 32, a.act_frag[0], tile * 4, wave, NW, lane);
  // a thin output layer's weights travel to LDS while the hidden layers compute (rg_mlp_frag.h: out_lds_prefetch)
  const bool out_lds = !GROUPED && a.out_lds;
  char* wo = (char*)(act + FB_BM * pitch);
  if (out_lds) out_lds_prefetch(a.wfrag[a.n_layers - 1], (a.dims[a.n_layers - 1] + 15) / 16, wo, wave, NW, lane);

  for (int l = 0; l < a.n_layers; ++l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    const int KC = (K + 15) / 16;
    if (l < a.n_layers - 1) {  // hidden layer, N == 32 * TN * NW
      f32x16 acc[4][TN];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
      const long nt_stride = (long)KC * 512;
      wide_mainloop<TN, RING, SWAP>(act, pitch, KC, a.wfrag[l] + (long)(wave * TN) * nt_stride, nt_stride, acc, lane,
                                    k_rotation(blockIdx.x, wave, KC), wave / (NW / 2));
      RG_STAMP(2 + 4 * l);
      unsigned PK[4][TN][8];
      if constexpr (SWAP) {
        RG_DISPATCH_ACT(a.acts[l], (fwd_hidden_pack_swapped<TN, A_>(acc, a.bias[l], wave, lane, PK)));
      } else {
        unsigned* sign_dst = a.save ? a.act_sign[l + 1] : nullptr;  // plane base; the lane offset is applied at the store
        RG_DISPATCH_ACT(a.acts[l], (fwd_hidden_pack<TN, A_>(acc, a.bias[l], fwd_save_dst(a, l),
                                                            sign_dst, N / 32, tile * 4, wave, lane, PK)));
      }
      RG_STAMP(3 + 4 * l);
      __syncthreads();  // every wave is done reading the layer input
      RG_STAMP(4 + 4 * l);
      if constexpr (SWAP) store_packed_tiles_swapped<TN>(act, pitch, PK, wave, lane);
      else store_packed_tiles<TN>(act, pitch, PK, wave, lane);
      if (out_lds) RG_WAIT_VMCNT(0);  // this wave's share of the output layer's weights has landed (long ago) ...
      __syncthreads();                // ... and after the barrier every wave's has
      RG_STAMP(5 + 4 * l);
    } else {  // output layer: 32x32 tiles spread over the waves, fp32 result to HBM
      const int NTo = (N + 31) / 32;
      const int out_act = a.acts[l];
      const bool stream_out = N > 64;
      // grouped output layer: the rows of the tile are cut into segments, one per group (rg_mlp_frag.h: next_segment; a
      // tile inside one group's range is one segment), and each segment's group selects the weight / bias slice; rows that
      // belong to no group have no output.  The plain layer is the one segment [0, 128) of "group 0".
      int seg_g = GROUPED ? a.tile_key[tile] : 0;
      RowSegment seg{0, 0, FB_BM};
      // (grouped: the lane-derived addresses of the segment loop are worked out HERE — hoisted out of that loop and above the
      // hidden layers' main loops they cost the 512-wide kernel 30 spilled registers)
      const int o_lane = GROUPED ? opaque(lane) : lane, o_tid = GROUPED ? opaque(tid) : tid;
      while (!GROUPED || next_segment(a.row_begin, a.n_groups, row_base, FB_BM, seg_g, seg)) {
        const int lane = o_lane, tid = o_tid, lr = lane & 31, lg = lane >> 5;
        const int grp = seg.grp;
        const bf16_t* wf_out = a.wfrag[l] + (GROUPED ? (long)grp * a.group_stride : 0);
        const float* b_out = a.bias[l] ? a.bias[l] + (GROUPED ? (long)grp * N : 0) : nullptr;
        const int tm0 = GROUPED ? seg.lo >> 5 : 0, tm1 = GROUPED ? (seg.hi + 31) >> 5 : 4;  // the segment's 32-row tiles
        // the bias of column tile 0, requested BEFORE the K loop: loaded at the point of use it put one more L2 round trip
        // (~2000 cycles) at the very end of every workgroup whose output is one column tile (16 Q-values, a critic's scalar)
        const float b_tile0 = (b_out && lr < N) ? b_out[lr] : 0.f;
        auto store_tile = [&](const f32x16& acc, int tm, int nt) {
          const int col = nt * 32 + lr;
          if (col < N) {
            const float b = nt == 0 ? b_tile0 : (b_out ? b_out[col] : 0.f);
            float ov[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) ov[r] = acc[r] + b;
            act_apply_n(ov, out_act);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rel = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
              if (GROUPED && (rel < seg.lo || rel >= seg.hi)) continue;  // another segment's row
              int row = row_base + rel;
              if (a.out_scatter) row = a.rowmap[row];  // back to batch order; padding rows (-1) are dropped
              if (row >= 0 && (a.out_scatter || row < a.batch)) {
                const float o = ov[r];
                // a wide output (QR-DQN's 200 quantiles per row: 54 MB per launch, read once by the loss head) streams
                // past the caches (same-box C3 step -1 %); a thin one is a few MB and its reader is next
                if (stream_out) stream_store(o, a.out32 + (long)row * a.ldo + col);
                else a.out32[(long)row * a.ldo + col] = o;
              }
            }
          }
        };
        // (A pipelined variant for 4..8 output column tiles — wave w running wide_mainloop<1> on column tile w, for all four
        // row tiles with a ring of 8 / 16 chunks — was measured on the C3 grouped forward in rounds 2 and 3: 171 us against
        // 169 us for this loop, C3 step 1.119 / 1.121 against 1.101 ms; its burst of output stores at the end costs more than
        // the re-read fragments.  Ablations of the 189 us target forward: head MFMAs removed -43 us, output stores -20..-28.)
        if (!GROUPED && NTo == 1 && NW == 8 && KC >= 8) {
          // one column tile (<= 32 outputs, e.g. 16 Q-values): four 32x32 tiles for eight waves.  The loop is a chain
          // of L2 round trips (7 % of a workgroup's life with four waves idle), so two waves share a tile, each
          // summing half of K; the upper four hand their accumulators over through the activation tile, dead by then.
          const int tm = wave & 3, half = wave >> 2, kc_mid = (KC / 2 + 3) / 4 * 4;
          f32x16 acc = out_lds ? tile_kloop_ldsb(act, pitch, wo, tm, lane, half ? kc_mid : 0, half ? KC : kc_mid)
                               : tile_kloop(act, pitch, KC, wf_out, tm, 0, lane, half ? kc_mid : 0, half ? KC : kc_mid);
          RG_STAMP(16);
          __syncthreads();  // every wave is done reading the layer input
          RG_STAMP(17);
          // RG_OUT_ROWSTORE (round 5): the 128 x N outputs leave as whole rows, 16 bytes per lane, through the (dead) activation
          // tile — 8 wave stores of 1 KB for 16 Q-values instead of 64 four-byte ones from the accumulators that each touch
          // 32 half-lines (fwd_phases: "sum + stores" 4.6k of a workgroup's 83k ticks -> 3.1k, the launch 84.6 -> 79.2 us).
          // Both halves of K put their partial sums into the staging area ([half][row][N] floats) and the row-store pass adds
          // them — (lower + upper) + bias, the order of the accumulator hand-off below — one barrier instead of two.
          // (everything the row store needs is worked out HERE, from an opaque copy of the lane: hoisted above the hidden layers'
          // main loops it cost the 512-wide kernel 5 spilled registers)
          const int o_ln = opaque(lane), o_lr = o_ln & 31, o_lg = o_ln >> 5;
          // two forms: rows of whole 16-byte pieces (N % 4 == 0), or — a dense output (ldo == N: e.g. a critic's single column)
          // and a full tile — the tile's 128 x N block as ONE contiguous run, whatever N is
          const bool aligned16 = (reinterpret_cast<uintptr_t>(a.out32) & 15) == 0;
          const bool dense_run = a.ldo == N && row_base + FB_BM <= a.batch;
          const bool rowstore = RG_OUT_ROWSTORE && !a.out_scatter && aligned16 && (dense_run || ((N & 3) == 0 && (a.ldo & 3) == 0));
          if (rowstore) {  // (workgroup-uniform)
            float* outs = (float*)act + half * (FB_BM * 32);
            float* bias_s = (float*)act + 2 * (FB_BM * 32);  // the bias (requested before the K loop) travels through LDS too
            if (o_lr < N) {
#pragma unroll
              for (int r = 0; r < 16; ++r) outs[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * o_lg) * N + o_lr] = acc[r];
              if (wave == 0 && o_lg == 0) bias_s[o_lr] = b_tile0;
            }
            __syncthreads();
            RG_STAMP(18);
            RG_STAMP(19);
            const float* lo_ = (const float*)act;
            const float* hi_ = lo_ + FB_BM * 32;
            if (dense_run) {
              float* dst = a.out32 + (long)row_base * N;
              for (int it = wave * 64 + o_ln; it < (FB_BM * N) >> 2; it += THREADS) {
                const f32x4 l4 = *(const f32x4*)(lo_ + it * 4), h4 = *(const f32x4*)(hi_ + it * 4);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (l4[e] + h4[e]) + bias_s[(it * 4 + e) % N];
                act_apply_n(o, out_act);
                *(f32x4*)(dst + it * 4) = f32x4{o[0], o[1], o[2], o[3]};
              }
            }
            const int np = N >> 2;  // 16-byte pieces per row
            for (int it = wave * 64 + o_ln; !dense_run && it < FB_BM * np; it += THREADS) {  // (tid, rebuilt from the live lane: tid itself is dead by now)
              const int rel = it / np, c4 = it - rel * np;
              const int row = row_base + rel;
              if (row < a.batch) {
                const f32x4 l4 = *(const f32x4*)(lo_ + rel * N + c4 * 4), h4 = *(const f32x4*)(hi_ + rel * N + c4 * 4);
                const f32x4 b4 = *(const f32x4*)(bias_s + c4 * 4);
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (l4[e] + h4[e]) + b4[e];
                act_apply_n(o, out_act);
                *(f32x4*)(a.out32 + (long)row * a.ldo + c4 * 4) = f32x4{o[0], o[1], o[2], o[3]};
              }
            }
          } else {
            float* hand = (float*)act + (tm * 64 + lane) * 16;
            if (half) {
#pragma unroll
              for (int r = 0; r < 16; r += 4) *(f32x4*)(hand + r) = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
            }
            __syncthreads();
            RG_STAMP(18);
            if (!half) {
#pragma unroll
              for (int r = 0; r < 16; r += 4) {
                const f32x4 o = *(const f32x4*)(hand + r);
                acc[r] += o[0]; acc[r + 1] += o[1]; acc[r + 2] += o[2]; acc[r + 3] += o[3];
              }
              RG_STAMP(19);
              store_tile(acc, tm, 0);
            }
          }
        } else if (GROUPED && RG_GROUPED_STAGE_OUT && NTo <= NW && a.stage_out) {
          // Grouped output layer, wide (QR-DQN: 200 quantiles per row).  Stored straight from the accumulators a wave
          // instruction writes two 128-byte row segments that start 32 * row bytes off a cache line (rows are 800 bytes):
          // partial lines, 54 MB of them per launch (-20..-28 us with the stores removed).  Here the output leaves as
          // whole rows, 16 bytes per lane: one 32-row tile at a time (wave w computes its column tile w) through a staging
          // area BEHIND the activation tile (32 x (32 NTo + 4) floats, 29 KB of the 30 KB the tile leaves of the CU's LDS).
          const int P = NTo * 32 + 4;  // floats per staged row
          const int np = N >> 2;       // 16-byte pieces per row
#if RG_GROUPED_WHOLE
          // Round 6.  The per-row-tile loop below is a chain of L2 round trips: each of its four passes runs tile_kloop (4 weight
          // chunks in flight per wave, 8 dependent groups) and two barriers — ~45k of the workgroup's cycles for 8k of MFMAs.
          // The LAST segment of a tile (the only one of a tile inside one group's range: all but <= n_groups - 1 tiles of a
          // launch) leaves the activation tile dead after its K loop, so there: wave w sums column tile w for ALL FOUR row tiles
          // in the software-pipelined main loop (one weight stream per wave, ring of 8 chunks), and the 128 x N outputs are staged
          // in the dead activation tile (128 x P floats) and leave as whole rows — two barriers instead of eight.
          bool last_segment = false;
          if ((size_t)FB_BM * P * sizeof(float) <= (size_t)FB_BM * pitch * sizeof(bf16_t)) {
            int g2 = seg_g;
            RowSegment s2;
            last_segment = !next_segment(a.row_begin, a.n_groups, row_base, FB_BM, g2, s2);  // (workgroup-uniform)
          }
          if (last_segment) {
            grouped_whole_tile_out<NW>(act, pitch, KC, wf_out, b_out, N, NTo, out_act, seg.lo, seg.hi, row_base, a.out_scatter ? a.rowmap : nullptr,
                                       a.batch, a.out32, a.ldo);
            break;  // (the last segment)
          }
#endif
          float* stage = (float*)(act + FB_BM * pitch);
          for (int tm = tm0; tm < tm1; ++tm) {
            if (wave < NTo) {
              const f32x16 acc = tile_kloop(act, pitch, KC, wf_out, tm, wave, lane);
              const int col = wave * 32 + lr;
              const float b = (b_out && col < N) ? b_out[col] : 0.f;
              float v[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] = acc[r] + b;
              act_apply_n(v, out_act);
#pragma unroll
              for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * lg) * P + col] = v[r];
            }
            __syncthreads();
            for (int it = tid; it < 32 * np; it += THREADS) {
              const int r = it / np, c4 = it - r * np;
              const int rel = tm * 32 + r;
              if (rel < seg.lo || rel >= seg.hi) continue;  // another segment's row
              int row = row_base + rel;
              if (a.out_scatter) row = a.rowmap[row];  // back to batch order; padding rows (-1) are dropped
              if (row >= 0 && (a.out_scatter || row < a.batch))
                stream_store(*(const f32x4*)(stage + r * P + c4 * 4), (f32x4*)(a.out32 + (long)row * a.ldo + c4 * 4));
            }
            __syncthreads();
          }
        } else {
          for (int t = wave; t < 4 * NTo; t += NW) {
            const int tm = t & 3, nt = t >> 2;
            if (tm < tm0 || tm >= tm1) continue;
            store_tile(tile_kloop(act, pitch, KC, wf_out, tm, nt, lane), tm, nt);
          }
        }
        if (!GROUPED) break;
      }
      RG_STAMP(2 + 4 * l);
    }
  }
}

template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_fwd_fused_kernel(MlpArgs a) {
  mlp_fwd_fused_body<TN, NW, PITCH, false>(a);
}
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_fwd_grouped_kernel(MlpArgs a) {
  mlp_fwd_fused_body<TN, NW, PITCH, true>(a);
}
#if RG_FWD_SWAP
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_fwd_swap_kernel(MlpArgs a) {  // save == 0 only
  mlp_fwd_fused_body<TN, NW, PITCH, false, true>(a);
}
#endif

// DX_ONLY: a frozen stack — only the input gradient is produced, no dZ fragments are written (rg_mlp_desc.dx_only)
// GROUPED: the stack's output layer takes per-group weights (rg_mlp_desc.tile_key / row_begin): its own instantiation, the
// plain kernel does not carry the segment walk.  The grouped layer's step — dH = dZ . W_g per row's group — runs once per
// segment of the tile on a copy of the dZ tile with the other segments' rows zeroed (they add nothing to the shared
// accumulators); that copy is also what leaves as the segment's dZ fragments (group g's blocks g blocks late: fill_args /
// grouped_dz_rows) and what the segment's bias-gradient partial sums (row tile + g of db_part).  A tile that lies inside one
// group's range is one segment and uses the tile itself.
template <int TN, int NW, int PITCH, bool DX_ONLY, bool GROUPED = false>
__device__ __forceinline__ void mlp_bwd_fused_body(const MlpArgs& a) {
  constexpr int THREADS = MlpCfg<NW>::THREADS, RING = MlpCfg<NW>::RING;
  RG_DYN_LDS(smem);
  bf16_t* act = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  const int tile = GROUPED ? grouped_tile(blockIdx.x, (a.batch + FB_BM - 1) / FB_BM) : (int)blockIdx.x;
  if (tile * FB_BM >= round_up(a.batch, FB_BM)) return;  // padding blocks of a grouped launch (workgroup-uniform)
  const int row_base = tile * FB_BM;
  constexpr int pitch = PITCH;
  const int L = a.n_layers;
  const int nop = round_up(a.dims[L], 32);
  RG_BSTAMP(0);
  // The sign bits of H_l are requested ahead of step l's main loop so that its epilogue finds them in registers.  The FIRST
  // step's main loop is one or two K chunks long (K = the output width), too short to cover their HBM round trip
  // (bwd_phases: its pack 7.9k cycles against 6.1-6.8k for the other steps), so its request leaves before the dout tile's.
  auto request_signs = [&](int l, unsigned (&sg)[2 * TN]) {
    if (a.act_sign[l] != nullptr) {
      const u32x2* sp = (const u32x2*)(a.act_sign[l] + sign_offset(tile, wave, lane, TN, a.dims[l]));
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const u32x2 t = sp[i];
        sg[2 * i] = t[0];
        sg[2 * i + 1] = t[1];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2 * TN; ++i) sg[i] = 0u;
    }
  };
  unsigned sg_first[2 * TN];
  if (RG_BWD_SIGNS_EARLY && L >= 2) request_signs(L - 1, sg_first);
  // ... and when that step's K is ONE chunk (<= 16 outputs: Q-values, a critic's scalar) so do its weight fragments: the step is
  // then four LDS reads and 4 x TN MFMAs instead of an L2 round trip behind the tile's barrier (bwd_phases: 2.7k cycles).
  const bool first_one_chunk = RG_BWD_SIGNS_EARLY && !GROUPED && L >= 2 && a.dims[L] <= 16;
  u16x8 w_first[TN];
  if (first_one_chunk) {
    const bf16_t* wl = a.wfrag[L - 1] + (long)(wave * TN) * 512 + lane * 8;  // KC = 1: one 512-element record per n-tile
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) w_first[tn] = *(const u16x8*)(wl + tn * 512);
  }
  load_tile_to_lds<float, THREADS>(act, pitch, a.dout32, a.lddo, row_base, a.batch, a.dims[L], nop, tid);
  __syncthreads();
  RG_BSTAMP(1);
  if (!GROUPED) {
    if (!DX_ONLY) emit_frags_from_lds(act, pitch, nop / 32, a.dz_frag[L - 1], tile * 4, wave, NW, lane);
    if (a.db_part[L - 1] && tid < a.dims[L]) {
      float s = 0.f;
      for (int r = 0; r < FB_BM; ++r) s += bf16_to_f32(act[r * pitch + tid]);
      a.db_part[L - 1][(long)tile * a.dims[L] + tid] = s;
    }
  }
  RG_BSTAMP(2);

  for (int l = L - 1; l >= 1; --l) {
    // dH = dZ_l (LDS, width dims[l+1]) . W_l -> [128, dims[l]] ; dZ_{l-1} = dH * act'(H_l)
    const int K = a.dims[l + 1], N = a.dims[l];
    const int KC = (K + 15) / 16;
    f32x16 acc[4][TN];
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    const long nt_stride = (long)KC * 512;
    // sign bits of H_l, requested before the main loop so they are in registers at the epilogue
    unsigned sg[2 * TN];
    const bool use_sign = a.act_sign[l] != nullptr;
    if (RG_BWD_SIGNS_EARLY && l == L - 1) {
#pragma unroll
      for (int i = 0; i < 2 * TN; ++i) sg[i] = sg_first[i];
    } else {
      request_signs(l, sg);
    }
    // The grouped layer's FIRST segment (the only one of a tile inside a group's range) goes through the main loop like a
    // plain layer's tile — one call site, no loop around it (the software-pipelined loop inside a loop over segments spilled
    // 390-460 registers in the 512-wide kernel, accumulator tiles among them) — any further segment through
    // segment_accumulate.  What a segment contributes besides its products — its rows of dZ as fragments for the weight
    // gradient, its bias-gradient partial — is read from the dZ tile with the row range as a mask; only the MFMA operand of a
    // segment that is not the whole tile needs a copy with the other rows zeroed.
    const bool grouped_layer = GROUPED && l == L - 1;
    bf16_t* cp = act + masked_copy_offset<PITCH>(FB_BM * PITCH);
    int seg_g = grouped_layer ? a.tile_key[tile] : 0;
    RowSegment seg{0, 0, FB_BM};
    bool more = grouped_layer ? next_segment(a.row_begin, a.n_groups, row_base, FB_BM, seg_g, seg) : true;
    auto segment_side = [&]() -> const bf16_t* {  // -> the segment's MFMA operand
      const bool whole = seg.lo == 0 && seg.hi == FB_BM;
      if (!whole) {
        copy_rows_masked<THREADS, FB_BM>(act, cp, pitch, nop, seg.lo, seg.hi, tid);
        __syncthreads();
      }
      const bf16_t* src = whole ? act : cp;
      emit_frags_from_lds(src, pitch, nop / 32, a.dz_frag[L - 1], tile * 4 + seg.grp, wave, NW, lane, seg.lo >> 5,
                          (seg.hi + 31) >> 5);
      if (a.db_part[L - 1] && tid < a.dims[L]) {
        float s = 0.f;
        for (int r = seg.lo; r < seg.hi; ++r) s += bf16_to_f32(act[r * pitch + tid]);
        a.db_part[L - 1][(long)(tile + seg.grp) * a.dims[L] + tid] = s;
      }
      return src;
    };
    if (more) {
      const bf16_t* src = act;
      if (grouped_layer) src = segment_side();
      const bf16_t* wl = a.wfrag[l] + (grouped_layer ? (long)seg.grp * a.group_stride : 0);  // the group's slice of W^T
      if (first_one_chunk && l == L - 1) {
        const bf16_t* arow = src + lr * pitch + lg * 8;
        u16x8 af[4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) af[tm] = *(const u16x8*)(arow + tm * 32 * pitch);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_main(af[tm], w_first[tn], acc[tm][tn]);
      } else {
        wide_mainloop<TN, RING>(src, pitch, KC, wl + (long)(wave * TN) * nt_stride, nt_stride, acc, lane,
                                k_rotation(blockIdx.x, wave, KC), wave / (NW / 2));
      }
    }
    if (grouped_layer) {
      while (more && next_segment(a.row_begin, a.n_groups, row_base, FB_BM, seg_g, seg)) {  // a boundary tile's other groups
        __syncthreads();  // every wave is done with the previous segment's copy
        const bf16_t* src = segment_side();
        segment_accumulate<4, TN>(src, pitch, KC, a.wfrag[l] + (long)seg.grp * a.group_stride + (long)(wave * TN) * nt_stride,
                                  nt_stride, acc, lane);
      }
    }
    float* dbp = a.db_part[l - 1] ? a.db_part[l - 1] + (long)tile * N : nullptr;
    RG_BSTAMP(3 + 4 * (L - 1 - l));
    unsigned PK[4][TN][8];
    if (use_sign) {
      RG_DISPATCH_ACT(a.acts[l - 1], (bwd_hidden_pack<TN, A_, true, !DX_ONLY>(acc, a.act_frag[l], sg, a.dz_frag[l - 1], dbp,
                                                                             N / 32, tile * 4, wave, lane, PK)));
    } else {
      RG_DISPATCH_ACT(a.acts[l - 1], (bwd_hidden_pack<TN, A_, false, !DX_ONLY>(acc, a.act_frag[l], sg, a.dz_frag[l - 1], dbp,
                                                                              N / 32, tile * 4, wave, lane, PK)));
    }
    RG_BSTAMP(4 + 4 * (L - 1 - l));
    __syncthreads();  // every wave is done reading dZ_l
    RG_BSTAMP(5 + 4 * (L - 1 - l));
    store_packed_tiles<TN>(act, pitch, PK, wave, lane);
    __syncthreads();
    RG_BSTAMP(6 + 4 * (L - 1 - l));
  }
  if (a.dx32) {  // gradient w.r.t. the network input (e.g. the critic's action input in SAC)
    const int K = a.dims[1], N = a.dims[0];
    const int KC = (K + 15) / 16, NTi = (N + 31) / 32, nt0 = a.dx_col0 / 32;  // only the tiles from dx_col0 on
    for (int t = wave; t < 4 * (NTi - nt0); t += NW) {
      const int tm = t & 3, nt = nt0 + (t >> 2);
      const f32x16 acc = tile_kloop(act, pitch, KC, a.wfrag[0], tm, nt, lane);
      const int col = nt * 32 + lr;
      if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_base + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
          if (row < a.batch) a.dx32[(long)row * a.lddx + col - a.dx_col0] = acc[r];
        }
      }
    }
  }
}

template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_bwd_fused_kernel(MlpArgs a) {
  mlp_bwd_fused_body<TN, NW, PITCH, false>(a);
}
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_bwd_dx_kernel(MlpArgs a) {
  mlp_bwd_fused_body<TN, NW, PITCH, true>(a);
}
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_bwd_grouped_kernel(MlpArgs a) {
  mlp_bwd_fused_body<TN, NW, PITCH, false, true>(a);
}

// ---- weight gradient from fragment-ordered operands ------------------------------------------
// dW[n][k] = sum_m dZ[m][n] X[m][k].  A = dz_frag tiles (lane = n), B = x_frag tiles (lane = k):
// per 32-row block and half, one MFMA per (n-tile, k-tile) pair.  Workgroup tile 256(n) x 256(k) — or another shape of
// <= 64 tiles, wgrad_shape_core — over 8 waves, each 4x2 MFMA tiles; operands staged HBM -> LDS by DMA (lane-linear
// 16-byte units, conflict-free reads), one 32-row block per stage, a ring of stages, one barrier per stage.
constexpr int WG_MB_STAGE = 1;                      // 32-row blocks per stage

struct WgradFragArgs {
  const bf16_t* a_frag;
  const bf16_t* b_frag;
  int NTa, NTb;       // tiles per 32-row block in each operand
  int MB;             // 32-row blocks in total (= the END of this entry's block range)
  int mb_base;        // first block of this entry's split 0 (0 but for the second class of an unevenly split layer)
  int mb_per_split;   // multiple of WG_MB_STAGE
  int splits;
  float* partial;     // [splits][N*K]
  long slab;
  int N, K;           // valid extents of dW
  // split-bf16 operands (x3 != 0): every stage carries the hi AND lo planes of both operands and a tile pair takes
  // three MFMAs — a_lo.b_hi + a_hi.b_lo + a_hi.b_hi into one accumulator (wgrad_x3_shape_core).  The kernel is bound by its
  // operand stream (§3.2), so the three products share ONE pass over the four planes instead of three passes over
  // two planes each (the round-2 first version: three partial slabs per split, 354 us per C2 launch).
  int x3;
  long a_lo, b_lo;    // element offsets of the lo planes
  int shape;          // workgroup tile shape (WG_SHAPE_*, wgrad_shape_core / wgrad_x3_shape_core); 0 = 8 x 8 tiles
  // How a split's partial tile leaves the workgroup.  0: fp32, row-major [N][K] (slab = N * K floats).  1 (round 5, the bf16
  // stack launch): bf16, one 2 KB record per 32 x 32 MFMA tile in ACCUMULATOR order — record (tn * NTb + tk), lane's 16 values
  // contiguous (32 bytes) — so a tile leaves as two 16-byte stores per lane instead of sixteen 4-byte ones and the launch
  // writes (and its reduce reads) half the bytes; slab = NTa * NTb * 512 floats' worth.  The reduce launch undoes the order.
  int part_mode;
  int part_nt;        // part_mode 1: bit 0 = the partial tiles leave as non-temporal stores (RG_WGRAD_PART_NT, A/B switch)
};


// ---- workgroup tile shapes ------------------------------------------------------------------------------------------
// The kernel is bound by its operand stream L2 -> LDS (round 2-3 ablations), and a 256 x 256 tile streams 8 + 8 fragment
// tiles per 32-row block whatever the layer looks like: for dW0 [512 x 128] half of the B operand was a clamped re-read,
// for a thin output layer's dW [16 x 512] seven eighths of the A operand — 93 MB of the 804 MB a C2 launch moved into LDS
// were such junk, and both layers read their long operand through two tiles.  A workgroup now takes GA n-tiles x GB
// k-tiles (GA * GB <= 64 accumulator tiles over 8 waves = WN x WK, each TA x TB), chosen per layer by the host to minimise
// the bytes staged (wgrad_pick_shape):
//   8 x 8   (256 x 256)  square layers                      4 DMAs per thread and 32-row block
//   16 x 4  (512 x 128)  wide-out / narrow-in (dW0)          5   (dZ0 read by ONE tile)
//   4 x 16  (128 x 512)  narrow-out / wide-in                5
//   2 x 16, 1 x 16       thin output layers (<= 64 / <= 32 outputs): 5, with 2 / 1 accumulator tile(s) per wave
// A stage is (GA + GB) tiles x 2 KB, padded to whole 16-byte units per thread; stages of more than 32 KB ring through
// three slots (two in flight: measured equal to three in round 2), the 8 x 8 shape keeps four.
enum { WG_SHAPE_8x8 = 0, WG_SHAPE_16x4 = 1, WG_SHAPE_4x16 = 2, WG_SHAPE_2x16 = 3, WG_SHAPE_1x16 = 4, WG_N_SHAPES = 5 };

template <int GA_, int GB_, int WN_, int WK_> struct WgShape {
  static constexpr int GA = GA_, GB = GB_, WN = WN_, WK = WK_;
  static constexpr int TA = GA / WN, TB = GB / WK;
  static_assert(WN * WK == WG_THREADS / 64 && TA * WN == GA && TB * WK == GB && TA * TB <= 8, "wave layout");
  static constexpr int DMA = ((GA + GB) * 128 + WG_THREADS - 1) / WG_THREADS;  // 16-byte units per thread and stage
  static constexpr int STAGE_BYTES = DMA * WG_THREADS * 16;
  static constexpr int SLOTS = STAGE_BYTES <= 32 * 1024 ? 4 : 3;
  static constexpr int LDS_BYTES = SLOTS * STAGE_BYTES;
};
using WgS8x8 = WgShape<8, 8, 2, 4>;
using WgS16x4 = WgShape<16, 4, 4, 2>;
using WgS4x16 = WgShape<4, 16, 1, 8>;
using WgS2x16 = WgShape<2, 16, 1, 8>;
using WgS1x16 = WgShape<1, 16, 1, 8>;
constexpr int WG_SHAPED_LDS = 128 * 1024;  // max over the shapes (8x8: 4 x 32 KB; the 5-DMA shapes: 3 x 40 KB)
static_assert(WgS8x8::LDS_BYTES <= WG_SHAPED_LDS && WgS16x4::LDS_BYTES <= WG_SHAPED_LDS && WgS4x16::LDS_BYTES <= WG_SHAPED_LDS &&
              WgS2x16::LDS_BYTES <= WG_SHAPED_LDS && WgS1x16::LDS_BYTES <= WG_SHAPED_LDS, "dynamic LDS of the weight-gradient launches");

template <int N> __device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}

// one workgroup: dW tile (n-group ng, k-group kg) of shape S over the 32-row blocks [mb_begin, mb_end), written to `part`
template <typename S>
__device__ __forceinline__ void wgrad_shape_core(const WgradFragArgs& g, int ng, int kg, int mb_begin, int mb_end, float* part,
                                                 char* smem) {
  constexpr int GA = S::GA, GB = S::GB, TA = S::TA, TB = S::TB, DMA = S::DMA, SLOTS = S::SLOTS, FLY = S::SLOTS - 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  const int wn = wave / S::WK, wk = wave % S::WK;
  const int ta0 = ng * GA, tb0 = kg * GB;
  const int na = (g.NTa - ta0 < GA) ? g.NTa - ta0 : GA, nb = (g.NTb - tb0 < GB) ? g.NTb - tb0 : GB;
  const bf16_t* ga_frag = g.a_frag;
  const bf16_t* gb_frag = g.b_frag;

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // LDS image of a stage: tile t (A tiles 0 .. GA-1, then the B tiles, then padding) at t * 2 KB, lane-linear 16-byte
  // units — what the DMA writes (wave-uniform base + lane * 16).  Unit u = tid + i * 512 belongs to tile u / 128, which is
  // the same for the 64 lanes of a wave; out-of-range tiles / blocks read a clamped valid address (never used), so every
  // wave issues exactly DMA loads per stage and the vmcnt arithmetic below is exact.
  const int mb_last = mb_end - 1;
  // per DMA of a stage (compile-time i): this wave's tile is fixed, only the 32-row block moves — base pointer and block
  // stride are worked out once, from values already in registers (selecting between the A and the B operand's FIELDS of
  // the argument block inside the loop made the compiler index them through scratch)
  const int nta = g.NTa, ntb = g.NTb;
  const int within = tid & 127;  // (tid + i * 512) & 127: the same for every i
  const bf16_t* src0[DMA];
  long blk_stride[DMA];
  static_for<0, DMA>([&](auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    const int t = i * (WG_THREADS / 128) + (wave >> 1);  // tile of this wave's units (wave-uniform)
    const bool is_a = t < GA;
    const int ta = ta0 + (t < na ? t : na - 1);
    const int tb = tb0 + ((t - GA) < nb ? (t - GA < 0 ? 0 : t - GA) : nb - 1);
    const long tile = is_a ? (long)ta : (long)tb;
    const bf16_t* base = is_a ? ga_frag : gb_frag;
    src0[i] = base + tile * 1024 + within * 8;
    blk_stride[i] = (long)(is_a ? nta : ntb) * 1024;
  });
  auto issue = [&](int blk, int slot) {
    const int mb = blk < mb_last ? blk : mb_last;
    static_for<0, DMA>([&](auto i_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      global_load_lds_b128(src0[i] + (long)mb * blk_stride[i], smem + slot * S::STAGE_BYTES + (wave * 64 + i * WG_THREADS) * 16);
    });
  };
  const bool wave_has_tiles = wn * TA < na && wk * TB < nb;  // a wave whose tiles are all padding skips its MFMAs
  // Round 4 (profiles/microbench/out/r04a/wgrad_phases.txt): an iteration of this loop took ~2700 cycles whatever the stage
  // held — 16 MFMAs per wave (1024 cycles per SIMD), the rest LDS latency, DMA issue and the barrier: the loop is bound
  // by its own per-iteration chain, not by the memory system (8 % of it waits for data).  So (a) the fragments of the next
  // 16-row half are requested from LDS before the MFMAs of the current one (RG_WGRAD_PIPE), and (b) the square shape,
  // whose ring has four slots, takes TWO blocks per barrier (RG_WGRAD_PAIR): one wait, one barrier and one burst of DMA
  // issues per 32 MFMAs of a wave instead of per 16, the next pair in flight meanwhile.
  auto load_half = [&](int slot, int h, u16x8 (&af)[TA], u16x8 (&bf)[TB]) {
    const char* base = smem + slot * S::STAGE_BYTES + h * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < TA; ++i) af[i] = *(const u16x8*)(base + (wn * TA + i) * 2048);
#pragma unroll
    for (int j = 0; j < TB; ++j) bf[j] = *(const u16x8*)(base + (GA + wk * TB + j) * 2048);
  };
  auto mma_half = [&](const u16x8 (&af)[TA], const u16x8 (&bf)[TB]) {
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) acc[i][j] = mfma_32x32x16_bf16(af[i], bf[j], acc[i][j]);
  };
  // the halves of `nb_` consecutive blocks whose first slot is slot0 (slots advance modulo the ring)
  auto compute = [&](int slot0, auto nb_c) __attribute__((always_inline)) {
    constexpr int NH = 2 * decltype(nb_c)::value;
    if (!wave_has_tiles) return;
#if RG_WGRAD_PIPE
    u16x8 af[2][TA], bf[2][TB];
    load_half(slot0, 0, af[0], bf[0]);
    static_for<0, NH>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      if constexpr (q + 1 < NH) load_half((slot0 + (q + 1) / 2) % SLOTS, (q + 1) & 1, af[(q + 1) & 1], bf[(q + 1) & 1]);
      sched_fence();  // (without the fences the scheduler requests every half's fragments up front: 256 registers + scratch)
      mma_half(af[q & 1], bf[q & 1]);
      sched_fence();
    });
#else
    static_for<0, NH>([&](auto q_c) __attribute__((always_inline)) {
      constexpr int q = decltype(q_c)::value;
      u16x8 af[TA], bf[TB];
      load_half((slot0 + q / 2) % SLOTS, q & 1, af, bf);
      mma_half(af, bf);
    });
#endif
  };
  using one_t = std::integral_constant<int, 1>;
  using two_t = std::integral_constant<int, 2>;
  constexpr bool PAIR = RG_WGRAD_PAIR && SLOTS == 4;

  RG_PHASE_INIT();
  if (mb_begin < mb_end) {
    const int n_blk = mb_end - mb_begin;
    if constexpr (PAIR) {
      // pair p = blocks 2p, 2p + 1 in slots (2p) % 4, (2p + 1) % 4; while it is computed pair p + 1 travels into the other
      // two slots, which every wave left before this iteration's barrier
      const int n_pair = n_blk >> 1;
      issue(mb_begin, 0);
      issue(mb_begin + 1, 1);
      RG_PHASE(0);
      for (int p = 0; p < n_pair; ++p) {
        wait_vmcnt<0>();
        RG_PHASE(2);
        raw_barrier();
        RG_PHASE(3);
        issue(mb_begin + 2 * p + 2, (2 * p + 2) & 3);
        issue(mb_begin + 2 * p + 3, (2 * p + 3) & 3);
        compute((2 * p) & 3, two_t{});
        RG_PHASE(1);
      }
      wait_vmcnt<0>();
      if (n_blk & 1) {  // an odd tail block (it travelled as the first block of the pair after the last; outside the
        raw_barrier();  // loop: a second path through the accumulators INSIDE it doubled the kernel's registers)
        compute((2 * n_pair) & 3, one_t{});
      }
    } else {
      static_for<0, FLY>([&](auto f_c) __attribute__((always_inline)) { issue(mb_begin + decltype(f_c)::value, decltype(f_c)::value); });
      RG_PHASE(0);
      for (int t = 0; t < n_blk; ++t) {
        // FLY stages are outstanding: let the oldest land, then meet the other waves — past the barrier block t is complete
        // in LDS and every wave has finished reading block t-1, whose slot the DMA issued below overwrites
        wait_vmcnt<(FLY - 1) * DMA>();
        RG_PHASE(2);
        raw_barrier();
        RG_PHASE(3);
        issue(mb_begin + t + FLY, (t + FLY) % SLOTS);  // before the MFMAs: the requests leave a block time earlier
        compute(t % SLOTS, one_t{});
        RG_PHASE(1);
      }
      wait_vmcnt<0>();
    }
  }

  if (g.part_mode == 1) {
    typedef __attribute__((ext_vector_type(4))) unsigned pk4_t;
    bf16_t* pb = (bf16_t*)part;
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) {
        const int tn = ta0 + wn * TA + i, tk = tb0 + wk * TB + j;
        if (tn < nta && tk < ntb) {
          pk4_t* dst = (pk4_t*)(pb + ((long)tn * ntb + tk) * 1024 + lane * 16);
          const f32x16& c = acc[i][j];
          const pk4_t v0 = pk4_t{pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], c[3]), pack_bf16x2(c[4], c[5]), pack_bf16x2(c[6], c[7])};
          const pk4_t v1 = pk4_t{pack_bf16x2(c[8], c[9]), pack_bf16x2(c[10], c[11]), pack_bf16x2(c[12], c[13]), pack_bf16x2(c[14], c[15])};
          if (g.part_nt & 1) { stream_store(v0, dst); stream_store(v1, dst + 1); }
          else { dst[0] = v0; dst[1] = v1; }
        }
      }
    RG_PHASE(4);
    RG_PHASE_FLUSH();
    return;
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int col = (tb0 + wk * TB + j) * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (ta0 + wn * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        if (row < g.N && col < g.K) part[(long)row * g.K + col] = acc[i][j][r];
      }
    }
  RG_PHASE(4);
  RG_PHASE_FLUSH();
}

// tiles per group of a shape, for the host plan and the workgroup decode (value-returning selects: reference outputs of a
// switch put four dwords of the five-shape kernel into scratch)
__host__ __device__ __forceinline__ int wgrad_shape_ga(int shape) {
  return shape == WG_SHAPE_16x4 ? 16 : shape == WG_SHAPE_4x16 ? 4 : shape == WG_SHAPE_2x16 ? 2 : shape == WG_SHAPE_1x16 ? 1 : 8;
}
__host__ __device__ __forceinline__ int wgrad_shape_gb(int shape) {
  return shape == WG_SHAPE_16x4 ? 4 : shape == WG_SHAPE_8x8 ? 8 : 16;
}

// Split-bf16 operands.  A stage is HALF a 32-row block (one 16-row MFMA chunk) of all four planes,
// [a_hi (GA tiles) | a_lo (GA) | b_hi (GB) | b_lo (GB)] x 1 KB — the same bytes, ring and DMA count per thread as the bf16
// core's stage of the same shape, twice the stages, three MFMAs per tile pair: per operand byte 1.5x the MFMA work of the
// bf16 kernel.  DMA i of a stage moves the 1 KB planes 8 i .. 8 i + 7, one per wave.
// TWO (x3 == 2, rg_mlp_frag.h: x3_dz_planes() == 1): the A operand (dZ) is ONE plane — a stage is [a_hi (GA) | b_hi (GB) | b_lo (GB)],
// two MFMAs per tile pair (a_hi.b_lo + a_hi.b_hi).
template <typename S, bool TWO = false>
__device__ __forceinline__ void wgrad_x3_shape_core(const WgradFragArgs& g, int ng, int kg, int mb_begin, int mb_end, float* part,
                                                    char* smem) {
  constexpr int GA = S::GA, GB = S::GB, TA = S::TA, TB = S::TB;
  constexpr int PA = TWO ? GA : 2 * GA;                     // A planes of a stage
  constexpr int DMA = TWO ? (PA + 2 * GB + 7) / 8 : S::DMA; // 1 KB planes, eight (one per wave) per DMA
  constexpr int STAGE_BYTES = TWO ? DMA * 8 * 1024 : S::STAGE_BYTES;
  constexpr int SLOTS = TWO ? (STAGE_BYTES <= 32 * 1024 ? 4 : 3) : S::SLOTS, FLY = SLOTS - 1;
  static_assert(SLOTS * STAGE_BYTES <= WG_SHAPED_LDS, "LDS of the two-product stage ring");
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  const int wn = wave / S::WK, wk = wave % S::WK;
  const int ta0 = ng * GA, tb0 = kg * GB;
  const int na = (g.NTa - ta0 < GA) ? g.NTa - ta0 : GA, nb = (g.NTb - tb0 < GB) ? g.NTb - tb0 : GB;
  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int n_stage = 2 * (mb_end - mb_begin);
  // out-of-range tiles / stages read a clamped valid address (never used), so every wave has exactly DMA loads per stage;
  // base pointer and block stride of each are worked out once (see wgrad_shape_core)
  const int nta = g.NTa, ntb = g.NTb;
  const bf16_t* a_hi = g.a_frag;
  const bf16_t* b_hi = g.b_frag;
  const long a_lo = g.a_lo, b_lo = g.b_lo;
  const bf16_t* src0[DMA];
  long blk_stride[DMA];
#pragma unroll
  for (int i = 0; i < DMA; ++i) {
    const int q = i * 8 + wave;  // plane of this wave's DMA (wave-uniform)
    const bool is_a = q < PA;
    const int qa = q < GA ? q : q - GA;                  // tile within the A planes
    int qb = q - PA;                                     // within the B planes (hi, lo, padding)
    const bool b_is_lo = qb >= GB;
    qb = qb >= GB ? qb - GB : qb;
    qb = qb < 0 ? 0 : qb;
    const long tile = is_a ? (long)(ta0 + (qa < na ? qa : na - 1)) : (long)(tb0 + (qb < nb ? qb : nb - 1));
    const long plane = is_a ? (q >= GA ? a_lo : 0) : (b_is_lo ? b_lo : 0);
    const bf16_t* base = is_a ? a_hi : b_hi;
    src0[i] = base + plane + tile * 1024 + lane * 8;
    blk_stride[i] = (long)(is_a ? nta : ntb) * 1024;
  }
  auto issue = [&](int st, int slot) {
    const int sc = st < n_stage ? st : n_stage - 1;
    const long mb = mb_begin + (sc >> 1);
    const int h = sc & 1;
    static_for<0, DMA>([&](auto i_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      global_load_lds_b128(src0[i] + mb * blk_stride[i] + h * 512, smem + slot * STAGE_BYTES + (i * 8 + wave) * 1024);
    });
  };
  const bool wave_has_tiles = wn * TA < na && wk * TB < nb;
  auto compute = [&](int slot) {
    if (!wave_has_tiles) return;
    const char* base = smem + slot * STAGE_BYTES + lane * 16;
    u16x8 ah[TA], al[TA], bh[TB], bl[TB];
#pragma unroll
    for (int i = 0; i < TA; ++i) {
      ah[i] = *(const u16x8*)(base + (wn * TA + i) * 1024);
      if (!TWO) al[i] = *(const u16x8*)(base + (GA + wn * TA + i) * 1024);
    }
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      bh[j] = *(const u16x8*)(base + (PA + wk * TB + j) * 1024);
      bl[j] = *(const u16x8*)(base + (PA + GB + wk * TB + j) * 1024);
    }
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
      for (int j = 0; j < TB; ++j) {  // the small products first
        if (!TWO) acc[i][j] = mfma_32x32x16_bf16(al[i], bh[j], acc[i][j]);
        acc[i][j] = mfma_32x32x16_bf16(ah[i], bl[j], acc[i][j]);
        acc[i][j] = mfma_32x32x16_bf16(ah[i], bh[j], acc[i][j]);
      }
  };
  if (n_stage > 0) {
    static_for<0, FLY>([&](auto f_c) __attribute__((always_inline)) { issue(decltype(f_c)::value, decltype(f_c)::value); });
    for (int t = 0; t < n_stage; ++t) {
      wait_vmcnt<(FLY - 1) * DMA>();  // FLY stages outstanding: the oldest has landed
      raw_barrier();                  // ... for every wave, and all are done reading stage t-1, whose slot is refilled next
      issue(t + FLY, (t + FLY) % SLOTS);
      compute(t % SLOTS);
    }
    wait_vmcnt<0>();
  }
#pragma unroll
  for (int i = 0; i < TA; ++i)
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const int col = (tb0 + wk * TB + j) * 32 + lr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (ta0 + wn * TA + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
        if (row < g.N && col < g.K) part[(long)row * g.K + col] = acc[i][j][r];
      }
    }
}

// the workgroup's core for its layer's operand format and tile shape (all arguments workgroup-uniform)
__device__ __forceinline__ void wgrad_dispatch(const WgradFragArgs& g, int ng, int kg, int mb_begin, int mb_end, float* part,
                                               char* smem) {
  if (g.x3 == 2) {  // dZ as one plane (two products)
    switch (g.shape) {
      case WG_SHAPE_16x4: wgrad_x3_shape_core<WgS16x4, true>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      case WG_SHAPE_4x16: wgrad_x3_shape_core<WgS4x16, true>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      case WG_SHAPE_2x16: wgrad_x3_shape_core<WgS2x16, true>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      case WG_SHAPE_1x16: wgrad_x3_shape_core<WgS1x16, true>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      default: wgrad_x3_shape_core<WgS8x8, true>(g, ng, kg, mb_begin, mb_end, part, smem); break;
    }
    return;
  }
  if (g.x3) {
    switch (g.shape) {
      case WG_SHAPE_16x4: wgrad_x3_shape_core<WgS16x4>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      case WG_SHAPE_4x16: wgrad_x3_shape_core<WgS4x16>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      case WG_SHAPE_2x16: wgrad_x3_shape_core<WgS2x16>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      case WG_SHAPE_1x16: wgrad_x3_shape_core<WgS1x16>(g, ng, kg, mb_begin, mb_end, part, smem); break;
      default: wgrad_x3_shape_core<WgS8x8>(g, ng, kg, mb_begin, mb_end, part, smem); break;
    }
    return;
  }
  switch (g.shape) {
    case WG_SHAPE_16x4: wgrad_shape_core<WgS16x4>(g, ng, kg, mb_begin, mb_end, part, smem); break;
    case WG_SHAPE_4x16: wgrad_shape_core<WgS4x16>(g, ng, kg, mb_begin, mb_end, part, smem); break;
    case WG_SHAPE_2x16: wgrad_shape_core<WgS2x16>(g, ng, kg, mb_begin, mb_end, part, smem); break;
    case WG_SHAPE_1x16: wgrad_shape_core<WgS1x16>(g, ng, kg, mb_begin, mb_end, part, smem); break;
    default: wgrad_shape_core<WgS8x8>(g, ng, kg, mb_begin, mb_end, part, smem); break;
  }
}

__device__ __forceinline__ void wgrad_frag_body(const WgradFragArgs& g, int bid, char* smem) {
  const int shape = g.shape;
  const int GA = wgrad_shape_ga(shape), GB = wgrad_shape_gb(shape);
  const int n_groups = (g.NTa + GA - 1) / GA, k_groups = (g.NTb + GB - 1) / GB;
  // workgroups that read the same 32-row blocks (the tiles of one split) go to ONE XCD (hardware
  // places block b on XCD b % 8), so the second reader of a fragment hits that XCD's L2 instead of
  // HBM (PMC: 700 MB fetched per launch against 420 MB of unique operands without this)
  const int tiles = k_groups * n_groups;
  // the launch has tiles * 8 * ceil(splits / 8) workgroups per layer; those of a split past the end return at once
  const int xcd = bid & 7, slot = bid >> 3;
  const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
  if (split >= g.splits) return;  // padding workgroups of a grouped launch (uniform for the workgroup)
  const int kg = tile % k_groups;
  const int ng = tile / k_groups;
  const int mb_begin0 = g.mb_base + split * g.mb_per_split;
  const int mb_begin = mb_begin0 < g.MB ? mb_begin0 : g.MB;
  const int mb_end = (mb_begin + g.mb_per_split < g.MB) ? mb_begin + g.mb_per_split : g.MB;
  wgrad_dispatch(g, ng, kg, mb_begin, mb_end, g.partial + (long)split * g.slab, smem);
}

__global__ void RG_LAUNCH_BOUNDS(512, 1) wgrad_frag_kernel(WgradFragArgs g) {
  RG_DYN_LDS(smem);
  wgrad_frag_body(g, (int)blockIdx.x, smem);
}

// Weight gradient of a GROUPED layer (QR-DQN's A x N output layer seen as A independent [N, K] layers, one per
// action; rows of the batch sorted by action, qr_grouped.hip): group a owns the rows [row_begin[a], row_begin[a + 1]), i.e.
// the 32-row blocks [row_begin[a] / 32, ceil(row_begin[a + 1] / 32)) of the activation fragments and the same blocks + a of
// the dZ fragments (where the backward launch put group a's copy of each block, the other groups' rows zeroed:
// grouped_dz_rows) — no row masks here.  The block range is cut into `splits` parts.  Workgroup = (group, k-group, split);
// partial slab index group * splits + split.  The ranges live in HBM (they depend on the sampled batch): no host round trip.
struct WgradGroupedArgs {
  WgradFragArgs g;       // a_frag: dZ fragments (NTa tiles of 32 columns), b_frag: activation fragments; N = rows of a group's dW
  const int* row_begin;  // [n_groups + 1], in rows of the grouped space
  int n_groups, splits;
};

__global__ void RG_LAUNCH_BOUNDS(512, 1) wgrad_grouped_kernel(WgradGroupedArgs G) {
  RG_DYN_LDS(smem);
  const int k_groups = (G.g.NTb + 7) / 8;
  const int bid = blockIdx.x;
  const int kg = bid % k_groups, s = (bid / k_groups) % G.splits, a = bid / (k_groups * G.splits);
  const int r0 = G.row_begin[a], r1 = G.row_begin[a + 1];
  const int mb0 = r0 >> 5, mb1 = r1 > r0 ? (r1 + 31) >> 5 : mb0;
  const int per = (mb1 - mb0 + G.splits - 1) / G.splits;
  int b0 = mb0 + s * per, b1 = b0 + per;
  if (b1 > mb1) b1 = mb1;
  if (b0 > mb1) b0 = mb1;
  float* part = G.g.partial + ((long)a * G.splits + s) * G.g.slab;
  WgradFragArgs g = G.g;
  g.a_frag += (long)a * g.NTa * 1024;  // this group's dZ blocks sit `a` blocks late (one block = NTa tiles of 2 KB)
  if (g.x3) wgrad_x3_shape_core<WgS8x8>(g, 0, kg, b0, b1, part, smem);  // split-bf16: both planes of dZ and of the activations
  else wgrad_shape_core<WgS8x8>(g, 0, kg, b0, b1, part, smem);
}

// out[a * slab + e] = sum_s partial[(a * splits + s) * slab + e]
__global__ void reduce_grouped_kernel(const float* __restrict__ partial, long slab, int splits, int n_groups,
                                      float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= slab * n_groups) return;
  const long a = i / slab, e = i % slab;
  float s0 = 0.f, s1 = 0.f;
  int k = 0;
  for (; k + 1 < splits; k += 2) {
    s0 += stream_load(partial + (a * splits + k) * slab + e);
    s1 += stream_load(partial + (a * splits + k + 1) * slab + e);
  }
  if (k < splits) s0 += stream_load(partial + (a * splits + k) * slab + e);
  out[i] = s0 + s1;
}

// all layers of a stack in ONE launch (workgroups of the small layers fill the CUs the big ones
// leave idle; 4 launches + 4 reduces become 1 + 1)
// (an ENTRY is a layer, or one of the two classes of splits of an unevenly split layer: rg_mlp_wgrad_fused)
constexpr int WG_MAXV = FB_MAXL + 4;
struct WgradGroupArgs {
  int n;
  int wg_begin[WG_MAXV + 1];
  WgradFragArgs layer[WG_MAXV];
};

__global__ void RG_LAUNCH_BOUNDS(512, 1) wgrad_group_kernel(WgradGroupArgs G) {
  RG_DYN_LDS(smem);
  const int bid = blockIdx.x;
  WgradFragArgs g = G.layer[0];
  int base = 0;
#pragma unroll
  for (int i = 1; i < WG_MAXV; ++i)
    if (i < G.n && bid >= G.wg_begin[i]) {
      g = G.layer[i];
      base = G.wg_begin[i];
    }
  wgrad_frag_body(g, bid - base, smem);
}

struct ReduceGroupArgs {
  int n;
  long elem_begin[FB_MAXL + 1];  // in THREADS: one per element (row-major fp32 partials), 256 per 32 x 32 tile (part_mode 1)
  const float* partial[FB_MAXL];
  long slab[FB_MAXL];
  int splits[FB_MAXL];
  float* out[FB_MAXL];
  // part_mode 1 layers (WgradFragArgs.part_mode): bf16 partial tiles in accumulator order; N x K = valid extents of dW
  int mode[FB_MAXL], NTb[FB_MAXL], N[FB_MAXL], K[FB_MAXL];
  int nt_loads;  // part_mode 1: the partial tiles are read with non-temporal loads (RG_WGRAD_PART_NT bit 1)
};

// One 256-thread workgroup = one 32 x 32 tile: wave w sums the lane records (16 values, 32 contiguous bytes per split) of the
// w-th quarter of the splits — all of a quarter's records requested before the first is added (RG_REDUCE_FLY_BF16 at a
// time): the launch runs on loads in flight, 148 workgroups of one wave each were 16 dependent round trips — the four
// quarter sums meet in LDS and are added in a fixed order.  Writes the row-major dW.
template <bool NT>
__device__ __forceinline__ void reduce_tiles_bf16(const float* part, long slab, int splits, float* out, int NTb, int N, int K,
                                                  long t, float (*red)[16][64]) {
  typedef __attribute__((ext_vector_type(4))) unsigned pk4_t;
  auto stream_load = [](const pk4_t* p) { return NT ? rg::stream_load(p) : *p; };  // (the partials are read exactly once)
  const int lane = (int)(t & 63), w = (int)((t >> 6) & 3);
  const long tile = t >> 8;
  const int tn = (int)(tile / NTb), tk = (int)(tile % NTb);
  const int lr = lane & 31, lg = lane >> 5;
  float s0[16], s1[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) s0[r] = s1[r] = 0.f;
  const char* p = (const char*)part + (tile * 1024 + lane * 16) * 2;
  const long stride = slab * 4;  // bytes between the splits' slabs
  auto add = [&](float (&s)[16], const pk4_t a, const pk4_t b) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s[2 * q] += __builtin_bit_cast(float, a[q] << 16);
      s[2 * q + 1] += __builtin_bit_cast(float, a[q] & 0xffff0000u);
      s[8 + 2 * q] += __builtin_bit_cast(float, b[q] << 16);
      s[8 + 2 * q + 1] += __builtin_bit_cast(float, b[q] & 0xffff0000u);
    }
  };
  const int per = (splits + 3) >> 2;
  int k = w * per;
  const int k_end = (k + per < splits) ? k + per : splits;
  constexpr int U = RG_REDUCE_FLY_BF16;
  for (; k + U <= k_end; k += U) {
    pk4_t a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      a[u] = stream_load((const pk4_t*)(p + (k + u) * stride));
      b[u] = stream_load((const pk4_t*)(p + (k + u) * stride) + 1);
    }
#pragma unroll
    for (int u = 0; u < U; u += 2) {
      add(s0, a[u], b[u]);
      add(s1, a[u + 1], b[u + 1]);
    }
  }
  for (; k + 1 < k_end; k += 2) {
    const pk4_t a0 = stream_load((const pk4_t*)(p + k * stride)), b0 = stream_load((const pk4_t*)(p + k * stride) + 1);
    const pk4_t a1 = stream_load((const pk4_t*)(p + (k + 1) * stride)), b1 = stream_load((const pk4_t*)(p + (k + 1) * stride) + 1);
    add(s0, a0, b0);
    add(s1, a1, b1);
  }
  if (k < k_end) add(s0, stream_load((const pk4_t*)(p + k * stride)), stream_load((const pk4_t*)(p + k * stride) + 1));
#pragma unroll
  for (int r = 0; r < 16; ++r) red[w][r][lane] = s0[r] + s1[r];
  __syncthreads();
  const int col = tk * 32 + lr;
  if (col >= K) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {  // this wave finishes the accumulator registers 4 w .. 4 w + 3: rows 8 w + q + 4 lg of the tile
    const int r = 4 * w + q;
    const int row = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
    if (row < N) out[(long)row * K + col] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
  }
}

__device__ __forceinline__ void reduce_group_body(const ReduceGroupArgs& R, long i) {
  if (i >= R.elem_begin[R.n]) return;
  const float* part = R.partial[0];
  long slab = R.slab[0], base = 0;
  int splits = R.splits[0];
  float* out = R.out[0];
  int mode = R.mode[0], NTb = R.NTb[0], N = R.N[0], K = R.K[0];
#pragma unroll
  for (int k = 1; k < FB_MAXL; ++k)
    if (k < R.n && i >= R.elem_begin[k]) {
      part = R.partial[k]; slab = R.slab[k]; splits = R.splits[k]; out = R.out[k]; base = R.elem_begin[k];
      mode = R.mode[k]; NTb = R.NTb[k]; N = R.N[k]; K = R.K[k];
    }
  if (mode == 1) {  // (whole workgroups: a layer's range is 256 threads per tile)
    __shared__ float red[4][16][64];
    if (R.nt_loads) reduce_tiles_bf16<true>(part, slab, splits, out, NTb, N, K, i - base, red);
    else reduce_tiles_bf16<false>(part, slab, splits, out, NTb, N, K, i - base, red);
    return;
  }
  const long e = i - base;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
#define RG_LDP(p) stream_load(p)  // the split partials are read exactly once
  // RG_REDUCE_FLY loads in flight per thread (round 5; see reduce_tiles_bf16): 32 splits were 8 dependent round trips per wave.
  // Same four chains, same order of additions: bit-identical to the 4-deep loop below.
  for (; k + RG_REDUCE_FLY <= splits; k += RG_REDUCE_FLY) {
    float v[RG_REDUCE_FLY];
#pragma unroll
    for (int u = 0; u < RG_REDUCE_FLY; ++u) v[u] = RG_LDP(part + (long)(k + u) * slab + e);
#pragma unroll
    for (int u = 0; u < RG_REDUCE_FLY; u += 4) {
      s0 += v[u];
      s1 += v[u + 1];
      s2 += v[u + 2];
      s3 += v[u + 3];
    }
  }
  for (; k + 3 < splits; k += 4) {
    s0 += RG_LDP(part + (long)k * slab + e);
    s1 += RG_LDP(part + (long)(k + 1) * slab + e);
    s2 += RG_LDP(part + (long)(k + 2) * slab + e);
    s3 += RG_LDP(part + (long)(k + 3) * slab + e);
  }
  for (; k < splits; ++k) s0 += part[(long)k * slab + e];
  out[e] = (s0 + s1) + (s2 + s3);
}

__global__ void reduce_group_kernel(ReduceGroupArgs R) { reduce_group_body(R, (long)blockIdx.x * blockDim.x + threadIdx.x); }

struct StageGroupArgs {
  int n;
  int x3;  // also write the lo planes (bf16(w - hi)) behind the hi planes
  long begin[FB_MAXL + 1];
  const float* w[FB_MAXL];
  int N[FB_MAXL], K[FB_MAXL];
  bf16_t* wf[FB_MAXL];
  bf16_t* wb[FB_MAXL];
};

// out[c] = sum_s partials[s][c], S x N row-major: 32 columns x 8 row-groups per workgroup, each
// thread sums rows g, g+8, ... (independent loads in flight), groups combined in fixed order
__device__ __forceinline__ void reduce_cols_body(const float* __restrict__ partials, int S, int N,
                                                 float* __restrict__ out, int block) {
  __shared__ float red[8][33];
  const int c = block * 32 + (threadIdx.x & 31), g = threadIdx.x >> 5;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  if (c < N) {
    // eight rows in flight per thread: the launch is a chain of dependent HBM round trips (S = 512 rows: 8 rounds)
    int r = g;
    for (; r + 248 < S; r += 256) {  // 32 in flight (round 5), added in the order of four passes of the 8-deep loop below
      float v[32];
#pragma unroll
      for (int u = 0; u < 32; ++u) v[u] = partials[(long)(r + 8 * u) * N + c];
#pragma unroll
      for (int u = 0; u < 32; u += 8) {
        s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; s4 += v[u + 4]; s5 += v[u + 5]; s6 += v[u + 6]; s7 += v[u + 7];
      }
    }
    for (; r + 56 < S; r += 64) {
      s0 += partials[(long)r * N + c];
      s1 += partials[(long)(r + 8) * N + c];
      s2 += partials[(long)(r + 16) * N + c];
      s3 += partials[(long)(r + 24) * N + c];
      s4 += partials[(long)(r + 32) * N + c];
      s5 += partials[(long)(r + 40) * N + c];
      s6 += partials[(long)(r + 48) * N + c];
      s7 += partials[(long)(r + 56) * N + c];
    }
    for (; r < S; r += 8) s0 += partials[(long)r * N + c];
  }
  red[g][threadIdx.x & 31] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  __syncthreads();
  if (g == 0 && c < N) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x & 31];
    out[c] = t;
  }
}

// bias gradients of every layer in one launch (four ~7 us launches were 4 % of a C2 step)
struct ReduceColsGroupArgs {
  int n;
  int block_begin[FB_MAXL + 1];
  const float* partials[FB_MAXL];
  float* out[FB_MAXL];
  int N[FB_MAXL];
  int S;
};
__global__ void reduce_cols_group_kernel(ReduceColsGroupArgs G) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < FB_MAXL; ++i)
    if (i < G.n && (int)blockIdx.x >= G.block_begin[i]) l = i;
  reduce_cols_body(G.partials[l], G.S, G.N[l], G.out[l], (int)blockIdx.x - G.block_begin[l]);
}

// The weight gradient's split reduce, the bias gradients' column reduce and (optionally) one scaled sum — the mean loss
// of a step — in ONE launch: blocks [0, elem_blocks) are reduce_group_kernel's, the next cols.block_begin[cols.n] are
// reduce_cols_group_kernel's, the last one is reduce_sum_kernel's (heads.hip), each with its own arithmetic unchanged:
// bit-identical to the three launches (5.6 + 4.6 us of launch-bound tails per C2 step).
struct ReduceTailArgs {
  ReduceGroupArgs splits;
  ReduceColsGroupArgs cols;
  int elem_blocks;
  const float* sum_in;
  int sum_n;
  float sum_scale;
  float* sum_out;
};

__global__ void reduce_tail_kernel(ReduceTailArgs T) {
  const int b = blockIdx.x;
  if (b < T.elem_blocks) {
    reduce_group_body(T.splits, (long)b * blockDim.x + threadIdx.x);
    return;
  }
  const int cb = b - T.elem_blocks;
  if (cb < T.cols.block_begin[T.cols.n]) {
    int l = 0;
#pragma unroll
    for (int i = 1; i < FB_MAXL; ++i)
      if (i < T.cols.n && cb >= T.cols.block_begin[i]) l = i;
    reduce_cols_body(T.cols.partials[l], T.cols.S, T.cols.N[l], T.cols.out[l], cb - T.cols.block_begin[l]);
    return;
  }
  // reduce_sum_kernel (heads.hip): strided partial sums, block_sum_256's fixed order
  __shared__ float scratch[4];
  float acc = strided_sum_256(T.sum_in, T.sum_n, threadIdx.x);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) acc += shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) T.sum_out[0] = ((scratch[0] + scratch[1]) + (scratch[2] + scratch[3])) * T.sum_scale;
}

__global__ void reduce_splits2_kernel(const float* __restrict__ partials, long slab, int splits,
                                      float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partials[(long)k * slab + i];
  out[i] = s;
}

__global__ void stage_group_kernel(StageGroupArgs G) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= G.begin[G.n]) return;
  const float* w = G.w[0];
  int N = G.N[0], K = G.K[0];
  bf16_t* wf = G.wf[0];
  bf16_t* wb = G.wb[0];
  long base = 0;
#pragma unroll
  for (int k = 1; k < FB_MAXL; ++k)
    if (k < G.n && i >= G.begin[k]) {
      w = G.w[k]; N = G.N[k]; K = G.K[k]; wf = G.wf[k]; wb = G.wb[k]; base = G.begin[k];
    }
  stage_weight_elem(w, N, K, wf, wb, i - base, G.x3);
}

__global__ void stage_weights_frag_kernel(const float* __restrict__ w, int N, int K, bf16_t* __restrict__ wf,
                                          bf16_t* __restrict__ wb) {
  const int NTf = (N + 31) / 32, KCf = (K + 15) / 16;
  const int NTb = (K + 31) / 32, KCb = (N + 15) / 16;
  const long tf = (long)NTf * KCf * 512, tb = (long)NTb * KCb * 512;
  const long total = tf > tb ? tf : tb;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    stage_weight_elem(w, N, K, wf, wb, i);
  }
}

// ---- optimizer step fused with weight staging ----------------------------------------------------
// After the backward pass a DQN step runs Adam on the online network, the soft update of the target
// network and the bf16 re-staging of both networks' weights: four launches of ~5-8 us each for
// 600 K parameters (they are launch-bound, not bandwidth-bound).  This kernel walks the flat
// parameter slab (coalesced on the five fp32 arrays) and does all of it per element: the Adam and
// soft-update arithmetic of rg_optim.h, then, for weight elements, the bf16 value goes to its three
// fragment slots (online forward / backward, target forward; 2-byte scattered stores into 1.2 MB).
struct UpdateArgs {
  int n;
  long total;  // slab elements
  int N[FB_MAXL], K[FB_MAXL];
  long w_off[FB_MAXL], b_off[FB_MAXL];
  bf16_t* wf[FB_MAXL];
  bf16_t* wb[FB_MAXL];
  bf16_t* twf[FB_MAXL];
  float* p;
  const float* g;
  float* m;
  float* v;
  float* t;
  AdamCoef c;
  float tau, one_minus_tau;
  const double* sched;  // device-resident Adam schedule (rg_optim.h) or null: coefficients as launch arguments
  // split-bf16 stacks: every fragment buffer is [hi plane | lo plane], lo = bf16(x - hi) (stage_weight_elem); the lo
  // plane of layer l starts wfrag_elems(N, K) (forward, target) / wfrag_elems(K, N) (backward) elements in
  int x3;
  // grouped layers (qr_grouped.hip: QR-DQN's A x N output layer as A independent [Ng, K] layers): Ng[l] > 0 = the rows
  // of layer l fall into groups of Ng, group g's fragments start g * per_f[l] (forward, target) / g * per_b[l] (backward)
  // elements in — what rg_group_weights_stage writes (split-bf16: a group's set is [hi plane | lo plane], per_* covers both).
  int Ng[FB_MAXL];
  long per_f[FB_MAXL], per_b[FB_MAXL];
  // replayed steps (runtime._GraphedLoop): the sampler launch has already counted this step in sched[0]
  // (sched_pre_ticked), and this launch advances the index pool's cursor for the next one — workgroup 0, when it is
  // done; no other workgroup of this launch touches it
  int pre_ticked;
  long long* post_tick;
  int post_tick_mod;
};

// the three fragment slots of W[n][k] (online forward / backward, target forward), both planes in split-bf16 mode —
// element for element what stage_weight_elem writes
__device__ __forceinline__ void update_store_frags(const UpdateArgs& U, int l, int n, int k, float pn, float tn) {
  int N = U.N[l];
  const int K = U.K[l];
  long gf = 0, gb = 0;
  if (U.Ng[l] > 0) {  // this row's group, its row inside the group
    const int g = n / U.Ng[l];
    n -= g * U.Ng[l];
    N = U.Ng[l];
    gf = g * U.per_f[l];
    gb = g * U.per_b[l];
  }
  const int KCf = (K + 15) / 16, KCb = (N + 15) / 16;
  const long jf = gf + ((((long)(n >> 5) * KCf + (k >> 4)) * 64) + ((n & 31) + 32 * ((k & 15) >> 3))) * 8 + (k & 7);
  const long tf = (long)((N + 31) / 32) * KCf * 512, tb = (long)((K + 31) / 32) * KCb * 512;
  const bf16_t ph = f32_to_bf16(pn), th = f32_to_bf16(tn);
  if (U.wf[l]) {
    U.wf[l][jf] = ph;
    if (U.x3) U.wf[l][tf + jf] = f32_to_bf16(pn - bf16_to_f32(ph));
  }
  if (U.twf[l]) {
    U.twf[l][jf] = th;
    if (U.x3) U.twf[l][tf + jf] = f32_to_bf16(tn - bf16_to_f32(th));
  }
  if (U.wb[l]) {
    const long jb = gb + ((((long)(k >> 5) * KCb + (n >> 4)) * 64) + ((k & 31) + 32 * ((n & 15) >> 3))) * 8 + (n & 7);
    U.wb[l][jb] = ph;
    if (U.x3) U.wb[l][tb + jb] = f32_to_bf16(pn - bf16_to_f32(ph));
  }
}

__global__ void mlp_update_kernel(UpdateArgs U) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= U.total) return;
  // which tensor of the slab does element i belong to?  (alignment gaps between tensors: none)
  int l = -1, is_w = 0;
  long rel = 0;
#pragma unroll
  for (int k = 0; k < FB_MAXL; ++k) {
    if (k < U.n) {
      const long wn = (long)U.N[k] * U.K[k];
      if (i >= U.w_off[k] && i < U.w_off[k] + wn) { l = k; is_w = 1; rel = i - U.w_off[k]; }
      if (i >= U.b_off[k] && i < U.b_off[k] + U.N[k]) { l = k; is_w = 0; rel = i - U.b_off[k]; }
    }
  }
  if (l < 0) return;
  if (U.post_tick && i == 0) U.post_tick[0] = (U.post_tick[0] + 1) % U.post_tick_mod;
  const AdamCoef coef = sched_coef(U.c, U.sched, U.pre_ticked);
  float mi = U.m[i], vi = U.v[i];
  const float pn = adam_element(coef, U.p[i], U.g[i], mi, vi);
  U.p[i] = pn;
  U.m[i] = mi;
  U.v[i] = vi;
  float tn = 0.f;
  if (U.t) {
    tn = soft_update_element(U.tau, U.one_minus_tau, pn, U.t[i]);
    U.t[i] = tn;
  }
  if (!is_w) return;
  const int K = U.K[l];
  // B-fragment slot of W[n][k] (forward) and of W^T[k][n] (backward); padding slots were zeroed by
  // the first staging and are never touched
  update_store_frags(U, l, (int)(rel / K), (int)(rel % K), pn, tn);
}

// Tiled form of the update for weight matrices whose rows can be read in 32-byte pieces (in_features a
// multiple of 8, slab offset a multiple of 4): a workgroup owns a 32 (out) x 32 (in) tile, every thread
// four consecutive in-features of one row.  Against one element per thread this turns
//   * the fp32 traffic (p, g, m, v, target) into float4 requests,
//   * the forward fragments (online and target) into one 8-byte store per thread (half a fragment
//     record), and
//   * the backward fragments (W^T: 8 consecutive OUT-features of one in-feature are a record) into one
//     16-byte store per thread after a transpose through LDS, instead of 2-byte stores 16 bytes apart.
// The arithmetic is adam_element / soft_update_element on the same values: bit-identical results.
constexpr int UT_ROWS = 32, UT_COLS = 32, UT_PITCH = UT_COLS + 2;  // pitch: 17 dwords, conflict-free columns

struct UpdateTileArgs {
  UpdateArgs u;
  int tile_begin[FB_MAXL + 1];  // first workgroup of each layer's tiles; [n] = first "rest" workgroup
  int tiled[FB_MAXL];           // layer's weight handled by tiles
  long rest_begin[2 * FB_MAXL + 1];  // prefix sums of the element ranges left to the per-element path
};

__device__ __forceinline__ void update_one(const UpdateArgs& U, const AdamCoef& coef, long i, float& pn, float& tn) {
  float mi = U.m[i], vi = U.v[i];
  pn = adam_element(coef, U.p[i], U.g[i], mi, vi);
  U.p[i] = pn;
  U.m[i] = mi;
  U.v[i] = vi;
  tn = 0.f;
  if (U.t) {
    tn = soft_update_element(U.tau, U.one_minus_tau, pn, U.t[i]);
    U.t[i] = tn;
  }
}

__global__ void mlp_update_tiles_kernel(UpdateTileArgs T) {
  const UpdateArgs& U = T.u;
  __shared__ bf16_t tile[UT_ROWS * UT_PITCH];
  __shared__ bf16_t tile_lo[UT_ROWS * UT_PITCH];  // split-bf16: lo plane of the tile
  const int wg = blockIdx.x, tid = threadIdx.x;
  const AdamCoef coef = sched_coef(U.c, U.sched, U.pre_ticked);
  if (U.post_tick && wg == 0 && tid == 0) U.post_tick[0] = (U.post_tick[0] + 1) % U.post_tick_mod;
  if (wg >= T.tile_begin[U.n]) {
    // everything the tiles do not cover (biases; weights with odd shapes): one element per thread
    const long j = (long)(wg - T.tile_begin[U.n]) * blockDim.x + tid;
    int r = -1;
    for (int q = 0; q < 2 * U.n; ++q)
      if (j >= T.rest_begin[q] && j < T.rest_begin[q + 1]) r = q;
    if (r < 0) return;
    const int l = r >> 1, is_w = r & 1;
    const long rel = j - T.rest_begin[r];
    float pn, tn;
    update_one(U, coef, (is_w ? U.w_off[l] : U.b_off[l]) + rel, pn, tn);
    if (!is_w) return;
    const int K = U.K[l];
    update_store_frags(U, l, (int)(rel / K), (int)(rel % K), pn, tn);
    return;
  }
  int l = 0;
  for (int q = 1; q < U.n; ++q)
    if (wg >= T.tile_begin[q]) l = q;
  const int N = U.N[l], K = U.K[l];
  const int tiles_k = (K + UT_COLS - 1) / UT_COLS;
  const int tw = wg - T.tile_begin[l];
  const int n0 = (tw / tiles_k) * UT_ROWS, k0 = (tw % tiles_k) * UT_COLS;
  const int r = tid >> 3, c = (tid & 7) * 4;  // row of the tile, first of this thread's 4 in-features
  const int n = n0 + r, k = k0 + c;
  const bool live = n < N && k < K;  // K % 8 == 0: a piece is entirely inside or outside
  float pn[4], tn[4], pl[4] = {0.f, 0.f, 0.f, 0.f}, tl[4] = {0.f, 0.f, 0.f, 0.f};
  if (live) {
    const long i = U.w_off[l] + (long)n * K + k;
    f32x4 P = *(const f32x4*)(U.p + i), G = *(const f32x4*)(U.g + i), M = *(const f32x4*)(U.m + i);
    f32x4 V = *(const f32x4*)(U.v + i), Tg = U.t ? *(const f32x4*)(U.t + i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float mi = M[e], vi = V[e];
      pn[e] = adam_element(coef, P[e], G[e], mi, vi);
      P[e] = pn[e];
      M[e] = mi;
      V[e] = vi;
      tn[e] = 0.f;
      if (U.t) {
        tn[e] = soft_update_element(U.tau, U.one_minus_tau, pn[e], Tg[e]);
        Tg[e] = tn[e];
      }
    }
    *(f32x4*)(U.p + i) = P;
    *(f32x4*)(U.m + i) = M;
    *(f32x4*)(U.v + i) = V;
    if (U.t) *(f32x4*)(U.t + i) = Tg;
    const int KCf = (K + 15) / 16;
    int nl = n;  // row inside its group (grouped layer) / the row itself
    long gf = 0;
    if (U.Ng[l] > 0) {
      const int g = n / U.Ng[l];
      nl = n - g * U.Ng[l];
      gf = g * U.per_f[l];
    }
    const long jf = gf + ((((long)(nl >> 5) * KCf + (k >> 4)) * 64) + ((nl & 31) + 32 * ((k & 15) >> 3))) * 8 + (k & 7);
    // lo plane of the forward fragments (split-bf16): behind the layer's — a grouped layer: the group's — hi plane
    const long tf = (long)(((U.Ng[l] > 0 ? U.Ng[l] : N) + 31) / 32) * KCf * 512;
    if (U.wf[l]) *(uint2*)(U.wf[l] + jf) = uint2{pack_bf16x2(pn[0], pn[1]), pack_bf16x2(pn[2], pn[3])};
    if (U.twf[l]) *(uint2*)(U.twf[l] + jf) = uint2{pack_bf16x2(tn[0], tn[1]), pack_bf16x2(tn[2], tn[3])};
    if (U.x3) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pl[e] = pn[e] - bf16_to_f32(f32_to_bf16(pn[e]));
        tl[e] = tn[e] - bf16_to_f32(f32_to_bf16(tn[e]));
      }
      if (U.wf[l]) *(uint2*)(U.wf[l] + tf + jf) = uint2{pack_bf16x2(pl[0], pl[1]), pack_bf16x2(pl[2], pl[3])};
      if (U.twf[l]) *(uint2*)(U.twf[l] + tf + jf) = uint2{pack_bf16x2(tl[0], tl[1]), pack_bf16x2(tl[2], tl[3])};
    }
  }
  if (!U.wb[l]) return;  // workgroup-uniform
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    tile[r * UT_PITCH + c + e] = live ? f32_to_bf16(pn[e]) : (bf16_t)0;
    if (U.x3) tile_lo[r * UT_PITCH + c + e] = live ? f32_to_bf16(pl[e]) : (bf16_t)0;
  }
  __syncthreads();
  // W^T fragments: thread -> (in-feature kk, group of 8 out-features); a record = 8 consecutive n
  const int kk = tid & 31, ng = tid >> 5;
  const int kt = k0 + kk, nt = n0 + ng * 8;
  if (ng < UT_ROWS / 8 && kt < K && nt < N) {  // N may end inside a record: those slots are padding and stay zero-written
    unsigned short h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (nt + e < N) ? tile[(ng * 8 + e) * UT_PITCH + kk] : (bf16_t)0;
    // grouped layer (Ng % 8 == 0, checked by the host): a record of 8 out-features lies inside one group
    const int grp = U.Ng[l] > 0 ? nt / U.Ng[l] : 0;
    const int ntl = nt - grp * (U.Ng[l] > 0 ? U.Ng[l] : 0);
    const int KCb = ((U.Ng[l] > 0 ? U.Ng[l] : N) + 15) / 16;
    const long jb = grp * (U.Ng[l] > 0 ? U.per_b[l] : 0) +
                    ((((long)(kt >> 5) * KCb + (ntl >> 4)) * 64) + ((kt & 31) + 32 * ((ntl & 15) >> 3))) * 8;
    *(u32x4*)(U.wb[l] + jb) = u32x4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                                    (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
    if (U.x3) {
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (nt + e < N) ? tile_lo[(ng * 8 + e) * UT_PITCH + kk] : (bf16_t)0;
      const long tb = (long)((K + 31) / 32) * KCb * 512;
      *(u32x4*)(U.wb[l] + tb + jb) = u32x4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                                           (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
    }
  }
}

// the three (hidden width, pitch) instantiations of a fused kernel template
#define RG_LAUNCH_FUSED(KERNEL, hidden, pitch, grid, lds, stream, args)                                      \
  do {                                                                                                       \
    const dim3 block_(FB_NW * 64);                                                                           \
    if ((hidden) == 256 && (pitch) == 264) {                                                                 \
      RG_ALLOW_LDS((KERNEL<256 / (32 * FB_NW), FB_NW, 264>), lds);                                           \
      RG_LAUNCH_DYN((KERNEL<256 / (32 * FB_NW), FB_NW, 264>), grid, block_, lds, (hipStream_t)stream, args); \
    } else if ((hidden) == 256) {                                                                            \
      RG_ALLOW_LDS((KERNEL<256 / (32 * FB_NW), FB_NW, 520>), lds);                                           \
      RG_LAUNCH_DYN((KERNEL<256 / (32 * FB_NW), FB_NW, 520>), grid, block_, lds, (hipStream_t)stream, args); \
    } else {                                                                                                 \
      RG_ALLOW_LDS((KERNEL<512 / (32 * FB_NW), FB_NW, 520>), lds);                                           \
      RG_LAUNCH_DYN((KERNEL<512 / (32 * FB_NW), FB_NW, 520>), grid, block_, lds, (hipStream_t)stream, args); \
    }                                                                                                        \
  } while (0)

}  // namespace rg

namespace rg {
static int padded_wgs(const rg_mlp_desc* d, int batch) {
  const int bm = d->x3 ? X3_BM : FB_BM;
  return (batch + 127) / 128 * (128 / bm);
}
}  // namespace rg

using namespace rg;

extern "C" {

int rg_mlp_fused_supported(const rg_mlp_desc* d) { return fused_supported(d) > 0; }

size_t rg_frag_elems(int rows, int cols) {
  return (size_t)((rows + 127) / 128 * 128) * (size_t)((cols + 31) / 32 * 32);
}

size_t rg_sign_bytes(int rows, int cols) { return rg_frag_elems(rows, cols) / 8; }

size_t rg_wfrag_elems(int out_features, int in_features) {
  return (size_t)((out_features + 31) / 32) * (size_t)((in_features + 15) / 16) * 512;
}

int rg_stage_weights_frag(const float* w, int out_features, int in_features, void* wfrag_fwd, void* wfrag_bwd,
                          rg_stream_t stream) {
  if (!w || out_features <= 0 || in_features <= 0 || (!wfrag_fwd && !wfrag_bwd)) return RG_EINVAL;
  const size_t tf = rg_wfrag_elems(out_features, in_features), tb = rg_wfrag_elems(in_features, out_features);
  const size_t total = tf > tb ? tf : tb;
  long blocks = (long)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  RG_LAUNCH(stage_weights_frag_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, w, out_features,
            in_features, (bf16_t*)wfrag_fwd, (bf16_t*)wfrag_bwd);
  return (int)hipGetLastError();
}

int rg_mlp_forward_fused(const rg_mlp_desc* d, const void* x, int x_dtype, int64_t ldx, int batch, float* out32,
                         int64_t ldo, int save, rg_stream_t stream) {
  const int tn = fused_supported(d);
  if (!tn) return RG_EUNSUPPORTED;
  if (!x || !out32 || batch <= 0) return RG_EINVAL;
  MlpArgs a;
  int rc = fill_args(d, batch, a, 0);
  if (rc) return rc;
  if (save < 0 || save > 2) return RG_EINVAL;
  if (save)
    for (int l = (save == 2 ? 1 : 0); l < d->n_layers; ++l)
      if (!d->act_frag[l]) return RG_EINVAL;  // (save = 2 writes it only where there is no usable sign plane)
  if (d->rowmap && (d->x2 || (d->x3 && !d->tile_key) || (batch % 128) != 0)) return RG_EUNSUPPORTED;
  if (d->tile_key && (!d->rowmap || !d->row_begin || d->n_groups <= 0)) return RG_EINVAL;
  if (d->x2 && (d->x_split <= 0 || d->x_split >= d->dims[0] || (d->x_split % 32) != 0)) return RG_EINVAL;
  a.x = x; a.ldx = ldx; a.x_is_f32 = (x_dtype == RG_DT_F32); a.out32 = out32; a.ldo = ldo; a.save = save;
  if (d->x3) return x3_forward_launch(d, a, (hipStream_t)stream);
  size_t lds = (size_t)FB_BM * a.pitch * sizeof(bf16_t);
  const int n_tiles = (batch + FB_BM - 1) / FB_BM;
  const dim3 grid(d->tile_key ? (n_tiles + 7) / 8 * 8 : n_tiles);  // grouped: whole eighths of the tile list (grouped_tile)
  a.stage_out = a.out_lds = 0;
  {
    // a thin output layer (one column tile, the K-split path of the 8-wave kernel) reads its weights from LDS
    const int L = d->n_layers, KCo = (d->dims[L - 1] + 15) / 16;
    if (RG_OUT_LDS && !d->tile_key && L >= 2 && d->dims[L] <= 16 && FB_NW == 8 && KCo >= 8 && (KCo & 1) == 0 &&
        lds + (size_t)KCo * 512 <= 160 * 1024) {
      a.out_lds = 1;
      lds += (size_t)KCo * 512;
    }
  }
  if (d->tile_key) {
    // a wide grouped output leaves as whole rows through a staging area behind the activation tile (mlp_fwd_fused_body)
    const int No = d->dims[d->n_layers], NTo = (No + 31) / 32;
    const size_t stage = (size_t)32 * (NTo * 32 + 4) * sizeof(float);
    if (No > 64 && (No & 3) == 0 && (ldo & 3) == 0 && (((uintptr_t)out32) & 15) == 0 && lds + stage <= 160 * 1024) {
      a.stage_out = 1;
      lds += stage;
    }
  }
  if (d->tile_key) RG_LAUNCH_FUSED(mlp_fwd_grouped_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
#if RG_FWD_SWAP
  else if (save == 0) RG_LAUNCH_FUSED(mlp_fwd_swap_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
#endif
  else RG_LAUNCH_FUSED(mlp_fwd_fused_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
  return (int)hipGetLastError();
}

size_t rg_mlp_backward_fused_workspace_bytes(const rg_mlp_desc* d, int batch) {
  if (!d || batch <= 0) return 0;
  size_t cols = 0;
  for (int l = 0; l < d->n_layers; ++l) cols += (size_t)d->dims[l + 1];
  // (a grouped output layer's partial rows are indexed workgroup + group: n_groups more rows of the LAST block)
  const size_t extra = d->n_groups > 0 ? (size_t)d->n_groups * d->dims[d->n_layers] : 0;
  return ((size_t)padded_wgs(d, batch) * cols + extra) * sizeof(float);
}

int rg_mlp_backward_fused(const rg_mlp_desc* d, const float* dout32, int64_t lddo, int batch, float* dx32,
                          int64_t lddx, void* workspace, size_t workspace_bytes, rg_stream_t stream) {
  // also a TRUNK: every layer hidden-wide, the "output" being the last hidden layer (dout32 = the gradient of its
  // pre-activation, [batch, H]) — what the grouped output layer of qr_grouped.hip hands back
  if (!fused_supported(d) && !fused_trunk(d)) return RG_EUNSUPPORTED;
  if (!dout32 || batch <= 0) return RG_EINVAL;
  MlpArgs a;
  int rc = fill_args(d, batch, a, 1);
  if (rc) return rc;
  bool want_db = false;
  for (int l = 0; l < d->n_layers; ++l) {
    if (!d->dx_only && !d->dz_frag[l]) return RG_EINVAL;
    if (l >= 1 && !d->act_frag[l]) return RG_EINVAL;
    if (d->db[l]) want_db = true;
  }
  if (dx32 && !d->wfrag_bwd[0]) return RG_EINVAL;
  if (d->dx_only && (!dx32 || want_db || d->tile_key)) return RG_EINVAL;
  if (d->tile_key && (!d->row_begin || d->n_groups <= 0 || d->dims[d->n_layers] > 256)) return RG_EINVAL;
  if (d->dx_col0 < 0 || d->dx_col0 >= d->dims[0] || (d->dx_col0 % 32) != 0) return RG_EINVAL;
  const int n_wg = padded_wgs(d, batch);
  if (want_db) {
    if (!workspace || workspace_bytes < rg_mlp_backward_fused_workspace_bytes(d, batch)) return RG_EWORKSPACE;
    float* p = (float*)workspace;
    for (int l = 0; l < d->n_layers; ++l) {
      a.db_part[l] = d->db[l] ? p : nullptr;
      p += (size_t)n_wg * d->dims[l + 1];
    }
  }
  a.dout32 = dout32; a.lddo = lddo; a.dx32 = dx32; a.lddx = lddx;
  if (d->x3) {
    rc = x3_backward_launch(d, a, (hipStream_t)stream);
  } else {
    size_t lds = (size_t)FB_BM * a.pitch * sizeof(bf16_t);
    const dim3 grid(d->tile_key ? (n_wg + 7) / 8 * 8 : n_wg);
    if (d->tile_key && a.pitch < 2 * 256 + 8) lds *= 2;  // a boundary tile's masked dZ copy lives behind a 264-wide tile
    if (d->dx_only) RG_LAUNCH_FUSED(mlp_bwd_dx_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
    else if (d->tile_key) RG_LAUNCH_FUSED(mlp_bwd_grouped_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
    else RG_LAUNCH_FUSED(mlp_bwd_fused_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
    rc = (int)hipGetLastError();
  }
  if (rc) return rc;
  if (d->defer_db) {  // the partials stay in the workspace: rg_mlp_wgrad_fused (db_partials) sums them in its reduce launch
    if (!want_db) return RG_EINVAL;
    // (a grouped output layer's per-group sums are not that launch's: reduced here; the trunk's partials wait for the
    // trunk's weight gradient, whose descriptor is this one's first n_layers - 1 layers — same workspace layout)
    const int l = d->n_layers - 1;
    if (d->tile_key && a.db_part[l])
      grouped_bias_reduce_launch(a.db_part[l], d->row_begin, d->n_groups, d->dims[l + 1], d->db[l], d->x3 ? X3_BM : FB_BM,
                                 (hipStream_t)stream);
    return (int)hipGetLastError();
  }
  ReduceColsGroupArgs G;
  G.n = 0;
  G.S = n_wg;
  int blocks = 0;
  for (int l = 0; l < d->n_layers; ++l) {
    if (!a.db_part[l]) continue;
    if (d->tile_key && l == d->n_layers - 1) {  // grouped output layer: per-group sums over each group's segments
      // (a workgroup covers 128 rows of the grouped space in the bf16 kernel, 64 in the split-bf16 kernel)
      grouped_bias_reduce_launch(a.db_part[l], d->row_begin, d->n_groups, d->dims[l + 1], d->db[l], d->x3 ? X3_BM : FB_BM,
                                 (hipStream_t)stream);
      continue;
    }
    const int i = G.n++;
    G.block_begin[i] = blocks;
    G.partials[i] = a.db_part[l];
    G.out[i] = d->db[l];
    G.N[i] = d->dims[l + 1];
    blocks += (d->dims[l + 1] + 31) / 32;
  }
  for (int i = G.n; i <= FB_MAXL; ++i) G.block_begin[i] = blocks;
  for (int i = G.n; i < FB_MAXL; ++i) { G.partials[i] = nullptr; G.out[i] = nullptr; G.N[i] = 0; }
  if (G.n > 0) RG_LAUNCH(reduce_cols_group_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, G);
  return (int)hipGetLastError();
}

struct WgradFragPlan {
  int NTa, NTb, MB, splits, mb_per_split;
  long slab;
  int shape, tiles;  // workgroup tile shape (WG_SHAPE_*) and the number of such tiles that cover dW
};
// RG_WGRAD_SHAPES = 0: every layer on 256 x 256 tiles (rounds 1-3; same-box A/B switch)
#ifndef RG_WGRAD_SHAPES
#define RG_WGRAD_SHAPES 1
#endif
// workgroups per layer whose dW is ONE tile of its shape (dW0, thin output layers): their partial slab is the whole dW, so
// the 128 of the multi-tile layers would double the partial bytes they had as two tiles x 64 splits
#ifndef RG_WGRAD_TARGET_THIN
#define RG_WGRAD_TARGET_THIN 64
#endif
// the shape that stages the fewest bytes for an NTa x NTb-tile dW: groups x 16-byte units per thread and 32-row block
static int wgrad_pick_shape(int NTa, int NTb, int x3, int* tiles_out) {
  static const int dma[WG_N_SHAPES] = {WgS8x8::DMA, WgS16x4::DMA, WgS4x16::DMA, WgS2x16::DMA, WgS1x16::DMA};
  int best = WG_SHAPE_8x8, best_cost = 0, best_tiles = 0;
  for (int sh = 0; sh < WG_N_SHAPES; ++sh) {
    if (sh != WG_SHAPE_8x8 && !RG_WGRAD_SHAPES) continue;
    (void)x3;  // (the split-bf16 core takes the same shapes: its stage is the same bytes)
    const int ga = wgrad_shape_ga(sh), gb = wgrad_shape_gb(sh);
    const int tiles = ((NTa + ga - 1) / ga) * ((NTb + gb - 1) / gb);
    const int cost = tiles * dma[sh];
    if (sh == WG_SHAPE_8x8 || cost < best_cost) { best = sh; best_cost = cost; best_tiles = tiles; }
  }
  *tiles_out = best_tiles;
  return best;
}
static WgradFragPlan wgrad_frag_plan(int out_f, int in_f, int batch, int x3 = 0) {
  WgradFragPlan p;
  p.NTa = (out_f + 31) / 32;
  p.NTb = (in_f + 31) / 32;
  p.MB = (batch + 127) / 128 * 4;
  p.shape = wgrad_pick_shape(p.NTa, p.NTb, x3, &p.tiles);
  const int tiles = p.tiles;
  int want = (256 + tiles - 1) / tiles;  // ~one workgroup per CU
  const int max_splits = (p.MB + WG_MB_STAGE - 1) / WG_MB_STAGE;
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  int per = (p.MB + want - 1) / want;
  per = (per + WG_MB_STAGE - 1) / WG_MB_STAGE * WG_MB_STAGE;
  p.mb_per_split = per;
  p.splits = (p.MB + per - 1) / per;
  p.slab = (long)out_f * in_f;
  return p;
}

size_t rg_fc_wgrad_frag_workspace_bytes(int out_features, int in_features, int batch) {
  const WgradFragPlan p = wgrad_frag_plan(out_features, in_features, batch > 0 ? batch : 1);
  return (size_t)p.splits * p.slab * sizeof(float);
}

int rg_fc_wgrad_frag(const void* dz_frag, const void* x_frag, int out_features, int in_features, int batch,
                     float* dw, void* workspace, size_t workspace_bytes, rg_stream_t stream) {
  if (!dz_frag || !x_frag || !dw || out_features <= 0 || in_features <= 0 || batch <= 0) return RG_EINVAL;
  const WgradFragPlan p = wgrad_frag_plan(out_features, in_features, batch);
  if (!workspace || workspace_bytes < (size_t)p.splits * p.slab * sizeof(float)) return RG_EWORKSPACE;
  WgradFragArgs g;
  g.a_frag = (const bf16_t*)dz_frag; g.b_frag = (const bf16_t*)x_frag;
  g.NTa = p.NTa; g.NTb = p.NTb; g.MB = p.MB; g.mb_base = 0; g.mb_per_split = p.mb_per_split; g.splits = p.splits;
  g.partial = (float*)workspace; g.slab = p.slab; g.N = out_features; g.K = in_features;
  g.x3 = 0; g.a_lo = g.b_lo = 0; g.shape = p.shape; g.part_mode = 0; g.part_nt = 0;
  const int grid = p.tiles * ((p.splits + 7) / 8 * 8);
  const size_t lds = (size_t)WG_SHAPED_LDS;
  RG_ALLOW_LDS(wgrad_frag_kernel, lds);
  RG_LAUNCH_DYN(wgrad_frag_kernel, dim3(grid), dim3(WG_THREADS), lds, (hipStream_t)stream, g);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  const long n = p.slab;
  RG_LAUNCH(reduce_splits2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)stream,
            (const float*)g.partial, p.slab, p.splits, dw, n);
  return (int)hipGetLastError();
}

size_t rg_group_head_wgrad_workspace_bytes(int n_groups, int group_rows, int in_features, int splits) {
  return (size_t)n_groups * splits * group_rows * in_features * sizeof(float);
}

/* dw [n_groups * group_rows, in_features] of a grouped layer (qr_grouped.hip): group g's rows are the 32-row blocks
 * [row_begin[g] / 32, ceil(row_begin[g + 1] / 32)) of h_frag and the same blocks + g of dz_frag (wgrad_grouped_kernel) */
int rg_group_head_wgrad(const void* dz_frag, const void* h_frag, const int32_t* row_begin, int n_groups, int group_rows,
                        int in_features, int splits, int x3, int rows, float* dw, void* workspace, size_t workspace_bytes,
                        rg_stream_t stream) {
  if (!dz_frag || !h_frag || !row_begin || !dw || n_groups <= 0 || group_rows <= 0 || in_features <= 0 || splits <= 0 ||
      (x3 && rows <= 0))
    return RG_EINVAL;
  if (group_rows > 256) return RG_EUNSUPPORTED;  // one n-group of the 256 x 256 workgroup tile per action
  if (!workspace || workspace_bytes < rg_group_head_wgrad_workspace_bytes(n_groups, group_rows, in_features, splits))
    return RG_EWORKSPACE;
  WgradGroupedArgs G;
  G.g.a_frag = (const bf16_t*)dz_frag; G.g.b_frag = (const bf16_t*)h_frag;
  G.g.NTa = (group_rows + 31) / 32; G.g.NTb = (in_features + 31) / 32; G.g.MB = 0; G.g.mb_base = 0; G.g.mb_per_split = 0; G.g.splits = splits;
  G.g.partial = (float*)workspace; G.g.slab = (long)group_rows * in_features; G.g.N = group_rows; G.g.K = in_features;
  // split-bf16: each operand is [hi plane | lo plane] over the `rows` rows of the grouped space (rg_frag_elems apart)
  G.g.x3 = x3 ? 1 : 0;
  G.g.a_lo = x3 ? (long)frag_elems(grouped_dz_rows(rows, n_groups), group_rows) : 0;
  G.g.b_lo = x3 ? (long)frag_elems(rows, in_features) : 0;
  G.g.shape = WG_SHAPE_8x8; G.g.part_mode = 0; G.g.part_nt = 0;
  G.row_begin = row_begin; G.n_groups = n_groups; G.splits = splits;
  const int k_groups = (G.g.NTb + 7) / 8;
  const size_t lds = (size_t)WgS8x8::LDS_BYTES;
  RG_ALLOW_LDS(wgrad_grouped_kernel, lds);
  RG_LAUNCH_DYN(wgrad_grouped_kernel, dim3(n_groups * splits * k_groups), dim3(WG_THREADS), lds, (hipStream_t)stream, G);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  const long n = G.g.slab * n_groups;
  RG_LAUNCH(reduce_grouped_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (hipStream_t)stream,
            (const float*)workspace, G.g.slab, splits, n_groups, dw);
  return (int)hipGetLastError();
}

/* all layers' weights of a stack -> fragment order in one launch */
int rg_mlp_stage_weights_fused(const rg_mlp_desc* d, int need_bwd, rg_stream_t stream) {
  if (!d || d->n_layers < 1 || d->n_layers > FB_MAXL) return RG_EINVAL;
  StageGroupArgs G;
  G.n = d->n_layers;
  G.x3 = d->x3;
  long off = 0;
  for (int l = 0; l < FB_MAXL; ++l) {
    G.begin[l] = off;
    if (l < d->n_layers) {
      if (!d->w[l] || !d->wfrag_fwd[l]) return RG_EINVAL;
      const int N = d->dims[l + 1], K = d->dims[l];
      G.w[l] = d->w[l]; G.N[l] = N; G.K[l] = K;
      G.wf[l] = (bf16_t*)d->wfrag_fwd[l];
      G.wb[l] = need_bwd ? (bf16_t*)d->wfrag_bwd[l] : nullptr;
      const size_t tf = rg_wfrag_elems(N, K), tb = G.wb[l] ? rg_wfrag_elems(K, N) : 0;
      off += (long)(tf > tb ? tf : tb);
    } else {
      G.w[l] = nullptr; G.N[l] = G.K[l] = 0; G.wf[l] = G.wb[l] = nullptr;
    }
  }
  G.begin[FB_MAXL] = off;
  for (int l = d->n_layers; l <= FB_MAXL; ++l) G.begin[l] = off;
  RG_LAUNCH(stage_group_kernel, dim3((unsigned)((off + 255) / 256)), dim3(256), (hipStream_t)stream, G);
  return (int)hipGetLastError();
}

// ---- how many splits per layer --------------------------------------------------------------------------------------
// Round 4 (profiles/microbench/out/r04a: hbm_roof.txt, wgrad_model.txt).  The staging mechanism alone — the LDS-DMA ring of
// one workgroup per CU — draws 7.2 TB/s from HBM when every workgroup streams its own bytes (28 B/ns per CU) and 11.7 TB/s
// into LDS (46 B/ns per CU, 5.9 TB/s of unique bytes) when the tiles of a split sit on one XCD and find each other's operand
// in its L2; LDS fragment reads and MFMAs cost nothing on top, the partial tiles do: 67 MB of them, written when the
// workgroups of a round finish together, are 20 us.  The launch of rounds 1-3 gave every layer 128 workgroups: 512 of
// uneven length (a dW0 workgroup half as long as a hidden layer's) in dispatch order — the CU that drew dW0 then a hidden
// layer finished last, at 3/2 of a balanced schedule — and 86 MB of partials.  Balanced plan: the splits of each layer are
// chosen so that ALL workgroups of the launch are one round of the chip (RG_WGRAD_TOTAL, default = the CU count) and
// take the same time by the rates above: fewer partial bytes, no tail.  (Round 3's "256 in all" experiment lost because a
// split count that is not a multiple of 8 fell off the XCD-grouped decode: every byte then came from HBM twice.)
struct WgradTuning { int balanced, total; double shared, unshared; int thin, long_first, bf16_part, uneven, part_nt; };
static const WgradTuning& wgrad_tuning() {
  static const WgradTuning t = [] {
    WgradTuning v{0, 0, 46.0, 28.0, RG_WGRAD_TARGET_THIN, 1, RG_WGRAD_BF16_PART, RG_WGRAD_UNEVEN, RG_WGRAD_PART_NT};
    if (const char* e = getenv("RG_WGRAD_PART_NT")) v.part_nt = atoi(e);
    if (const char* e = getenv("RG_WGRAD_BF16_PART")) v.bf16_part = atoi(e);
    if (const char* e = getenv("RG_WGRAD_UNEVEN")) v.uneven = atoi(e);
    if (const char* e = getenv("RG_WGRAD_THIN")) v.thin = atoi(e);
    if (const char* e = getenv("RG_WGRAD_ORDER")) v.long_first = atoi(e);
    if (const char* e = getenv("RG_WGRAD_PLAN")) v.balanced = (e[0] == 'b');  // "balanced": one round by the cost model
    if (const char* e = getenv("RG_WGRAD_TOTAL")) v.total = atoi(e);
    if (const char* e = getenv("RG_WGRAD_SHARED")) v.shared = atof(e);
    if (const char* e = getenv("RG_WGRAD_UNSHARED")) v.unshared = atof(e);
    if (v.total <= 0) {
      int dev = 0;
      hipDeviceProp_t pr;
      v.total = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0)
                    ? pr.multiProcessorCount : 256;
    }
    return v;
  }();
  return t;
}

static WgradFragPlan wgrad_group_plan(int out_f, int in_f, int batch, int target_wgs, int x3 = 0) {
  WgradFragPlan p = wgrad_frag_plan(out_f, in_f, batch, x3);
  const int tiles = p.tiles;
  if (tiles == 1 && p.shape != WG_SHAPE_8x8 && wgrad_tuning().thin < target_wgs) target_wgs = wgrad_tuning().thin;
  int want = (target_wgs + tiles - 1) / tiles;
  const int max_splits = (p.MB + WG_MB_STAGE - 1) / WG_MB_STAGE;
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  if (want >= 8) want = want / 8 * 8;
  int per = (p.MB + want - 1) / want;
  per = (per + WG_MB_STAGE - 1) / WG_MB_STAGE * WG_MB_STAGE;
  p.mb_per_split = per;
  p.splits = (p.MB + per - 1) / per;
  return p;
}

static void wgrad_stack_plan(const rg_mlp_desc* d, int batch, WgradFragPlan* out) {
  const WgradTuning& T = wgrad_tuning();
  if (!T.balanced) {
    for (int l = 0; l < d->n_layers; ++l) out[l] = wgrad_group_plan(d->dims[l + 1], d->dims[l], batch, RG_WGRAD_TARGET, d->x3);
    return;
  }
  static const int dma[WG_N_SHAPES] = {WgS8x8::DMA, WgS16x4::DMA, WgS4x16::DMA, WgS2x16::DMA, WgS1x16::DMA};
  double work[FB_MAXL], sum = 0.0;
  for (int l = 0; l < d->n_layers; ++l) {
    out[l] = wgrad_frag_plan(d->dims[l + 1], d->dims[l], batch, d->x3);
    const double stage_bytes = dma[out[l].shape] * WG_THREADS * 16.0;
    work[l] = (double)out[l].tiles * out[l].MB * stage_bytes / (out[l].tiles > 1 ? T.shared : T.unshared);
    sum += work[l];
  }
  for (int l = 0; l < d->n_layers; ++l) {
    WgradFragPlan& p = out[l];
    int want = (int)(T.total * work[l] / sum / p.tiles + 0.5);
    if (want > p.MB) want = p.MB;
    if (want < 1) want = 1;
    const int per = (p.MB + want - 1) / want;
    p.mb_per_split = per;
    p.splits = (p.MB + per - 1) / per;
  }
}

// floats a split's slab takes in the stack launch: N * K row-major, or NTa * NTb bf16 tile records of 2 KB (a thin layer's
// padded tiles can be the larger)
static long wgrad_slab_floats(const WgradFragPlan& p) {
  const long tiles = (long)p.NTa * p.NTb * 512;
  return tiles > p.slab ? tiles : p.slab;
}

size_t rg_mlp_wgrad_fused_workspace_bytes(const rg_mlp_desc* d, int batch) {
  if (!d || batch <= 0 || d->n_layers < 1 || d->n_layers > FB_MAXL) return 0;
  WgradFragPlan plan[FB_MAXL];
  wgrad_stack_plan(d, batch, plan);
  size_t total = 0;  // in floats; a layer's slab is the larger of its two partial forms (WgradFragArgs.part_mode)
  for (int l = 0; l < d->n_layers; ++l) total += (size_t)plan[l].splits * wgrad_slab_floats(plan[l]);
  return total * sizeof(float);
}

/* dw[l] = dz_frag[l]^T act_frag[l] for every layer, one wgrad launch + one reduce launch */
int rg_mlp_wgrad_fused(const rg_mlp_desc* d, int batch, void* workspace, size_t workspace_bytes,
                       rg_stream_t stream) {
  if (!d || batch <= 0 || d->n_layers < 1 || d->n_layers > FB_MAXL) return RG_EINVAL;
  if (!workspace || workspace_bytes < rg_mlp_wgrad_fused_workspace_bytes(d, batch)) return RG_EWORKSPACE;
  WgradGroupArgs G;
  ReduceGroupArgs R;
  G.n = R.n = d->n_layers;
  WgradFragPlan plan[FB_MAXL];
  wgrad_stack_plan(d, batch, plan);
  float* part = (float*)workspace;
  int wg = 0;
  long el = 0;
  // bf16 stacks: the splits' partial tiles as bf16 in accumulator order (half the bytes written here and read by the
  // reduce; error 2^-9 of a PARTIAL sum, far inside what bf16 operands cost the gradient).  Split-bf16 stacks: fp32.
  const int part_mode = (!d->x3 && wgrad_tuning().bf16_part) ? 1 : 0;
  // ---- entries of the launch.  An entry is a layer, or one of the two CLASSES of splits of an unevenly split layer.
  // Round 4: the layers whose workgroups run longest go first (workgroups are dispatched in id order, one per CU: with dW0's
  // short ones first the launch ended at 3/2 of a balanced schedule).  Round 5: that order still leaves a staircase — at C2
  // 256 hidden-layer workgroups of 64 blocks take every CU, then the 128 single-tile ones (dW0, the output layer: 32 blocks)
  // run on half of the chip while the other half idles: 96 block times for 80 of work per CU.  The splits of the multi-tile
  // layers are therefore cut UNEVENLY: a fraction f = (single-tile workgroups) / (multi-tile workgroups) of each layer's
  // splits is shorter by what a single-tile workgroup costs (b blocks, weighted by RG_WGRAD_UNEVEN percent: its stage is a
  // single-reader stream, dearer per block), L1 = L - (1 - f) b, the others longer, L2 = L + f b.  Launch order L2 | L1 |
  // single-tile: the CUs that drew an L1 workgroup are the ones that free up for a single-tile one, and every CU ends at ~L2.
  struct Entry { int layer, mb_base, mb_end, per, splits, split_base; };
  Entry ent[WG_MAXV];
  int n_ent = 0;
  {
    const WgradTuning& T = wgrad_tuning();
    // Applies to the launch it was measured on: every multi-tile layer's workgroups are ONE round of the chip and only
    // single-tile layers follow (C2's stack, either precision: 101 -> 94.7 us, split-bf16 221 -> 203).  Measured and NOT
    // extended (round 5, same box): counting a narrower multi-tile first layer among the followers (C4's critic, 512 x 288:
    // 116 -> 119 us), and any launch that shares the chip with another one (C3's trunk beside the head's weight gradient on
    // the second stream: 113.6 -> 122.6 us — the dispatch order this plan leans on is then not the launch's own; such
    // callers set rg_mlp_desc.wgrad_flags & 1).
    int n_multi = 0, n_single = 0;
    double single_blocks = 0.0;
    for (int l = 0; l < d->n_layers; ++l) {
      if (plan[l].tiles > 1) n_multi += plan[l].tiles * plan[l].splits;
      else { n_single += plan[l].splits; single_blocks += (double)plan[l].splits * plan[l].mb_per_split; }
    }
    const bool uneven = T.uneven > 0 && !T.balanced && !(d->wgrad_flags & 1) && n_multi == T.total && n_single > 0 &&
                        n_single <= n_multi && d->n_layers + 2 <= WG_MAXV;
    const double f = uneven ? (double)n_single / n_multi : 0.0;
    const double b = uneven ? single_blocks / n_single * T.uneven / 100.0 : 0.0;
    for (int l = 0; l < d->n_layers; ++l) {
      const WgradFragPlan& p = plan[l];
      int s_short = (uneven && p.tiles > 1) ? ((int)(f * p.splits + 0.5) + 4) / 8 * 8 : 0;  // whole XCD rows of splits (wgrad_frag_body)
      // (an entry is kept in reserve for every layer still to come: the table has WG_MAXV slots)
      if (s_short <= 0 || s_short >= p.splits || n_ent + 2 + (d->n_layers - 1 - l) > WG_MAXV) s_short = 0;
      if (!s_short) {
        ent[n_ent++] = Entry{l, 0, p.MB, p.mb_per_split, p.splits, 0};
        continue;
      }
      const int s_long = p.splits - s_short;
      const double fl = (double)s_short / p.splits;
      int L1 = (int)((double)p.MB / p.splits - (1.0 - fl) * b + 0.5);
      if (L1 < 1) L1 = 1;
      int L2 = (p.MB - s_short * L1 + s_long - 1) / s_long;
      const int cut = s_long * L2 < p.MB ? s_long * L2 : p.MB;
      L1 = (p.MB - cut + s_short - 1) / s_short;  // the short class covers exactly what is left
      ent[n_ent++] = Entry{l, 0, cut, L2, s_long, 0};
      ent[n_ent++] = Entry{l, cut, p.MB, L1 > 0 ? L1 : 1, s_short, s_long};
    }
    if (T.long_first)
      for (int i = 1; i < n_ent; ++i)  // stable insertion sort by descending blocks per split
        for (int j = i; j > 0 && ent[j].per > ent[j - 1].per; --j) { const Entry t = ent[j]; ent[j] = ent[j - 1]; ent[j - 1] = t; }
  }
  if (getenv("RG_WGRAD_DEBUG")) {  // the launch plan, once per distinct shape (diagnostics: profiles/scripts)
    static int shown_batch = -1, shown_layers = -1;
    if (shown_batch != batch || shown_layers != d->n_layers) {
      shown_batch = batch; shown_layers = d->n_layers;
      for (int j = 0; j < n_ent; ++j)
        fprintf(stderr, "rg_mlp_wgrad_fused: entry %d layer %d (dW %d x %d, %d tile%s) blocks [%d, %d) in %d splits of %d\n", j, ent[j].layer,
                d->dims[ent[j].layer + 1], d->dims[ent[j].layer], plan[ent[j].layer].tiles, plan[ent[j].layer].tiles > 1 ? "s" : "",
                ent[j].mb_base, ent[j].mb_end, ent[j].splits, ent[j].per);
    }
  }
  float* part_of[FB_MAXL];
  long slab_of[FB_MAXL];
  for (int l = 0; l < FB_MAXL; ++l) {
    R.elem_begin[l] = el;
    if (l < d->n_layers) {
      if (!d->dz_frag[l] || !d->act_frag[l] || !d->dw[l]) return RG_EINVAL;
      const int out_f = d->dims[l + 1], in_f = d->dims[l];
      const WgradFragPlan& p = plan[l];
      const long slab = wgrad_slab_floats(p);
      part_of[l] = part; slab_of[l] = slab;
      R.partial[l] = part; R.slab[l] = slab; R.splits[l] = p.splits; R.out[l] = d->dw[l];
      R.mode[l] = part_mode; R.NTb[l] = p.NTb; R.N[l] = out_f; R.K[l] = in_f;
#ifdef RG_WGRAD_LAYER_MASK  // timing ablation only (profiles/scripts): layers outside the mask get no workgroups, dW = 0
      if (!((RG_WGRAD_LAYER_MASK >> l) & 1)) R.splits[l] = 0;
#endif
      part += (size_t)p.splits * slab;
      el += part_mode == 1 ? (long)p.NTa * p.NTb * 256 : p.slab;  // threads of the reduce launch: a workgroup per tile, or one per element
    } else {
      part_of[l] = nullptr; slab_of[l] = 0;
      R.partial[l] = nullptr; R.slab[l] = 0; R.splits[l] = 0; R.out[l] = nullptr;
      R.mode[l] = 0; R.NTb[l] = 1; R.N[l] = 0; R.K[l] = 0;
    }
  }
  G.n = n_ent;
  R.nt_loads = (wgrad_tuning().part_nt >> 1) & 1;
  for (int j = 0; j < WG_MAXV; ++j) {  // workgroup ranges in launch order
    G.wg_begin[j] = wg;
    if (j >= n_ent) {
      G.layer[j] = G.layer[0];
      continue;
    }
    const Entry& e = ent[j];
    const int l = e.layer;
    const WgradFragPlan& p = plan[l];
    WgradFragArgs& g = G.layer[j];
    g.a_frag = (const bf16_t*)d->dz_frag[l]; g.b_frag = (const bf16_t*)d->act_frag[l];
    g.NTa = p.NTa; g.NTb = p.NTb; g.MB = e.mb_end; g.mb_base = e.mb_base; g.mb_per_split = e.per; g.splits = e.splits;
    g.partial = part_of[l] + (size_t)e.split_base * slab_of[l]; g.slab = slab_of[l]; g.N = d->dims[l + 1]; g.K = d->dims[l];
    g.x3 = d->x3 ? (x3_dz_planes() == 1 ? 2 : 1) : 0;
    g.shape = p.shape;
    g.a_lo = d->x3 ? (long)frag_elems(batch, d->dims[l + 1]) : 0;
    g.b_lo = d->x3 ? (long)frag_elems(batch, d->dims[l]) : 0;
    g.part_mode = part_mode;
    g.part_nt = wgrad_tuning().part_nt;
    int splits = e.splits;
#ifdef RG_WGRAD_LAYER_MASK
    if (!((RG_WGRAD_LAYER_MASK >> l) & 1)) splits = g.splits = 0;
#endif
    wg += p.tiles * ((splits + 7) / 8 * 8);  // the tiles of a split on ONE XCD (wgrad_frag_body), eight splits abreast
  }
  G.wg_begin[WG_MAXV] = wg;
  R.elem_begin[FB_MAXL] = el;
  for (int l = d->n_layers; l <= FB_MAXL; ++l) R.elem_begin[l] = el;
  const size_t lds = (size_t)WG_SHAPED_LDS;
  RG_ALLOW_LDS(wgrad_group_kernel, lds);
  RG_LAUNCH_DYN(wgrad_group_kernel, dim3(wg), dim3(WG_THREADS), lds, (hipStream_t)stream, G);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  const int elem_blocks = (int)((el + 255) / 256);
  if (!d->db_partials && !d->sum_in) {
    RG_LAUNCH(reduce_group_kernel, dim3((unsigned)elem_blocks), dim3(256), (hipStream_t)stream, R);
    return (int)hipGetLastError();
  }
  // tails folded into this launch: the bias partials rg_mlp_backward_fused(defer_db) left in ITS workspace (layout as
  // there: [n_wg][dims[l+1]] per layer with a bias gradient, in layer order) and one scaled sum
  ReduceTailArgs T;
  T.splits = R;
  T.elem_blocks = elem_blocks;
  ReduceColsGroupArgs& C = T.cols;
  C.n = 0;
  C.S = padded_wgs(d, batch);
  int blocks = 0;
  if (d->db_partials) {
    const float* p = d->db_partials;
    for (int l = 0; l < d->n_layers; ++l) {
      if (d->db[l]) {
        const int i = C.n++;
        C.block_begin[i] = blocks;
        C.partials[i] = p;
        C.out[i] = d->db[l];
        C.N[i] = d->dims[l + 1];
        blocks += (d->dims[l + 1] + 31) / 32;
      }
      p += (size_t)C.S * d->dims[l + 1];
    }
  }
  for (int i = C.n; i <= FB_MAXL; ++i) C.block_begin[i] = blocks;
  for (int i = C.n; i < FB_MAXL; ++i) { C.partials[i] = nullptr; C.out[i] = nullptr; C.N[i] = 0; }
  if (d->sum_in && (!d->sum_out || d->sum_n <= 0)) return RG_EINVAL;
  T.sum_in = d->sum_in; T.sum_n = d->sum_n; T.sum_scale = (float)d->sum_scale; T.sum_out = d->sum_out;
  RG_LAUNCH(reduce_tail_kernel, dim3((unsigned)(elem_blocks + blocks + (d->sum_in ? 1 : 0))), dim3(256), (hipStream_t)stream, T);
  return (int)hipGetLastError();
}

static int mlp_update_launch(const rg_mlp_update_desc* d, double lr, double beta1, double beta2, double eps,
                             double weight_decay, double bias_correction1, double bias_correction2_sqrt,
                             double grad_scale, double tau, const double* sched, rg_stream_t stream) {
  if (!d || d->n_layers < 1 || d->n_layers > FB_MAXL || !d->param || !d->grad || !d->exp_avg || !d->exp_avg_sq ||
      bias_correction1 == 0.0 || (d->target && (tau < 0.0 || tau > 1.0)))
    return RG_EINVAL;
  UpdateArgs U;
  U.n = d->n_layers;
  long total = 0;
  for (int l = 0; l < FB_MAXL; ++l) {
    if (l < d->n_layers) {
      const int K = d->dims[l], N = d->dims[l + 1];
      U.N[l] = N; U.K[l] = K;
      U.w_off[l] = d->w_off[l]; U.b_off[l] = d->b_off[l];
      U.wf[l] = (bf16_t*)d->wfrag_fwd[l]; U.wb[l] = (bf16_t*)d->wfrag_bwd[l]; U.twf[l] = (bf16_t*)d->target_wfrag_fwd[l];
      const int Ng = d->group_rows[l];
      if (Ng < 0 || (Ng > 0 && N % Ng != 0)) return RG_EINVAL;
      U.Ng[l] = Ng;
      // a group's fragment set: [hi plane] (bf16) or [hi plane | lo plane] (split-bf16), what rg_group_weights_stage writes
      U.per_f[l] = Ng > 0 ? (long)wfrag_elems(Ng, K) * (d->x3 ? 2 : 1) : 0;
      U.per_b[l] = Ng > 0 ? (long)wfrag_elems(K, Ng) * (d->x3 ? 2 : 1) : 0;
      const long we = d->w_off[l] + (long)N * K, be = d->b_off[l] + N;
      total = we > total ? we : total;
      total = be > total ? be : total;
    } else {
      U.N[l] = U.K[l] = 0; U.w_off[l] = U.b_off[l] = 0; U.wf[l] = U.wb[l] = U.twf[l] = nullptr;
      U.Ng[l] = 0; U.per_f[l] = U.per_b[l] = 0;
    }
  }
  U.total = total;
  U.p = d->param; U.g = d->grad; U.m = d->exp_avg; U.v = d->exp_avg_sq; U.t = d->target;
  const double step_size = lr / bias_correction1;
  U.c = AdamCoef{(float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay,
                 (float)(-step_size), (float)bias_correction2_sqrt, (float)grad_scale};
  U.tau = (float)tau; U.one_minus_tau = (float)(1.0 - tau);
  U.sched = sched;
  U.pre_ticked = (sched && d->sched_pre_ticked) ? 1 : 0;
  U.post_tick = (long long*)d->post_tick;
  U.post_tick_mod = d->post_tick_mod > 0 ? d->post_tick_mod : 1;
  if (d->sched_pre_ticked && !sched) return RG_EINVAL;
  U.x3 = d->x3 ? 1 : 0;
  // weights with 32-byte-addressable rows go to the tiled kernel, the rest of the slab to its
  // per-element workgroups (same launch)
  UpdateTileArgs T;
  T.u = U;
  int wgs = 0, any_tiled = 0;
  long rest = 0;
  for (int l = 0; l < FB_MAXL; ++l) {
    T.tile_begin[l] = wgs;
    T.tiled[l] = 0;
    if (l < d->n_layers) {
      const int K = U.K[l], N = U.N[l];
      const bool ok = (K % 8) == 0 && (U.Ng[l] % 8) == 0 && (U.w_off[l] % 4) == 0 && ((((uintptr_t)U.p | (uintptr_t)U.g | (uintptr_t)U.m |
                                                                  (uintptr_t)U.v | (uintptr_t)U.t) & 15) == 0) &&
                      ((((uintptr_t)U.wf[l] | (uintptr_t)U.wb[l] | (uintptr_t)U.twf[l]) & 15) == 0);
      if (ok) {
        T.tiled[l] = 1;
        any_tiled = 1;
        wgs += ((N + UT_ROWS - 1) / UT_ROWS) * ((K + UT_COLS - 1) / UT_COLS);
      }
      T.rest_begin[2 * l] = rest;
      rest += N;  // bias
      T.rest_begin[2 * l + 1] = rest;
      if (!ok) rest += (long)N * K;
    } else {
      T.rest_begin[2 * l] = T.rest_begin[2 * l + 1] = rest;
    }
  }
  T.tile_begin[FB_MAXL] = wgs;
  for (int l = d->n_layers; l <= FB_MAXL; ++l) T.tile_begin[l] = wgs;
  T.rest_begin[2 * FB_MAXL] = rest;
  for (int q = 2 * d->n_layers; q <= 2 * FB_MAXL; ++q) T.rest_begin[q] = rest;
  if (!any_tiled) {
    RG_LAUNCH(mlp_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (hipStream_t)stream, U);
    return (int)hipGetLastError();
  }
  const int rest_wgs = (int)((rest + 255) / 256);
  RG_LAUNCH(mlp_update_tiles_kernel, dim3((unsigned)(wgs + rest_wgs)), dim3(256), (hipStream_t)stream, T);
  return (int)hipGetLastError();
}

int rg_mlp_update_fused(const rg_mlp_update_desc* d, double lr, double beta1, double beta2, double eps,
                        double weight_decay, double bias_correction1, double bias_correction2_sqrt,
                        double grad_scale, double tau, rg_stream_t stream) {
  return mlp_update_launch(d, lr, beta1, beta2, eps, weight_decay, bias_correction1, bias_correction2_sqrt, grad_scale,
                           tau, nullptr, stream);
}

int rg_mlp_update_fused_sched(const rg_mlp_update_desc* d, double beta1, double beta2, double eps,
                              double weight_decay, double grad_scale, double tau, const double* sched,
                              rg_stream_t stream) {
  if (!sched) return RG_EINVAL;
  return mlp_update_launch(d, 0.0, beta1, beta2, eps, weight_decay, 1.0, 1.0, grad_scale, tau, sched, stream);
}

}  // extern "C"

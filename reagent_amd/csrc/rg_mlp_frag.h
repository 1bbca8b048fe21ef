// rg_mlp_frag.h — what the fused FullyConnected-stack kernels share (mlp_fused.hip: bf16 operands;
// mlp_fused_x3.hip: split-bf16 "bf16x3" operands): the kernel argument block, the MFMA C-fragment order of
// saved activations, the LDS tile helpers and the two-phase epilogue pieces.
#pragma once
#include "rg_gemm.h"
#include "rg_optim.h"
#include <type_traits>
#include "../../include/reagent_hip.h"

// phase-timing hook: expands to nothing here; profiles/microbench/fwd_phases.hip defines it to an
// s_memtime stamp before including this file
#ifndef RG_STAMP
#define RG_STAMP(slot)
#endif
#ifndef RG_BSTAMP  // the backward kernel's hooks (profiles/microbench/bwd_phases.hip)
#define RG_BSTAMP(slot)
#endif
#ifndef RG_PHASE_INIT  // accumulating variant for loops (profiles/microbench/wgrad_phases.hip)
#define RG_PHASE_INIT()
#define RG_PHASE(i)
#define RG_PHASE_FLUSH()
#endif

namespace rg {

constexpr int FB_BM = 128;
constexpr int WG_THREADS = 512;  // weight-gradient kernels
#ifndef RG_FUSED_WAVES
#define RG_FUSED_WAVES 8
#endif
constexpr int FB_NW = RG_FUSED_WAVES;  // waves per workgroup of the forward / backward kernels (4 or 8)
#ifndef RG_SIGN_STORE16
#define RG_SIGN_STORE16 1  // saving forward, 512-wide stacks: a lane's sign words of both column tiles leave as one 16-byte store
#endif
#ifndef RG_OUT_ROWSTORE
#define RG_OUT_ROWSTORE 1  // forwards: a thin output layer's [rows, N] result leaves as whole 16-byte pieces through LDS
// (round 5, same box: fwd_phases 84.6 -> 79.2 us per launch, C2 step 0.495-0.508 -> 0.482 ms; a critic's single dense column as one
// run per tile: C4 step 1.750 -> 1.708 ms.  Split-bf16 forward: a first form with an extra barrier lost 2 %; with every part's partial
// sums staged — one barrier fewer — and the bias requested before the K loop: C2 step 1.136 -> 1.125 ms, C4 4.15 -> 4.115.)
#endif
#ifndef RG_SAVE_NT
#define RG_SAVE_NT 1  // saved fragments leave as non-temporal stores (store_packed_frags; same-box A/B switch)
#endif
constexpr int FB_MAXL = RG_MLP_MAX_LAYERS;

// Split-bf16 stacks: how many bf16 planes of dZ the STACK'S weight gradient multiplies (RG_X3_DZ_PLANES, default below).
// 2: dW = dZ_lo x_hi + dZ_hi x_lo + dZ_hi x_hi (three MFMAs per tile pair; gradients ~1e-6 of their largest entry from fp64).
// 1 (round 5): dZ travels as ONE plane, dW = dZ_hi x_lo + dZ_hi x_hi — the backward launch writes no lo plane (205 MB per C2
// step) and the weight gradient streams three planes instead of four; dZ's rounding (2^-9 per element, independent over the
// batch rows a dW entry sums) leaves dW within ~1.4e-3 of its largest entry (profiles/microbench/two_product.py: the
// first-Adam-step direction flips it adds stay under 0.05 %).  north_star's 1e-4 binds Q-values / logits — the forward and
// dgrad stay three-product either way.
#ifndef RG_X3_DZ_PLANES
#define RG_X3_DZ_PLANES 2
#endif
inline int x3_dz_planes() {
  static const int v = [] {
    const char* e = getenv("RG_X3_DZ_PLANES");
    const int n = e ? atoi(e) : RG_X3_DZ_PLANES;
    return n == 1 ? 1 : 2;
  }();
  return v;
}

struct MlpArgs {
  int n_layers, batch;
  int dims[FB_MAXL + 1];
  int acts[FB_MAXL];
  const bf16_t* wfrag[FB_MAXL];  // forward: B fragments of W_l; backward: B fragments of W_l^T
  const float* bias[FB_MAXL];
  bf16_t* act_frag[FB_MAXL + 1];  // [l] = input of layer l in C-fragment order ([0] = network input)
  bf16_t* dz_frag[FB_MAXL];       // [l] = d loss / d (pre-activation output of layer l)
  unsigned* act_sign[FB_MAXL];    // [l] = (act_frag[l] > 0) bits, one per element, private order (nullable)
  float* db_part[FB_MAXL];        // backward: [n_workgroups][dims[l+1]] bias-gradient partials (nullable)
  const void* x;                  // forward input [batch, dims[0]] row-major, bf16 or fp32
  long ldx;
  const void* x2;                 // optional second input panel: columns [x_split, dims[0]) come from here
  long ldx2;
  int x_split;
  int dx_col0;                    // backward: first input column whose gradient is produced (dx32[0])
  const int* rowmap;              // forward: tile row r reads input row rowmap[r] (-1: zeros); null = identity
  const int* tile_key;            // grouped output layer: FIRST group with rows in each 128-row tile (-1: none), null = plain layer
  const int* row_begin;           // grouped output layer: group g owns rows [row_begin[g], row_begin[g + 1]) of the grouped space
  int n_groups;
  long group_stride;              // elements between the groups' fragment sets of the last layer (this direction)
  int out_scatter;                // forward: output row r goes to out32[rowmap[r]]
  int stage_out;                  // grouped forward: the output leaves as whole rows through the LDS behind the activation tile
  int out_lds;                    // forward: a thin output layer's weight fragments are copied (LDS-DMA) behind the activation tile
  int x_is_f32, x2_is_f32;
  float* out32;  // forward output [batch, dims[L]] fp32
  long ldo;
  const float* dout32;  // backward input [batch, dims[L]] fp32
  long lddo;
  float* dx32;  // backward: optional gradient w.r.t. the network input, fp32 [batch, dims[0]]
  long lddx;
  int pitch;  // LDS row pitch (elements)
  int save;   // forward: store act_frag[]
  // split-bf16 ("bf16x3") mode: element offset of the lo plane behind the hi plane of each fragment buffer
  long wfrag_lo[FB_MAXL];
  long act_lo[FB_MAXL + 1];
  long dz_lo[FB_MAXL];
};

__device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// compile-time activation (a runtime `switch` per element would bloat the unrolled epilogues until
// the unroller gives up and the accumulator arrays fall into scratch)
template <int ACT> __device__ __forceinline__ float act_t(float z) { return act_apply(z, ACT); }

template <int ACT> __device__ __forceinline__ float act_grad_t(float h) { return act_grad_from_output(h, ACT); }
#define RG_DISPATCH_ACT(act, ...)                                                   \
  switch (act) {                                                                    \
    case ACT_RELU: { constexpr int A_ = ACT_RELU; __VA_ARGS__; } break;             \
    case ACT_LEAKY_RELU: { constexpr int A_ = ACT_LEAKY_RELU; __VA_ARGS__; } break; \
    case ACT_TANH: { constexpr int A_ = ACT_TANH; __VA_ARGS__; } break;             \
    case ACT_SIGMOID: { constexpr int A_ = ACT_SIGMOID; __VA_ARGS__; } break;       \
    case ACT_SOFTPLUS: { constexpr int A_ = ACT_SOFTPLUS; __VA_ARGS__; } break;     \
    default: { constexpr int A_ = ACT_LINEAR; __VA_ARGS__; } break;                 \
  }

// C-fragment order: element (row, col) lives in block (row/32, col/32), half h, lane, e with
//   col%32 = lane&31,  row%32 = (r&3) + 8*(r>>2) + 4*(lane>>5),  r = 8*h + e   (MFMA 32x32 D layout)
__device__ __forceinline__ long frag_offset(int mb, int nt, int NT, int h, int lane) {
  return ((((long)mb * NT + nt) * 2 + h) * 64 + lane) * 8;
}
__device__ __forceinline__ int frag_row(int h, int e, int lg) {
  const int r = 8 * h + e;
  return (r & 3) + 8 * (r >> 2) + 4 * lg;
}

// ---- LDS tile helpers -----------------------------------------------------------------------
// rows [row_base, row_base+128) x cols [0, ncols_pad) of a row-major global matrix -> bf16 LDS tile
template <typename T, int THREADS>
__device__ __forceinline__ void load_tile_to_lds(bf16_t* act, int pitch, const T* src, long ld, int row_base,
                                                 int nrows, int ncols, int ncols_pad, int tid) {
  const int cpr = ncols_pad / 8;  // 8-element chunks per row
  const int total = FB_BM * cpr;
  const bool vec = ((ld % 8) == 0) && ((((uintptr_t)src) & 15) == 0);
  if (vec && (ncols & 7) == 0 && ncols > 0) {
    // aligned rows: four chunks per thread in flight (all loads issued before the first use; the
    // addresses of out-of-range chunks are clamped and their result replaced by zeros).  Chunks of the padding
    // columns [ncols, ncols_pad) count as out of range (QR-DQN's dZ tile: 200 of 224 columns — on the one-chunk-at-a-
    // time path below its seven dependent HBM round trips per thread were ~20 us of the C3 backward).
    constexpr int U = 4;
    const int kmax = ncols - 8;
    for (int c0 = tid; c0 < total; c0 += THREADS * U) {
      f32x4 raw[U][sizeof(T) == 4 ? 2 : 1];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int cu = c0 + u * THREADS, c = cu < total ? cu : total - 1;
        const int gr = row_base + c / cpr, grow = gr < nrows ? gr : nrows - 1;
        const int kk = (c % cpr) * 8;
        const T* p = src + (long)grow * ld + (kk < kmax ? kk : kmax);
        raw[u][0] = *(const f32x4*)p;
        if (sizeof(T) == 4) raw[u][sizeof(T) == 4 ? 1 : 0] = *(const f32x4*)(p + 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * THREADS;
        if (c >= total) break;
        const int r = c / cpr, k0 = (c % cpr) * 8;
        u16x8 v;
        if (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = f32_to_bf16(raw[u][0][e]);
            v[4 + e] = f32_to_bf16(raw[u][sizeof(T) == 4 ? 1 : 0][e]);
          }
        } else {
          v = __builtin_bit_cast(u16x8, raw[u][0]);
        }
        if (row_base + r >= nrows || k0 >= ncols) v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
        *(u16x8*)&act[r * pitch + k0] = v;
      }
    }
    return;
  }
  for (int c = tid; c < total; c += THREADS) {
    const int r = c / cpr, k0 = (c % cpr) * 8;
    const int grow = row_base + r;
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (grow < nrows && k0 < ncols) {
      const T* p = src + (long)grow * ld + k0;
      if (sizeof(T) == 2 && vec && k0 + 8 <= ncols) {
        v = *(const u16x8*)p;
      } else if (sizeof(T) == 4 && vec && k0 + 8 <= ncols) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = cvt_out<bf16_t>(a[e]);
          v[4 + e] = cvt_out<bf16_t>(b[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e < ncols) v[e] = cvt_out<bf16_t>(cvt_in(p[e]));
      }
    }
    *(u16x8*)&act[r * pitch + k0] = v;
  }
}

// the same through a row map: tile row r <- src row rowmap[row_base + r] (-1: zeros)
template <typename T, int THREADS>
__device__ __forceinline__ void load_tile_rows_mapped(bf16_t* act, int pitch, const T* src, long ld, const int* rowmap,
                                                      int row_base, int ncols, int ncols_pad, int tid) {
  const int cpr = ncols_pad / 8;
  const int total = FB_BM * cpr;
  const bool vec = ((ld % 8) == 0) && ((((uintptr_t)src) & 15) == 0);
  for (int c = tid; c < total; c += THREADS) {
    const int r = c / cpr, k0 = (c % cpr) * 8;
    const int grow = rowmap[row_base + r];
    u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (grow >= 0 && k0 < ncols) {
      const T* p = src + (long)grow * ld + k0;
      if (sizeof(T) == 2 && vec && k0 + 8 <= ncols) {
        v = *(const u16x8*)p;
      } else if (sizeof(T) == 4 && vec && k0 + 8 <= ncols) {
        const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = cvt_out<bf16_t>(a[e]);
          v[4 + e] = cvt_out<bf16_t>(b[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e < ncols) v[e] = cvt_out<bf16_t>(cvt_in(p[e]));
      }
    }
    *(u16x8*)&act[r * pitch + k0] = v;
  }
}

// LDS tile (128 rows x ntiles*32 cols) -> C-fragment order in global memory
// (only the 32-row blocks [mbl0, mbl1) of the tile; block mbl goes to fragment block mb_base + mbl)
__device__ __forceinline__ void emit_frags_from_lds(const bf16_t* act, int pitch, int ntiles, bf16_t* dst,
                                                    int mb_base, int wave, int n_waves, int lane, int mbl0 = 0, int mbl1 = 4) {
  const int lr = lane & 31, lg = lane >> 5;
  const int total = (mbl1 - mbl0) * ntiles * 2;
  for (int f = wave; f < total; f += n_waves) {
    const int h = f & 1, nt = (f >> 1) % ntiles, mbl = mbl0 + (f >> 1) / ntiles;
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = act[(mbl * 32 + frag_row(h, e, lg)) * pitch + nt * 32 + lr];
    *(u16x8*)(dst + frag_offset(mb_base + mbl, nt, ntiles, h, lane)) = v;
  }
}

// One 32x32 accumulator tile (values final, fp32) as 8 packed bf16 pairs P[i] = (v[2i], v[2i+1]):
// rows 2i and 2i+1 of the lane's column.  P[0..3] / P[4..7] ARE the two 16-byte C-fragment records
// of the tile, so saving for backward costs no further conversion.
__device__ __forceinline__ void pack_tile(const float (&v)[16], unsigned (&P)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) P[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
}

// Packed tile -> LDS activation tile.  Neighbouring lanes hold neighbouring columns: swap the pair
// with the neighbour (DPP) and let a byte permute build the dword this lane stores — the even lane
// writes row 2i (its low half + the neighbour's low half), the odd lane row 2i+1 (high halves).
// Per value pair: 1 cvt_pk (pack_tile) + 1 DPP move + 1 v_perm + 1 ds_write_b32.
__device__ __forceinline__ void store_packed_to_lds(bf16_t* act, int pitch, int row0_tile, int col, int lane,
                                                    const unsigned (&P)[8]) {
  const int lg = lane >> 5, odd = lane & 1;
  const unsigned sel = odd ? 0x03020706u : 0x05040100u;  // {hi = neighbour's pair, lo = own pair}
  bf16_t* base = act + (row0_tile + 4 * lg + odd) * pitch + (col & ~1);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = 2 * i;
    const unsigned other = swap_adjacent_lanes(P[i]);
    const unsigned word = perm_bytes(other, P[i], sel);
    *(unsigned*)(base + ((r & 3) + 8 * (r >> 2)) * pitch) = word;
  }
}

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void store_packed_frags(bf16_t* dst, int mb, int nt, int NT, int lane,
                                                   const unsigned (&P)[8]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // streaming store: the fragments are read next by another launch (the weight gradient), never again by this one,
    // and should not push the weights out of L2 (same-box A/B: the consuming wgrad launch 137 -> 131 us)
    const u32x4 v = u32x4{P[4 * h], P[4 * h + 1], P[4 * h + 2], P[4 * h + 3]};
#if RG_SAVE_NT
    stream_store(v, (u32x4*)(dst + frag_offset(mb, nt, NT, h, lane)));
#else
    *(u32x4*)(dst + frag_offset(mb, nt, NT, h, lane)) = v;
#endif
  }
}

// bit r = (v[r] > 0).  NONNEG (ReLU outputs: v is +0 or a positive float, never -0/NaN): the sign
// bit of (0 - bits(v)) is the answer, shifted in by a funnel shift (v_sub + v_alignbit, 2 VALU per
// element against 3 for compare/select/or).
template <bool NONNEG>
__device__ __forceinline__ unsigned positive_bits(const float (&v)[16]) {
  unsigned bits = 0u;
#pragma unroll
  for (int r = 15; r >= 0; --r) {
    if (NONNEG) bits = (bits << 1) | ((0u - __builtin_bit_cast(unsigned, v[r])) >> 31);
    else bits = (bits << 1) | (v[r] > 0.f ? 1u : 0u);
  }
  return bits;
}

// Grouped launches (rg_mlp_desc.tile_key, qr_grouped.hip): the tiles are sorted by group and every group has its own
// slice of the output layer's weights (QR-DQN C3: 16 actions x 229 KB of fragments, 3.7 MB next to the trunk's 1.2 MB —
// more than one XCD's 4 MB L2 when every XCD sees every group).  The hardware places block b on XCD b % 8, so XCD x takes
// the x-th EIGHTH of the tile list: its L2 then serves the slices of ~G/8 groups.  The launch has 8 * ceil(n_tiles / 8)
// blocks; those whose tile is past the end return at once.
#ifndef RG_GROUPED_WHOLE
#define RG_GROUPED_WHOLE 1  // grouped forward, a tile's last segment: pipelined K loop per column tile + whole-tile staging (round 6)
#endif
#ifndef RG_GROUPED_XCD
#define RG_GROUPED_XCD 1
#endif
__device__ __forceinline__ int grouped_tile(int block, int n_tiles) {
#if RG_GROUPED_XCD
  const int per = (n_tiles + 7) >> 3;
  return (block & 7) * per + (block >> 3);
#else
  return block;
#endif
}

// Grouped row space (qr_grouped.hip): group g owns the rows [row_begin[g], row_begin[g + 1]) — padded per group to whole
// 128-row tiles (then a tile belongs to one group) or DENSE (round 4: no padding between groups, B / 128 tiles exactly; a tile
// that holds the end of one group and the start of the next is cut into SEGMENTS and the grouped layer runs once per
// segment).  A window of `rows` rows from row_base is walked segment by segment: g = the window's first group (tile_key),
// then next_segment() until it returns false.  Everything here is workgroup-uniform.
struct RowSegment {
  int grp, lo, hi;  // window-relative rows [lo, hi) of group grp
};
__device__ __forceinline__ bool next_segment(const int* row_begin, int n_groups, int row_base, int rows, int& g, RowSegment& s) {
  while (g >= 0 && g < n_groups) {
    const int b = row_begin[g], e = row_begin[g + 1];
    if (b >= row_base + rows) return false;
    const int lo = b > row_base ? b : row_base, hi = e < row_base + rows ? e : row_base + rows;
    const int grp = g++;
    if (hi > lo) {
      s.grp = grp;
      s.lo = lo - row_base;
      s.hi = hi - row_base;
      return true;
    }
  }
  return false;
}

// dst[r][0 .. ncols) = lo <= r < hi ? src[r][0 .. ncols) : 0 for the ROWS rows of an LDS tile (ncols a multiple of 8; src and dst
// 16-byte aligned with the same pitch): the operand of ONE segment of a boundary tile — the other groups' rows contribute zero
template <int THREADS, int ROWS>
__device__ __forceinline__ void copy_rows_masked(const bf16_t* src, bf16_t* dst, int pitch, int ncols, int lo, int hi, int tid) {
  const int cpr = ncols >> 3;
  for (int c = tid; c < ROWS * cpr; c += THREADS) {
    const int r = c / cpr, k = (c - r * cpr) * 8;
    u16x8 v = *(const u16x8*)(src + r * pitch + k);
    if (r < lo || r >= hi) v = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
    *(u16x8*)(dst + r * pitch + k) = v;
  }
}
// where that copy lives: behind the operand inside the rows of a 520-wide tile (the grouped layer is <= 256 columns wide), behind
// the whole tile of a 264-wide one (the launch asks for the extra LDS)
template <int PITCH> __device__ __forceinline__ constexpr int masked_copy_offset(int tile_elems) {
  return PITCH >= 2 * 256 + 8 ? 256 : tile_elems;
}

// ---- main loop of a wide layer: this wave's [128 x 32*TN] slice over K ------------------------
// `rot` rotates the order in which the K chunks are visited (a sum may be taken in any order):
// every workgroup streams the SAME weight fragments, and without de-phasing all 256 CUs would
// hammer one L2 channel at a time (measured on MI355X, C2 forward: 103.5 us with, 107 us without).
// Software pipeline: the weight (B) fragments come from L2 — ~2000 cycles under this load, measured
// with profiles/microbench/fwd_phases — and are prefetched RING-1 chunks ahead through a ring of
// RING register sets; the activation (A) fragments come from LDS, one chunk ahead.
// wf_wave, rot and every chunk offset are wave-uniform: the weight address math stays on the
// scalar unit (SGPR base + lane*16 B), only the LDS reads need a vector add per chunk.
// The K-rotation of (workgroup, wave) for a layer with KC chunks
__device__ __forceinline__ int k_rotation(int wg, int wave, int KC) { return (wg * 5 + wave * 11) % KC; }

// (Hoisting the ring fill of a layer ahead of the previous epilogue / the input-tile load was
// measured with profiles/microbench/fwd_phases: the epilogues got 1.2k cycles slower each and the
// main loops no faster, so the fill stays at the top of the main loop.)
// Accumulators pinned to AccVGPRs (RG_ACC_AGPR, rounds 3 and 6) are gone from this file.  The experiment rested on
// profiles/microbench/mfma_feed showing this loop ALONE 19-22 % faster with them; round 6 found that micro-benchmark's arch-VGPR
// variant compiled to 660-692 bytes of scratch per lane (its K is a compile-time constant, the fully unrolled loop parked every
// chunk's address in VGPRs) — spill-free both forms run at 0.50 of the nominal peak.  In these kernels (no scratch, run-time K)
// the AccVGPR form measured slower both times: 20 / 14 cold spills out of the 128 arch registers it leaves, a copy per
// accumulator on its way to the VALU, forward 78.9 -> 91.4 us (profiles/NOTES_r06.md §5, §9).
__device__ __forceinline__ f32x16 mfma_main(u16x8 a, u16x8 b, f32x16 c) { return mfma_32x32x16_bf16(a, b, c); }

// SWAP: the MFMA takes the WEIGHT fragment as its A operand and the activation fragment as B (the two fragment layouts are
// the same registers: lane = row / column index, 8 consecutive k) — the accumulator tile is then the TRANSPOSE: lane =
// batch row, register r = feature (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile.  Same products, same order over k.
template <int TN, int RING, bool SWAP = false>
__device__ __forceinline__ void wide_mainloop(const bf16_t* act, int pitch, int KC, const bf16_t* wf_wave,
                                              long nt_stride, f32x16 (&acc)[4][TN], int lane, int rot,
                                              int prio_phase = 0) {
  static_assert(RING >= 2 && RING % 2 == 0, "the A double buffer alternates with the ring slot parity");
  const int lr = lane & 31, lg = lane >> 5;
  auto kx = [&](int kc) { const int k = kc + rot; return k >= KC ? k - KC : k; };
  const bf16_t* arow = act + lr * pitch + lg * 8;
  const int tm_stride = 32 * pitch;
  u16x8 a[2][4], b[RING][TN];
  auto loadB = [&](u16x8 (&bf)[TN], int kc) {
    const bf16_t* chunk = wf_wave + (long)kx(kc) * 512;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bf[tn] = *(const u16x8*)(chunk + tn * nt_stride + lane * 8);
  };
  auto loadA = [&](u16x8 (&af)[4], int kc) {
    const int off = kx(kc) * 16;
#pragma unroll
    for (int tm = 0; tm < 4; ++tm) af[tm] = *(const u16x8*)(arow + tm * tm_stride + off);
  };
  auto mma = [&](const u16x8 (&af)[4], const u16x8 (&bf)[TN]) {
#pragma unroll
    for (int tm = 0; tm < 4; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = SWAP ? mfma_main(bf[tn], af[tm], acc[tm][tn]) : mfma_main(af[tm], bf[tn], acc[tm][tn]);
  };
  if (KC % RING == 0) {
    // fast path: no conditionals around the loads in the steady state, so the compiler keeps exact
    // s_waitcnt vmcnt(N)/lgkmcnt(N) counts and (RING-1)*TN weight loads stay in flight.
    // profiles/microbench/mfma_feed.hip (this loop without epilogues or barriers, random data, whole chip; round 6's spill-free
    // build — the L2-fed modes of rounds 2-5 carried 660-692 bytes of scratch per lane and read 0.39-0.45): MFMAs fed from
    // registers only 0.65-0.71 of the nominal 2.5 PFLOP/s (the chip clocks down under dense MFMA load), + A fragments from LDS
    // 0.60-0.65, + B fragments from L2 with RING = 2 0.50-0.51.  In the step RING = 2 measured 168-171 us for the two
    // non-saving forwards against 173-179 with RING = 4 (same box), so 2 is the default for the 8-wave kernels.
    // Measured alternatives (profiles/microbench/fwd_phases, K=512 main loop, cycles of the early /
    // late wave of a SIMD): this loop 12.5k / 20.3k; RING=8 13.4k / 21.1k; rotation per block of 4
    // chunks with all addresses as immediates (a quarter of the scalar instructions) 16.4k / 22.9k;
    // one wave per SIMD with 16 accumulator tiles 28k.  The loop is bound by how evenly the weight
    // reads of 256 CUs spread over the L2 channels, not by instruction issue or by load latency.
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) loadB(b[s], s);
    loadA(a[0], 0);
    int kc = 0;
    for (; kc < KC - RING; kc += RING) {
      // The SIMD arbitrates MFMA issue by priority, then age: left alone, the older wave of a pair
      // takes ~63 % of the pipe and finishes its loop ~8k cycles before its partner, which then runs
      // the tail alone at half rate.  The younger wave (prio_phase = 1) therefore runs the first half
      // of its loop at raised priority and hands the advantage back for the second half, so the two
      // finish together.
      // (profiles/microbench/fwd_phases, early / late wave of a SIMD, K=512 loop: this hand-off 16.8k /
      // 19.3k; swapping the priority every ring block 14.7k / 20.3k, every two blocks 15.0k / 20.2k; no
      // priorities 13.0k / 20.5k — what counts is when the LATER wave gets out.)
      if (prio_phase && kc * 2 < KC) RG_SETPRIO(1);
      else RG_SETPRIO(0);
#pragma unroll
      for (int s = 0; s < RING; ++s) {
        loadB(b[(s + RING - 1) % RING], kc + s + RING - 1);
        loadA(a[(s + 1) & 1], kc + s + 1);
        sched_fence();
        mma(a[s & 1], b[s]);
        sched_fence();
      }
    }
    RG_SETPRIO(0);
#pragma unroll
    for (int s = 0; s < RING; ++s) {  // last block: only the loads that are still in range
      if (s == 0) loadB(b[RING - 1], kc + RING - 1);
      if (s < RING - 1) loadA(a[(s + 1) & 1], kc + s + 1);
      sched_fence();
      mma(a[s & 1], b[s]);
      sched_fence();
    }
    return;
  }
  // generic K: same ring with guarded loads
#pragma unroll
  for (int s = 0; s < RING - 1; ++s)
    if (s < KC) loadB(b[s], s);
  loadA(a[0], 0);
  for (int kc = 0; kc < KC; kc += RING) {
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      if (kc + s < KC) {
        if (kc + s + RING - 1 < KC) loadB(b[(s + RING - 1) % RING], kc + s + RING - 1);
        if (kc + s + 1 < KC) loadA(a[(s + 1) & 1], kc + s + 1);
        sched_fence();
        mma(a[s & 1], b[s]);
        sched_fence();
      }
    }
  }
}

// acc += A . B for one MORE segment of a boundary tile of a grouped layer (the first segment went through wide_mainloop):
// the same products on a plain loop, one chunk of fragments at a time.  Deliberately small — with the software-pipelined
// main loop inside a loop over segments the 512-wide backward kernel spilled 390-460 registers, accumulator tiles among
// them inside the K loop; as a second, conditional call site it keeps its registers.  At most one tile per group boundary
// (<= n_groups - 1 of the B / 128 tiles of a launch) comes here, for the grouped layer's K only.
template <int TM, int TN>
__device__ __forceinline__ void segment_accumulate(const bf16_t* act, int pitch, int KC, const bf16_t* wf_wave, long nt_stride,
                                                   f32x16 (&acc)[TM][TN], int lane) {
  const bf16_t* arow = act + (lane & 31) * pitch + (lane >> 5) * 8;
  for (int kc = 0; kc < KC; ++kc) {
    u16x8 af[TM], bf[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bf[tn] = *(const u16x8*)(wf_wave + (long)kc * 512 + tn * nt_stride + lane * 8);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) af[tm] = *(const u16x8*)(arow + tm * 32 * pitch + kc * 16);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_main(af[tm], bf[tn], acc[tm][tn]);
  }
}

// Thin output layer (<= 16 outputs: DQN's Q-values, a critic's scalar) with its weights resident in LDS.  Every workgroup
// needs the same 16 KB of B fragments at the end of its life, and tile_kloop fetches them from L2 in a chain of ~2000-cycle
// round trips.  The fragments' valid half (lanes of columns 0..15: 512 bytes per 16-chunk) is instead copied once, by
// LDS-DMA at the top of the kernel, into the 30 KB of LDS behind the activation tile (out_lds_prefetch), and the loop below
// reads both operands from LDS.  A lane of a padding column (16..31) reads its neighbour's record: it only feeds output
// columns that are not stored.  Measured (round 3, profiles/microbench/fwd_phases with its stamps inside the output layer):
// the K loop of the output layer 3.8k -> 2.2k cycles, the forward launch 84.0 -> 82.8 us (-1.3 %; C2 step same box
// 0.538 -> 0.531-0.538 ms).  The phase table's "output layer 8.7k" overstates what there was to win: its closing stamp waits
// (vmcnt(0)) for the output stores, which a wave of the product kernel does not.
__device__ __forceinline__ void out_lds_prefetch(const bf16_t* wf, int KC, char* wo, int wave, int n_waves, int lane) {
  const int cl = lane & 31, src_lane = (cl & 15) + 32 * (cl >> 4);
  for (int i = wave; i < KC / 2; i += n_waves) {  // one DMA = the records of chunks 2i (lanes 0..31) and 2i+1 (lanes 32..63)
    const int chunk = 2 * i + (lane >> 5);
    global_load_lds_b128_cached(wf + (long)chunk * 512 + src_lane * 8, wo + i * 1024);
  }
}
__device__ __forceinline__ f32x16 tile_kloop_ldsb(const bf16_t* act, int pitch, const char* wo, int tm, int lane, int kc_lo,
                                                  int kc_hi) {
  const int lr = lane & 31, lg = lane >> 5;
  const bf16_t* arow = act + (tm * 32 + lr) * pitch + lg * 8;
  const char* brec = wo + ((lr & 15) + 16 * lg) * 16;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int kc = kc_lo;
  // (Round 6: eight chunks of both operands requested at once, twice — 64 + 64 registers, the hidden layers' accumulators being
  // dead by now — measured with fwd_phases: K loop 2252 -> 2446 cycles, not kept: the loop is not a chain of exposed LDS round trips.)
  for (; kc + 4 <= kc_hi; kc += 4) {
    u16x8 af[4], bf[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      af[i] = *(const u16x8*)(arow + (kc + i) * 16);
      bf[i] = *(const u16x8*)(brec + (kc + i) * 512);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = mfma_32x32x16_bf16(af[i], bf[i], acc);
  }
  for (; kc < kc_hi; ++kc)
    acc = mfma_32x32x16_bf16(*(const u16x8*)(arow + kc * 16), *(const u16x8*)(brec + kc * 512), acc);
  return acc;
}

// one 32x32 output tile (row tile tm, weight n-tile nt) over K; used for narrow / irregular widths.
// Groups of 4 chunks, next group's fragments in flight during the current group's MFMAs.
// (Measured alternatives, profiles/microbench/fwd_phases, cycles per wave averaged over the 8 waves:
// this loop 3.9k; 4 independent accumulators 5.5k; all <= 32 weight chunks requested up front 9.2k —
// every workgroup reads the same 32 KB, and a burst on that region queues in the L2 channels.)
// [kc_lo, kc_hi): the K chunks this call sums (a thin output layer splits K over two waves per tile, below)
__device__ __forceinline__ f32x16 tile_kloop(const bf16_t* act, int pitch, int KC, const bf16_t* wf, int tm,
                                             int nt, int lane, int kc_lo = 0, int kc_hi = -1) {
  const int lr = lane & 31, lg = lane >> 5;
  const bf16_t* arow = act + (tm * 32 + lr) * pitch + lg * 8 + kc_lo * 16;
  const bf16_t* wl = wf + ((long)nt * KC + kc_lo) * 512 + lane * 8;
  KC = (kc_hi < 0 ? KC : kc_hi) - kc_lo;  // from here on: the number of chunks of this call
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  u16x8 a0[4], b0[4], a1[4], b1[4];
  auto load = [&](u16x8 (&af)[4], u16x8 (&bf)[4], int kc0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kc = kc0 + i < KC ? kc0 + i : KC - 1;  // clamped; the extra products are skipped below
      bf[i] = *(const u16x8*)(wl + (long)kc * 512);
      af[i] = *(const u16x8*)(arow + kc * 16);
    }
  };
  auto mma = [&](const u16x8 (&af)[4], const u16x8 (&bf)[4], int kc0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (kc0 + i < KC) acc = mfma_32x32x16_bf16(af[i], bf[i], acc);
  };
  load(a0, b0, 0);
  for (int kc = 0; kc < KC; kc += 8) {
    if (kc + 4 < KC) load(a1, b1, kc + 4);
    sched_fence();
    mma(a0, b0, kc);
    sched_fence();
    if (kc + 4 < KC) {
      if (kc + 8 < KC) load(a0, b0, kc + 8);
      sched_fence();
      mma(a1, b1, kc + 4);
      sched_fence();
    }
  }
  return acc;
}

// Sign bits of a wave's [128 x 32*TN] slice: 2*TN dwords per lane, dword = tn*2 + tm/2,
// bit = (tm&1)*16 + r for accumulator element r of tile (tm, tn).  For ReLU-family activations the
// derivative depends on nothing else, so backward prefetches these 16 bytes per lane ahead of its
// main loop instead of waiting on 8 KB of saved activations per wave in the epilogue.
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__device__ __forceinline__ long sign_offset(int wg, int wave, int lane, int TN, int width) {
  // a workgroup's plane: 128 rows x width bits = 4*width dwords = n_waves * 64 lanes * 2*TN dwords
  return (long)wg * (4 * width) + ((long)wave * 64 + lane) * (2 * TN);
}
template <int ACT> constexpr bool act_is_sign_based() { return ACT == ACT_RELU || ACT == ACT_LEAKY_RELU; }
// what a saving forward stores of layer l's output: save = 1 everything backward + wgrad read; save = 2 (a dx_only
// backward follows) the fragments only where the activation gradient needs the values (the sign plane otherwise)
__device__ __forceinline__ bf16_t* fwd_save_dst(const MlpArgs& a, int l) {
  const bool sign_based = a.acts[l] == ACT_RELU || a.acts[l] == ACT_LEAKY_RELU;
  return (a.save == 1 || (a.save == 2 && !(sign_based && a.act_sign[l + 1]))) ? a.act_frag[l + 1] : nullptr;
}

// The hidden-layer epilogues run in two phases around the barrier that protects the in-place LDS
// tile.  PACK (before the barrier; touches no LDS): bias/activation (or the activation-gradient
// mask), bf16 packing, the global stores of what backward needs.  STORE (after the barrier): the
// packed pairs go to the LDS tile.  The two waves of a SIMD do not finish a main loop together (the
// older one wins the MFMA arbitration and is ~8k cycles early, profiles/microbench/fwd_phases), so
// the early wave's PACK runs under its partner's MFMAs, and the late wave packs with the SIMD's
// VALU to itself, instead of both competing for the VALU after the barrier.
template <int TN>
__device__ __forceinline__ void store_packed_tiles(bf16_t* act, int pitch, const unsigned (&PK)[4][TN][8], int wave,
                                                   int lane) {
  const int lr = lane & 31;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int col = (wave * TN + tn) * 32 + lr;
    static_for<0, 4>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      store_packed_to_lds(act, pitch, tm * 32, col, lane, PK[tm][tn]);
    });
  });
}

// (Starting the accumulators at the bias instead of adding it here was tried: the adds are already
// packed (v_pk_add_f32, 64 per wave and layer), and accumulators that are not a rematerialisable zero
// cost the TN = 2 kernel 68 spilled registers.)
template <int TN, int ACT>
__device__ __forceinline__ void fwd_hidden_pack(f32x16 (&acc)[4][TN], const float* bias, bf16_t* save_dst,
                                                unsigned* sign_dst, int NT, int mb_base, int wave, int lane,
                                                unsigned (&PK)[4][TN][8]) {
  lane = opaque(lane);
  const int lr = lane & 31;
  unsigned sg_prev0 = 0u, sg_prev1 = 0u;  // (TN == 2: the first column tile's sign words wait for the second's — one 16-byte store)
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int nt = wave * TN + tn, col = nt * 32 + lr;
    const float b = bias ? bias[col] : 0.f;
    unsigned sg0 = 0u, sg1 = 0u;
    static_for<0, 4>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = act_t<ACT>(acc[tm][tn][r] + b);
      pack_tile(v, PK[tm][tn]);
      if (save_dst) store_packed_frags(save_dst, mb_base + tm, nt, NT, lane, PK[tm][tn]);
      if (act_is_sign_based<ACT>() && sign_dst) {
        const unsigned bits = positive_bits<ACT == ACT_RELU>(v);
        if (tm < 2) sg0 |= bits << ((tm & 1) * 16);
        else sg1 |= bits << ((tm & 1) * 16);
      }
    });
    if (act_is_sign_based<ACT>() && sign_dst) {
      if constexpr (TN == 2 && RG_SIGN_STORE16) {
        // a lane's four sign words (two per column tile) are contiguous: ONE 16-byte store per lane and layer — 1 KB per wave
        // instruction — instead of two 8-byte ones at a lane stride of 16 bytes
        if constexpr (tn == 0) { sg_prev0 = sg0; sg_prev1 = sg1; }
        else *(u32x4*)(sign_dst + sign_offset(mb_base >> 2, wave, lane, TN, NT * 32)) = u32x4{sg_prev0, sg_prev1, sg0, sg1};
      } else {
        ((u32x2*)(sign_dst + sign_offset(mb_base >> 2, wave, lane, TN, NT * 32)))[tn] = u32x2{sg0, sg1};
      }
    }
  });
}

// Non-saving forward with transposed accumulator tiles (wide_mainloop<.., SWAP>): a lane holds ONE batch row and 16 features
// of a tile in four runs of four consecutive ones, so the bf16 pairs are column neighbours already and the LDS tile takes
// them as 8-byte writes — no neighbour swap (DPP + v_perm per pair) and half the LDS write instructions of the
// lane-per-column epilogue.  Nothing is saved in this layout (the fragment records backward and the weight gradient read
// are lane-per-column): it serves save = 0 launches only.  Bias: 16 values per tile and half-wave, four 16-byte loads.
template <int TN, int ACT>
__device__ __forceinline__ void fwd_hidden_pack_swapped(f32x16 (&acc)[4][TN], const float* bias, int wave, int lane,
                                                        unsigned (&PK)[4][TN][8]) {
  lane = opaque(lane);
  const int lg = lane >> 5;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int f0 = (wave * TN + tn) * 32 + 4 * lg;
    // the tile's 16 bias values (four runs of four features) are re-requested per row tile — 16-byte loads of a 2 KB
    // vector every workgroup of the CU reads, L1-resident — so that they live across ONE tile: kept across the four row
    // tiles of a column tile their 16 registers were what the 512-wide kernel (128 accumulators) spilled
    static_for<0, 4>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      sched_fence();  // one tile at a time: 16 accumulator registers die as 8 packed ones are born
      f32x4 b4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) b4[q] = bias ? *(const f32x4*)(bias + f0 + 8 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = act_t<ACT>(acc[tm][tn][r] + b4[r >> 2][r & 3]);
      pack_tile(v, PK[tm][tn]);
      pin_packed(PK[tm][tn]);
    });
    sched_fence();
  });
}

template <int TN>
__device__ __forceinline__ void store_packed_tiles_swapped(bf16_t* act, int pitch, const unsigned (&PK)[4][TN][8], int wave,
                                                           int lane) {
  const int lr = lane & 31, lg = lane >> 5;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    static_for<0, 4>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      bf16_t* row = act + (tm * 32 + lr) * pitch + (wave * TN + tn) * 32 + 4 * lg;
#pragma unroll
      for (int q = 0; q < 4; ++q) *(uint2*)(row + 8 * q) = uint2{PK[tm][tn][2 * q], PK[tm][tn][2 * q + 1]};
    });
  });
}

// dZ_below = dH * act'(H_below); column sums of dZ_below (bias gradient) for this workgroup
// STORE_DZ is a compile-time property (a run-time test of dz_dst inside the unrolled tile loops cost the 512-wide
// kernel 245 spilled registers and 124 MB of scratch traffic per launch)
template <int TN, int ACT, bool USE_SIGN, bool STORE_DZ = true>
__device__ __forceinline__ void bwd_hidden_pack(f32x16 (&acc)[4][TN], const bf16_t* h_frag,
                                                const unsigned (&sg)[2 * TN], bf16_t* dz_dst, float* db_part, int NT,
                                                int mb_base, int wave, int lane, unsigned (&PK)[4][TN][8]) {
  lane = opaque(lane);
  const int lr = lane & 31;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int nt = wave * TN + tn, col = nt * 32 + lr;
    float colsum = 0.f;
    static_for<0, 4>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      float v[16];
      if (USE_SIGN && act_is_sign_based<ACT>()) {
        const unsigned bits = sg[tn * 2 + (tm >> 1)] >> ((tm & 1) * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float g = ((bits >> r) & 1u) ? 1.f : (ACT == ACT_RELU ? 0.f : 0.01f);
          v[r] = acc[tm][tn][r] * g;
          colsum += v[r];
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u16x8 hf = *(const u16x8*)(h_frag + frag_offset(mb_base + tm, nt, NT, h, lane));
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[8 * h + e] = acc[tm][tn][8 * h + e] * act_grad_t<ACT>(bf16_to_f32(hf[e]));
            colsum += v[8 * h + e];
          }
        }
      }
      pack_tile(v, PK[tm][tn]);
      if (STORE_DZ) store_packed_frags(dz_dst, mb_base + tm, nt, NT, lane, PK[tm][tn]);
      else pin_packed(PK[tm][tn]);  // the stores force the bf16 packing here; without them the compiler keeps all
                                    // eight tiles in fp32 until the LDS pass after the barrier and spills 497 registers
    });
    colsum += shfl_xor(colsum, 32);
    if (db_part && lane < 32) db_part[col] = colsum;
  });
}

static inline int fused_supported(const rg_mlp_desc* d) {
  if (!d || d->n_layers < 2 || d->n_layers > FB_MAXL) return 0;
  const int H = d->dims[1];
  if (H != 256 && H != 512) return 0;
  for (int l = 1; l < d->n_layers; ++l)
    if (d->dims[l] != H) return 0;
  if (d->dims[0] < 1 || d->dims[0] > 512) return 0;
  if (d->dims[d->n_layers] < 1 || d->dims[d->n_layers] > 256) return 0;
  return H / 256;
}

// all layers (the last one included) of one hidden width in {256, 512}: the trunk of a stack whose output layer is
// handled elsewhere (backward only)
static inline int fused_trunk(const rg_mlp_desc* d) {
  if (!d || d->n_layers < 2 || d->n_layers > FB_MAXL || d->x3) return 0;
  const int H = d->dims[1];
  if (H != 256 && H != 512) return 0;
  for (int l = 1; l <= d->n_layers; ++l)
    if (d->dims[l] != H) return 0;
  return d->dims[0] >= 1 && d->dims[0] <= 512;
}

// LDS row pitch: widest layer + 8 elements (row stride = 4 banks mod 64: conflict-free 16-byte reads)
static inline int fused_pitch(const rg_mlp_desc* d) {
  int m = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    const int w = (d->dims[l] + 31) / 32 * 32;
    if (w > m) m = w;
  }
  return m <= 256 ? 264 : 520;
}



// fp32 master weights -> B-fragment order for forward (W) and backward (W^T), zero padded
__device__ __forceinline__ void stage_weight_elem(const float* __restrict__ w, int N, int K, bf16_t* __restrict__ wf,
                                                  bf16_t* __restrict__ wb, long i, int x3 = 0) {
  const int KCf = (K + 15) / 16, NTf = (N + 31) / 32;
  const int KCb = (N + 15) / 16, NTb = (K + 31) / 32;
  const long tf = (long)NTf * KCf * 512, tb = (long)NTb * KCb * 512;
  const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
  const long blk = i >> 9;
  if (wf && i < tf) {
    const int kc = (int)(blk % KCf), nt = (int)(blk / KCf);
    const int n = nt * 32 + (lane & 31), k = kc * 16 + (lane >> 5) * 8 + e;
    const float v = (n < N && k < K) ? w[(long)n * K + k] : 0.f;
    const bf16_t hi = f32_to_bf16(v);
    wf[i] = hi;
    if (x3) wf[tf + i] = f32_to_bf16(v - bf16_to_f32(hi));
  }
  if (wb && i < tb) {
    const int kc = (int)(blk % KCb), nt = (int)(blk / KCb);
    const int k = nt * 32 + (lane & 31), n = kc * 16 + (lane >> 5) * 8 + e;  // "weight" = W^T [K][N]
    const float v = (n < N && k < K) ? w[(long)n * K + k] : 0.f;
    const bf16_t hi = f32_to_bf16(v);
    wb[i] = hi;
    if (x3) wb[tb + i] = f32_to_bf16(v - bf16_to_f32(hi));
  }
}


static inline size_t frag_elems(int rows, int cols) {
  return (size_t)((rows + 127) / 128 * 128) * (size_t)((cols + 31) / 32 * 32);
}
static inline size_t wfrag_elems(int out_features, int in_features) {
  return (size_t)((out_features + 31) / 32) * (size_t)((in_features + 15) / 16) * 512;
}

// Rows of the fragment matrix that holds a GROUPED layer's dZ: a 32-row block that two groups share is written once per group
// (the other group's rows zeroed), group g's copy at block (row / 32) + g — so group g's blocks are contiguous,
// [row_begin[g] / 32 + g, ceil(row_begin[g + 1] / 32) + g), aligned with the UNSHIFTED blocks of the layer's input fragments,
// and the weight gradient needs no row masks (rg_group_head_wgrad)
static inline int grouped_dz_rows(int rows, int n_groups) { return rows + 32 * n_groups; }

static inline int fill_args(const rg_mlp_desc* d, int batch, MlpArgs& a, int backward) {
  a.n_layers = d->n_layers;
  a.batch = batch;
  for (int l = 0; l <= d->n_layers; ++l) a.dims[l] = d->dims[l];
  for (int l = 0; l < d->n_layers; ++l) {
    a.acts[l] = d->acts[l];
    a.wfrag[l] = (const bf16_t*)(backward ? d->wfrag_bwd[l] : d->wfrag_fwd[l]);
    a.bias[l] = d->bias[l];
    a.dz_frag[l] = d->dx_only ? nullptr : (bf16_t*)d->dz_frag[l];  // dx_only: the dZ fragments have no reader
    // the sign plane of layer l's input only exists when layer l-1 has a sign-based activation
    const bool sign_ok = l >= 1 && (d->acts[l - 1] == RG_ACT_RELU || d->acts[l - 1] == RG_ACT_LEAKY_RELU);
    a.act_sign[l] = sign_ok ? (unsigned*)d->act_sign[l] : nullptr;
    a.db_part[l] = nullptr;
    if (!a.wfrag[l] && !(backward && l == 0)) return RG_EINVAL;
    a.wfrag_lo[l] = d->x3 ? (long)(backward ? wfrag_elems(d->dims[l], d->dims[l + 1]) : wfrag_elems(d->dims[l + 1], d->dims[l])) : 0;
    // (a grouped output layer's dZ fragments: group g's 32-row blocks start g blocks late, grouped_dz_rows)
    const bool grouped_out = d->tile_key && l == d->n_layers - 1;
    a.dz_lo[l] = d->x3 ? (long)frag_elems(grouped_out ? grouped_dz_rows(batch, d->n_groups) : batch, d->dims[l + 1]) : 0;
    // one-plane dZ for the stack's weight gradient (x3_dz_planes): the backward launch skips the lo-plane stores; a grouped
    // output layer's dZ keeps both planes (rg_group_head_wgrad reads them)
    if (d->x3 && !grouped_out && x3_dz_planes() == 1) a.dz_lo[l] = -1;
  }
  for (int l = 0; l <= d->n_layers; ++l) {
    a.act_frag[l] = (bf16_t*)(l < d->n_layers ? d->act_frag[l] : nullptr);
    a.act_lo[l] = d->x3 ? (long)frag_elems(batch, d->dims[l]) : 0;
  }
  a.pitch = fused_pitch(d);
  a.rowmap = d->rowmap;
  a.tile_key = d->tile_key; a.row_begin = d->row_begin; a.n_groups = d->n_groups; a.group_stride = backward ? d->group_stride_bwd : d->group_stride_fwd;
  a.out_scatter = d->tile_key ? d->out_scatter : 0;
  a.x2 = d->x2; a.ldx2 = d->ldx2; a.x_split = d->x2 ? d->x_split : 0; a.dx_col0 = d->dx_col0;
  a.x2_is_f32 = d->x2_dtype == RG_DT_F32;
  a.x = nullptr; a.ldx = 0; a.x_is_f32 = 0; a.out32 = nullptr; a.ldo = 0; a.dout32 = nullptr; a.lddo = 0;
  a.dx32 = nullptr; a.lddx = 0; a.save = 0;
  return RG_OK;
}

// db[g * Ng + n] = sum over the segments of group g of db_part[unit + g][n] (qr_grouped.hip; db_part row pitch Ng): the
// backward kernel's workgroup `unit` (unit_rows rows of the grouped space: 128 for the bf16 kernels, 64 for the split-bf16 ones)
// writes the column sums of its rows of group g to row unit + g — distinct for distinct segments, contiguous per group
void grouped_bias_reduce_launch(const float* db_part, const int* row_begin, int n_groups, int Ng, float* db, int unit_rows,
                                hipStream_t stream);

// split-bf16 kernels (mlp_fused_x3.hip); `a` filled by fill_args, launch geometry decided there
int x3_forward_launch(const rg_mlp_desc* d, MlpArgs& a, hipStream_t stream);
int x3_backward_launch(const rg_mlp_desc* d, MlpArgs& a, hipStream_t stream);
constexpr int X3_BM = 64;  // rows per workgroup of the split-bf16 kernels

}  // namespace rg

// mlp_fused_x3.hip — the fused FullyConnected stack in split-bf16 ("bf16x3") arithmetic: fp32-class
// results from the bf16 MFMA pipe.
//
// Why: BASELINE.json's north_star asks for Q-values within 1e-4 of the fp32 reference AND for the dense layers
// on the bf16 matrix cores.  Plain bf16 operands are ~2e-2 away (8 mantissa bits); exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32) peaks at 157 TFLOP/s.  Here every operand x travels as two bf16 numbers,
//     hi = bf16(x),  lo = bf16(x - hi)          (x - hi is exact in fp32; hi + lo carries 16 mantissa bits)
// and a product a*b is evaluated as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  by three v_mfma_f32_32x32x16_bf16
// into the same fp32 accumulator (the dropped lo*lo term and the truncation of lo are ~2^-16 relative).
// Measured on the reference's C2 network (BASELINE.md §2 probe): max |dQ| 4e-5.
//
// Layout differences from mlp_fused.hip (everything else — C-fragment order of saved activations, weights in
// B-fragment order streamed L2 -> VGPR with a register ring, two-phase epilogue, sign planes, K rotation — is
// shared through rg_mlp_frag.h):
//   * the LDS tile holds BOTH planes, so a workgroup owns 64 rows instead of 128:
//       hi plane 64 x PITCH bf16 at offset 0, lo plane at offset 64*PITCH   (2 * 64 * 520 * 2 B = 133 KB)
//   * every fragment buffer in HBM (weights, saved activations, dZ) is [hi plane | lo plane];
//   * a wave still owns 32*TN output columns, now 2 x TN accumulator tiles; per K chunk it issues
//     3 * 2 * TN MFMAs for 2*TN weight fragments (hi, lo) and 4 LDS fragments: the MFMA : load ratio is 1.5x
//     that of the bf16 kernel, weight traffic per MFMA 4/3 of it.
// Replaces: FullyConnectedNetwork.forward (reagent/models/fully_connected_network.py:157-163) and its
// autograd backward, in the accuracy class of the reference's fp32 CPU arithmetic.
#include "rg_mlp_frag.h"

// weight-fragment ring depth of the 8-wave kernels.  Round 3, same box, C2 step: ring 2 1.128-1.139 ms against 1.143-1.144
// with ring 4 (saving forward 234 -> 227 us, backward 192 -> 188) — as for the bf16 kernels, a deeper ring only lengthens
// the L2 queues.
#ifndef RG_X3_RING
#define RG_X3_RING 2
#endif

namespace rg {

constexpr int X3_TM = X3_BM / 32;

// (v0, v1) -> packed hi pair and packed lo pair
__device__ __forceinline__ void split_pack(float v0, float v1, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2(v0, v1);
  const float h0 = __builtin_bit_cast(float, hi << 16), h1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  lo = pack_bf16x2(v0 - h0, v1 - h1);
}

// rows [row_base, row_base+64) x cols [0, ncols_pad) of a row-major matrix -> hi / lo LDS planes
template <typename T, int THREADS, int LO>
__device__ __forceinline__ void load_tile_split(bf16_t* act, int pitch, const T* src, long ld, int row_base, int nrows,
                                                int ncols, int ncols_pad, int tid) {
  const int cpr = ncols_pad / 8;
  const int total = X3_BM * cpr;
  const bool vec = ((ld % 8) == 0) && ((((uintptr_t)src) & 15) == 0);
  if (vec && (ncols & 7) == 0 && ncols > 0) {
    // aligned rows: every chunk of this thread requested before the first is used (C2: two dependent HBM round trips
    // per thread became one); out-of-range chunks read a clamped address and are replaced by zeros
    constexpr int U = 2;
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
        float f[8];
        if (sizeof(T) == 4) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = raw[u][0][e];
            f[4 + e] = raw[u][sizeof(T) == 4 ? 1 : 0][e];
          }
        } else {
          const u16x8 v = __builtin_bit_cast(u16x8, raw[u][0]);
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
        }
        if (row_base + r >= nrows || k0 >= ncols) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = 0.f;
        }
        u32x4 h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned hh, ll;
          split_pack(f[2 * e], f[2 * e + 1], hh, ll);
          h[e] = hh;
          l[e] = ll;
        }
        *(u32x4*)&act[r * pitch + k0] = h;
        *(u32x4*)&act[LO + r * pitch + k0] = l;
      }
    }
    return;
  }
  for (int c = tid; c < total; c += THREADS) {
    const int r = c / cpr, k0 = (c % cpr) * 8;
    const int grow = row_base + r;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    if (grow < nrows && k0 < ncols) {
      const T* p = src + (long)grow * ld + k0;
      if (vec && k0 + 8 <= ncols) {
        if (sizeof(T) == 4) {
          const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = b[e];
          }
        } else {
          const u16x8 v = *(const u16x8*)p;
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e < ncols) f[e] = cvt_in(p[e]);
      }
    }
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, ll;
      split_pack(f[2 * e], f[2 * e + 1], hh, ll);
      h[e] = hh;
      l[e] = ll;
    }
    *(u32x4*)&act[r * pitch + k0] = h;
    *(u32x4*)&act[LO + r * pitch + k0] = l;
  }
}

// the same through a row map (grouped space, qr_grouped.hip): tile row r <- src row rowmap[row_base + r] (-1: zeros).
// Two chunks per thread in flight, like the aligned path above.
template <typename T, int THREADS, int LO>
__device__ __forceinline__ void load_tile_split_mapped(bf16_t* act, int pitch, const T* src, long ld, const int* rowmap,
                                                       int row_base, int ncols, int ncols_pad, int tid) {
  const int cpr = ncols_pad / 8;
  const int total = X3_BM * cpr;
  const bool vec = ((ld % 8) == 0) && ((((uintptr_t)src) & 15) == 0);
  for (int c = tid; c < total; c += THREADS) {
    const int r = c / cpr, k0 = (c % cpr) * 8;
    const int grow = rowmap[row_base + r];
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
    if (grow >= 0 && k0 < ncols) {
      const T* p = src + (long)grow * ld + k0;
      if (vec && k0 + 8 <= ncols) {
        if (sizeof(T) == 4) {
          const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f[e] = a[e];
            f[4 + e] = b[e];
          }
        } else {
          const u16x8 v = *(const u16x8*)p;
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32(v[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (k0 + e < ncols) f[e] = cvt_in(p[e]);
      }
    }
    u32x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned hh, ll;
      split_pack(f[2 * e], f[2 * e + 1], hh, ll);
      h[e] = hh;
      l[e] = ll;
    }
    *(u32x4*)&act[r * pitch + k0] = h;
    *(u32x4*)&act[LO + r * pitch + k0] = l;
  }
}

// LDS plane (64 rows x ntiles*32 cols) -> C-fragment order in global memory (two 32-row blocks per workgroup)
// (only the blocks [mbl0, mbl1) of the plane; block mbl goes to fragment block mb_base + mbl)
__device__ __forceinline__ void emit_frags_x3(const bf16_t* plane, int pitch, int ntiles, bf16_t* dst, int mb_base,
                                              int wave, int n_waves, int lane, int mbl0 = 0, int mbl1 = X3_TM) {
  const int lr = lane & 31, lg = lane >> 5;
  const int total = (mbl1 - mbl0) * ntiles * 2;
  for (int f = wave; f < total; f += n_waves) {
    const int h = f & 1, nt = (f >> 1) % ntiles, mbl = mbl0 + (f >> 1) / ntiles;
    u16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = plane[(mbl * 32 + frag_row(h, e, lg)) * pitch + nt * 32 + lr];
    *(u16x8*)(dst + frag_offset(mb_base + mbl, nt, ntiles, h, lane)) = v;
  }
}

// ---- main loop: this wave's [64 x 32*TN] slice over K, three MFMAs per (A, B) fragment pair -------------
// Same software pipeline as wide_mainloop (rg_mlp_frag.h): weight fragments (hi and lo) prefetched RING-1
// chunks ahead from L2 into a register ring, activation fragments (hi and lo) one chunk ahead from LDS.
// The three terms of a chunk are issued term-major (all tiles' lo*hi, then hi*lo, then hi*hi), so that
// consecutive MFMAs never wait on each other's accumulator.
template <int TN, int RING, int LO>
__device__ __forceinline__ void x3_mainloop(const bf16_t* act, int pitch, int KC, const bf16_t* wf_wave, long wlo,
                                            long nt_stride, f32x16 (&acc)[X3_TM][TN], int lane, int rot,
                                            int prio_phase = 0) {
  static_assert(RING >= 2 && RING % 2 == 0, "the A double buffer alternates with the ring slot parity");
  const int lr = lane & 31, lg = lane >> 5;
  auto kx = [&](int kc) { const int k = kc + rot; return k >= KC ? k - KC : k; };
  const bf16_t* arow = act + lr * pitch + lg * 8;
  const int tm_stride = 32 * pitch;
  u16x8 ah[2][X3_TM], al[2][X3_TM], bh[RING][TN], bl[RING][TN];
  auto loadB = [&](int s, int kc) {
    const bf16_t* chunk = wf_wave + (long)kx(kc) * 512;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      bh[s][tn] = *(const u16x8*)(chunk + tn * nt_stride + lane * 8);
      bl[s][tn] = *(const u16x8*)(chunk + wlo + tn * nt_stride + lane * 8);
    }
  };
  auto loadA = [&](int s, int kc) {
    const int off = kx(kc) * 16;
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm) {
      ah[s][tm] = *(const u16x8*)(arow + tm * tm_stride + off);
      al[s][tm] = *(const u16x8*)(arow + LO + tm * tm_stride + off);
    }
  };
  auto mma = [&](int sa, int sb) {
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(al[sa][tm], bh[sb][tn], acc[tm][tn]);
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(ah[sa][tm], bl[sb][tn], acc[tm][tn]);
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(ah[sa][tm], bh[sb][tn], acc[tm][tn]);
  };
  if (KC % RING == 0) {
#pragma unroll
    for (int s = 0; s < RING - 1; ++s) loadB(s, s);
    loadA(0, 0);
    int kc = 0;
    for (; kc < KC - RING; kc += RING) {
      // (round 3, same-box A/B in the C2 step: no priorities at all, or waves 0-3 at priority 3 for the whole loop,
      // measured 1.126-1.128 ms/step against 1.129-1.133 with this hand-off — within the run-to-run spread)
      if (prio_phase && kc * 2 < KC) RG_SETPRIO(1);
      else RG_SETPRIO(0);
#pragma unroll
      for (int s = 0; s < RING; ++s) {
        loadB((s + RING - 1) % RING, kc + s + RING - 1);
        loadA((s + 1) & 1, kc + s + 1);
        sched_fence();
        mma(s & 1, s);
        sched_fence();
      }
    }
    RG_SETPRIO(0);
#pragma unroll
    for (int s = 0; s < RING; ++s) {  // last block: only the loads that are still in range
      if (s == 0) loadB(RING - 1, kc + RING - 1);
      if (s < RING - 1) loadA((s + 1) & 1, kc + s + 1);
      sched_fence();
      mma(s & 1, s);
      sched_fence();
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < RING - 1; ++s)
    if (s < KC) loadB(s, s);
  loadA(0, 0);
  for (int kc = 0; kc < KC; kc += RING) {
#pragma unroll
    for (int s = 0; s < RING; ++s) {
      if (kc + s < KC) {
        if (kc + s + RING - 1 < KC) loadB((s + RING - 1) % RING, kc + s + RING - 1);
        if (kc + s + 1 < KC) loadA((s + 1) & 1, kc + s + 1);
        sched_fence();
        mma(s & 1, s);
        sched_fence();
      }
    }
  }
}

// acc += A . B (three MFMAs per fragment pair) for one MORE segment of a boundary unit of a grouped layer — see
// segment_accumulate (rg_mlp_frag.h): a plain loop, one chunk at a time, kept small on purpose
template <int TN, int LO>
__device__ __forceinline__ void x3_segment_accumulate(const bf16_t* act, int pitch, int KC, const bf16_t* wf_wave, long wlo,
                                                      long nt_stride, f32x16 (&acc)[X3_TM][TN], int lane) {
  const bf16_t* arow = act + (lane & 31) * pitch + (lane >> 5) * 8;
  for (int kc = 0; kc < KC; ++kc) {
    u16x8 ah[X3_TM], al[X3_TM], bh[TN], bl[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      bh[tn] = *(const u16x8*)(wf_wave + (long)kc * 512 + tn * nt_stride + lane * 8);
      bl[tn] = *(const u16x8*)(wf_wave + wlo + (long)kc * 512 + tn * nt_stride + lane * 8);
    }
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm) {
      ah[tm] = *(const u16x8*)(arow + tm * 32 * pitch + kc * 16);
      al[tm] = *(const u16x8*)(arow + LO + tm * 32 * pitch + kc * 16);
    }
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        acc[tm][tn] = mfma_32x32x16_bf16(al[tm], bh[tn], acc[tm][tn]);
        acc[tm][tn] = mfma_32x32x16_bf16(ah[tm], bl[tn], acc[tm][tn]);
        acc[tm][tn] = mfma_32x32x16_bf16(ah[tm], bh[tn], acc[tm][tn]);
      }
  }
}

// one 32x32 output tile over K (narrow output layers, the input-gradient layer): pairs of chunks, the next
// pair's four fragments per chunk in flight during the current pair's MFMAs
template <int LO>
__device__ __forceinline__ f32x16 x3_tile_kloop(const bf16_t* act, int pitch, int KC, const bf16_t* wf, long wlo, int tm,
                                                int nt, int lane, int kc_lo = 0, int kc_hi = -1) {
  const int lr = lane & 31, lg = lane >> 5;
  const bf16_t* arow = act + (tm * 32 + lr) * pitch + lg * 8 + kc_lo * 16;
  const bf16_t* wl = wf + ((long)nt * KC + kc_lo) * 512 + lane * 8;
  KC = (kc_hi < 0 ? KC : kc_hi) - kc_lo;  // from here on: the chunks [kc_lo, kc_hi) of this call
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int G = 2;
  u16x8 ah[2][G], al[2][G], bh[2][G], bl[2][G];
  auto load = [&](int s, int kc0) {
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const int kc = kc0 + i < KC ? kc0 + i : KC - 1;  // clamped; the extra products are skipped below
      bh[s][i] = *(const u16x8*)(wl + (long)kc * 512);
      bl[s][i] = *(const u16x8*)(wl + wlo + (long)kc * 512);
      ah[s][i] = *(const u16x8*)(arow + kc * 16);
      al[s][i] = *(const u16x8*)(arow + LO + kc * 16);
    }
  };
  auto mma = [&](int s, int kc0) {
#pragma unroll
    for (int i = 0; i < G; ++i)
      if (kc0 + i < KC) {
        acc = mfma_32x32x16_bf16(al[s][i], bh[s][i], acc);
        acc = mfma_32x32x16_bf16(ah[s][i], bl[s][i], acc);
        acc = mfma_32x32x16_bf16(ah[s][i], bh[s][i], acc);
      }
  };
  load(0, 0);
  for (int kc = 0; kc < KC; kc += 2 * G) {
    if (kc + G < KC) load(1, kc + G);
    sched_fence();
    mma(0, kc);
    sched_fence();
    if (kc + G < KC) {
      if (kc + 2 * G < KC) load(0, kc + 2 * G);
      sched_fence();
      mma(1, kc + G);
      sched_fence();
    }
  }
  return acc;
}

// sign bits of a wave's [64 x 32*TN] slice: TN dwords per lane, dword tn: bit tm*16 + r
__device__ __forceinline__ long x3_sign_offset(int wg, int wave, int lane, int TN, int width) {
  return (long)wg * (2 * width) + ((long)wave * 64 + lane) * TN;
}

template <int TN>
__device__ __forceinline__ void x3_store_packed_tiles(bf16_t* plane, int pitch, const unsigned (&PK)[X3_TM][TN][8],
                                                      int wave, int lane) {
  const int lr = lane & 31;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int col = (wave * TN + tn) * 32 + lr;
    static_for<0, X3_TM>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      store_packed_to_lds(plane, pitch, tm * 32, col, lane, PK[tm][tn]);
    });
  });
}

template <int TN, int ACT>
__device__ __forceinline__ void x3_fwd_pack(f32x16 (&acc)[X3_TM][TN], const float* bias, bf16_t* save_dst, long save_lo,
                                            unsigned* sign_dst, int NT, int mb_base, int wave, int lane,
                                            unsigned (&PH)[X3_TM][TN][8], unsigned (&PL)[X3_TM][TN][8]) {
  lane = opaque(lane);
  const int lr = lane & 31;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int nt = wave * TN + tn, col = nt * 32 + lr;
    const float b = bias ? bias[col] : 0.f;
    unsigned sg = 0u;
    static_for<0, X3_TM>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = act_t<ACT>(acc[tm][tn][r] + b);
#pragma unroll
      for (int i = 0; i < 8; ++i) split_pack(v[2 * i], v[2 * i + 1], PH[tm][tn][i], PL[tm][tn][i]);
      if (save_dst) {
        store_packed_frags(save_dst, mb_base + tm, nt, NT, lane, PH[tm][tn]);
        store_packed_frags(save_dst + save_lo, mb_base + tm, nt, NT, lane, PL[tm][tn]);
      }
      if (act_is_sign_based<ACT>() && sign_dst) sg |= positive_bits<ACT == ACT_RELU>(v) << (tm * 16);
    });
    if (act_is_sign_based<ACT>() && sign_dst) sign_dst[x3_sign_offset(mb_base / X3_TM, wave, lane, TN, NT * 32) + tn] = sg;
  });
}

// dZ_below = dH * act'(H_below); column sums of dZ_below (bias gradient) for this workgroup
template <int TN, int ACT, bool USE_SIGN, bool STORE_DZ = true>
__device__ __forceinline__ void x3_bwd_pack(f32x16 (&acc)[X3_TM][TN], const bf16_t* h_frag, long h_lo,
                                            const unsigned (&sg)[TN], bf16_t* dz_dst, long dz_lo, float* db_part, int NT,
                                            int mb_base, int wave, int lane, unsigned (&PH)[X3_TM][TN][8],
                                            unsigned (&PL)[X3_TM][TN][8]) {
  lane = opaque(lane);
  const int lr = lane & 31;
  static_for<0, TN>([&](auto tn_c) __attribute__((always_inline)) {
    constexpr int tn = decltype(tn_c)::value;
    const int nt = wave * TN + tn, col = nt * 32 + lr;
    float colsum = 0.f;
    static_for<0, X3_TM>([&](auto tm_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value;
      float v[16];
      if (USE_SIGN && act_is_sign_based<ACT>()) {
        const unsigned bits = sg[tn] >> (tm * 16);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float g = ((bits >> r) & 1u) ? 1.f : (ACT == ACT_RELU ? 0.f : 0.01f);
          v[r] = acc[tm][tn][r] * g;
          colsum += v[r];
        }
      } else {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const long off = frag_offset(mb_base + tm, nt, NT, h, lane);
          const u16x8 hf = *(const u16x8*)(h_frag + off), lf = *(const u16x8*)(h_frag + h_lo + off);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[8 * h + e] = acc[tm][tn][8 * h + e] * act_grad_t<ACT>(bf16_to_f32(hf[e]) + bf16_to_f32(lf[e]));
            colsum += v[8 * h + e];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) split_pack(v[2 * i], v[2 * i + 1], PH[tm][tn][i], PL[tm][tn][i]);
      if (STORE_DZ) {
        store_packed_frags(dz_dst, mb_base + tm, nt, NT, lane, PH[tm][tn]);
        // dz_lo < 0: the weight gradient reads dZ as ONE bf16 plane (rg_mlp_frag.h: x3_dz_planes) — nobody reads a lo plane
        if (dz_lo >= 0) store_packed_frags(dz_dst + dz_lo, mb_base + tm, nt, NT, lane, PL[tm][tn]);
      } else {
        pin_packed(PH[tm][tn]);
        pin_packed(PL[tm][tn]);
      }
    });
    colsum += shfl_xor(colsum, 32);
    if (db_part && lane < 32) db_part[col] = colsum;
  });
}

// Grouped forward, the LAST segment of a 64-row unit (mlp_fwd_x3_body; the bf16 kernel's grouped_whole_tile_out on two planes):
// wave w sums column tile w of the group's [N, K] layer for both row tiles in the pipelined main loop, the 64 x N outputs are
// staged in the activation planes — dead once every wave has left its K loop — and leave as whole rows, a wave per row.
#ifndef RG_X3_GROUPED_RING
#define RG_X3_GROUPED_RING 4
#endif
template <int NW, int LO>
__device__ __forceinline__ void x3_grouped_whole_unit_out(bf16_t* act, int pitch, int KC, const bf16_t* wf_out, long wlo, const float* b_out,
                                                          int N, int NTo, int out_act, int lo, int hi, int row_base, const int* scatter,
                                                          int batch, float* out32, long ldo) {
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6), lr = lane & 31, lg = lane >> 5;
  const int P = NTo * 32 + 4, np = N >> 2;
  f32x16 acc2[X3_TM][1];
#pragma unroll
  for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[tm][0][r] = 0.f;
  const int col = wave * 32 + lr;
  float b = 0.f;
  if (wave < NTo) {
    if (b_out && col < N) b = b_out[col];  // (requested before the K loop)
    x3_mainloop<1, RG_X3_GROUPED_RING, LO>(act, pitch, KC, wf_out + (long)wave * KC * 512, wlo, 0, acc2, lane, k_rotation(blockIdx.x, wave, KC));
  }
  __syncthreads();  // every wave is done reading the layer input
  float* stage = (float*)act;
  if (wave < NTo) {
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc2[tm][0][r] + b;
      act_apply_n(v, out_act);
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * P + col] = v[r];
    }
  }
  __syncthreads();
  for (int rel = lo + wave; rel < hi; rel += NW) {  // a wave per row: np <= 64 16-byte pieces
    int row = row_base + rel;
    if (scatter) row = scatter[row];  // back to batch order; padding rows (-1) are dropped
    if (row >= 0 && (scatter || row < batch) && lane < np)
      stream_store(*(const f32x4*)(stage + rel * P + lane * 4), (f32x4*)(out32 + (long)row * ldo + lane * 4));
  }
}

// GROUPED: the launch of a stack whose output layer takes per-tile weights (rg_mlp_desc.tile_key, qr_grouped.hip: QR-DQN's
// wide layer, one action's [N, H] slice per 128-row tile of the grouped row space).  A 64-row workgroup is HALF such a tile:
// unit u = rows [64 u, 64 u + 64), tile u / 2.  Units go to the XCDs in eighths of the (group-sorted) unit list, as the
// bf16 kernel's tiles do (grouped_tile), and the launch is its own instantiation.
template <int TN, int NW, int PITCH, bool GROUPED>
__device__ __forceinline__ void mlp_fwd_x3_body(const MlpArgs& a) {
  constexpr int THREADS = NW * 64, RING = RG_X3_RING, LO = X3_BM * PITCH;
  RG_DYN_LDS(smem);
  bf16_t* act = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  const int n_units = (a.batch + X3_BM - 1) / X3_BM;
  const int unit = GROUPED ? grouped_tile(blockIdx.x, n_units) : (int)blockIdx.x;
  if (GROUPED && unit >= n_units) return;  // padding blocks (workgroup-uniform)
  const int row_base = unit * X3_BM;
  constexpr int pitch = PITCH;
  const int k0p = round_up(a.dims[0], 32);
  RG_STAMP(0);
  if (GROUPED) {  // grouped space: rows gathered through the map
    if (a.x_is_f32)
      load_tile_split_mapped<float, THREADS, LO>(act, pitch, (const float*)a.x, a.ldx, a.rowmap, row_base, a.dims[0], k0p, tid);
    else
      load_tile_split_mapped<bf16_t, THREADS, LO>(act, pitch, (const bf16_t*)a.x, a.ldx, a.rowmap, row_base, a.dims[0], k0p, tid);
  } else if (a.x2) {  // two panels (state | action): columns [0, x_split) from x, the rest from x2
    const int n2 = a.dims[0] - a.x_split;
    if (a.x_is_f32)
      load_tile_split<float, THREADS, LO>(act, pitch, (const float*)a.x, a.ldx, row_base, a.batch, a.x_split, a.x_split, tid);
    else
      load_tile_split<bf16_t, THREADS, LO>(act, pitch, (const bf16_t*)a.x, a.ldx, row_base, a.batch, a.x_split, a.x_split, tid);
    if (a.x2_is_f32)
      load_tile_split<float, THREADS, LO>(act + a.x_split, pitch, (const float*)a.x2, a.ldx2, row_base, a.batch, n2, k0p - a.x_split, tid);
    else
      load_tile_split<bf16_t, THREADS, LO>(act + a.x_split, pitch, (const bf16_t*)a.x2, a.ldx2, row_base, a.batch, n2, k0p - a.x_split, tid);
  } else if (a.x_is_f32)
    load_tile_split<float, THREADS, LO>(act, pitch, (const float*)a.x, a.ldx, row_base, a.batch, a.dims[0], k0p, tid);
  else
    load_tile_split<bf16_t, THREADS, LO>(act, pitch, (const bf16_t*)a.x, a.ldx, row_base, a.batch, a.dims[0], k0p, tid);
  __syncthreads();
  RG_STAMP(1);
  if (a.save == 1 && a.act_frag[0]) {
    emit_frags_x3(act, pitch, k0p / 32, a.act_frag[0], unit * X3_TM, wave, NW, lane);
    emit_frags_x3(act + LO, pitch, k0p / 32, a.act_frag[0] + a.act_lo[0], unit * X3_TM, wave, NW, lane);
  }

  for (int l = 0; l < a.n_layers; ++l) {
    const int K = a.dims[l], N = a.dims[l + 1];
    const int KC = (K + 15) / 16;
    if (l < a.n_layers - 1) {  // hidden layer, N == 32 * TN * NW
      f32x16 acc[X3_TM][TN];
#pragma unroll
      for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
      const long nt_stride = (long)KC * 512;
      x3_mainloop<TN, RING, LO>(act, pitch, KC, a.wfrag[l] + (long)(wave * TN) * nt_stride, a.wfrag_lo[l], nt_stride, acc,
                                lane, k_rotation(blockIdx.x, wave, KC), wave / (NW / 2));
      RG_STAMP(2 + 4 * l);
      unsigned PH[X3_TM][TN][8], PL[X3_TM][TN][8];
      unsigned* sign_dst = a.save ? a.act_sign[l + 1] : nullptr;
      RG_DISPATCH_ACT(a.acts[l], (x3_fwd_pack<TN, A_>(acc, a.bias[l], fwd_save_dst(a, l),
                                                      a.act_lo[l + 1], sign_dst, N / 32, unit * X3_TM, wave, lane, PH,
                                                      PL)));
      RG_STAMP(3 + 4 * l);
      __syncthreads();  // every wave is done reading the layer input
      RG_STAMP(4 + 4 * l);
      x3_store_packed_tiles<TN>(act, pitch, PH, wave, lane);
      x3_store_packed_tiles<TN>(act + LO, pitch, PL, wave, lane);
      __syncthreads();
      RG_STAMP(5 + 4 * l);
    } else if (GROUPED) {  // grouped output layer: once per segment of the unit's rows, that group's weight / bias slice
      const int NTo = (N + 31) / 32;
      const int out_act = a.acts[l];
      int seg_g = a.tile_key[unit >> 1];
      RowSegment seg;
      while (next_segment(a.row_begin, a.n_groups, row_base, X3_BM, seg_g, seg)) {  // (workgroup-uniform)
        const int grp = seg.grp;
        const int tm0 = seg.lo >> 5, tm1 = (seg.hi + 31) >> 5;  // the segment's 32-row tiles
        const bf16_t* wf_out = a.wfrag[l] + (long)grp * a.group_stride;
        const float* b_out = a.bias[l] ? a.bias[l] + (long)grp * N : nullptr;
        if (a.stage_out && NTo <= NW) {
          // a wide output (QR-DQN: 200 quantiles, 800-byte rows) leaves as WHOLE ROWS, 16 bytes per lane, through a staging
          // area behind the two activation planes — one 32-row tile at a time, wave w computing its column tile w (what
          // mlp_fwd_fused_body does for the bf16 stack: stored straight from the accumulators every wave instruction would
          // write partial cache lines)
          const int P = NTo * 32 + 4;  // floats per staged row
          const int np = N >> 2;       // 16-byte pieces per row
#if RG_GROUPED_WHOLE
          // Round 6 (as mlp_fwd_fused_body): the LAST segment of a unit leaves both activation planes dead after its K loop —
          // wave w sums column tile w for both row tiles in the pipelined main loop (one weight stream per wave instead of a
          // chain of x3_tile_kloop round trips per row tile), the 64 x N outputs are staged in the dead planes and leave as whole rows
          bool last_segment = false;
          if ((size_t)X3_BM * P * sizeof(float) <= (size_t)2 * LO * sizeof(bf16_t)) {
            int g2 = seg_g;
            RowSegment s2;
            last_segment = !next_segment(a.row_begin, a.n_groups, row_base, X3_BM, g2, s2);  // (workgroup-uniform)
          }
          if (last_segment) {
            x3_grouped_whole_unit_out<NW, LO>(act, pitch, KC, wf_out, a.wfrag_lo[l], b_out, N, NTo, out_act, seg.lo, seg.hi, row_base,
                                              a.out_scatter ? a.rowmap : nullptr, a.batch, a.out32, a.ldo);
            break;
          }
#endif
          float* stage = (float*)(act + 2 * LO);
          for (int tm = tm0; tm < tm1; ++tm) {
            if (wave < NTo) {
              const f32x16 acc = x3_tile_kloop<LO>(act, pitch, KC, wf_out, a.wfrag_lo[l], tm, wave, lane);
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
          for (int t = wave; t < X3_TM * NTo; t += NW) {
            const int tm = t % X3_TM, nt = t / X3_TM;
            if (tm < tm0 || tm >= tm1) continue;
            const f32x16 acc = x3_tile_kloop<LO>(act, pitch, KC, wf_out, a.wfrag_lo[l], tm, nt, lane);
            const int col = nt * 32 + lr;
            if (col < N) {
              const float b = b_out ? b_out[col] : 0.f;
              float ov[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) ov[r] = acc[r] + b;
              act_apply_n(ov, out_act);
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int rel = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
                if (rel < seg.lo || rel >= seg.hi) continue;  // another segment's row
                int row = row_base + rel;
                if (a.out_scatter) row = a.rowmap[row];
                if (row >= 0 && (a.out_scatter || row < a.batch)) a.out32[(long)row * a.ldo + col] = ov[r];
              }
            }
          }
        }
      }
      RG_STAMP(2 + 4 * l);
    } else {  // output layer: 32x32 tiles spread over the waves, fp32 result to HBM
      const int NTo = (N + 31) / 32;
      const int out_act = a.acts[l];
      constexpr int PARTS = NW / X3_TM;  // waves per tile when there is one column tile
      const bool split = NTo == 1 && KC >= 4 * PARTS;
      // RG_OUT_ROWSTORE (round 5, as in mlp_fwd_fused_body): a thin output layer's [64, N] result leaves as whole 16-byte pieces
      // — rows (N % 4 == 0) or, for a dense output and a full tile, the tile's block as one run (a critic's single column) —
      // through a staging area in the dead activation planes: every part puts its partial sums there and the store pass adds
      // them, ((p0 + p1) + p2) + p3 + bias as the accumulator hand-off below does; the bias is requested BEFORE the K loop.
      const bool aligned16 = (reinterpret_cast<uintptr_t>(a.out32) & 15) == 0;
      const bool dense_run = a.ldo == N && row_base + X3_BM <= a.batch;
      const bool rowstore = RG_OUT_ROWSTORE && split && !a.out_scatter && aligned16 && (dense_run || ((N & 3) == 0 && (a.ldo & 3) == 0));
      if (rowstore) {  // (workgroup-uniform; NW == PARTS * X3_TM waves, one (part, row tile) each)
        const int tm = wave % X3_TM, part = wave / X3_TM, per = (KC / PARTS + 1) / 2 * 2;
        const int lo = part * per, hi = part == PARTS - 1 ? KC : lo + per;
        const float b0 = (a.bias[l] && lr < N) ? a.bias[l][lr] : 0.f;
        const f32x16 acc = x3_tile_kloop<LO>(act, pitch, KC, a.wfrag[l], a.wfrag_lo[l], tm, 0, lane, lo, hi);
        __syncthreads();  // every wave is done reading the layer input
        float* stage = (float*)act;
        float* bias_s = stage + PARTS * (X3_BM * 32);
        if (lr < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) stage[part * (X3_BM * 32) + (tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg) * N + lr] = acc[r];
          if (wave == 0 && lg == 0) bias_s[lr] = b0;
        }
        __syncthreads();
        auto emit = [&](int i0, float* dst) {  // four consecutive staged floats i0 .. i0 + 3 -> dst
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = stage[i0 + e];
#pragma unroll
            for (int p = 1; p < PARTS; ++p) v += stage[p * (X3_BM * 32) + i0 + e];
            o[e] = v + bias_s[(i0 + e) % N];
          }
          act_apply_n(o, out_act);
          *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
        };
        if (dense_run) {
          for (int it = tid; it < (X3_BM * N) >> 2; it += THREADS) emit(it * 4, a.out32 + (long)row_base * N + it * 4);
        } else {
          const int np = N >> 2;
          for (int it = tid; it < X3_BM * np; it += THREADS) {
            const int rel = it / np, c4 = it - rel * np;
            if (row_base + rel < a.batch) emit(rel * N + c4 * 4, a.out32 + (long)(row_base + rel) * a.ldo + c4 * 4);
          }
        }
      }
      for (int t = wave; !rowstore && t < (split ? NW : X3_TM * NTo); t += NW) {
        const int tm = t % X3_TM, nt = split ? 0 : t / X3_TM;
        f32x16 acc;
        if (split) {
          // one column tile (<= 32 outputs): two 32x32 tiles for eight waves, a chain of L2 round trips.  Four waves
          // share a tile, a quarter of K each; parts 1..3 hand their sums over through the (by then dead)
          // activation planes and part 0 adds them in a fixed order.
          const int part = wave / X3_TM, per = (KC / PARTS + 1) / 2 * 2;
          const int lo = part * per, hi = part == PARTS - 1 ? KC : lo + per;
          acc = x3_tile_kloop<LO>(act, pitch, KC, a.wfrag[l], a.wfrag_lo[l], tm, 0, lane, lo, hi);
          __syncthreads();  // every wave is done reading the layer input
          float* hand = (float*)act + ((part * X3_TM + tm) * 64 + lane) * 16;
          if (part) {
#pragma unroll
            for (int r = 0; r < 16; r += 4) *(f32x4*)(hand + r) = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
          }
          __syncthreads();
          if (part) continue;
#pragma unroll
          for (int p = 1; p < PARTS; ++p) {
            const float* other = (const float*)act + ((p * X3_TM + tm) * 64 + lane) * 16;
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
              const f32x4 o = *(const f32x4*)(other + r);
              acc[r] += o[0]; acc[r + 1] += o[1]; acc[r + 2] += o[2]; acc[r + 3] += o[3];
            }
          }
        } else {
          acc = x3_tile_kloop<LO>(act, pitch, KC, a.wfrag[l], a.wfrag_lo[l], tm, nt, lane);
        }
        const int col = nt * 32 + lr;
        if (col < N) {
          const float b = a.bias[l] ? a.bias[l][col] : 0.f;
          float ov[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) ov[r] = acc[r] + b;
          act_apply_n(ov, out_act);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = row_base + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
            if (row < a.batch) a.out32[(long)row * a.ldo + col] = ov[r];
          }
        }
      }
      RG_STAMP(2 + 4 * l);
    }
  }
}

template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_fwd_x3_kernel(MlpArgs a) {
  mlp_fwd_x3_body<TN, NW, PITCH, false>(a);
}
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_fwd_x3_grouped_kernel(MlpArgs a) {
  mlp_fwd_x3_body<TN, NW, PITCH, true>(a);
}

// GROUPED: see mlp_bwd_fused_body (mlp_fused.hip) — the grouped layer's step once per segment of the unit's 64 rows, on
// masked copies of both dZ planes
template <int TN, int NW, int PITCH, bool DX_ONLY, bool GROUPED = false>
__device__ __forceinline__ void mlp_bwd_x3_body(const MlpArgs& a) {
  constexpr int THREADS = NW * 64, RING = RG_X3_RING, LO = X3_BM * PITCH;
  RG_DYN_LDS(smem);
  bf16_t* act = (bf16_t*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  // grouped launch (rg_mlp_desc.tile_key): 64-row units of the group-sorted row space, shared out to the XCDs in eighths
  const int n_units = round_up(a.batch, 128) / X3_BM;
  const int unit = GROUPED ? grouped_tile(blockIdx.x, n_units) : (int)blockIdx.x;
  if (unit >= n_units) return;  // padding blocks of a grouped launch (workgroup-uniform)
  const int row_base = unit * X3_BM;
  constexpr int pitch = PITCH;
  const int L = a.n_layers;
  const int nop = round_up(a.dims[L], 32);
  load_tile_split<float, THREADS, LO>(act, pitch, a.dout32, a.lddo, row_base, a.batch, a.dims[L], nop, tid);
  __syncthreads();
  if (!GROUPED) {
    if (!DX_ONLY) {
      emit_frags_x3(act, pitch, nop / 32, a.dz_frag[L - 1], unit * X3_TM, wave, NW, lane);
      if (a.dz_lo[L - 1] >= 0) emit_frags_x3(act + LO, pitch, nop / 32, a.dz_frag[L - 1] + a.dz_lo[L - 1], unit * X3_TM, wave, NW, lane);
    }
    if (a.db_part[L - 1] && tid < a.dims[L]) {
      float s = 0.f;
      for (int r = 0; r < X3_BM; ++r) s += bf16_to_f32(act[r * pitch + tid]) + bf16_to_f32(act[LO + r * pitch + tid]);
      a.db_part[L - 1][(long)unit * a.dims[L] + tid] = s;
    }
  }

  for (int l = L - 1; l >= 1; --l) {
    // dH = dZ_l (LDS, width dims[l+1]) . W_l -> [64, dims[l]] ; dZ_{l-1} = dH * act'(H_l)
    const int K = a.dims[l + 1], N = a.dims[l];
    const int KC = (K + 15) / 16;
    f32x16 acc[X3_TM][TN];
#pragma unroll
    for (int tm = 0; tm < X3_TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
    const long nt_stride = (long)KC * 512;
    unsigned sg[TN];
    const bool use_sign = a.act_sign[l] != nullptr;
#pragma unroll
    for (int i = 0; i < TN; ++i) sg[i] = use_sign ? a.act_sign[l][x3_sign_offset(unit, wave, lane, TN, N) + i] : 0u;
    // the grouped layer's first segment through the main loop like a plain layer's unit, further segments of a boundary unit
    // through x3_segment_accumulate (mlp_bwd_fused_body has the reasons)
    const bool grouped_layer = GROUPED && l == L - 1;
    bf16_t* cp = act + masked_copy_offset<PITCH>(2 * LO);  // [hi copy | lo copy], LO apart like the planes
    int seg_g = grouped_layer ? a.tile_key[unit >> 1] : 0;
    RowSegment seg{0, 0, X3_BM};
    bool more = grouped_layer ? next_segment(a.row_begin, a.n_groups, row_base, X3_BM, seg_g, seg) : true;
    auto segment_side = [&]() -> const bf16_t* {  // -> the segment's MFMA operand (hi plane; lo plane LO behind it)
      const bool whole = seg.lo == 0 && seg.hi == X3_BM;
      if (!whole) {
        copy_rows_masked<THREADS, X3_BM>(act, cp, pitch, nop, seg.lo, seg.hi, tid);
        copy_rows_masked<THREADS, X3_BM>(act + LO, cp + LO, pitch, nop, seg.lo, seg.hi, tid);
        __syncthreads();
      }
      const bf16_t* src = whole ? act : cp;
      emit_frags_x3(src, pitch, nop / 32, a.dz_frag[L - 1], unit * X3_TM + seg.grp, wave, NW, lane, seg.lo >> 5,
                    (seg.hi + 31) >> 5);
      emit_frags_x3(src + LO, pitch, nop / 32, a.dz_frag[L - 1] + a.dz_lo[L - 1], unit * X3_TM + seg.grp, wave, NW, lane,
                    seg.lo >> 5, (seg.hi + 31) >> 5);
      if (a.db_part[L - 1] && tid < a.dims[L]) {
        float s = 0.f;
        for (int r = seg.lo; r < seg.hi; ++r) s += bf16_to_f32(act[r * pitch + tid]) + bf16_to_f32(act[LO + r * pitch + tid]);
        a.db_part[L - 1][(long)(unit + seg.grp) * a.dims[L] + tid] = s;
      }
      return src;
    };
    if (more) {
      const bf16_t* src = act;
      if (grouped_layer) src = segment_side();
      const bf16_t* wl = a.wfrag[l] + (grouped_layer ? (long)seg.grp * a.group_stride : 0);  // the group's slice of W^T
      x3_mainloop<TN, RING, LO>(src, pitch, KC, wl + (long)(wave * TN) * nt_stride, a.wfrag_lo[l], nt_stride, acc, lane,
                                k_rotation(blockIdx.x, wave, KC), wave / (NW / 2));
    }
    if (grouped_layer) {
      while (more && next_segment(a.row_begin, a.n_groups, row_base, X3_BM, seg_g, seg)) {  // a boundary unit's other groups
        __syncthreads();  // every wave is done with the previous segment's copies
        const bf16_t* src = segment_side();
        x3_segment_accumulate<TN, LO>(src, pitch, KC, a.wfrag[l] + (long)seg.grp * a.group_stride + (long)(wave * TN) * nt_stride,
                                      a.wfrag_lo[l], nt_stride, acc, lane);
      }
    }
    float* dbp = a.db_part[l - 1] ? a.db_part[l - 1] + (long)unit * N : nullptr;
    unsigned PH[X3_TM][TN][8], PL[X3_TM][TN][8];
    if (use_sign) {
      RG_DISPATCH_ACT(a.acts[l - 1], (x3_bwd_pack<TN, A_, true, !DX_ONLY>(acc, a.act_frag[l], a.act_lo[l], sg, a.dz_frag[l - 1],
                                                               a.dz_lo[l - 1], dbp, N / 32, unit * X3_TM, wave, lane,
                                                               PH, PL)));
    } else {
      RG_DISPATCH_ACT(a.acts[l - 1], (x3_bwd_pack<TN, A_, false, !DX_ONLY>(acc, a.act_frag[l], a.act_lo[l], sg, a.dz_frag[l - 1],
                                                                a.dz_lo[l - 1], dbp, N / 32, unit * X3_TM, wave, lane,
                                                                PH, PL)));
    }
    __syncthreads();  // every wave is done reading dZ_l
    x3_store_packed_tiles<TN>(act, pitch, PH, wave, lane);
    x3_store_packed_tiles<TN>(act + LO, pitch, PL, wave, lane);
    __syncthreads();
  }
  if (a.dx32) {  // gradient w.r.t. the network input (e.g. the critic's action input in SAC)
    const int K = a.dims[1], N = a.dims[0];
    const int KC = (K + 15) / 16, NTi = (N + 31) / 32, nt0 = a.dx_col0 / 32;  // only the tiles from dx_col0 on
    for (int t = wave; t < X3_TM * (NTi - nt0); t += NW) {
      const int tm = t % X3_TM, nt = nt0 + t / X3_TM;
      const f32x16 acc = x3_tile_kloop<LO>(act, pitch, KC, a.wfrag[0], a.wfrag_lo[0], tm, nt, lane);
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
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_bwd_x3_kernel(MlpArgs a) {
  mlp_bwd_x3_body<TN, NW, PITCH, false>(a);
}
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_bwd_x3_dx_kernel(MlpArgs a) {
  mlp_bwd_x3_body<TN, NW, PITCH, true>(a);
}
template <int TN, int NW, int PITCH>
__global__ void RG_LAUNCH_BOUNDS(NW * 64, 1) mlp_bwd_x3_grouped_kernel(MlpArgs a) {
  mlp_bwd_x3_body<TN, NW, PITCH, false, true>(a);
}

#define RG_LAUNCH_X3(KERNEL, hidden, pitch, grid, lds, stream, args)                                         \
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

// The saved fragment matrices are padded to 128 rows (rg_frag_elems) and the weight-gradient kernel reads every
// 32-row block of them: a saving forward and the backward therefore cover the padded row count (the extra
// workgroup sees only out-of-range rows: zero inputs, zero dZ).
static inline int x3_grid(int batch, bool padded) {
  return padded ? (batch + 127) / 128 * (128 / X3_BM) : (batch + X3_BM - 1) / X3_BM;
}

int x3_forward_launch(const rg_mlp_desc* d, MlpArgs& a, hipStream_t stream) {
  size_t lds = (size_t)2 * X3_BM * a.pitch * sizeof(bf16_t);
  a.stage_out = a.out_lds = 0;
  if (d->tile_key) {
    // grouped: 64-row units of the padded (multiple of 128) row space, whole eighths of the unit list (grouped_tile)
    const int n_units = (a.batch + X3_BM - 1) / X3_BM;
    const dim3 grid((n_units + 7) / 8 * 8);
    const int No = d->dims[d->n_layers], NTo = (No + 31) / 32;
    const size_t stage = (size_t)32 * (NTo * 32 + 4) * sizeof(float);
    if (No > 64 && (No & 3) == 0 && (a.ldo & 3) == 0 && (((uintptr_t)a.out32) & 15) == 0 && lds + stage <= 160 * 1024) {
      a.stage_out = 1;
      lds += stage;
    }
    RG_LAUNCH_X3(mlp_fwd_x3_grouped_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
    return (int)hipGetLastError();
  }
  const dim3 grid(x3_grid(a.batch, a.save != 0));
  RG_LAUNCH_X3(mlp_fwd_x3_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
  return (int)hipGetLastError();
}

int x3_backward_launch(const rg_mlp_desc* d, MlpArgs& a, hipStream_t stream) {
  size_t lds = (size_t)2 * X3_BM * a.pitch * sizeof(bf16_t);
  const int n_wg = x3_grid(a.batch, true);
  const dim3 grid(d->tile_key ? (n_wg + 7) / 8 * 8 : n_wg);
  if (d->tile_key && a.pitch < 2 * 256 + 8) lds *= 2;  // a boundary unit's masked dZ copies live behind 264-wide planes
  if (d->dx_only) RG_LAUNCH_X3(mlp_bwd_x3_dx_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
  else if (d->tile_key) RG_LAUNCH_X3(mlp_bwd_x3_grouped_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
  else RG_LAUNCH_X3(mlp_bwd_x3_kernel, d->dims[1], a.pitch, grid, lds, stream, a);
  return (int)hipGetLastError();
}

}  // namespace rg

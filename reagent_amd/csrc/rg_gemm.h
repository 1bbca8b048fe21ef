// rg_gemm.h — the one GEMM core behind every FullyConnected layer op (forward, dgrad, wgrad).
//
// All three are expressed as the same "NT" product   C[M,N] = A[M,K] · B[N,K]^T   with BOTH
// operands contiguous along the reduction axis, which is the friendly layout for CDNA4 MFMA
// fragments (each lane supplies 8 consecutive-k bf16 / one f32 per operand):
//   forward : Y   = X   · W^T        A = X   [batch, in]    B = W    [out, in]   (nn.Linear layout)
//   dgrad   : dX  = dZ  · (W^T)^T    A = dZ  [batch, out]   B = W^T  [in, out]
//   wgrad   : dW  = dZ^T· (X^T)^T    A = dZ^T[out, batch]   B = X^T  [in, batch]   (split over batch)
// The transposed activation copies that wgrad needs are written by the producing kernel's
// epilogue (MFMA accumulators hold 4 consecutive rows of one column per lane, so the transposed
// store is the natural vector store), not by a separate transpose pass.
//
// Reference semantics replaced: torch.nn.Linear + activation inside
// reagent/models/fully_connected_network.py:101-153 and its autograd backward.
//
// Two arithmetic modes:
//   PrecF32  — v_mfma_f32_32x32x2_f32, exact fp32 products/accumulate (parity mode)
//   PrecBF16 — v_mfma_f32_32x32x16_bf16, bf16 operands, fp32 accumulate (throughput mode)
//
// Workgroup = 256 threads = 4 waves.  LDS single-buffered with register prefetch of the next
// K-slab (global loads in flight during the MFMA block), 2 barriers per slab, 37 KB (bf16) /
// 17 KB (f32) of LDS so 3-4 workgroups fit a CU.
#pragma once
#include <rg_platform.h>  // resolved via -I (product: csrc/, CPU test harness: tests/emu/)
#include <cstdint>
#include <type_traits>

namespace rg {

// compile-time loop: the body is instantiated once per index, so accumulator arrays indexed by it
// stay in registers even where `#pragma unroll` gives up ("unrolled size is too large")
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

enum { ACT_LINEAR = 0, ACT_RELU = 1, ACT_LEAKY_RELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_SOFTPLUS = 5 };

__device__ __forceinline__ float act_apply(float z, int act) {
  switch (act) {
    case ACT_RELU: return z > 0.f ? z : 0.f;
    case ACT_LEAKY_RELU: return z > 0.f ? z : 0.01f * z;
    case ACT_TANH: return tanhf(z);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case ACT_SOFTPLUS: return z > 20.f ? z : log1pf(expf(z));
    default: return z;
  }
}
// v[i] = act(v[i]) for a register array, the run-time activation tested ONCE.  (Round 6: `act == ACT_LINEAR ? z : act_apply(z, act)`
// per element of an unrolled loop puts a copy of every activation's code — tanhf, expf, log1pf — between two consecutive elements:
// the grouped forward's 64 staged outputs per lane were 17k instructions, each copy skipped by a branch the instruction cache
// pays for: 16.7k of a workgroup's cycles for 448 LDS writes.)
template <int N> __device__ __forceinline__ void act_apply_n(float (&v)[N], int act) {
  if (act == ACT_LINEAR) return;
  if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    return;
  }
#pragma unroll
  for (int i = 0; i < N; ++i) v[i] = act_apply(v[i], act);
}
// d act(z) / dz expressed through the activation OUTPUT h = act(z) (all six are invertible enough)
__device__ __forceinline__ float act_grad_from_output(float h, int act) {
  switch (act) {
    case ACT_RELU: return h > 0.f ? 1.f : 0.f;
    case ACT_LEAKY_RELU: return h > 0.f ? 1.f : 0.01f;
    case ACT_TANH: return 1.f - h * h;
    case ACT_SIGMOID: return h * (1.f - h);
    case ACT_SOFTPLUS: return 1.f - expf(-h);
    default: return 1.f;
  }
}

template <typename T> __device__ __forceinline__ T cvt_out(float v);
template <> __device__ __forceinline__ float cvt_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t cvt_out<bf16_t>(float v) { return f32_to_bf16(v); }
__device__ __forceinline__ float cvt_in(float v) { return v; }
__device__ __forceinline__ float cvt_in(bf16_t v) { return bf16_to_f32(v); }

// ---------------------------------------------------------------------------------------------
// tile configurations
template <int BM_, int BN_, int WM_, int WN_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
  static constexpr int THREADS = WM * WN * 64;
};
typedef TileCfg<128, 128, 2, 2> TileWide;   // 4 waves, each 64x64 (2x2 MFMA tiles)
typedef TileCfg<256, 32, 4, 1> TileNarrow;  // 4 waves, each 64x32; for N <= 32 outputs

struct GemmArgs {
  const void* A;
  const void* B;
  long lda, ldb;
  int M, N, K;
  int k_per_split;  // multiple of BK; == K rounded up when unsplit
  int splits;
  int bias_rows_only;  // unused placeholder for ABI stability
};

// XCD-aware decode of the linear workgroup id: workgroups that share an operand panel (same
// `outer`) are placed on the same XCD (hardware dispatches id b to XCD b % 8) so the panel is
// fetched from HBM once and re-read from that XCD's L2.
__device__ __forceinline__ void decode_wg(int L, int n_outer, int n_inner, int& outer, int& inner) {
  if ((n_outer & 7) == 0) {
    const int xcd = L & 7, w = L >> 3;
    outer = xcd + 8 * (w / n_inner);
    inner = w % n_inner;
  } else {
    outer = L / n_inner;
    inner = L % n_inner;
  }
}

// ---------------------------------------------------------------------------------------------
// precision policies: staging registers, LDS image, fragment reads + MFMA issue
struct PrecBF16 {
  typedef bf16_t T;
  static constexpr int BK = 64;     // K-slab per iteration
  static constexpr int VEC = 8;     // elements per 16-byte chunk
  static constexpr int PITCH = 72;  // LDS row pitch in elements (144 B: conflict-free ds_read_b128)
  typedef u16x8 Chunk;
  template <int ROWS> static constexpr int lds_elems() { return ROWS * PITCH; }

  template <int ROWS, int THREADS>
  struct Stage {
    static constexpr int NCH = ROWS * (BK / VEC);
    static constexpr int PER = (NCH + THREADS - 1) / THREADS;
    Chunk r[PER];
  };

  template <int ROWS, int THREADS>
  static __device__ __forceinline__ void load_global(Stage<ROWS, THREADS>& s, const T* src, long ld,
                                                     int row0, int nrows, int k0, int k_end,
                                                     bool vec_ok, int tid) {
    typedef Stage<ROWS, THREADS> S;
#pragma unroll
    for (int i = 0; i < S::PER; ++i) {
      const int c = tid + i * THREADS;
      Chunk v = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < S::NCH) {
        const int row = c >> 3, kc = c & 7;
        const int grow = row0 + row, gk = k0 + kc * 8;
        if (grow < nrows && gk < k_end) {
          const T* p = src + (long)grow * ld + gk;
          if (vec_ok && gk + 8 <= k_end) {
            v = *(const Chunk*)p;
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
              if (gk + e < k_end) v[e] = p[e];
          }
        }
      }
      s.r[i] = v;
    }
  }
  template <int ROWS, int THREADS>
  static __device__ __forceinline__ void store_lds(const Stage<ROWS, THREADS>& s, T* lds, int tid) {
    typedef Stage<ROWS, THREADS> S;
#pragma unroll
    for (int i = 0; i < S::PER; ++i) {
      const int c = tid + i * THREADS;
      if (c < S::NCH) {
        const int row = c >> 3, kc = c & 7;
        *(Chunk*)&lds[row * PITCH + kc * 8] = s.r[i];
      }
    }
  }
  // one K-slab of MFMA work for this wave. a_row0/b_row0: first LDS row of the wave's sub-tile.
  // SWAP: the operands trade places in the MFMA, so the accumulator tile is the TRANSPOSE of the
  // output tile (a lane then holds 4 consecutive COLUMNS of one output row; same products, same k order,
  // same values) — see the swapped epilogue of gemm_nt_kernel
  template <class C, int BIAS_MODE, int SWAP = 0>
  static __device__ __forceinline__ void compute(const T* As, const T* Bs, int a_row0, int b_row0,
                                                 int lane, f32x16 (&acc)[C::TM][C::TN],
                                                 f32x16 (&accb)[(C::TM > C::TN ? C::TM : C::TN)],
                                                 bool do_bias) {
    constexpr int TM = C::TM, TN = C::TN;
    const Chunk ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const int lr = lane & 31, lk = (lane >> 5) * 8;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      Chunk af[TM], bf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        af[tm] = *(const Chunk*)&As[(a_row0 + tm * 32 + lr) * PITCH + kk * 16 + lk];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        bf[tn] = *(const Chunk*)&Bs[(b_row0 + tn * 32 + lr) * PITCH + kk * 16 + lk];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = SWAP ? mfma_32x32x16_bf16(bf[tn], af[tm], acc[tm][tn]) : mfma_32x32x16_bf16(af[tm], bf[tn], acc[tm][tn]);
      if (BIAS_MODE == 1 && do_bias) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) accb[tm] = mfma_32x32x16_bf16(af[tm], ones, accb[tm]);
      }
      if (BIAS_MODE == 2 && do_bias) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) accb[tn] = mfma_32x32x16_bf16(ones, bf[tn], accb[tn]);
      }
    }
  }
};

struct PrecF32 {
  typedef float T;
  static constexpr int BK = 16;
  static constexpr int VEC = 4;
  typedef f32x4 Chunk;
  // LDS image is k-major [BK][ROWS + 4]: a fragment read is 32 consecutive floats per half-wave
  template <int ROWS> static constexpr int lds_elems() { return BK * (ROWS + 4); }

  template <int ROWS, int THREADS>
  struct Stage {
    static constexpr int NCH = ROWS * (BK / VEC);
    static constexpr int PER = (NCH + THREADS - 1) / THREADS;
    Chunk r[PER];
  };

  template <int ROWS, int THREADS>
  static __device__ __forceinline__ void load_global(Stage<ROWS, THREADS>& s, const T* src, long ld,
                                                     int row0, int nrows, int k0, int k_end,
                                                     bool vec_ok, int tid) {
    typedef Stage<ROWS, THREADS> S;
#pragma unroll
    for (int i = 0; i < S::PER; ++i) {
      const int c = tid + i * THREADS;
      Chunk v = {0.f, 0.f, 0.f, 0.f};
      if (c < S::NCH) {
        const int row = c >> 2, kc = c & 3;
        const int grow = row0 + row, gk = k0 + kc * 4;
        if (grow < nrows && gk < k_end) {
          const T* p = src + (long)grow * ld + gk;
          if (vec_ok && gk + 4 <= k_end) {
            v = *(const Chunk*)p;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (gk + e < k_end) v[e] = p[e];
          }
        }
      }
      s.r[i] = v;
    }
  }
  template <int ROWS, int THREADS>
  static __device__ __forceinline__ void store_lds(const Stage<ROWS, THREADS>& s, T* lds, int tid) {
    typedef Stage<ROWS, THREADS> S;
#pragma unroll
    for (int i = 0; i < S::PER; ++i) {
      const int c = tid + i * THREADS;
      if (c < S::NCH) {
        const int row = c >> 2, kc = c & 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) lds[(kc * 4 + e) * (ROWS + 4) + row] = s.r[i][e];
      }
    }
  }
  template <class C, int BIAS_MODE, int SWAP = 0>
  static __device__ __forceinline__ void compute(const T* As, const T* Bs, int a_row0, int b_row0,
                                                 int lane, f32x16 (&acc)[C::TM][C::TN],
                                                 f32x16 (&accb)[(C::TM > C::TN ? C::TM : C::TN)],
                                                 bool do_bias) {
    static_assert(SWAP == 0, "the swapped epilogue is a bf16 path");
    constexpr int TM = C::TM, TN = C::TN, AROWS = C::BM, BROWS = C::BN;
    const int lr = lane & 31, lk = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float af[TM], bf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) af[tm] = As[(ks * 2 + lk) * (AROWS + 4) + a_row0 + tm * 32 + lr];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bf[tn] = Bs[(ks * 2 + lk) * (BROWS + 4) + b_row0 + tn * 32 + lr];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x2_f32(af[tm], bf[tn], acc[tm][tn]);
      if (BIAS_MODE == 1 && do_bias) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) accb[tm] = mfma_32x32x2_f32(af[tm], 1.0f, accb[tm]);
      }
      if (BIAS_MODE == 2 && do_bias) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) accb[tn] = mfma_32x32x2_f32(1.0f, bf[tn], accb[tn]);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// epilogues.  Called with 4 accumulator values = rows row0..row0+3 of column `col`.

// Stores of an accumulator quad (4 consecutive rows of one column per lane; consecutive lanes hold
// consecutive columns).  The kernels are store-ISSUE bound in their epilogues when every value is its
// own 2-byte store (65536 x 512 bf16 out + transposed copy: 256 store instructions per lane and tile),
// so: the transposed copy of a quad is ONE 8-byte store (its 4 rows are contiguous there), and the
// row-major bf16 copy is paired with the neighbouring lane's column — the even lane stores rows 0 and
// 2, the odd lane rows 1 and 3, each as one 4-byte (col, col+1) pair.  All lanes of the wave must call
// these (the pair exchange is a cross-lane move); out-of-range lanes pass live = false.
template <typename T>
__device__ __forceinline__ void store_quad_transposed(T* yt, long ldyt, int row0, int col, int M, bool live,
                                                      const float (&o)[4]) {
  if (!yt || !live) return;
  T* p = yt + (long)col * ldyt + row0;
  if (sizeof(T) == 2 && row0 + 3 < M && ((((uintptr_t)p) & 7) == 0)) {
    uint2 w;
    w.x = pack_bf16x2(o[0], o[1]);
    w.y = pack_bf16x2(o[2], o[3]);
    *(uint2*)p = w;
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (row0 + e < M) p[e] = cvt_out<T>(o[e]);
}

template <typename T>
__device__ __forceinline__ void store_quad_rowmajor(T* y, long ldy, int row0, int col, int M, int N, bool live,
                                                    const float (&o)[4]) {
  if (sizeof(T) == 2) {
    // every lane exchanges (also the ones with nothing to store): packed pairs of rows (0,1) and (2,3)
    const unsigned mine01 = pack_bf16x2(o[0], o[1]), mine23 = pack_bf16x2(o[2], o[3]);
    const unsigned other01 = swap_adjacent_lanes(mine01), other23 = swap_adjacent_lanes(mine23);
    if (!y) return;
    const int odd = col & 1;  // == lane & 1: tile columns start at multiples of 32
    const bool pair_ok = (col | 1) < N && ((ldy & 1) == 0) && ((((uintptr_t)y) & 3) == 0);
    if (pair_ok) {
      if (!live) return;
      // even lane: rows 0, 2 = low halves; odd lane: rows 1, 3 = high halves; {hi = odd column, lo = even column}
      const unsigned sel = odd ? 0x07060302u : 0x05040100u;
      const unsigned lo01 = odd ? other01 : mine01, hi01 = odd ? mine01 : other01;
      const unsigned lo23 = odd ? other23 : mine23, hi23 = odd ? mine23 : other23;
      const int r_a = row0 + odd, r_b = row0 + 2 + odd;
      if (r_a < M) *(unsigned*)(y + (long)r_a * ldy + (col & ~1)) = perm_bytes(hi01, lo01, sel);
      if (r_b < M) *(unsigned*)(y + (long)r_b * ldy + (col & ~1)) = perm_bytes(hi23, lo23, sel);
      return;
    }
  }
  if (!y || !live) return;
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (row0 + e < M) y[(long)(row0 + e) * ldy + col] = cvt_out<T>(o[e]);
}

template <typename T>
struct EpiForward {  // y = act(acc + bias); row-major and/or transposed stores
  const float* bias;
  T* y;        // row-major [M, ldy] in the compute type (nullable)
  float* y32;  // row-major fp32 (nullable) — used by the last layer (Q-values / head inputs)
  long ldy;
  T* yt;  // transposed [N, ldyt] (nullable) — saved for wgrad + activation-derivative mask
  long ldyt;
  int act, M, N;
  __device__ __forceinline__ void operator()(int row0, int col, const float (&v)[4]) const {
    const bool live = col < N && row0 < M;
    const float b = (bias && live) ? bias[col] : 0.f;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] + b;
    act_apply_n(o, act);
    if (y32 && live) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row0 + e < M) y32[(long)(row0 + e) * ldy + col] = o[e];
    }
    store_quad_rowmajor<T>(y, ldy, row0, col, M, N, live, o);
    store_quad_transposed<T>(yt, ldyt, row0, col, M, live, o);
  }
  // swapped accumulators: 4 consecutive columns col0..col0+3 of output row `row` — the fp32 copy is ONE
  // 16-byte store and the bf16 copy one 8-byte store per quad (against 4 and 2 the other way round);
  // the transposed copy is the expensive one here, so the caller swaps only when there is none
  __device__ __forceinline__ void cols(int row, int col0, const float (&v)[4]) const {
    if (row >= M || col0 >= N) return;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = v[e] + ((bias && col0 + e < N) ? bias[col0 + e] : 0.f);
    act_apply_n(o, act);
    const bool full = col0 + 3 < N;
    if (y32) {
      float* p = y32 + (long)row * ldy + col0;
      if (full && ((((uintptr_t)p) & 15) == 0)) {
        *(f32x4*)p = f32x4{o[0], o[1], o[2], o[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col0 + e < N) p[e] = o[e];
      }
    }
    if (y) {
      T* p = y + (long)row * ldy + col0;
      if (sizeof(T) == 2 && full && ((((uintptr_t)p) & 7) == 0)) {
        uint2 w;
        w.x = pack_bf16x2(o[0], o[1]);
        w.y = pack_bf16x2(o[2], o[3]);
        *(uint2*)p = w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (col0 + e < N) p[e] = cvt_out<T>(o[e]);
      }
    }
    if (yt) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col0 + e < N) yt[(long)(col0 + e) * ldyt + row] = cvt_out<T>(o[e]);
    }
  }
};

template <typename T>
struct EpiDgrad {  // dz_prev = acc * act'(h_prev); row-major and/or transposed stores
  const T* ht;     // transposed saved activation of the layer below [N, ldht] (nullable = linear)
  long ldht;
  T* dx;
  float* dx32;
  long lddx;
  T* dxt;
  long lddxt;
  int act, M, N;
  __device__ __forceinline__ void operator()(int row0, int col, const float (&v)[4]) const {
    const bool live = col < N && row0 < M;
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float g = 1.f;
      if (ht && live && row0 + e < M) g = act_grad_from_output(cvt_in(ht[(long)col * ldht + row0 + e]), act);
      o[e] = v[e] * g;
    }
    if (dx32 && live) {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (row0 + e < M) dx32[(long)(row0 + e) * lddx + col] = o[e];
    }
    store_quad_rowmajor<T>(dx, lddx, row0, col, M, N, live, o);
    store_quad_transposed<T>(dxt, lddxt, row0, col, M, live, o);
  }
};

struct EpiWgrad {  // fp32 split partials; optional transposed placement
  float* p;          // [splits][rows*ld] partial slabs
  long ld;           // leading dim of the (possibly transposed) destination
  long slab;         // elements per split slab
  int transposed;    // 0: p[row][col]   1: p[col][row]
  int M, N;
  int split;         // filled in by the kernel
  __device__ __forceinline__ void operator()(int row0, int col, const float (&v)[4]) const {
    if (col >= N || row0 >= M) return;
    float* base = p + (long)split * slab;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (row0 + e < M) {
        if (transposed) base[(long)col * ld + row0 + e] = v[e];
        else base[(long)(row0 + e) * ld + col] = v[e];
      }
    }
  }
};

template <class E> __device__ __forceinline__ void epi_set_split(E&, int) {}
__device__ __forceinline__ void epi_set_split(EpiWgrad& e, int split) { e.split = split; }

// ---------------------------------------------------------------------------------------------
// the kernel.  BIAS_MODE 0: none; 1: also emit rowsum(A) (bias grad when A = dZ^T);
// 2: also emit rowsum(B) (bias grad when B = dZ^T, the transposed narrow variant).
template <class P, class C, class Epi, int BIAS_MODE, int SWAP = 0>
__global__ void RG_LAUNCH_BOUNDS(256, 1)
    gemm_nt_kernel(GemmArgs g, Epi epi, float* bias_partials /*[splits][rows]*/, long bias_slab) {
  static_assert(SWAP == 0 || BIAS_MODE == 0, "bias row sums assume the unswapped accumulators");
  typedef typename P::T T;
  __shared__ __attribute__((aligned(16))) T As[P::template lds_elems<C::BM>()];
  __shared__ __attribute__((aligned(16))) T Bs[P::template lds_elems<C::BN>()];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;

  const int tiles_m = (g.M + C::BM - 1) / C::BM, tiles_n = (g.N + C::BN - 1) / C::BN;
  int split, tile, tile_m, tile_n;
  if (g.splits > 1) {
    decode_wg((int)blockIdx.x, g.splits, tiles_m * tiles_n, split, tile);
    tile_m = tile / tiles_n;
    tile_n = tile % tiles_n;
  } else {
    split = 0;
    decode_wg((int)blockIdx.x, tiles_m, tiles_n, tile_m, tile_n);
  }
  const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
  const int k_begin = split * g.k_per_split;
  const int k_end = (k_begin + g.k_per_split < g.K) ? k_begin + g.k_per_split : g.K;

  const T* A = (const T*)g.A;
  const T* B = (const T*)g.B;
  const bool a_vec = ((g.lda % P::VEC) == 0) && ((((uintptr_t)A) & 15) == 0);
  const bool b_vec = ((g.ldb % P::VEC) == 0) && ((((uintptr_t)B) & 15) == 0);

  f32x16 acc[C::TM][C::TN];
  f32x16 accb[(C::TM > C::TN ? C::TM : C::TN)];
#pragma unroll
  for (int i = 0; i < C::TM; ++i)
#pragma unroll
    for (int j = 0; j < C::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < (C::TM > C::TN ? C::TM : C::TN); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;

  const bool do_bias = (BIAS_MODE == 1) ? (tile_n == 0 && wn == 0)
                                        : (BIAS_MODE == 2 ? (tile_m == 0 && wm == 0) : false);

  typename P::template Stage<C::BM, C::THREADS> sa;
  typename P::template Stage<C::BN, C::THREADS> sb;

  if (k_begin < k_end) {
    P::template load_global<C::BM, C::THREADS>(sa, A, g.lda, m0, g.M, k_begin, k_end, a_vec, tid);
    P::template load_global<C::BN, C::THREADS>(sb, B, g.ldb, n0, g.N, k_begin, k_end, b_vec, tid);
    P::template store_lds<C::BM, C::THREADS>(sa, As, tid);
    P::template store_lds<C::BN, C::THREADS>(sb, Bs, tid);
    __syncthreads();
    for (int k0 = k_begin; k0 < k_end; k0 += P::BK) {
      const bool more = (k0 + P::BK) < k_end;
      if (more) {
        P::template load_global<C::BM, C::THREADS>(sa, A, g.lda, m0, g.M, k0 + P::BK, k_end, a_vec, tid);
        P::template load_global<C::BN, C::THREADS>(sb, B, g.ldb, n0, g.N, k0 + P::BK, k_end, b_vec, tid);
      }
      P::template compute<C, BIAS_MODE, SWAP>(As, Bs, wm * (C::TM * 32), wn * (C::TN * 32), lane, acc, accb, do_bias);
      __syncthreads();
      if (more) {
        P::template store_lds<C::BM, C::THREADS>(sa, As, tid);
        P::template store_lds<C::BN, C::THREADS>(sb, Bs, tid);
        __syncthreads();
      }
    }
  }

  epi_set_split(epi, split);
  const int lr = lane & 31, lh = lane >> 5;
  // compile-time indices (see gemm_nt_big_kernel): a `#pragma unroll` that gives up on a heavy epilogue
  // body leaves the accumulators indexed at run time, i.e. in scratch
  static_for<0, C::TM>([&](auto tm_c) __attribute__((always_inline)) {
    static_for<0, C::TN>([&](auto tn_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value, tn = decltype(tn_c)::value;
      const int col = n0 + wn * (C::TN * 32) + tn * 32 + lr;
      static_for<0, 4>([&](auto rq_c) __attribute__((always_inline)) {
        constexpr int rq = decltype(rq_c)::value;
        const float v[4] = {acc[tm][tn][rq * 4 + 0], acc[tm][tn][rq * 4 + 1], acc[tm][tn][rq * 4 + 2],
                            acc[tm][tn][rq * 4 + 3]};
        if constexpr (SWAP) {  // transposed accumulator tile: lane = output row, quad = 4 consecutive columns
          epi.cols(m0 + wm * (C::TM * 32) + tm * 32 + lr, n0 + wn * (C::TN * 32) + tn * 32 + 8 * rq + 4 * lh, v);
        } else {
          epi(m0 + wm * (C::TM * 32) + tm * 32 + 8 * rq + 4 * lh, col, v);
        }
      });
    });
  });

  if (BIAS_MODE == 1 && do_bias && lr == 0) {  // D[i][0] = sum_k A[i][k]
    float* bp = bias_partials + (long)split * bias_slab;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (C::TM * 32) + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (row < g.M) bp[row] = accb[tm][r];
      }
  }
  if (BIAS_MODE == 2 && do_bias && lh == 0) {  // D[0][j] = sum_k B[k][j]   (reg 0 of lanes 0..31)
    float* bp = bias_partials + (long)split * bias_slab;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
      const int col = n0 + wn * (C::TN * 32) + tn * 32 + lr;
      if (col < g.N) bp[col] = accb[tn][0];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The large-shape bf16 kernel: 256 x 256 workgroup tile, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA
// tiles), K in blocks of 32.  The 128 x 128 kernel above moves 32 KB through LDS per 2.1 MFLOP and has
// to cross two barriers per K-slab; at 65536 x 3200 x 512 it reaches a tenth of the MFMA peak.  Here:
//   * operands go HBM/L2 -> LDS by DMA (global_load_lds_dwordx4, no VGPR round trip) through a ring of
//     four 32 KB stages, three in flight, one s_waitcnt vmcnt(8) + one barrier per K block;
//   * a DMA instruction fills 1 KB of LDS linearly (lane j -> byte 16 j), so the image is unpadded,
//     64 B per row; bank conflicts are avoided by WHAT each lane fetches: LDS slot (row, s) holds the
//     16-byte K chunk s ^ ((row >> 1) & 3), and a fragment read of 8 consecutive rows then covers all
//     eight 16-byte positions of a 128-byte bank line;
//   * 6 fragment reads (6 KB) feed 8 MFMAs per wave and K step: 96 B/clk of LDS reads per CU at MFMA
//     peak, under the 128 B/clk limit (the 64 x 64 wave tile of the small kernel needs 128).
// Rows past M / N are clamped for the loads and masked by the epilogue; K must be a multiple of 32 and
// the operands 16-byte aligned with leading dimensions a multiple of 8 (the caller checks).  Every
// accumulator sees its K chunks of 16 in ascending order, exactly as in gemm_nt_kernel: the two
// kernels give bit-identical results.
constexpr int BIG_BM = 256, BIG_BN = 256, BIG_BK = 32, BIG_THREADS = 512;
constexpr int BIG_STAGE_BYTES = (BIG_BM + BIG_BN) * BIG_BK * 2, BIG_SLOTS = 4, BIG_DMA_PER_THREAD = 4;
static_assert(BIG_STAGE_BYTES == BIG_DMA_PER_THREAD * BIG_THREADS * 16, "one stage = 4 DMA requests per thread");

template <class Epi>
__global__ void RG_LAUNCH_BOUNDS(BIG_THREADS, 1) gemm_nt_big_kernel(GemmArgs g, Epi epi) {
  RG_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tiles_m = (g.M + BIG_BM - 1) / BIG_BM, tiles_n = (g.N + BIG_BN - 1) / BIG_BN;
  int tile_m, tile_n;
  decode_wg((int)blockIdx.x, tiles_m, tiles_n, tile_m, tile_n);
  const int m0 = tile_m * BIG_BM, n0 = tile_n * BIG_BN;
  const bf16_t* A = (const bf16_t*)g.A;
  const bf16_t* B = (const bf16_t*)g.B;
  const int n_blk = g.K / BIG_BK;

  // this thread's four DMA sources of a stage (requests 0,1: A rows; 2,3: B rows), as row pointers
  const bf16_t* src[BIG_DMA_PER_THREAD];
#pragma unroll
  for (int i = 0; i < BIG_DMA_PER_THREAD; ++i) {
    const int v = (tid + i * BIG_THREADS) & 1023;  // slot index inside the operand's 16 KB image
    const int row = v >> 2, chunk = (v & 3) ^ ((row >> 1) & 3);
    if (i < 2) {
      const int gr = m0 + row < g.M ? m0 + row : g.M - 1;
      src[i] = A + (long)gr * g.lda + chunk * 8;
    } else {
      const int gr = n0 + row < g.N ? n0 + row : g.N - 1;
      src[i] = B + (long)gr * g.ldb + chunk * 8;
    }
  }
  auto issue = [&](int blk, int slot) {
    const int kb = (blk < n_blk ? blk : n_blk - 1) * BIG_BK;  // past the end: refetch the last block (keeps vmcnt exact)
    static_for<0, BIG_DMA_PER_THREAD>([&](auto i_c) __attribute__((always_inline)) {
      constexpr int i = decltype(i_c)::value;
      global_load_lds_b128(src[i] + kb, smem + slot * BIG_STAGE_BYTES + (wave * 64 + i * BIG_THREADS) * 16);
    });
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int tm = 0; tm < 4; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

  const int lr = lane & 31, lg = lane >> 5, sw = (lr >> 1) & 3;
  auto compute = [&](int slot) {
    const char* As = smem + slot * BIG_STAGE_BYTES;
    const char* Bs = As + BIG_BM * BIG_BK * 2;
#pragma unroll
    for (int ks = 0; ks < BIG_BK / 16; ++ks) {
      const int off = ((ks * 2 + lg) ^ sw) * 16;
      u16x8 af[4], bf[2];
#pragma unroll
      for (int tm = 0; tm < 4; ++tm) af[tm] = *(const u16x8*)(As + (wm * 128 + tm * 32 + lr) * 64 + off);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) bf[tn] = *(const u16x8*)(Bs + (wn * 64 + tn * 32 + lr) * 64 + off);
#pragma unroll
      for (int tm = 0; tm < 4; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(af[tm], bf[tn], acc[tm][tn]);
    }
  };

  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  for (int t = 0; t < n_blk; ++t) {
    RG_WAIT_VMCNT(8);  // 12 requests of this thread in flight; the oldest 4 (block t) have landed
    raw_barrier();     // ... for every thread; and block t-1 is consumed, so its slot can be refilled
    issue(t + 3, (t + 3) & 3);
    compute(t & 3);
  }
  RG_WAIT_VMCNT(0);

  // compile-time indices: with an activation switch in the epilogue body `#pragma unroll` gives up
  // and the accumulators would be indexed at run time, i.e. live in scratch
  static_for<0, 4>([&](auto tm_c) __attribute__((always_inline)) {
    static_for<0, 2>([&](auto tn_c) __attribute__((always_inline)) {
      constexpr int tm = decltype(tm_c)::value, tn = decltype(tn_c)::value;
      const int col = n0 + wn * 64 + tn * 32 + lr;
      static_for<0, 4>([&](auto rq_c) __attribute__((always_inline)) {
        constexpr int rq = decltype(rq_c)::value;
        const int row0 = m0 + wm * 128 + tm * 32 + 8 * rq + 4 * lg;
        const float v[4] = {acc[tm][tn][rq * 4 + 0], acc[tm][tn][rq * 4 + 1], acc[tm][tn][rq * 4 + 2],
                            acc[tm][tn][rq * 4 + 3]};
        epi(row0, col, v);
      });
    });
  });
}

// Shapes the big kernel takes: enough tiles to fill the chip, K in whole blocks, DMA-able operands —
// and a LONG reduction.  Measured on MI355X (profiles/microbench/gemm_shapes.py, M = 65536, bf16 out):
// N=512 K=3200 297 us = 0.72 PFLOP/s, but N=512 K=512 127 us and N=3200 K=512 723 us = 0.27-0.30
// PFLOP/s: with 16 K blocks per tile the launch is bound by the epilogue's store issue (one workgroup
// per CU, nothing to overlap it with), and inside the QR-DQN step those two shapes ran SLOWER than on
// the 128 x 128 kernel (0.18 vs 0.14 ms, 0.98 vs 0.89 ms), whose 3-4 resident workgroups overlap each
// other's epilogues.  Until the epilogue goes through LDS (16-byte row stores) the kernel is used for
// K >= 1024 only.  (Its accumulators swapped like gemm_nt_kernel's SWAP mode — 16-byte row stores — were
// measured too: 203 vs 123 us at N=512 K=512 and 1180 vs 730 us at N=3200 K=512 against the swapped
// 128 x 128 kernel; with one workgroup per CU a row-scattered epilogue has nothing to hide behind.)
static inline bool gemm_big_ok(const GemmArgs& g) {
  return g.M >= 2048 && g.N >= 192 && g.K >= 1024 && (g.K % BIG_BK) == 0 && (g.lda % 8) == 0 && (g.ldb % 8) == 0 &&
         ((((uintptr_t)g.A) | ((uintptr_t)g.B)) & 15) == 0 && g.splits <= 1;
}

}  // namespace rg

// Prototype / calibration (not product code): the fused FullyConnected forward in TRANSPOSED, register-chained form.
//
//   Y^T[N, rows] = W[N, K] . X^T[K, rows]:  the WEIGHTS are the MFMA A operand, a wave's 32 batch rows the B operand.
//   The D fragment of v_mfma_f32_32x32x16_bf16 (lane <-> batch row, registers <-> output features) is, after
//   bias + activation + bf16 packing, bit for bit the B fragment of the NEXT layer (up to a fixed permutation of the
//   features inside each 16-chunk, which the weight staging absorbs).  So activations never leave the registers: no
//   LDS activation tile, no transposing epilogue, no barrier between layers.  The LDS is a ring of weight stages
//   filled by LDS-DMA (global_load_lds_dwordx4) and shared by the workgroup's 4 waves (one per SIMD, 512 registers).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc chain_fwd.hip -o chain_fwd
// Run:   ./chain_fwd [batch=65536] [check_rows=256]
#include "rg_platform.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace rg;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#ifndef CH_KK
#define CH_KK 4  // K chunks (of 16) per weight stage
#endif
#ifndef CH_NSLOT
#define CH_NSLOT 4
#endif
#ifndef CH_PF
#define CH_PF 8  // A fragments requested ahead of the MFMA that consumes them (must divide the stage's fragment count)
#endif
constexpr int NW = 4;                    // waves per workgroup (one per SIMD)
constexpr int FT = 8;                    // feature tiles (of 32) per accumulator chunk
constexpr int KK = CH_KK;
constexpr int SF = FT * KK;              // fragments (1 KB each) per stage
constexpr int STAGE_BYTES = SF * 1024;
constexpr int NSLOT = CH_NSLOT;
constexpr int PF = CH_PF;
constexpr int H = 512, K0 = 128, NOUT = 16;
constexpr int KC0 = K0 / 16, KCH = H / 16;
constexpr int BIAS_FLOATS = 3 * H + 32;

struct Args {
  const bf16_t* wstream;  // all stages of the network, consumption order
  int n_stages;
  const float* bias;      // [3*H + 32]
  const bf16_t* x;        // [batch, K0] bf16 row-major
  float* out;             // [batch, NOUT]
  int batch, n_hidden;    // hidden layers after the first (K = H)
};

// feature held by accumulator register r of lane group lg inside a 32-feature tile (MFMA 32x32 D layout)
__host__ __device__ inline int d_feature(int r, int lg) { return (r & 3) + 8 * (r >> 2) + 4 * lg; }

// at most `stages` whole stages of this wave's LDS-DMA requests still outstanding
template <int STAGES> __device__ __forceinline__ void wait_stages() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(STAGES * (SF / NW)) : "memory");
}

struct Ring {
  char* lds;             // ring base
  const char* src;       // weight stream base
  unsigned src_bytes;    // n_stages * STAGE_BYTES
  unsigned issue_off;    // byte offset of the next stage to request
  int issue_slot;        // its slot
  int cur_slot;          // slot of the stage being consumed
  int wave, lane;
};

__device__ __forceinline__ void ring_issue(Ring& r) {
#ifdef CH_NO_DMA  // ablation: the ring is never refilled after the prologue (MFMAs on stale weights)
  if (r.issue_off >= (unsigned)NSLOT * STAGE_BYTES) {
    r.issue_off += STAGE_BYTES;
    r.issue_slot = r.issue_slot + 1 == NSLOT ? 0 : r.issue_slot + 1;
    return;
  }
#endif
  // this wave's share of the next stage: SF / NW fragments
  const char* s = r.src + r.issue_off + (r.wave * (SF / NW)) * 1024 + r.lane * 16;
  char* d = r.lds + r.issue_slot * STAGE_BYTES + (r.wave * (SF / NW)) * 1024;
#pragma unroll
  for (int i = 0; i < SF / NW; ++i) global_load_lds_b128(s + i * 1024, d + i * 1024);
  r.issue_off += STAGE_BYTES;
  if (r.issue_off >= r.src_bytes) r.issue_off = 0;
  r.issue_slot = r.issue_slot + 1 == NSLOT ? 0 : r.issue_slot + 1;
}

// Stage hand-over.  On return: stage `cur` + 1 has landed for every wave (so reads may run ahead into it), the slot
// of stage `cur` - 1 is being refilled with stage cur + NSLOT - 1.
__device__ __forceinline__ void ring_advance(Ring& r) {
  // requested so far: stages .. cur+NSLOT-1; stage cur+2 (counting from the stage being left) must have landed:
  // only the NSLOT-3 younger stages may be outstanding
  wait_stages<NSLOT - 3>();
  raw_barrier();
  ring_issue(r);
  r.cur_slot = r.cur_slot + 1 == NSLOT ? 0 : r.cur_slot + 1;
}

__device__ __forceinline__ u16x8 lds_frag(const char* p) { return *(const u16x8*)p; }

// One stage of a wide layer: acc[ft] += A[kk][ft] . X[kk] for kk < KK, ft < FT.  `a` carries the first PF fragments of
// this stage on entry and the first PF fragments of the next stage on exit.
template <typename XT>
__device__ __forceinline__ void stage_wide(Ring& r, f32x16 (&acc)[FT], const XT& xk, u16x8 (&a)[PF]) {
  const char* cur = r.lds + r.cur_slot * STAGE_BYTES + r.lane * 16;
  const int ns = r.cur_slot + 1 == NSLOT ? 0 : r.cur_slot + 1;
  const char* nxt = r.lds + ns * STAGE_BYTES + r.lane * 16;
#pragma unroll
  for (int i = 0; i < SF; ++i) {
    const int kk = i / FT, ft = i % FT;
    const u16x8 af = a[i % PF];
#ifndef CH_NO_LDS  // ablation: without it the A fragments are never re-read (registers-only MFMA loop)
    a[i % PF] = (i + PF < SF) ? lds_frag(cur + (i + PF) * 1024) : lds_frag(nxt + (i + PF - SF) * 1024);
#endif
    acc[ft] = mfma_32x32x16_bf16(af, xk(kk), acc[ft]);
  }
}

template <int NCH>
struct XRegs {
  u32x4 v[NCH];  // B fragment of chunk c = 4 dwords
};

__device__ __forceinline__ u16x8 as_frag(u32x4 v) { return __builtin_bit_cast(u16x8, v); }

// bias-initialised accumulators of feature chunk fc of a layer whose bias starts at `b` (LDS)
__device__ __forceinline__ void acc_init(f32x16 (&acc)[FT], const float* b, int fc, int lg) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = *(const f32x4*)(b + (fc * FT + ft) * 32 + 8 * q + 4 * lg);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[ft][4 * q + e] = v[e];
    }
}

// ReLU + bf16 packing: tile ft becomes chunks 2*ft and 2*ft+1 of the next layer's B operand
__device__ __forceinline__ void relu_pack(const f32x16 (&acc)[FT], u32x4* dst) {
#pragma unroll
  for (int ft = 0; ft < FT; ++ft)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4 p;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        p[q] = pack_bf16x2(fmaxf(acc[ft][8 * h + 2 * q], 0.f), fmaxf(acc[ft][8 * h + 2 * q + 1], 0.f));
      dst[2 * ft + h] = p;
    }
}

__global__ void __launch_bounds__(NW * 64, 1) chain_forward(Args g) {
  RG_DYN_LDS(smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = wave_uniform(tid >> 6);
  const int lr = lane & 31, lg = lane >> 5;
  float* bias_lds = (float*)(smem + NSLOT * STAGE_BYTES);
  for (int i = tid; i < BIAS_FLOATS; i += NW * 64) bias_lds[i] = g.bias[i];
  Ring r;
  r.lds = smem;
  r.src = (const char*)g.wstream;
  r.src_bytes = (unsigned)g.n_stages * STAGE_BYTES;
  r.issue_off = 0;
  r.issue_slot = 0;
  r.cur_slot = 0;
  r.wave = wave;
  r.lane = lane;
  // prologue: every slot requested (stages 0 .. NSLOT-1), stages 0 and 1 landed
#pragma unroll
  for (int s = 0; s < NSLOT; ++s) ring_issue(r);
  wait_stages<NSLOT - 2>();
  __syncthreads();
  u16x8 a[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) a[i] = lds_frag(smem + lane * 16 + i * 1024);

  const int n_blocks = (g.batch + NW * 32 - 1) / (NW * 32);
  bool first = true;
  for (int blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
    const int row = blk * (NW * 32) + wave * 32 + lr;
    const int rowc = row < g.batch ? row : g.batch - 1;
    // ---- input rows -> B fragments (natural feature order: chunk kc, lane group lg holds features 16kc + 8lg ..+8)
    XRegs<KC0> x0;
    const bf16_t* xr = g.x + (long)rowc * K0 + lg * 8;
#pragma unroll
    for (int kc = 0; kc < KC0; ++kc) x0.v[kc] = *(const u32x4*)(xr + kc * 16);
    XRegs<KCH> xa, xb;
    f32x16 acc[FT];
    // ---- layer 0: K0 -> H
#pragma unroll
    for (int fc = 0; fc < H / (32 * FT); ++fc) {
      acc_init(acc, bias_lds, fc, lg);
#pragma unroll
      for (int kg = 0; kg < KC0 / KK; ++kg) {
        if (fc == 0 && kg == 0) {
          if (!first) ring_advance(r);
          first = false;
        } else {
          ring_advance(r);
        }
        stage_wide(r, acc, [&](int kk) { return as_frag(x0.v[kg * KK + kk]); }, a);
      }
      relu_pack(acc, &xa.v[fc * 2 * FT]);
    }
    // ---- hidden layers: H -> H
    for (int l = 0; l < g.n_hidden; ++l) {
#pragma unroll
      for (int fc = 0; fc < H / (32 * FT); ++fc) {
        acc_init(acc, bias_lds + (l + 1) * H, fc, lg);
#pragma unroll
        for (int kg = 0; kg < KCH / KK; ++kg) {
          ring_advance(r);
          stage_wide(r, acc, [&](int kk) { return as_frag(xa.v[kg * KK + kk]); }, a);
        }
        relu_pack(acc, &xb.v[fc * 2 * FT]);
      }
#pragma unroll
      for (int c = 0; c < KCH; ++c) xa.v[c] = xb.v[c];
    }
    // ---- output layer: H -> NOUT (one 32-feature tile, zero padded), K split over 4 accumulators
    {
      f32x16 o[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[p][e] = 0.f;
      const float* bo = bias_lds + (g.n_hidden + 1) * H;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *(const f32x4*)(bo + 8 * q + 4 * lg);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[0][4 * q + e] = v[e];
      }
#pragma unroll
      for (int sg = 0; sg < KCH / SF + (KCH % SF ? 1 : 0); ++sg) {
        ring_advance(r);
        const char* cur = r.lds + r.cur_slot * STAGE_BYTES + r.lane * 16;
        const int ns = r.cur_slot + 1 == NSLOT ? 0 : r.cur_slot + 1;
        const char* nxt = r.lds + ns * STAGE_BYTES + r.lane * 16;
#pragma unroll
        for (int i = 0; i < SF; ++i) {
          const int kc = sg * SF + i;
          const u16x8 af = a[i % PF];
          a[i % PF] = (i + PF < SF) ? lds_frag(cur + (i + PF) * 1024) : lds_frag(nxt + (i + PF - SF) * 1024);
          if (kc < KCH) o[i % 4] = mfma_32x32x16_bf16(af, as_frag(xa.v[kc]), o[i % 4]);
        }
      }
      if (row < g.batch) {
        float* orow = g.out + (long)row * NOUT;
#pragma unroll
        for (int q = 0; q < NOUT / 8; ++q) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = o[0][4 * q + e] + o[1][4 * q + e] + o[2][4 * q + e] + o[3][4 * q + e];
          *(f32x4*)(orow + 8 * q + 4 * lg) = v;
        }
      }
    }
  }
  RG_WAIT_VMCNT(0);
}

// ---------------------------------------------------------------- host side
static unsigned short f2bf(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float bf2f(unsigned short v) {
  unsigned u = (unsigned)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// A fragment of W[N, K] (row-major fp32) for feature tile nt, chunk kc; perm = the hidden-layer K permutation
static void put_frag(std::vector<unsigned short>& dst, const std::vector<float>& w, int N, int K, int nt, int kc, bool perm) {
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 8; ++e) {
      const int n = nt * 32 + (lane & 31), lgp = lane >> 5;
      const int k = kc * 16 + (perm ? (e & 3) + 8 * (e >> 2) + 4 * lgp : lgp * 8 + e);
      dst.push_back((n < N && k < K) ? f2bf(w[(size_t)n * K + k]) : 0);
    }
}

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 65536;
  const int check = argc > 2 ? atoi(argv[2]) : 256;
  const int n_hidden = 2;
  const int dims[5] = {K0, H, H, H, NOUT};
  srand(1);
  std::vector<std::vector<float>> W(4), Bv(4);
  for (int l = 0; l < 4; ++l) {
    W[l].resize((size_t)dims[l + 1] * dims[l]);
    Bv[l].resize(dims[l + 1]);
    const float s = 1.0f / sqrtf((float)dims[l]);
    for (auto& v : W[l]) v = bf2f(f2bf(((rand() & 0xffff) / 32768.0f - 1.0f) * s * 1.7f));
    for (auto& v : Bv[l]) v = ((rand() & 0xffff) / 32768.0f - 1.0f) * 0.1f;
  }
  // weight stream in consumption order
  std::vector<unsigned short> ws;
  for (int l = 0; l < 3; ++l) {
    const int K = dims[l], KC = K / 16;
    for (int fc = 0; fc < H / (32 * FT); ++fc)
      for (int kg = 0; kg < KC / KK; ++kg)
        for (int kk = 0; kk < KK; ++kk)
          for (int ft = 0; ft < FT; ++ft) put_frag(ws, W[l], H, K, fc * FT + ft, kg * KK + kk, l > 0);
  }
  {
    const int nsg = (KCH + SF - 1) / SF;
    for (int sg = 0; sg < nsg; ++sg)
      for (int i = 0; i < SF; ++i) {
        const int kc = sg * SF + i;
        if (kc < KCH) put_frag(ws, W[3], NOUT, H, 0, kc, true);
        else ws.insert(ws.end(), 512, 0);
      }
  }
  const int n_stages = (int)(ws.size() * 2 / STAGE_BYTES);
  std::vector<float> bias(BIAS_FLOATS, 0.f);
  for (int l = 0; l < 3; ++l) memcpy(&bias[l * H], Bv[l].data(), H * 4);
  memcpy(&bias[3 * H], Bv[3].data(), NOUT * 4);
  std::vector<unsigned short> x((size_t)batch * K0);
  for (auto& v : x) v = f2bf(((rand() & 0xffff) / 32768.0f - 1.0f) * 1.5f);

  bf16_t *d_ws, *d_x;
  float *d_bias, *d_out;
  hipMalloc(&d_ws, ws.size() * 2);
  hipMalloc(&d_x, x.size() * 2);
  hipMalloc(&d_bias, bias.size() * 4);
  hipMalloc(&d_out, (size_t)batch * NOUT * 4);
  hipMemcpy(d_ws, ws.data(), ws.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_x, x.data(), x.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_bias, bias.data(), bias.size() * 4, hipMemcpyHostToDevice);
  hipMemset(d_out, 0, (size_t)batch * NOUT * 4);
  Args g{d_ws, n_stages, d_bias, d_x, d_out, batch, n_hidden};
  const int lds = NSLOT * STAGE_BYTES + BIAS_FLOATS * 4;
  hipFuncSetAttribute((const void*)chain_forward, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int n_blocks = (batch + NW * 32 - 1) / (NW * 32);
  const int grid = n_blocks < 256 ? n_blocks : 256;
  chain_forward<<<grid, NW * 64, lds>>>(g);
  hipError_t err = hipDeviceSynchronize();
  printf("chain_fwd KK=%d NSLOT=%d PF=%d: %d stages (%.2f MB weight stream), grid %d, lds %d, launch: %s\n", KK, NSLOT, PF, n_stages,
         ws.size() * 2 / 1e6, grid, lds, hipGetErrorString(err));
  // ---- check against a CPU evaluation with bf16 rounding at the same points
  std::vector<float> out((size_t)batch * NOUT);
  hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost);
  double max_err = 0;
  for (int t = 0; t < check; ++t) {
    const int row = (int)(((long)t * 2654435761u) % batch);
    std::vector<float> h(K0), hn;
    for (int k = 0; k < K0; ++k) h[k] = bf2f(x[(size_t)row * K0 + k]);
    for (int l = 0; l < 4; ++l) {
      hn.assign(dims[l + 1], 0.f);
      for (int n = 0; n < dims[l + 1]; ++n) {
        double s = Bv[l][n];
        for (int k = 0; k < dims[l]; ++k) s += (double)W[l][(size_t)n * dims[l] + k] * h[k];
        hn[n] = l < 3 ? bf2f(f2bf(fmaxf((float)s, 0.f))) : (float)s;
      }
      h = hn;
    }
    for (int n = 0; n < NOUT; ++n) max_err = fmax(max_err, fabs(h[n] - out[(size_t)row * NOUT + n]));
  }
  printf("max |out - cpu| over %d rows: %.3e\n", check, max_err);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    const int n = 20;
    for (int i = 0; i < n; ++i) chain_forward<<<grid, NW * 64, lds>>>(g);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / n;
    const double flop = 2.0 * batch * ((double)K0 * H + 2.0 * H * H + (double)H * NOUT);
    printf("forward B=%d: %.1f us  %.1f TFLOP/s (%.3f of 2500)\n", batch, us, flop / us * 1e-6, flop / us * 1e-6 / 2500.0);
  }
  return 0;
}

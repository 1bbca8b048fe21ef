// Calibration microbenchmark (round 6, not product code): the fused forward's MAIN LOOP with its weight operand staged through an
// LDS-DMA ring instead of the register ring — the experiment VERDICT r5 item 5 named, measured instead of bounded.
//   variant 0: the product's loop — A fragments from the LDS activation tile, B fragments L2 -> VGPR (global_load_dwordx4), ring of 2
//   variant 1: B fragments L2 -> LDS by global_load_lds_dwordx4 into a PER-WAVE ring of NS 1-KB slots (a wave's weight operand is
//              private, so no barrier: its own vmcnt orders the ring), read back with one ds_read_b128 per fragment, one fragment
//              ahead of the MFMAs that use it.  NS = 3 is what the 27 KB behind the product's 128 x 520 tile hold (8 waves x 3 KB);
//              deeper rings run here on a 64-row tile (the same bank pattern, rows aliased) to show the trend.
// Same shapes as mfma_feed.hip: 8 waves, 4 x 2 accumulator tiles per wave, K = 512 x `layers`, 512 KB of weights per layer shared by
// every workgroup, K order rotated per (workgroup, wave).  Every variant sums the same products in the same order: the per-wave
// checksums must agree bit for bit with variant 0's (printed).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc mfma_feed_dma.hip -o mfma_feed_dma
#include "rg_platform.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace rg;

constexpr int PITCH = 520, KC = 32, TM = 4, TN = 2, NW = 8;
constexpr int FRAGS = KC * TN;  // fragments of a wave's layer

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int VARIANT, int NS, int AROWS>
__global__ void __launch_bounds__(NW * 64, 1) feed(const unsigned short* __restrict__ wf, float* out, float* sums, int layers) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* act = (unsigned short*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < AROWS * PITCH; i += NW * 64) act[i] = (unsigned short)(0x3c00 + ((i * 2654435761u) >> 20));
  __syncthreads();
  const int lr = lane & 31, lg = lane >> 5;
  const unsigned short* arow = act + lr * PITCH + lg * 8;
  const long nt_stride = (long)KC * 512;
  const unsigned short* wf_wave = wf + (long)(wave * TN) * nt_stride;
  const int rot = (blockIdx.x * 5 + wave * 11) % KC;
  auto kx = [&](int kc) { const int k = kc + rot; return k >= KC ? k - KC : k; };
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  u16x8 a[2][TM], b[2][TN];
  auto loadA = [&](u16x8 (&af)[TM], int kc) {
    const int off = kx(kc) * 16;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) af[tm] = *(const u16x8*)(arow + ((tm * 32) % AROWS) * PITCH + off);
  };
  if constexpr (VARIANT == 0 || VARIANT == 2) {
    // variant 2: the same loop with the weight fragments requested by raw buffer loads (V# over the weight array, lane * 16 as the
    // vector offset, the chunk as the scalar offset) so that a cache policy can ride on them — NS is the aux field here:
    // 0 default, 2 = nt, 16 = sc1 (agent scope: served by L2, no line allocated in the CU's vector L1), 17 = sc0 sc1, 18 = sc1 nt
    typedef __attribute__((ext_vector_type(4))) int i32x4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wf, 0, (int)((size_t)16 * KC * 512 * 2 * 2), 0x00020000);
    auto loadB = [&](u16x8 (&bf)[TN], int kc) {
      if constexpr (VARIANT == 2) {
        const int soff = ((wave * TN) * (int)nt_stride + kx(kc) * 512) * 2;  // bytes, wave-uniform
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, soff + tn * (int)nt_stride * 2, NS);
          bf[tn] = __builtin_bit_cast(u16x8, v);
        }
      } else {
        const unsigned short* chunk = wf_wave + (long)kx(kc) * 512;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bf[tn] = *(const u16x8*)(chunk + tn * nt_stride + lane * 8);
      }
    };
    for (int l = 0; l < layers; ++l) {
      loadB(b[0], 0);
      loadA(a[0], 0);
#pragma unroll 1
      for (int kc = 0; kc < KC; kc += 2) {  // (not unrolled: the product's K is a run-time value; fully unrolled the compiler parks every
                                            // chunk's addresses in VGPRs and spills — as mfma_feed.hip's modes 2 / 3 / 18 do: 660-692 bytes of scratch per lane)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (kc + s + 1 < KC) {
            loadB(b[(s + 1) & 1], kc + s + 1);
            loadA(a[(s + 1) & 1], kc + s + 1);
          }
          sched_fence();
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(a[s & 1][tm], b[s & 1][tn], acc[tm][tn]);
          sched_fence();
        }
      }
    }
  } else if constexpr (VARIANT == 4 || VARIANT == 5) {
    // variant 4: A from LDS every chunk, the two B fragments loaded once; variant 5: nothing reloaded (MFMAs from registers)
    {
      const unsigned short* chunk = wf_wave + (long)kx(0) * 512;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) b[0][tn] = *(const u16x8*)(chunk + tn * nt_stride + lane * 8);
    }
    loadA(a[0], 0);
    loadA(a[1], 1);
    for (int l = 0; l < layers; ++l) {
#pragma unroll 1
      for (int kc = 0; kc < KC; kc += 2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if (VARIANT == 4 && kc + s + 1 < KC) loadA(a[(s + 1) & 1], kc + s + 1);
          sched_fence();
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(a[s & 1][tm], b[0][tn], acc[tm][tn]);
          sched_fence();
        }
      }
    }
  } else if constexpr (VARIANT == 3) {
    // variant 3: the register ring at depth NS (the product's RING): weight fragments NS - 1 chunks ahead, A one chunk ahead
    u16x8 br[NS][TN];
    auto loadBr = [&](u16x8 (&bf)[TN], int kc) {
      const unsigned short* chunk = wf_wave + (long)kx(kc) * 512;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bf[tn] = *(const u16x8*)(chunk + tn * nt_stride + lane * 8);
    };
    for (int l = 0; l < layers; ++l) {
#pragma unroll
      for (int s = 0; s < NS - 1; ++s) loadBr(br[s], s);
      loadA(a[0], 0);
#pragma unroll 1
      for (int kc = 0; kc < KC; kc += NS) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          if (kc + s + NS - 1 < KC) loadBr(br[(s + NS - 1) % NS], kc + s + NS - 1);
          if (kc + s + 1 < KC) loadA(a[(s + 1) & 1], kc + s + 1);
          sched_fence();
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = mfma_32x32x16_bf16(a[s & 1][tm], br[s][tn], acc[tm][tn]);
          sched_fence();
        }
      }
    }
  } else {
    // fragment f of the whole run: layer f / FRAGS, chunk (f % FRAGS) / TN, column tile f % TN
    char* ring = smem + (size_t)AROWS * PITCH * 2 + (size_t)wave * NS * 1024;
    const int F = layers * FRAGS;
    auto src = [&](int f) {
      const int fl = f % FRAGS;
      return wf_wave + (long)kx(fl / TN) * 512 + (fl % TN) * nt_stride + lane * 8;
    };
    u16x8 bf[2];
    int slot = 0;  // slot of fragment f (wave-uniform), f % NS
#pragma unroll
    for (int f = 0; f < NS; ++f) global_load_lds_b128_cached(src(f), ring + f * 1024);
    wait_vm<NS - 1>();
    bf[0] = *(const u16x8*)(ring + lane * 16);
    loadA(a[0], 0);
    for (int f0 = 0; f0 < F; f0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int f = f0 + j;
        wait_lgkm0();  // fragment f is in registers: its slot is free
        if (f + NS < F) {
          global_load_lds_b128_cached(src(f + NS), ring + slot * 1024);
          wait_vm<NS - 1>();  // fragment f + 1 has landed
        } else {
          wait_vm<0>();
        }
        slot = slot + 1 == NS ? 0 : slot + 1;
        if (f + 1 < F) {
          bf[(j + 1) & 1] = *(const u16x8*)(ring + slot * 1024 + lane * 16);
          if (j & 1) loadA(a[((j + 1) >> 1) & 1], ((f + 1) % FRAGS) / TN);
        }
        sched_fence();
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) acc[tm][j & 1] = mfma_32x32x16_bf16(a[(j >> 1) & 1][tm], bf[j & 1], acc[tm][j & 1]);
        sched_fence();
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) sums[blockIdx.x * NW + wave] = s;
  if (s == 12345.678f) out[0] = s;
}

static std::vector<float> g_ref;
static int g_long = 0;  // argv[2]: repetitions of the timed loop (default 40); telemetry runs use a few thousand

template <int VARIANT, int NS, int AROWS>
void run(const char* what, const unsigned short* wf, float* out, float* sums, int grid) {
  const int lds = AROWS * PITCH * 2 + (VARIANT == 1 ? NW * NS * 1024 : 0);
  const int layers = 12;
  auto k = feed<VARIANT, NS, AROWS>;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipMemset(sums, 0, (size_t)grid * NW * 4);
  k<<<grid, NW * 64, lds>>>(wf, out, sums, layers);
  hipDeviceSynchronize();
  std::vector<float> h((size_t)grid * NW);
  hipMemcpy(h.data(), sums, h.size() * 4, hipMemcpyDeviceToHost);
  const char* check = "reference";
  if (AROWS == 128) {
    if (VARIANT == 0 && g_ref.empty()) g_ref = h;
    if (VARIANT == 0) check = "reference";
    if (VARIANT != 0 && VARIANT < 4 && g_ref.empty()) check = "(no reference in this run)";
    else if (VARIANT != 0 && VARIANT < 4) check = (g_ref.size() == h.size() && memcmp(g_ref.data(), h.data(), h.size() * 4) == 0) ? "checksums == variant 0" : "CHECKSUMS DIFFER";
  } else {
    check = "(64-row tile: no check)";
  }
  for (int r = 0; r < 200; ++r) k<<<grid, NW * 64, lds>>>(wf, out, sums, layers);  // sustained clock state
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  const int reps = g_long > 0 ? g_long : 40;
  for (int r = 0; r < reps; ++r) k<<<grid, NW * 64, lds>>>(wf, out, sums, layers);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps;
  const double flop = (double)grid * NW * layers * KC * TM * TN * 32768.0;
  printf("%-74s lds %6d B: %8.1f us  %7.1f TFLOP/s  (%.3f of 2500)  %s  err=%d\n", what, lds, us, flop / us * 1e-6, flop / us * 1e-6 / 2500.0,
         check, (int)hipGetLastError());
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // -1 = every variant; 0..3 = one of the four below, for clock / power sampling
  g_long = argc > 2 ? atoi(argv[2]) : 0;
  float *out, *sums;
  hipMalloc(&out, 4);
  hipMalloc(&sums, (size_t)512 * NW * 4);
  const size_t n = (size_t)16 * KC * 512 * 2;
  std::vector<unsigned short> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  unsigned short* wf;
  hipMalloc(&wf, n * 2);
  hipMemcpy(wf, h.data(), n * 2, hipMemcpyHostToDevice);
  if (only >= 0) {
    if (only == 0) run<0, 1, 128>("A from LDS + B L2 -> VGPR ring 2 (the product's loop)", wf, out, sums, 512);
    if (only == 1) run<4, 1, 128>("A from LDS only (B fragments stay in registers)", wf, out, sums, 512);
    if (only == 2) run<5, 1, 128>("registers only", wf, out, sums, 512);
    if (only == 3) run<1, 3, 128>("A from LDS + B L2 -> LDS-DMA ring of 3 slots per wave", wf, out, sums, 512);
    return 0;
  }
  run<0, 1, 128>("A from LDS + B L2 -> VGPR ring 2 (the product's loop)", wf, out, sums, 512);
  run<2, 0, 128>("  the same by raw buffer loads, default policy", wf, out, sums, 512);
  run<2, 2, 128>("  buffer loads, nt", wf, out, sums, 512);
  run<2, 16, 128>("  buffer loads, sc1 (L2-served, no L1 allocation)", wf, out, sums, 512);
  run<2, 17, 128>("  buffer loads, sc0 sc1", wf, out, sums, 512);
  run<2, 18, 128>("  buffer loads, sc1 nt", wf, out, sums, 512);
  run<0, 1, 128>("A from LDS + B L2 -> VGPR ring 2 (again)", wf, out, sums, 512);
  run<3, 2, 128>("  register ring, depth 2 (generic form)", wf, out, sums, 512);
  run<3, 4, 128>("  register ring, depth 4", wf, out, sums, 512);
  run<0, 1, 128>("A from LDS + B L2 -> VGPR ring 2 (again)", wf, out, sums, 512);
  run<1, 3, 128>("A from LDS + B L2 -> LDS-DMA ring of 3 slots per wave (fits the product)", wf, out, sums, 512);
  run<0, 1, 128>("A from LDS + B L2 -> VGPR ring 2 (again)", wf, out, sums, 512);
  run<1, 2, 128>("A from LDS + B L2 -> LDS-DMA ring of 2 slots per wave", wf, out, sums, 512);
  run<0, 1, 64>("64-row tile: B L2 -> VGPR ring 2", wf, out, sums, 512);
  run<1, 3, 64>("64-row tile: LDS-DMA ring of 3 slots", wf, out, sums, 512);
  run<1, 4, 64>("64-row tile: LDS-DMA ring of 4 slots", wf, out, sums, 512);
  run<1, 6, 64>("64-row tile: LDS-DMA ring of 6 slots", wf, out, sums, 512);
  run<1, 8, 64>("64-row tile: LDS-DMA ring of 8 slots", wf, out, sums, 512);
  run<1, 3, 128>("A from LDS + B L2 -> LDS-DMA ring of 3 slots per wave (again)", wf, out, sums, 512);
  return 0;
}

// wgrad_model.hip — the weight-gradient kernel's operand stream rebuilt piece by piece (measurement tool, not product code).
// hbm_roof shows the staging mechanism alone (LDS-DMA ring, 1 workgroup per CU) at 7.2 TB/s unshared / 11.1 TB/s into LDS
// when pairs of workgroups share a range; the product kernel moves 6.4 TB/s into LDS.  Which ingredient costs the rest?
// Model of layers 1 + 2 of C2 (the RG_WGRAD_LAYER_MASK=6 ablation: 70 us): 2 layers x 32 splits x 4 tiles (ng, kg) =
// 256 workgroups, 64 blocks each; stage = 16 KB of the A operand (tiles ng*8 .. +7 of block mb: contiguous, stride
// 32 KB) + 16 KB of B.  Flags add: the LDS fragment reads, the MFMAs, the 256 KB partial store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}

// MODE bits: 1 = LDS reads, 2 = MFMAs (implies reads), 4 = partial store, 8 = contiguous 32 KB per workgroup instead of
// the A / B interleave, 16 = ring of 2 in flight
template <int MODE>
__global__ void __launch_bounds__(512) k_model(const char* __restrict__ a, const char* __restrict__ b, float* __restrict__ part,
                                               int blocks, int splits, int xcd_map) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x;
  // in-situ decode: the 4 tiles of a split on one XCD
  int layer = bid / (4 * splits), lb = bid % (4 * splits), tile, split;
  if (xcd_map) { const int xcd = lb & 7, slot = lb >> 3; split = (slot / 4) * 8 + xcd; tile = slot % 4; }
  else { tile = lb % 4; split = lb / 4; }
  const int ng = tile >> 1, kg = tile & 1;
  const long layer_bytes = (long)splits * blocks * 32768;
  const char* pa = a + layer * layer_bytes + (long)split * blocks * 32768 + ng * 16384;
  const char* pb = b + layer * layer_bytes + (long)split * blocks * 32768 + kg * 16384;
  auto issue = [&](int s) {
    const int sc = s < blocks ? s : blocks - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned wb = lds0 + (s & 3) * 32768 + (unsigned)((wave * 64 + i * 512) * 16);
      const int u = (tid + i * 512) & 1023;
      // MODE 8: one contiguous 32 KB per stage — tiles 0, 1 read operand a's block, tiles 2, 3 operand b's (two readers each)
      const char* src = (MODE & 8) ? (ng ? pb - kg * 16384 : pa) + (long)sc * 32768 + (tid + i * 512) * 16
                                   : (i < 2 ? pa : pb) + (long)sc * 32768 + u * 16;
      glds16(src, wb);
    }
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int wn = wave >> 2, wk = wave & 3;
  constexpr int FLY = (MODE & 16) ? 2 : 3;
  issue(0); issue(1);
  if (FLY == 3) issue(2);
  unsigned sink = 0;
  for (int t = 0; t < blocks; ++t) {
    if (FLY == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(t + FLY);
    const char* base = smem + (t & 3) * 32768;
    if (MODE & 3) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        u16x8 af[4], bf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *(const u16x8*)(base + (wn * 4 + i) * 2048 + h * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *(const u16x8*)(base + 16384 + (wk * 2 + j) * 2048 + h * 1024 + lane * 16);
        if (MODE & 2) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, af[i]), __builtin_bit_cast(bf16x8_hw, bf[j]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) sink += af[i][0];
#pragma unroll
          for (int j = 0; j < 2; ++j) sink += bf[j][0];
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (MODE & 4) {
    float* p = part + ((long)layer * splits + split) * 262144;
    const int lr = lane & 31, lg = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = (kg * 8 + wk * 2 + j) * 32 + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (ng * 8 + wn * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lg;
          p[(long)row * 512 + col] = acc[i][j][r] + (float)sink;
        }
      }
  } else if (sink == 0x12345u || acc[0][0][0] == 1.2345f) part[0] = 1.f;
}

template <typename F> static float timeit(F f, int reps = 7) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best;
}
template <int MODE> static void run(const char* name, const char* a, const char* b, float* part, int layers, int splits, int blocks, int xcd) {
  hipFuncSetAttribute((const void*)k_model<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int wgs = layers * splits * 4;
  const float ms = timeit([&] { k_model<MODE><<<wgs, 512, 131072>>>(a, b, part, blocks, splits, xcd); });
  const double unique = 2.0 * layers * splits * blocks * 32768, lds = (double)wgs * blocks * 32768;
  printf("%-34s %4d wgs x %3d blocks xcd_map=%d: %7.1f us  unique %.2f TB/s  into LDS %.2f TB/s\n", name, wgs, blocks, xcd, ms * 1e3,
         unique / (ms * 1e-3) / 1e12, lds / (ms * 1e-3) / 1e12);
  fflush(stdout);
}
int main() {
  const int layers = 2, splits = 32, blocks = 64;  // B = 65536: 2048 32-row blocks per layer
  const size_t bytes = (size_t)layers * splits * blocks * 32768;
  char *a, *b; float* part;
  hipMalloc((void**)&a, bytes); hipMalloc((void**)&b, bytes); hipMalloc((void**)&part, (size_t)layers * 64 * 262144 * 4);
  hipMemset(a, 0x3c, bytes); hipMemset(b, 0x3c, bytes);
  for (int xcd = 1; xcd >= 0; --xcd) {
    run<0>("dma only", a, b, part, layers, splits, blocks, xcd);
    run<1>("+ LDS fragment reads", a, b, part, layers, splits, blocks, xcd);
    run<2>("+ MFMAs", a, b, part, layers, splits, blocks, xcd);
    run<6>("+ MFMAs + partial store", a, b, part, layers, splits, blocks, xcd);
    run<4>("dma + partial store", a, b, part, layers, splits, blocks, xcd);
    run<16>("dma only, two in flight", a, b, part, layers, splits, blocks, xcd);
    run<8>("dma only, one contiguous stream", a, b, part, layers, splits, blocks, xcd);
  }
  // two rounds of shorter workgroups (64 splits x 32 blocks) and one round of 128 workgroups x 128 blocks
  run<6>("full, 64 splits x 32 blocks", a, b, part, layers, 64, 32, 1);
  run<6>("full, 16 splits x 128 blocks", a, b, part, layers, 16, 128, 1);
  return 0;
}

// feasibility check (not product code): LDS-DMA (global_load_lds_dwordx4) issued from inline asm with
// manual vmcnt waits and a raw s_barrier; verifies that data lands lane-linearly at M0 + lane*16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}

__global__ void __launch_bounds__(512) k(const unsigned* src, unsigned* dst, int stages) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // ring of 4 slots x 32 KB; stage s -> slot s & 3; 4 DMAs per thread per stage
  auto issue = [&](int s) {
    for (int i = 0; i < 4; ++i) {
      const int u = tid + i * 512;
      const unsigned wave_base = lds0 + (s & 3) * 32768 + (unsigned)__builtin_amdgcn_readfirstlane((wave * 64 + i * 512) * 16);
      glds16(src + ((long)blockIdx.x * stages + s) * 8192 + u * 4, wave_base);
    }
  };
  issue(0); issue(1); issue(2);
  unsigned acc = 0;
  for (int s = 0; s < stages; ++s) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned* slot = (const unsigned*)(smem + (s & 3) * 32768);
    for (int i = 0; i < 16; ++i) acc += slot[(tid + i * 512) % 8192] * (i + 1);
    asm volatile("" ::: "memory");
    issue(s + 3 < stages ? s + 3 : stages - 1);  // keep the count uniform
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  dst[blockIdx.x * 512 + tid] = acc;
}

int main() {
  const int stages = 16, wgs = 512;
  const size_t n = (size_t)wgs * stages * 8192;
  std::vector<unsigned> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (unsigned)(i * 2654435761u >> 7);
  unsigned *src, *dst;
  hipMalloc((void**)&src, n * 4); hipMalloc((void**)&dst, wgs * 512 * 4);
  hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  k<<<wgs, 512, 131072>>>(src, dst, stages);
  std::vector<unsigned> out(wgs * 512);
  hipMemcpy(out.data(), dst, out.size() * 4, hipMemcpyDeviceToHost);
  long bad = 0;
  for (int b = 0; b < wgs; ++b)
    for (int t = 0; t < 512; ++t) {
      unsigned acc = 0;
      for (int s = 0; s < stages; ++s)
        for (int i = 0; i < 16; ++i) acc += h[((size_t)b * stages + s) * 8192 + (t + i * 512) % 8192] * (i + 1);
      bad += acc != out[b * 512 + t];
    }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) k<<<wgs, 512, 131072>>>(src, dst, stages);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mismatches %ld of %d ; %.1f us per launch, %.2f TB/s, err %d\n", bad, wgs * 512, ms * 50, n * 4 / (ms / 20 * 1e-3) / 1e12, (int)hipGetLastError());
  return bad != 0;
}

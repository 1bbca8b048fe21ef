// wgrad_mix.hip — do single-reader operand streams slow the shared ones down through the L2?  (measurement tool)
// The full weight-gradient launch runs its hidden-layer workgroups (four tiles of a split share operands through their
// XCD's L2) at 1.15 us per 32-row block; the same workgroups alone (layers 1 + 2 only) at 0.78.  Hypothesis: dW0 / the
// output layer's workgroups stream every byte ONCE, through the same L2s, and evict the shared lines before the sibling
// tile has read them.  256 workgroups, one round: 160 shared (40 quads x 102 blocks), 96 unshared (54 blocks of 32 KB),
//   mode 0 "mixed"     : every XCD hosts 20 shared + 12 unshared workgroups
//   mode 1 "dedicated" : XCDs 0-4 host the quads, XCDs 5-7 the single-reader streams
//   mode 2             : shared only (the unshared workgroups exit), mode 3: unshared only
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_wave_base) : "memory");
}
__global__ void __launch_bounds__(512) k_mix(const char* __restrict__ a, const char* __restrict__ b, const char* __restrict__ c,
                                             int mode, int sblocks, int ublocks, unsigned long long* life) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
  bool shared;
  int sidx = 0, uidx = 0;  // quad-major index of a shared workgroup: quad * 4 + tile; index of an unshared one
  if (mode == 1) { shared = xcd < 5; sidx = ((slot >> 2) * 5 + xcd) * 4 + (slot & 3); uidx = slot * 3 + (xcd - 5); }
  else { shared = slot < 20; sidx = ((slot >> 2) * 8 + xcd) * 4 + (slot & 3); uidx = (slot - 20) * 8 + xcd; }
  if ((mode == 2 && !shared) || (mode == 3 && shared)) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int blocks = shared ? sblocks : ublocks;
  const int quad = sidx >> 2, tile = sidx & 3, ng = tile >> 1, kg = tile & 1;
  const char* pa = shared ? a + (long)quad * sblocks * 32768 + ng * 16384 : c + (long)uidx * ublocks * 32768;
  const char* pb = shared ? b + (long)quad * sblocks * 32768 + kg * 16384 : pa + 16384;
  auto issue = [&](int s) {
    const int sc = s < blocks ? s : blocks - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned wb = lds0 + (s & 3) * 32768 + (unsigned)((wave * 64 + i * 512) * 16);
      glds16((i < 2 ? pa : pb) + (long)sc * 32768 + ((tid + i * 512) & 1023) * 16, wb);
    }
  };
  issue(0); issue(1); issue(2);
  unsigned sink = 0;
  for (int t = 0; t < blocks; ++t) {
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue(t + 3);
    sink += ((const unsigned*)(smem + (t & 3) * 32768))[tid];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) life[bid] = ((__builtin_amdgcn_s_memtime() - t0) << 1) | (shared ? 1 : 0) | (sink == 0x12345u ? 2 : 0);
}
int main() {
  const int sblocks = 102, ublocks = 54;
  char *a, *b, *c; unsigned long long* life;
  const size_t sb = (size_t)40 * sblocks * 32768, ub = (size_t)96 * ublocks * 32768;
  hipMalloc((void**)&a, sb); hipMalloc((void**)&b, sb); hipMalloc((void**)&c, ub); hipMalloc((void**)&life, 256 * 8);
  hipMemset(a, 0x3c, sb); hipMemset(b, 0x3c, sb); hipMemset(c, 0x3c, ub);
  hipFuncSetAttribute((const void*)k_mix, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"mixed", "dedicated XCDs", "shared only", "unshared only"};
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
      hipMemset(life, 0, 256 * 8);
      hipEventRecord(e0); k_mix<<<256, 512, 131072>>>(a, b, c, mode, sblocks, ublocks, life); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    unsigned long long h[256]; hipMemcpy(h, life, sizeof(h), hipMemcpyDeviceToHost);
    double ls = 0, lu = 0; int ns = 0, nu = 0;
    for (int i = 0; i < 256; ++i) { if (!h[i]) continue; if (h[i] & 1) { ls += (double)(h[i] >> 1); ++ns; } else { lu += (double)(h[i] >> 1); ++nu; } }
    printf("%-16s %6.1f us | shared: %3d wgs, %7.0f ticks each = %6.0f per block | unshared: %3d wgs, %7.0f ticks = %6.0f per block\n", names[mode],
           best * 1e3, ns, ns ? ls / ns : 0.0, ns ? ls / ns / sblocks : 0.0, nu, nu ? lu / nu : 0.0, nu ? lu / nu / ublocks : 0.0);
    fflush(stdout);
  }
  return 0;
}

// Calibration microbenchmark (not product code): sustained v_mfma_f32_32x32x16_bf16 rate of this
// box with the fused-MLP occupancy (512 threads / workgroup, 1 workgroup / CU, 8 independent
// accumulators per wave).  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void __launch_bounds__(512, 1) mfma_loop(float* out, int iters, int lds_bytes_touch) {
  extern __shared__ char smem[];
  if (lds_bytes_touch) smem[threadIdx.x] = 0;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
  f32x16 acc[8];
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int lds = 133 * 1024;
  hipFuncSetAttribute((const void*)mfma_loop, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int block : {512, 256}) {
    for (int iters : {1168}) {
      const int grid = 512;
      mfma_loop<<<grid, block, lds>>>(out, iters, 1);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) mfma_loop<<<grid, block, lds>>>(out, iters, 1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      const double flop = (double)grid * (block / 64) * iters * 8 * 32768.0;
      printf("block %d (waves/SIMD %d) grid %4d iters %5d : %8.2f us/launch  %7.1f TFLOP/s\n", block, block / 256, grid, iters, us, flop / us * 1e-6);
    }
  }
  for (int grid : {512}) {
    for (int iters : {1168}) {  // 1168*8... per wave: 146 iters x 8 = one fused forward's MFMAs
      mfma_loop<<<grid, 512, lds>>>(out, iters, 1);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r) mfma_loop<<<grid, 512, lds>>>(out, iters, 1);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / reps;
      const double flop = (double)grid * 8 * iters * 8 * 32768.0;
      printf("grid %4d iters %5d : %8.2f us/launch  %7.1f TFLOP/s\n", grid, iters, us, flop / us * 1e-6);
    }
  }
  return 0;
}

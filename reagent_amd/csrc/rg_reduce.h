// rg_reduce.h — workgroup reductions shared by the loss heads (heads.hip, crr.hip).
#pragma once
#include <rg_platform.h>

namespace rg {

// block-wide sum of a 256-thread workgroup in a fixed order (wave shuffles, then the 4 wave sums
// added in order): the same inputs give the same bits on every run
__device__ __forceinline__ float block_sum_256(float v, float* scratch /*[4]*/) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) scratch[wave] = v;
  __syncthreads();
  const float s = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
  __syncthreads();
  return s;
}

// this thread's share in[tid], in[tid + 256], ... of a sum over n values, as 16 independent partial sums combined in a fixed
// order: 16 loads in flight per thread.  (One load per loop turn is a chain of n / 256 dependent round trips: ~0.4 us each
// when the chip is busy — 100 us for the 65536 row terms of a C3 step's loss, measured in round 4.)
__device__ __forceinline__ float strided_sum_256(const float* __restrict__ in, int n, int tid) {
  float a[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = 0.f;
  int i = tid;
  for (; i + 15 * 256 < n; i += 16 * 256) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] += in[i + j * 256];
  }
  for (; i < n; i += 256) a[0] += in[i];
  return (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) +
         (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
}

}  // namespace rg

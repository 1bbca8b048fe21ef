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

}  // namespace rg

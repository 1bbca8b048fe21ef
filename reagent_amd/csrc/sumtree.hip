// sumtree.hip — device-resident sum tree for prioritized replay.
// Replaces reagent/replay_memory/sum_tree.py:30-189 (SumTree.set / get / sample /
// stratified_sample) and the per-index Python loops of PrioritizedReplayBuffer.set_priority /
// get_priority (reagent/replay_memory/prioritized_replay_buffer.py:146-180).
//
// Layout: one fp64 array in heap order, level d (d = 0 root .. depth = leaves) at offset 2^d - 1,
// 2^d nodes — the reference's list of power-of-two numpy levels laid end to end (fp64 like numpy).
// Two update paths:
//   * in-order walk (one thread; n <= 32 or no scratch given): the reference's arithmetic, operation
//     for operation — delta = value - leaf, then `+= delta` on the leaf and every ancestor
//     (sum_tree.py:180-187) — so scalar `set` calls and small batches are bit-identical to it;
//   * batched (n > 32): the last pair naming a leaf writes leaf = value, then every level is rebuilt
//     as node = left + right.  Deterministic in any execution order (the tree becomes a pure function
//     of its leaves) and free of the reference's accumulated rounding ("tolerable numerical
//     inaccuracies", :183); bit-identical to the sequential loop whenever the sums are exact in fp64,
//     last-place differences otherwise.
// The descent of `sample` is the reference's, operation for operation (sum_tree.py:115-131).
#include <rg_platform.h>
#include "../../include/reagent_hip.h"

namespace rg {

__device__ __forceinline__ double* level_ptr(double* tree, int d) { return tree + ((1L << d) - 1); }
__device__ __forceinline__ const double* level_ptr(const double* tree, int d) { return tree + ((1L << d) - 1); }

// one thread applies the updates in order, with SumTree.set's own arithmetic (sum_tree.py:180-187)
__global__ void sumtree_set_walk_kernel(double* tree, int depth, const int64_t* __restrict__ indices,
                                        const double* __restrict__ values, int n, long capacity) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  for (int j = 0; j < n; ++j) {
    long node = indices[j];
    if (node < 0 || node >= capacity) continue;  // the reference raises IndexError; checked on the host when it can be
    const double delta = values[j] - level_ptr(tree, depth)[node];
    for (int d = depth; d >= 0; --d) {
      level_ptr(tree, d)[node] += delta;
      node >>= 1;
    }
  }
}

// many updates: (1) the LAST position that names a leaf claims it, (2) the claimant writes the
// leaf and releases the claim, (3) the levels above are rebuilt bottom-up
__device__ __forceinline__ bool leaf_ok(long i, long capacity) { return i >= 0 && i < capacity; }

__global__ void sumtree_claim_kernel(const int64_t* __restrict__ indices, int n, int* __restrict__ claim,
                                     long capacity) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && leaf_ok(indices[j], capacity)) atomicMax(&claim[indices[j]], j);
}

__global__ void sumtree_scatter_kernel(double* tree, int depth, const int64_t* __restrict__ indices,
                                       const double* __restrict__ values, int n, const int* __restrict__ claim,
                                       long capacity) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && leaf_ok(indices[j], capacity) && claim[indices[j]] == j) level_ptr(tree, depth)[indices[j]] = values[j];
}

__global__ void sumtree_release_kernel(const int64_t* __restrict__ indices, int n, int* __restrict__ claim,
                                       long capacity) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n && leaf_ok(indices[j], capacity)) claim[indices[j]] = -1;
}

// one level: parent = left + right
__global__ void sumtree_level_kernel(double* tree, int d) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (1L << d)) return;
  const double* child = level_ptr(tree, d + 1);
  level_ptr(tree, d)[i] = child[2 * i] + child[2 * i + 1];
}

// levels d_top .. 0 (at most 1024 nodes wide) by one workgroup
__global__ void sumtree_top_kernel(double* tree, int d_top) {
  for (int d = d_top; d >= 0; --d) {
    const double* child = level_ptr(tree, d + 1);
    for (int i = threadIdx.x; i < (1 << d); i += blockDim.x) level_ptr(tree, d)[i] = child[2 * i] + child[2 * i + 1];
    __syncthreads();
  }
}

// SumTree.sample (sum_tree.py:97-131) for a batch of query values in [0, 1]
__global__ void sumtree_sample_kernel(const double* __restrict__ tree, int depth, const double* __restrict__ query01,
                                      int n, int64_t* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double q = query01[j] * tree[0];
  long node = 0;
  for (int d = 1; d <= depth; ++d) {
    const long left = node * 2;
    const double left_sum = level_ptr(tree, d)[left];
    if (q < left_sum) {
      node = left;
    } else {
      node = left + 1;
      q -= left_sum;
    }
  }
  out[j] = node;
}

__global__ void sumtree_get_kernel(const double* __restrict__ tree, int depth, const int64_t* __restrict__ indices,
                                   int n, float* __restrict__ out32, double* __restrict__ out64, long capacity) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double v = leaf_ok(indices[j], capacity) ? level_ptr(tree, depth)[indices[j]] : 0.0;
  if (out32) out32[j] = (float)v;  // get_priority returns float32 (prioritized_replay_buffer.py:176)
  if (out64) out64[j] = v;
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_sumtree_depth(int64_t capacity) {
  if (capacity <= 0) return -1;
  int depth = 0;
  while ((1L << depth) < capacity) ++depth;  // int(ceil(log2(capacity))), sum_tree.py:73
  return depth;
}

size_t rg_sumtree_nodes(int64_t capacity) {
  const int depth = rg_sumtree_depth(capacity);
  return depth < 0 ? 0 : (size_t)((1L << (depth + 1)) - 1);
}

int rg_sumtree_set(double* tree, int depth, int64_t capacity, const int64_t* indices, const double* values, int n,
                   int* claim, rg_stream_t stream) {
  if (!tree || depth < 0 || depth > 40 || n < 0 || (n > 0 && (!indices || !values))) return RG_EINVAL;
  if (capacity <= 0 || capacity > (1L << depth)) return RG_EINVAL;
  if (n == 0) return RG_OK;
  hipStream_t s = (hipStream_t)stream;
  if (n <= 32 || !claim) {
    RG_LAUNCH(sumtree_set_walk_kernel, dim3(1), dim3(64), s, tree, depth, indices, values, n, (long)capacity);
    return (int)hipGetLastError();
  }
  const dim3 grid((n + 255) / 256), block(256);
  RG_LAUNCH(sumtree_claim_kernel, grid, block, s, indices, n, claim, (long)capacity);
  RG_LAUNCH(sumtree_scatter_kernel, grid, block, s, tree, depth, indices, values, n, (const int*)claim,
            (long)capacity);
  RG_LAUNCH(sumtree_release_kernel, grid, block, s, indices, n, claim, (long)capacity);
  int d = depth - 1;
  for (; d > 10; --d)
    RG_LAUNCH(sumtree_level_kernel, dim3((unsigned)(((1L << d) + 255) / 256)), dim3(256), s, tree, d);
  if (d >= 0) RG_LAUNCH(sumtree_top_kernel, dim3(1), dim3(1024), s, tree, d);
  return (int)hipGetLastError();
}

int rg_sumtree_sample(const double* tree, int depth, const double* query01, int n, int64_t* out_indices,
                      rg_stream_t stream) {
  if (!tree || depth < 0 || n < 0 || (n > 0 && (!query01 || !out_indices))) return RG_EINVAL;
  if (n == 0) return RG_OK;
  RG_LAUNCH(sumtree_sample_kernel, dim3((n + 255) / 256), dim3(256), (hipStream_t)stream, tree, depth, query01, n,
            out_indices);
  return (int)hipGetLastError();
}

int rg_sumtree_get(const double* tree, int depth, int64_t capacity, const int64_t* indices, int n, float* out32,
                   double* out64, rg_stream_t stream) {
  if (!tree || depth < 0 || n < 0 || (n > 0 && (!indices || (!out32 && !out64)))) return RG_EINVAL;
  if (capacity <= 0 || capacity > (1L << depth)) return RG_EINVAL;
  if (n == 0) return RG_OK;
  RG_LAUNCH(sumtree_get_kernel, dim3((n + 255) / 256), dim3(256), (hipStream_t)stream, tree, depth, indices, n, out32,
            out64, (long)capacity);
  return (int)hipGetLastError();
}

}  // extern "C"

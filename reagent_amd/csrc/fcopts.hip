// fcopts.hip — the optional layer components of FullyConnectedNetwork that are off in every BASELINE configuration
// (reagent/models/fully_connected_network.py:101-153): batch normalisation of a layer's input (SlateBatchNorm1d on
// [batch, features] = nn.BatchNorm1d, :107-108) and dropout after the activation (:139-141).  Layers that use them run
// on the per-layer path in fp32 between the GEMMs (engine.GeneralFCStack); the residual wrapper's add is rg_add_cols.
// HBM-bound row sweeps: one lane per column (a wave reads 256 contiguous bytes of a row), column statistics as
// fixed-order fp64 partial sums (deterministic, no atomics).
#include "rg_gemm.h"
#include "../../include/reagent_hip.h"

namespace rg {

constexpr int BN_COLS = 64;       // columns per workgroup (one per lane)
constexpr int BN_WAVES = 4;       // row lanes of a workgroup
constexpr int BN_MAX_CHUNKS = 64; // row chunks (grid.y) of the partial-sum kernels

__host__ __device__ inline int bn_chunks(int batch) {
  const int c = (batch + 255) / 256;
  return c < 1 ? 1 : (c > BN_MAX_CHUNKS ? BN_MAX_CHUNKS : c);
}

// partial column sums over a row chunk: parts[chunk][0][c] = sum a, parts[chunk][1][c] = sum b where
//   STATS: a = x, b = x * x;     GRADS: a = g, b = g * xhat, xhat = (x - mean[c]) * rstd[c]
// (rstd null: 1 / sqrt(var[c] + eps), the frozen statistics of eval mode)
template <bool GRADS>
__global__ void bn_partial_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ g, long ldg,
                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                  const float* __restrict__ var, float eps, int batch, int n,
                                  double* __restrict__ parts) {
  __shared__ double red[2][BN_WAVES][BN_COLS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, c = blockIdx.x * BN_COLS + lane;
  const int chunks = gridDim.y, per = (batch + chunks - 1) / chunks;
  const int r0 = blockIdx.y * per, r1 = r0 + per < batch ? r0 + per : batch;
  double sa = 0.0, sb = 0.0;
  if (c < n) {
    const float m = GRADS ? mean[c] : 0.f, rs = GRADS ? (rstd ? rstd[c] : 1.f / sqrtf(var[c] + eps)) : 0.f;
    for (int r = r0 + w; r < r1; r += BN_WAVES) {
      const float xv = x[(long)r * ldx + c];
      if (GRADS) {
        const float gv = g[(long)r * ldg + c];
        sa += (double)gv;
        sb += (double)gv * (double)((xv - m) * rs);
      } else {
        sa += (double)xv;
        sb += (double)xv * (double)xv;
      }
    }
  }
  red[0][w][lane] = sa;
  red[1][w][lane] = sb;
  __syncthreads();
  if (w == 0 && c < n) {
    double ta = 0.0, tb = 0.0;
#pragma unroll
    for (int k = 0; k < BN_WAVES; ++k) {
      ta += red[0][k][lane];
      tb += red[1][k][lane];
    }
    parts[((long)blockIdx.y * 2 + 0) * n + c] = ta;
    parts[((long)blockIdx.y * 2 + 1) * n + c] = tb;
  }
}

// batch statistics of a training-mode forward (torch.nn.functional.batch_norm, training=True): the biased variance
// normalises, the unbiased one feeds running_var; running <- (1 - momentum) * running + momentum * batch
__global__ void bn_stats_finish_kernel(const double* __restrict__ parts, int chunks, int batch, int n, float eps,
                                       float momentum, int updates, float* __restrict__ save_mean,
                                       float* __restrict__ save_rstd,
                                       float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += parts[((long)k * 2 + 0) * n + c];
    q += parts[((long)k * 2 + 1) * n + c];
  }
  const double mean = s / (double)batch;
  double var = q / (double)batch - mean * mean;
  if (var < 0.0) var = 0.0;
  save_mean[c] = (float)mean;
  save_rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  // `updates` evaluations of the module on this batch (GaussianFullyConnectedActor runs its stack twice per forward)
  const double unbiased = batch > 1 ? var * (double)batch / (double)(batch - 1) : var;
  for (int u = 0; u < updates; ++u) {
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// y = (x - mean) * rstd * gamma + beta with rstd given, or 1 / sqrt(var + eps) from the running variance (eval mode)
__global__ void bn_apply_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ mean,
                                const float* __restrict__ rstd, const float* __restrict__ var, float eps,
                                const float* __restrict__ gamma, const float* __restrict__ beta, int batch, int n,
                                float* __restrict__ y, long ldy) {
  const long total = (long)batch * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    const float rs = rstd ? rstd[c] : 1.f / sqrtf(var[c] + eps);
    const float xh = (x[(long)r * ldx + c] - mean[c]) * rs;
    y[(long)r * ldy + c] = gamma ? xh * gamma[c] + (beta ? beta[c] : 0.f) : xh;
  }
}

__global__ void bn_grads_finish_kernel(const double* __restrict__ parts, int chunks, int n, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, float* __restrict__ sums /*[2][n]: sum g, sum g xhat*/) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < chunks; ++k) {
    s += parts[((long)k * 2 + 0) * n + c];
    q += parts[((long)k * 2 + 1) * n + c];
  }
  if (dbeta) dbeta[c] = (float)s;
  if (dgamma) dgamma[c] = (float)q;
  sums[c] = (float)s;
  sums[n + c] = (float)q;
}

// training: dx = gamma * rstd * (g - mean_b(g) - xhat * mean_b(g * xhat));  eval (frozen statistics): dx = gamma * rstd * g
__global__ void bn_dx_kernel(const float* __restrict__ g, long ldg, const float* __restrict__ x, long ldx,
                             const float* __restrict__ mean, const float* __restrict__ rstd,
                             const float* __restrict__ var, float eps, const float* __restrict__ gamma,
                             const float* __restrict__ sums, int training, int batch, int n, float* __restrict__ dx,
                             long lddx) {
  const long total = (long)batch * n;
  const float inv_b = 1.f / (float)batch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    const float rs = rstd ? rstd[c] : 1.f / sqrtf(var[c] + eps);
    const float ga = gamma ? gamma[c] : 1.f;
    const float gv = g[(long)r * ldg + c];
    float d;
    if (training) {
      const float xh = (x[(long)r * ldx + c] - mean[c]) * rs;
      d = ga * rs * (gv - sums[c] * inv_b - xh * (sums[n + c] * inv_b));
    } else {
      d = ga * rs * gv;
    }
    dx[(long)r * lddx + c] = d;
  }
}

// ---- dropout -----------------------------------------------------------------------------------------------------
// Philox4x32-10 keyed by (seed), counter (element group, offset): four uniforms per call
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint64_t seed, uint64_t group, uint64_t offset, uint32_t (&out)[4]) {
  uint32_t c[4] = {(uint32_t)group, (uint32_t)(group >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) out[i] = c[i];
}

// y = x * keep / (1 - p); keep[b, c] (bytes, [batch, n] contiguous) drawn here (GENERATE) or read (the backward)
template <bool GENERATE>
__global__ void dropout_kernel(const float* __restrict__ x, long ldx, int batch, int n, float p, float scale,
                               uint64_t seed, uint64_t offset, uint8_t* __restrict__ keep, float* __restrict__ y,
                               long ldy) {
  const long total = (long)batch * n, groups = (total + 3) / 4;
  for (long gi = (long)blockIdx.x * blockDim.x + threadIdx.x; gi < groups; gi += (long)gridDim.x * blockDim.x) {
    uint32_t u[4] = {0, 0, 0, 0};
    if (GENERATE) philox4x32_10(seed, (uint64_t)gi, offset, u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long i = gi * 4 + j;
      if (i >= total) break;
      uint8_t k;
      if (GENERATE) {
        // 24 random bits -> [0, 1): keep with probability 1 - p
        k = ((float)(u[j] >> 8) * (1.f / 16777216.f)) >= p ? 1 : 0;
        keep[i] = k;
      } else {
        k = keep[i];
      }
      const int r = (int)(i / n), c = (int)(i - (long)r * n);
      y[(long)r * ldy + c] = k ? x[(long)r * ldx + c] * scale : 0.f;
    }
  }
}

static inline unsigned sweep_blocks(long work) {
  long b = (work + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace rg

using namespace rg;

extern "C" {

size_t rg_batch_norm_workspace_bytes(int batch, int n) {
  if (batch <= 0 || n <= 0) return 0;
  return (size_t)bn_chunks(batch) * 2 * n * sizeof(double) + (size_t)2 * n * sizeof(float);
}

int rg_batch_norm_forward(const float* x, int64_t ldx, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int training, int stat_updates, double momentum, double eps, int batch, int n, float* y,
                          int64_t ldy, float* save_mean, float* save_rstd, void* workspace, size_t workspace_bytes,
                          rg_stream_t stream) {
  if (!x || !y || batch <= 0 || n <= 0 || stat_updates < 0) return RG_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if (training) {
    if (!save_mean || !save_rstd) return RG_EINVAL;
    if (!workspace || workspace_bytes < rg_batch_norm_workspace_bytes(batch, n)) return RG_EWORKSPACE;
    const int chunks = bn_chunks(batch);
    double* parts = (double*)workspace;
    RG_LAUNCH((bn_partial_kernel<false>), dim3((n + BN_COLS - 1) / BN_COLS, chunks), dim3(64 * BN_WAVES), s, x, (long)ldx,
              (const float*)nullptr, 0L, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.f, batch,
              n, parts);
    RG_LAUNCH(bn_stats_finish_kernel, dim3((n + 255) / 256), dim3(256), s, (const double*)parts, chunks, batch, n, (float)eps,
              (float)momentum, stat_updates, save_mean, save_rstd, running_mean, running_var);
    RG_LAUNCH(bn_apply_kernel, dim3(sweep_blocks((long)batch * n)), dim3(256), s, x, (long)ldx, (const float*)save_mean,
              (const float*)save_rstd, (const float*)nullptr, (float)eps, gamma, beta, batch, n, y, (long)ldy);
  } else {
    if (!running_mean || !running_var) return RG_EINVAL;
    RG_LAUNCH(bn_apply_kernel, dim3(sweep_blocks((long)batch * n)), dim3(256), s, x, (long)ldx, (const float*)running_mean,
              (const float*)nullptr, (const float*)running_var, (float)eps, gamma, beta, batch, n, y, (long)ldy);
  }
  return (int)hipGetLastError();
}

int rg_batch_norm_backward(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* gamma,
                           const float* mean, const float* rstd, const float* running_var, int training, double eps,
                           int batch, int n, float* dx, int64_t lddx, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, rg_stream_t stream) {
  if (!g || !x || !mean || batch <= 0 || n <= 0 || (!dx && !dgamma && !dbeta)) return RG_EINVAL;
  if (training ? !rstd : !running_var) return RG_EINVAL;
  if (!workspace || workspace_bytes < rg_batch_norm_workspace_bytes(batch, n)) return RG_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = bn_chunks(batch);
  double* parts = (double*)workspace;
  float* sums = (float*)((char*)workspace + (size_t)chunks * 2 * n * sizeof(double));
  const float* rs = training ? rstd : nullptr;  // eval mode: `mean` is the running mean, the variance the running one
  RG_LAUNCH((bn_partial_kernel<true>), dim3((n + BN_COLS - 1) / BN_COLS, chunks), dim3(64 * BN_WAVES), s, x, (long)ldx, g,
            (long)ldg, mean, rs, running_var, (float)eps, batch, n, parts);
  RG_LAUNCH(bn_grads_finish_kernel, dim3((n + 255) / 256), dim3(256), s, (const double*)parts, chunks, n, dgamma, dbeta, sums);
  if (dx)
    RG_LAUNCH(bn_dx_kernel, dim3(sweep_blocks((long)batch * n)), dim3(256), s, g, (long)ldg, x, (long)ldx, mean, rs,
              running_var, (float)eps, gamma, (const float*)sums, training, batch, n, dx, (long)lddx);
  return (int)hipGetLastError();
}

int rg_dropout(const float* x, int64_t ldx, int batch, int n, double p, int generate, uint64_t seed, uint64_t offset,
               uint8_t* keep, float* y, int64_t ldy, rg_stream_t stream) {
  if (!x || !y || !keep || batch <= 0 || n <= 0 || !(p >= 0.0 && p < 1.0)) return RG_EINVAL;
  const long groups = ((long)batch * n + 3) / 4;
  const float scale = (float)(1.0 / (1.0 - p));
  if (generate)
    RG_LAUNCH((dropout_kernel<true>), dim3(sweep_blocks(groups)), dim3(256), (hipStream_t)stream, x, (long)ldx, batch, n,
              (float)p, scale, seed, offset, keep, y, (long)ldy);
  else
    RG_LAUNCH((dropout_kernel<false>), dim3(sweep_blocks(groups)), dim3(256), (hipStream_t)stream, x, (long)ldx, batch, n,
              (float)p, scale, seed, offset, keep, y, (long)ldy);
  return (int)hipGetLastError();
}

}  // extern "C"

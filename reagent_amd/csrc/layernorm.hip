// layernorm.hip — nn.LayerNorm between a Linear and its activation, the `use_layer_norm` option of
// FullyConnectedNetwork (reagent/models/fully_connected_network.py:128-130: Linear -> LayerNorm(out_dim) -> activation).
// The layer runs on the per-layer path (rg_fc_forward with a linear epilogue writes the pre-norm z in fp32, these
// kernels normalise): one wave per row, the row held in registers between the two passes, fp32 statistics exactly as
// torch computes them (biased variance, eps inside the square root).  HBM-bound: 4 B in + 2..4 B out per element.
#include "rg_gemm.h"
#include "../../include/reagent_hip.h"

namespace rg {

constexpr int LN_MAX_PER_LANE = 32;  // rows up to 2048 wide
constexpr int LN_ROWS_PER_WG = 4;    // one wave each

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += shfl_xor(v, off);
  return v;
}

// y = act(((z - mean) * rstd) * gamma + beta); mean / rstd [batch] are kept for the backward
template <typename TY>
__global__ void layer_norm_fwd_kernel(const float* __restrict__ z, long ldz, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float eps, int act, int batch, int n,
                                      TY* __restrict__ y, long ldy, float* __restrict__ y32, long ldy32,
                                      float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = lane_id(), row = blockIdx.x * LN_ROWS_PER_WG + (threadIdx.x >> 6);
  if (row >= batch) return;
  const float* zr = z + (long)row * ldz;
  float v[LN_MAX_PER_LANE];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    v[i] = c < n ? zr[c] : 0.f;
    s += v[i];
  }
  const float mean = wave_sum(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    const float d = c < n ? v[i] - mean : 0.f;
    q += d * d;
  }
  const float rstd = 1.f / sqrtf(wave_sum(q) / (float)n + eps);
#pragma unroll
  for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
    const int c = lane + 64 * i;
    if (c < n) {
      const float o = act_apply((v[i] - mean) * rstd * gamma[c] + beta[c], act);
      if (y) y[(long)row * ldy + c] = cvt_out<TY>(o);
      if (y32) y32[(long)row * ldy32 + c] = o;
    }
  }
  if (lane == 0) {
    if (mean_out) mean_out[row] = mean;
    if (rstd_out) rstd_out[row] = rstd;
  }
}

// g [batch, n] = d loss / d (LayerNorm output, before the activation); xhat = (z - mean) * rstd
//   dz = rstd * (g*gamma - mean_n(g*gamma) - xhat * mean_n(g*gamma*xhat))
//   dgamma = sum_b g * xhat, dbeta = sum_b g     (per-workgroup partials [n_wg, 2n], summed by ln_param_grad_kernel)
template <typename TD>
__global__ void layer_norm_bwd_kernel(const float* __restrict__ g, long ldg, const float* __restrict__ z, long ldz,
                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                      const float* __restrict__ gamma, int batch, int n, TD* __restrict__ dz, long lddz,
                                      float* __restrict__ dz32, long lddz32, float* __restrict__ partials) {
  RG_DYN_LDS(smem);
  float* col = (float*)smem;  // [LN_ROWS_PER_WG][2 * n]: each row's (g * xhat | g), summed over the rows below
  const int lane = lane_id(), w = threadIdx.x >> 6, row = blockIdx.x * LN_ROWS_PER_WG + w;
  float* mine = col + (long)w * 2 * n;
  if (row < batch) {
    const float m = mean[row], r = rstd[row];
    const float* gr = g + (long)row * ldg;
    const float* zr = z + (long)row * ldz;
    float gg[LN_MAX_PER_LANE], xh[LN_MAX_PER_LANE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < n) {
        const float gv = gr[c];
        xh[i] = (zr[c] - m) * r;
        gg[i] = gv * gamma[c];
        s1 += gg[i];
        s2 += gg[i] * xh[i];
        mine[c] = gv * xh[i];
        mine[n + c] = gv;
      } else {
        xh[i] = gg[i] = 0.f;
      }
    }
    const float a = wave_sum(s1) / (float)n, b = wave_sum(s2) / (float)n;
#pragma unroll
    for (int i = 0; i < LN_MAX_PER_LANE; ++i) {
      const int c = lane + 64 * i;
      if (c < n) {
        const float d = r * (gg[i] - a - xh[i] * b);
        if (dz) dz[(long)row * lddz + c] = cvt_out<TD>(d);
        if (dz32) dz32[(long)row * lddz32 + c] = d;
      }
    }
  } else {
    for (int c = lane; c < 2 * n; c += 64) mine[c] = 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * n; c += blockDim.x) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < LN_ROWS_PER_WG; ++k) t += col[(long)k * 2 * n + c];
    partials[(long)blockIdx.x * 2 * n + c] = t;
  }
}

// dgamma [n] | dbeta [n] = column sums of the [n_wg, 2n] partials, in a fixed order
__global__ void ln_param_grad_kernel(const float* __restrict__ partials, int n_wg, int n2, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n2) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 3 < n_wg; k += 4) {
    s0 += partials[(long)k * n2 + c];
    s1 += partials[(long)(k + 1) * n2 + c];
    s2 += partials[(long)(k + 2) * n2 + c];
    s3 += partials[(long)(k + 3) * n2 + c];
  }
  for (; k < n_wg; ++k) s0 += partials[(long)k * n2 + c];
  const float t = (s0 + s1) + (s2 + s3);
  const int n = n2 / 2;
  if (c < n) dgamma[c] = t;
  else dbeta[c - n] = t;
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_layer_norm_forward(const float* z, int64_t ldz, const float* gamma, const float* beta, double eps, int act,
                          int batch, int n, void* y, int y_dtype, int64_t ldy, float* y32, int64_t ldy32, float* mean,
                          float* rstd, rg_stream_t stream) {
  if (!z || !gamma || !beta || batch <= 0 || n <= 0 || (!y && !y32)) return RG_EINVAL;
  if (n > 64 * LN_MAX_PER_LANE) return RG_EUNSUPPORTED;
  const dim3 grid((batch + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG), block(64 * LN_ROWS_PER_WG);
  if (y && y_dtype == RG_DT_BF16)
    RG_LAUNCH((layer_norm_fwd_kernel<bf16_t>), grid, block, (hipStream_t)stream, z, (long)ldz, gamma, beta, (float)eps, act,
              batch, n, (bf16_t*)y, (long)ldy, y32, (long)ldy32, mean, rstd);
  else
    RG_LAUNCH((layer_norm_fwd_kernel<float>), grid, block, (hipStream_t)stream, z, (long)ldz, gamma, beta, (float)eps, act,
              batch, n, (float*)y, (long)ldy, y32, (long)ldy32, mean, rstd);
  return (int)hipGetLastError();
}

size_t rg_layer_norm_backward_workspace_bytes(int batch, int n) {
  if (batch <= 0 || n <= 0) return 0;
  return (size_t)((batch + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG) * 2 * n * sizeof(float);
}

int rg_layer_norm_backward(const float* g, int64_t ldg, const float* z, int64_t ldz, const float* mean, const float* rstd,
                           const float* gamma, int batch, int n, void* dz, int dz_dtype, int64_t lddz, float* dz32,
                           int64_t lddz32, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                           rg_stream_t stream) {
  if (!g || !z || !mean || !rstd || !gamma || !dgamma || !dbeta || batch <= 0 || n <= 0 || (!dz && !dz32)) return RG_EINVAL;
  if (n > 64 * LN_MAX_PER_LANE) return RG_EUNSUPPORTED;
  if (!workspace || workspace_bytes < rg_layer_norm_backward_workspace_bytes(batch, n)) return RG_EWORKSPACE;
  const int n_wg = (batch + LN_ROWS_PER_WG - 1) / LN_ROWS_PER_WG;
  const dim3 grid(n_wg), block(64 * LN_ROWS_PER_WG);
  const size_t lds = (size_t)LN_ROWS_PER_WG * 2 * n * sizeof(float);
  float* parts = (float*)workspace;
  RG_ALLOW_LDS((layer_norm_bwd_kernel<bf16_t>), lds);
  RG_ALLOW_LDS((layer_norm_bwd_kernel<float>), lds);
  if (dz && dz_dtype == RG_DT_BF16)
    RG_LAUNCH_DYN((layer_norm_bwd_kernel<bf16_t>), grid, block, lds, (hipStream_t)stream, g, (long)ldg, z, (long)ldz, mean,
                  rstd, gamma, batch, n, (bf16_t*)dz, (long)lddz, dz32, (long)lddz32, parts);
  else
    RG_LAUNCH_DYN((layer_norm_bwd_kernel<float>), grid, block, lds, (hipStream_t)stream, g, (long)ldg, z, (long)ldz, mean,
                  rstd, gamma, batch, n, (float*)dz, (long)lddz, dz32, (long)lddz32, parts);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  RG_LAUNCH(ln_param_grad_kernel, dim3((2 * n + 255) / 256), dim3(256), (hipStream_t)stream, (const float*)parts, n_wg, 2 * n,
            dgamma, dbeta);
  return (int)hipGetLastError();
}

}  // extern "C"

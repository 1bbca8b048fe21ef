// preprocess.hip — dense feature normalization (HBM-bound elementwise).
// Replaces Preprocessor.forward, reagent/preprocessing/preprocessor.py:115-170 and the
// per-type bodies :197-525: one descriptor per OUTPUT column, evaluated in registers while the
// row streams through; presence multiply and the +-11.513 clamp are fused.
#include "rg_norm.h"

namespace rg {

__global__ void normalize_dense_kernel(const float* __restrict__ x, long ldx,
                                       const uint8_t* __restrict__ presence, long ldp,
                                       const rg_norm_col* __restrict__ cols, int n_out,
                                       const float* __restrict__ quantiles,
                                       float* __restrict__ out, long ldo, int batch) {
  const long total = (long)batch * n_out;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const long b = i / n_out;
    const int j = (int)(i % n_out);
    const rg_norm_col c = cols[j];
    const float v = x[b * ldx + c.in_col];
    const float p = presence ? (float)presence[b * ldp + c.in_col] : 1.f;
    out[b * ldo + j] = normalize_value(c, v, p, quantiles);
  }
}

}  // namespace rg

using namespace rg;

extern "C" int rg_normalize_dense(const float* x, int64_t ldx, const uint8_t* presence, int64_t ldp,
                                  const rg_norm_col* cols, int n_out, const float* quantiles,
                                  float* out, int64_t ldo, int batch, rg_stream_t stream) {
  if (!x || !cols || !out || n_out <= 0 || batch < 0) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  const long total = (long)batch * n_out;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  RG_LAUNCH(normalize_dense_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, x,
            (long)ldx, presence, (long)ldp, cols, n_out, quantiles, out, (long)ldo, batch);
  return (int)hipGetLastError();
}

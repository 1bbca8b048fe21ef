// fc.hip — FullyConnected layer ops (forward / dgrad / wgrad) on the rg_gemm.h core + C ABI.
#include "rg_gemm.h"
#include "../../include/reagent_hip.h"

namespace rg {

// ---- second-stage reduction of split partials (deterministic: fixed order) --------------------
__global__ void reduce_splits_kernel(const float* __restrict__ partials, long slab, int splits,
                                     float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partials[(long)k * slab + i];
  out[i] = s;
}

// ---- transpose + cast -------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ void transpose_cast_kernel(const TS* __restrict__ src, long ld_src, int rows, int cols,
                                      TD* __restrict__ dst, long ld_dst, TD* __restrict__ dst_t,
                                      long ld_t) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + i * 8, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) {
      v = cvt_in(src[(long)r * ld_src + c]);
      if (dst) dst[(long)r * ld_dst + c] = cvt_out<TD>(v);
    }
    tile[ty + i * 8][tx] = v;
  }
  __syncthreads();
  if (dst_t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + i * 8, r = r0 + tx;  // write dst_t[c][r]
      if (r < rows && c < cols) dst_t[(long)c * ld_t + r] = cvt_out<TD>(tile[tx][ty + i * 8]);
    }
  }
}

template <class P, class C, class Epi, int BIAS, int SWAP = 0>
static int launch_gemm(const GemmArgs& g, const Epi& epi, float* bias_partials, long bias_slab,
                       hipStream_t stream) {
  const int tiles = ((g.M + C::BM - 1) / C::BM) * ((g.N + C::BN - 1) / C::BN);
  const int grid = tiles * (g.splits > 1 ? g.splits : 1);
  if (grid <= 0) return RG_OK;
  RG_LAUNCH((gemm_nt_kernel<P, C, Epi, BIAS, SWAP>), dim3(grid), dim3(256), stream, g, epi, bias_partials,
            bias_slab);
  return (int)hipGetLastError();
}

// large bf16 shapes: the 256 x 256 DMA kernel (rg_gemm.h); same results bit for bit
template <class Epi>
static int launch_gemm_big(const GemmArgs& g, const Epi& epi, hipStream_t stream) {
  const int tiles = ((g.M + BIG_BM - 1) / BIG_BM) * ((g.N + BIG_BN - 1) / BIG_BN);
  const size_t lds = (size_t)BIG_SLOTS * BIG_STAGE_BYTES;
  RG_ALLOW_LDS((gemm_nt_big_kernel<Epi>), lds);
  RG_LAUNCH_DYN((gemm_nt_big_kernel<Epi>), dim3(tiles), dim3(BIG_THREADS), lds, stream, g, epi);
  return (int)hipGetLastError();
}
template <class P> struct IsBF16 { static constexpr bool value = false; };
template <> struct IsBF16<PrecBF16> { static constexpr bool value = true; };

template <class P>
static int fc_forward_t(const void* x, long ldx, const void* w, long ldw, const float* bias, void* y,
                        float* y32, long ldy, void* yt, long ldyt, int batch, int out_f, int in_f,
                        int act, hipStream_t stream) {
  typedef typename P::T T;
  GemmArgs g;
  g.A = x; g.B = w; g.lda = ldx; g.ldb = ldw;
  g.M = batch; g.N = out_f; g.K = in_f;
  g.k_per_split = ((in_f + P::BK - 1) / P::BK) * P::BK;
  g.splits = 1; g.bias_rows_only = 0;
  EpiForward<T> e;
  e.bias = bias; e.y = (T*)y; e.y32 = y32; e.ldy = ldy; e.yt = (T*)yt; e.ldyt = ldyt;
  e.act = act; e.M = batch; e.N = out_f;
  if (out_f <= 32) return launch_gemm<P, TileNarrow, EpiForward<T>, 0>(g, e, nullptr, 0, stream);
  if constexpr (IsBF16<P>::value) {
    if (gemm_big_ok(g)) return launch_gemm_big(g, e, stream);
    // no transposed copy wanted (inference, the last layer): swapped accumulators, 16-byte fp32 / 8-byte
    // bf16 row stores — the epilogue of these launches is bound by store issue, not by bytes
    if (!yt) return launch_gemm<P, TileWide, EpiForward<T>, 0, 1>(g, e, nullptr, 0, stream);
  }
  return launch_gemm<P, TileWide, EpiForward<T>, 0>(g, e, nullptr, 0, stream);
}

template <class P>
static int fc_dgrad_t(const void* dz, long lddz, const void* wt, long ldwt, const void* ht, long ldht,
                      int act_below, void* dx, float* dx32, long lddx, void* dxt, long lddxt,
                      int batch, int in_f, int out_f, hipStream_t stream) {
  typedef typename P::T T;
  GemmArgs g;
  g.A = dz; g.B = wt; g.lda = lddz; g.ldb = ldwt;
  g.M = batch; g.N = in_f; g.K = out_f;
  g.k_per_split = ((out_f + P::BK - 1) / P::BK) * P::BK;
  g.splits = 1; g.bias_rows_only = 0;
  EpiDgrad<T> e;
  e.ht = (const T*)ht; e.ldht = ldht; e.dx = (T*)dx; e.dx32 = dx32; e.lddx = lddx;
  e.dxt = (T*)dxt; e.lddxt = lddxt; e.act = act_below; e.M = batch; e.N = in_f;
  if (in_f <= 32) return launch_gemm<P, TileNarrow, EpiDgrad<T>, 0>(g, e, nullptr, 0, stream);
  if constexpr (IsBF16<P>::value)
    if (gemm_big_ok(g)) return launch_gemm_big(g, e, stream);
  return launch_gemm<P, TileWide, EpiDgrad<T>, 0>(g, e, nullptr, 0, stream);
}

// wgrad split plan (shared by the workspace query and the launcher)
struct WgradPlan {
  int narrow;       // 1: transposed variant (A = x^T, B = dz^T), used when out_features <= 32
  int splits;
  int k_per_split;
  long slab;        // floats per split slab of dw partials
  long bias_slab;   // floats per split slab of db partials
};
static WgradPlan wgrad_plan(int out_f, int in_f, int batch, int bk) {
  WgradPlan p;
  p.narrow = out_f <= 32;
  const int tiles = p.narrow ? ((in_f + 255) / 256) : ((out_f + 127) / 128) * ((in_f + 127) / 128);
  int want = (1024 + tiles - 1) / tiles;  // aim at ~1024 workgroups
  const int max_splits = (batch + bk - 1) / bk;
  if (want > max_splits) want = max_splits;
  if (want < 1) want = 1;
  if (want >= 8) want = (want / 8) * 8;  // multiple of 8 keeps the XCD decode bijective
  int kps = (batch + want - 1) / want;
  kps = ((kps + bk - 1) / bk) * bk;
  p.k_per_split = kps;
  p.splits = (batch + kps - 1) / kps;
  if (p.splits >= 8 && (p.splits % 8) != 0) p.splits = ((p.splits + 7) / 8) * 8;  // empty tail splits write zeros
  p.slab = (long)out_f * in_f;
  p.bias_slab = out_f;
  return p;
}

template <class P>
static int fc_wgrad_t(const void* dzt, long lddzt, const void* xt, long ldxt, float* dw, float* db,
                      void* ws, size_t ws_bytes, int out_f, int in_f, int batch, hipStream_t stream) {
  const WgradPlan p = wgrad_plan(out_f, in_f, batch, P::BK);
  const size_t need = (size_t)p.splits * (p.slab + p.bias_slab) * sizeof(float);
  if (ws_bytes < need || !ws) return RG_EWORKSPACE;
  float* part_w = (float*)ws;
  float* part_b = part_w + (long)p.splits * p.slab;
  GemmArgs g;
  g.K = batch; g.k_per_split = p.k_per_split; g.splits = p.splits; g.bias_rows_only = 0;
  EpiWgrad e;
  e.p = part_w; e.ld = in_f; e.slab = p.slab; e.split = 0;
  int rc;
  if (!p.narrow) {
    g.A = dzt; g.lda = lddzt; g.B = xt; g.ldb = ldxt; g.M = out_f; g.N = in_f;
    e.transposed = 0; e.M = out_f; e.N = in_f;
    rc = db ? launch_gemm<P, TileWide, EpiWgrad, 1>(g, e, part_b, p.bias_slab, stream)
            : launch_gemm<P, TileWide, EpiWgrad, 0>(g, e, part_b, p.bias_slab, stream);
  } else {
    g.A = xt; g.lda = ldxt; g.B = dzt; g.ldb = lddzt; g.M = in_f; g.N = out_f;
    e.transposed = 1; e.M = in_f; e.N = out_f;
    rc = db ? launch_gemm<P, TileNarrow, EpiWgrad, 2>(g, e, part_b, p.bias_slab, stream)
            : launch_gemm<P, TileNarrow, EpiWgrad, 0>(g, e, part_b, p.bias_slab, stream);
  }
  if (rc) return rc;
  const long n = p.slab;
  RG_LAUNCH(reduce_splits_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), stream,
            (const float*)part_w, p.slab, p.splits, dw, n);
  if (db)
    RG_LAUNCH(reduce_splits_kernel, dim3((unsigned)((out_f + 255) / 256)), dim3(256), stream,
              (const float*)part_b, p.bias_slab, p.splits, db, (long)out_f);
  return (int)hipGetLastError();
}

// dz = dy * act'(z) expressed through the activation output y = act(z): lets a stack whose LAST
// layer is non-linear (e.g. the tanh action head of FullyConnectedActor, reagent/models/actor.py:71-75)
// be trained by the kernels that assume a linear output layer
__global__ void act_backward_kernel(const float* __restrict__ dy, long ld_dy, const float* __restrict__ y, long ld_y,
                                    int act, float* __restrict__ dz, long ld_dz, int rows, int cols) {
  const long total = (long)rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols;
    const int c = (int)(i % cols);
    dz[r * ld_dz + c] = dy[r * ld_dy + c] * act_grad_from_output(y[r * ld_y + c], act);
  }
}

}  // namespace rg

using namespace rg;

extern "C" {

int rg_fc_forward(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y,
                  float* y32, int64_t ldy, void* yt, int64_t ldyt, int batch, int out_features,
                  int in_features, int act, int precision, rg_stream_t stream) {
  if (!x || !w || batch < 0 || out_features <= 0 || in_features <= 0) return RG_EINVAL;
  if (act < 0 || act > RG_ACT_SOFTPLUS) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  if (precision == RG_PREC_F32)
    return fc_forward_t<PrecF32>(x, ldx, w, ldw, bias, y, y32, ldy, yt, ldyt, batch, out_features,
                                 in_features, act, (hipStream_t)stream);
  if (precision == RG_PREC_BF16)
    return fc_forward_t<PrecBF16>(x, ldx, w, ldw, bias, y, y32, ldy, yt, ldyt, batch, out_features,
                                  in_features, act, (hipStream_t)stream);
  return RG_EUNSUPPORTED;
}

int rg_fc_dgrad(const void* dz, int64_t lddz, const void* wt, int64_t ldwt, const void* ht,
                int64_t ldht, int act_below, void* dx, float* dx32, int64_t lddx, void* dxt,
                int64_t lddxt, int batch, int in_features, int out_features, int precision,
                rg_stream_t stream) {
  if (!dz || !wt || batch < 0 || out_features <= 0 || in_features <= 0) return RG_EINVAL;
  if (batch == 0) return RG_OK;
  if (precision == RG_PREC_F32)
    return fc_dgrad_t<PrecF32>(dz, lddz, wt, ldwt, ht, ldht, act_below, dx, dx32, lddx, dxt, lddxt,
                               batch, in_features, out_features, (hipStream_t)stream);
  if (precision == RG_PREC_BF16)
    return fc_dgrad_t<PrecBF16>(dz, lddz, wt, ldwt, ht, ldht, act_below, dx, dx32, lddx, dxt, lddxt,
                                batch, in_features, out_features, (hipStream_t)stream);
  return RG_EUNSUPPORTED;
}

size_t rg_fc_wgrad_workspace_bytes(int out_features, int in_features, int batch, int precision) {
  const int bk = precision == RG_PREC_BF16 ? PrecBF16::BK : PrecF32::BK;
  if (batch <= 0) batch = 1;
  const WgradPlan p = wgrad_plan(out_features, in_features, batch, bk);
  return (size_t)p.splits * (p.slab + p.bias_slab) * sizeof(float);
}

int rg_fc_wgrad(const void* dzt, int64_t lddzt, const void* xt, int64_t ldxt, float* dw, float* db,
                void* workspace, size_t workspace_bytes, int out_features, int in_features, int batch,
                int precision, rg_stream_t stream) {
  if (!dzt || !xt || !dw || batch <= 0 || out_features <= 0 || in_features <= 0) return RG_EINVAL;
  if (precision == RG_PREC_F32)
    return fc_wgrad_t<PrecF32>(dzt, lddzt, xt, ldxt, dw, db, workspace, workspace_bytes, out_features,
                               in_features, batch, (hipStream_t)stream);
  if (precision == RG_PREC_BF16)
    return fc_wgrad_t<PrecBF16>(dzt, lddzt, xt, ldxt, dw, db, workspace, workspace_bytes,
                                out_features, in_features, batch, (hipStream_t)stream);
  return RG_EUNSUPPORTED;
}

int rg_act_backward(const float* dy, int64_t ld_dy, const float* y, int64_t ld_y, int act, float* dz,
                    int64_t ld_dz, int rows, int cols, rg_stream_t stream) {
  if (!dy || !y || !dz || rows < 0 || cols < 0) return RG_EINVAL;
  if (rows == 0 || cols == 0) return RG_OK;
  const long total = (long)rows * cols;
  long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  RG_LAUNCH(act_backward_kernel, dim3((unsigned)blocks), dim3(256), (hipStream_t)stream, dy, (long)ld_dy, y,
            (long)ld_y, act, dz, (long)ld_dz, rows, cols);
  return (int)hipGetLastError();
}

int rg_transpose_cast(const void* src, int src_dt, int64_t ld_src, int rows, int cols, void* dst,
                      int64_t ld_dst, void* dst_t, int64_t ld_t, int dst_dt, rg_stream_t stream) {
  if (!src || rows < 0 || cols < 0) return RG_EINVAL;
  if (rows == 0 || cols == 0) return RG_OK;
  const dim3 grid((cols + 31) / 32, (rows + 31) / 32), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (src_dt == RG_DT_F32 && dst_dt == RG_DT_F32)
    RG_LAUNCH((transpose_cast_kernel<float, float>), grid, block, s, (const float*)src, (long)ld_src,
              rows, cols, (float*)dst, (long)ld_dst, (float*)dst_t, (long)ld_t);
  else if (src_dt == RG_DT_F32 && dst_dt == RG_DT_BF16)
    RG_LAUNCH((transpose_cast_kernel<float, bf16_t>), grid, block, s, (const float*)src, (long)ld_src,
              rows, cols, (bf16_t*)dst, (long)ld_dst, (bf16_t*)dst_t, (long)ld_t);
  else if (src_dt == RG_DT_BF16 && dst_dt == RG_DT_BF16)
    RG_LAUNCH((transpose_cast_kernel<bf16_t, bf16_t>), grid, block, s, (const bf16_t*)src,
              (long)ld_src, rows, cols, (bf16_t*)dst, (long)ld_dst, (bf16_t*)dst_t, (long)ld_t);
  else if (src_dt == RG_DT_BF16 && dst_dt == RG_DT_F32)
    RG_LAUNCH((transpose_cast_kernel<bf16_t, float>), grid, block, s, (const bf16_t*)src,
              (long)ld_src, rows, cols, (float*)dst, (long)ld_dst, (float*)dst_t, (long)ld_t);
  else
    return RG_EUNSUPPORTED;
  return (int)hipGetLastError();
}

}  // extern "C"

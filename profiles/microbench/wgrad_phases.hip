// Phase timing of the grouped weight-gradient kernel (measurement tool, not product code).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc -I../../include wgrad_phases.hip -o wgrad_phases
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_stamps;
#define RG_PHASE_INIT() unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_t0 = ph_t, ph_acc[5] = {0, 0, 0, 0, 0}
#define RG_PHASE(i)                                              \
  do {                                                           \
    const unsigned long long n_ = __builtin_amdgcn_s_memtime(); \
    ph_acc[i] += n_ - ph_t;                                      \
    ph_t = n_;                                                   \
  } while (0)
#define RG_PHASE_FLUSH()                                                              \
  do {                                                                                \
    if ((threadIdx.x & 63) == 0) {                                                    \
      unsigned long long* o_ = g_stamps + ((long)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8; \
      for (int i_ = 0; i_ < 5; ++i_) o_[i_] = ph_acc[i_];                             \
      o_[5] = ph_t0;                                                                  \
      o_[6] = ph_t;                                                                   \
      o_[7] = ((unsigned long long)g.N << 32) | (unsigned)g.K;                        \
    }                                                                                 \
  } while (0)
#include "../../reagent_amd/csrc/mlp_fused.hip"
// microbench stubs: entry points of the library that live in other translation units and are not exercised here
namespace rg {
int x3_forward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
int x3_backward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
void grouped_bias_reduce_launch(const float*, const int*, int, int, float*, int, hipStream_t) {}
}
#include <cstdio>
#include <cstdlib>
#include <vector>

int main() {
  const int B = 65536, dims[5] = {128, 512, 512, 512, 16};
  rg_mlp_desc d{};
  d.n_layers = 4;
  for (int i = 0; i < 5; ++i) d.dims[i] = dims[i];
  for (int l = 0; l < 4; ++l) {
    void *af, *dz; float* dw;
    hipMalloc(&af, rg_frag_elems(B, dims[l]) * 2); hipMemset(af, 0x3c, rg_frag_elems(B, dims[l]) * 2);
    hipMalloc(&dz, rg_frag_elems(B, dims[l + 1]) * 2); hipMemset(dz, 0x3c, rg_frag_elems(B, dims[l + 1]) * 2);
    hipMalloc((void**)&dw, (size_t)dims[l] * dims[l + 1] * 4);
    d.act_frag[l] = af; d.dz_frag[l] = dz; d.dw[l] = dw;
  }
  const size_t wsb = rg_mlp_wgrad_fused_workspace_bytes(&d, B);
  void* ws; hipMalloc(&ws, wsb);
  const int max_wg = 1024;
  unsigned long long* stamps;
  hipMalloc((void**)&stamps, (size_t)max_wg * 8 * 8 * 8); hipMemset(stamps, 0, (size_t)max_wg * 8 * 8 * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) rg_mlp_wgrad_fused(&d, B, ws, wsb, nullptr);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) rg_mlp_wgrad_fused(&d, B, ws, wsb, nullptr);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("wgrad group + reduce: %.2f us per call (stamps on), err=%d, workspace %.1f MB\n", ms * 1e3 / 20, (int)hipGetLastError(), wsb / 1e6);
  hipMemset(stamps, 0, (size_t)max_wg * 8 * 8 * 8);
  rg_mlp_wgrad_fused(&d, B, ws, wsb, nullptr);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)max_wg * 8 * 8);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[5] = {"prologue (ring fill)", "compute + next issue", "vmcnt wait (HBM)", "barrier", "partial-tile store"};
  // every wave's record carries its layer's (N, K); workgroups that returned at once (padding) left zeros
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int g = 0; g < max_wg; ++g) for (int w = 0; w < 8; ++w) { const unsigned long long* s = &h[((size_t)g * 8 + w) * 8]; if (!s[6]) continue; if (s[5] < tmin) tmin = s[5]; if (s[6] > tmax) tmax = s[6]; }
  const double span = (double)(tmax - tmin);
  printf("kernel span (first start -> last end): %.0f ticks; plan: RG_WGRAD_PLAN=%s RG_WGRAD_TOTAL=%s RG_WGRAD_UNSHARED=%s\n", span,
         getenv("RG_WGRAD_PLAN") ? getenv("RG_WGRAD_PLAN") : "-", getenv("RG_WGRAD_TOTAL") ? getenv("RG_WGRAD_TOTAL") : "-",
         getenv("RG_WGRAD_UNSHARED") ? getenv("RG_WGRAD_UNSHARED") : "-");
  for (int layer = 0; layer < 4; ++layer) {
    const unsigned long long key = ((unsigned long long)dims[layer + 1] << 32) | (unsigned)dims[layer];
    double tot[5] = {0}, life = 0, start = 0, end = 0, last = 0, first_end = 1e30;
    int n = 0, wgs = 0;
    for (int g = 0; g < max_wg; ++g) {
      bool any = false;
      for (int w = 0; w < 8; ++w) {
        const unsigned long long* s = &h[((size_t)g * 8 + w) * 8];
        if (!s[6] || s[7] != key) continue;
        any = true;
        for (int p = 0; p < 5; ++p) tot[p] += (double)s[p];
        life += (double)(s[6] - s[5]); start += (double)(s[5] - tmin); end += (double)(s[6] - tmin);
        if ((double)(s[6] - tmin) > last) last = (double)(s[6] - tmin);
        if ((double)(s[6] - tmin) < first_end) first_end = (double)(s[6] - tmin);
        ++n;
      }
      wgs += any;
    }
    if (!n) continue;
    printf("layer %d (dW %dx%d): %d workgroups, lifetime %.0f ticks (%.1f %% of the span), mean start %.0f, mean end %.0f, first end %.0f, last end %.0f\n",
           layer, dims[layer + 1], dims[layer], wgs, life / n, 100.0 * life / n / span, start / n, end / n, first_end, last);
    for (int p = 0; p < 5; ++p) printf("    %-24s %9.0f  %5.1f %%\n", names[p], tot[p] / n, 100.0 * tot[p] / life);
  }
  return 0;
}

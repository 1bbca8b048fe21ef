// Phase timing of the fused BACKWARD kernel (round 6; measurement tool, not product code): the product kernel with its RG_BSTAMP hooks
// as s_memtime stamps.  C2 shapes: B = 65536, 128-512-512-512-16, dout [B, 16] fp32, sign planes for the ReLU layers, dZ fragments and
// bias-gradient partials written as in the training step.  argv[1] = output width (16 = C2; 200 = the width of C3's grouped layer,
// run here as a plain 200-wide output layer to see what a wide dout tile costs).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc -I../../include bwd_phases.hip -o bwd_phases
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_stamps;
#define RG_BSTAMP(slot)                                                                                  \
  do {                                                                                                   \
    if ((threadIdx.x & 63) == 0)                                                                         \
      g_stamps[((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 20 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#include "../../reagent_amd/csrc/mlp_fused.hip"
namespace rg {
int x3_forward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
int x3_backward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
void grouped_bias_reduce_launch(const float*, const int*, int, int, float*, int, hipStream_t) {}
}
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  using namespace rg;
  const int NO = argc > 1 ? atoi(argv[1]) : 16;
  const int B = 65536, dims[5] = {128, 512, 512, 512, NO};
  MlpArgs a{};
  a.n_layers = 4; a.batch = B;
  for (int i = 0; i < 5; ++i) a.dims[i] = dims[i];
  const int n_wg = B / 128;
  for (int l = 0; l < 4; ++l) {
    a.acts[l] = l < 3 ? ACT_RELU : ACT_LINEAR;
    void* w; const size_t n = rg_wfrag_elems(dims[l], dims[l + 1]);  // fragments of W_l^T
    hipMalloc(&w, n * 2); hipMemset(w, 0x3c, n * 2);
    a.wfrag[l] = (const bf16_t*)w;
    void* f; hipMalloc(&f, rg_frag_elems(B, dims[l + 1]) * 2); a.dz_frag[l] = (bf16_t*)f;
    float* dbp; hipMalloc((void**)&dbp, (size_t)n_wg * dims[l + 1] * 4); a.db_part[l] = dbp;
    if (l >= 1) { void* sg; hipMalloc(&sg, rg_sign_bytes(B, dims[l])); hipMemset(sg, 0x5a, rg_sign_bytes(B, dims[l])); a.act_sign[l] = (unsigned*)sg; }
  }
  float* dout;
  hipMalloc((void**)&dout, (size_t)B * NO * 4); hipMemset(dout, 0x3c, (size_t)B * NO * 4);
  a.dout32 = dout; a.lddo = NO; a.pitch = 520;
  const int NPH = 20, NWV = FB_NW;
  unsigned long long* stamps;
  hipMalloc((void**)&stamps, (size_t)n_wg * NWV * NPH * 8);
  hipMemset(stamps, 0, (size_t)n_wg * NWV * NPH * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
  const size_t lds = (size_t)128 * 520 * 2;
  auto kern = mlp_bwd_fused_kernel<512 / (32 * FB_NW), FB_NW, 520>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) kern<<<n_wg, FB_NW * 64, lds>>>(a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) kern<<<n_wg, FB_NW * 64, lds>>>(a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("backward NW=%d out=%d: %.2f us/launch (stamps on), err=%d\n", FB_NW, NO, ms * 1e3 / 20, (int)hipGetLastError());
  std::vector<unsigned long long> h((size_t)n_wg * NWV * NPH);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  // 0 start | 1 dout tile in LDS | 2 its dZ fragments + bias partial | per layer step s = 0..2 (l = 3, 2, 1): 3+4s main loop, 4+4s pack
  // (+ dZ fragment stores, column sums), 5+4s barrier wait, 6+4s LDS store + barrier
  const char* names[15] = {"", "dout tile load+barrier", "dZ_out frags + bias partial", "S0 mainloop (K=out)", "S0 pack", "S0 barrier wait", "S0 LDS store+barrier",
                           "S1 mainloop (K=512)", "S1 pack", "S1 barrier wait", "S1 LDS store+barrier", "S2 mainloop (K=512)", "S2 pack", "S2 barrier wait",
                           "S2 LDS store+barrier"};
  double tot[15] = {0}, span = 0;
  for (int g = 0; g < n_wg; ++g)
    for (int w = 0; w < NWV; ++w) {
      const unsigned long long* s = &h[((size_t)g * NWV + w) * NPH];
      for (int p = 1; p <= 14; ++p) tot[p] += (double)(s[p] - s[p - 1]);
      span += (double)(s[14] - s[0]);
    }
  const double nw = (double)n_wg * NWV;
  printf("avg s_memtime ticks per wave: %.0f\n", span / nw);
  for (int p = 1; p <= 14; ++p) printf("  %-30s %9.0f ticks  %5.1f %%\n", names[p], tot[p] / nw, 100.0 * tot[p] / span);
  return 0;
}

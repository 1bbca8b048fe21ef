// Phase timing of the fused forward kernel (measurement tool, not product code): compiles the
// product kernel with its RG_STAMP hooks turned into s_memtime stamps (one per wave per phase).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc -I../../include fwd_phases.hip -o fwd_phases
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_stamps;
#define RG_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if ((threadIdx.x & 63) == 0)                                                                         \
      g_stamps[((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 20 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#include "../../reagent_amd/csrc/mlp_fused.hip"
// microbench stubs: entry points of the library that live in other translation units and are not exercised here
namespace rg {
int x3_forward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
int x3_backward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
void grouped_bias_reduce_launch(const float*, const int*, int, int, float*, int, hipStream_t) {}
}
#include <cstdio>
#include <vector>

int main(int argc, char** argv) {
  using namespace rg;
  const int save = argc > 1 ? atoi(argv[1]) : 0;
  const int B = 65536, dims[5] = {128, 512, 512, 512, 16};
  MlpArgs a{};
  a.n_layers = 4; a.batch = B;
  for (int i = 0; i < 5; ++i) a.dims[i] = dims[i];
  for (int l = 0; l < 4; ++l) {
    a.acts[l] = l < 3 ? ACT_RELU : ACT_LINEAR;
    void* w; const size_t n = rg_wfrag_elems(dims[l + 1], dims[l]);
    hipMalloc(&w, n * 2); hipMemset(w, 0x3c, n * 2);
    a.wfrag[l] = (const bf16_t*)w;
    float* b; hipMalloc((void**)&b, dims[l + 1] * 4); hipMemset(b, 0, dims[l + 1] * 4);
    a.bias[l] = b;
    if (save) {
      void* f; hipMalloc(&f, rg_frag_elems(B, dims[l]) * 2); a.act_frag[l] = (bf16_t*)f;
      if (l >= 1) { void* sg; hipMalloc(&sg, rg_sign_bytes(B, dims[l])); a.act_sign[l] = (unsigned*)sg; }
    }
  }
  float *x, *out;
  hipMalloc((void**)&x, (size_t)B * 128 * 4); hipMemset(x, 0x3c, (size_t)B * 128 * 4);
  hipMalloc((void**)&out, (size_t)B * 16 * 4);
  a.x = x; a.ldx = 128; a.x_is_f32 = 1; a.out32 = out; a.ldo = 16; a.pitch = 520; a.save = save;
  const int n_wg = argc > 2 ? atoi(argv[2]) : B / 128, NPH = 20, NWV = FB_NW;  // argv[2]: fewer workgroups (clock / power experiments)
  unsigned long long* stamps;
  hipMalloc((void**)&stamps, (size_t)n_wg * NWV * NPH * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
  a.out_lds = argc > 3 ? atoi(argv[3]) : 0;  // argv[3] = 1: the output layer's weights resident in LDS (round 3)
  const size_t lds = (size_t)128 * 520 * 2 + (a.out_lds ? 32 * 512 : 0);
  hipFuncSetAttribute((const void*)mlp_fwd_fused_kernel<512 / (32 * FB_NW), FB_NW, 520>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) mlp_fwd_fused_kernel<512 / (32 * FB_NW), FB_NW, 520><<<n_wg, FB_NW * 64, lds>>>(a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) mlp_fwd_fused_kernel<512 / (32 * FB_NW), FB_NW, 520><<<n_wg, FB_NW * 64, lds>>>(a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("forward NW=%d RING=%d save=%d: %.2f us/launch (stamps on), err=%d\n", FB_NW, MlpCfg<FB_NW>::RING, save, ms * 1e3 / 20, (int)hipGetLastError());
  std::vector<unsigned long long> h((size_t)n_wg * NWV * NPH);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const char* names[NPH] = {"", "x tile load+barrier", "L0 mainloop(K=128)", "L0 pack", "L0 barrier wait", "L0 LDS store+barrier",
                            "L1 mainloop(K=512)", "L1 pack", "L1 barrier wait", "L1 LDS store+barrier",
                            "L2 mainloop(K=512)", "L2 pack", "L2 barrier wait", "L2 LDS store+barrier", "output layer", ""};
  double tot[NPH] = {0}, span = 0;
  for (int g = 0; g < n_wg; ++g)
    for (int w = 0; w < NWV; ++w) {
      const unsigned long long* s = &h[((size_t)g * NWV + w) * NPH];
      for (int p = 1; p <= 14; ++p) tot[p] += (double)(s[p] - s[p - 1]);
      span += (double)(s[14] - s[0]);
    }
  const double nw = (double)n_wg * NWV;
  printf("workgroups %d, shader clock %.3f GHz (ticks of one round / launch time)\n", n_wg, (span / nw) * ((n_wg + 255) / 256) / (ms * 1e6 / 20));
  printf("avg s_memtime ticks per wave: %.0f\n", span / nw);
  for (int p = 1; p <= 14; ++p) printf("  %-22s %9.0f ticks  %5.1f %%\n", names[p], tot[p] / nw, 100.0 * tot[p] / span);
  // per wave slot: duration of the L1 main loop (stamps 5 -> 6) and of its wait at the barrier (6 -> 7)
  {  // inside the output layer (one column tile): K loop | wait for the workgroup | hand-off through LDS + barrier | add + stores
    double t[4] = {0, 0, 0, 0}, t19 = 0, n19 = 0;
    for (int g = 0; g < n_wg; ++g)
      for (int w = 0; w < NWV; ++w) {
        const unsigned long long* s = &h[((size_t)g * NWV + w) * NPH];
        if (!s[16]) continue;
        if (s[19]) { t19 += (double)(s[19] - s[18]); n19 += 1; }
        t[0] += (double)(s[16] - s[13]); t[1] += (double)(s[17] - s[16]); t[2] += (double)(s[18] - s[17]); t[3] += (double)(s[14] - s[18]);
      }
    printf("  storing waves: LDS sum %.0f ticks (of their sum + stores)\n", t19 / (n19 > 0 ? n19 : 1));
    printf("  output layer split: K loop %.0f | barrier wait %.0f | hand-off + barrier %.0f | sum + stores %.0f ticks\n", t[0] / nw, t[1] / nw, t[2] / nw, t[3] / nw);
  }
  printf("per-wave L1 main loop / barrier wait (ticks), averaged over workgroups:\n");
  for (int w = 0; w < NWV; ++w) {
    double ml = 0, bw = 0, ep = 0;
    for (int g = 0; g < n_wg; ++g) {
      const unsigned long long* s = &h[((size_t)g * NWV + w) * NPH];
      ml += (double)(s[6] - s[5]); bw += (double)(s[7] - s[6]); ep += (double)(s[8] - s[7]);
    }
    printf("  wave %d: mainloop %8.0f  pack %8.0f  wait %8.0f\n", w, ml / n_wg, bw / n_wg, ep / n_wg);
  }
  return 0;
}

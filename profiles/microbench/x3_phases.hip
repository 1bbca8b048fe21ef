// Phase timing of the split-bf16 (bf16x3) fused forward kernel (measurement tool, not product code): the product kernel
// compiled with its RG_STAMP hooks turned into s_memtime stamps (one per wave per phase), random operands.
// Build: profiles/microbench/build.sh
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_stamps;
#define RG_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if ((threadIdx.x & 63) == 0)                                                                         \
      g_stamps[((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#include "../../reagent_amd/csrc/mlp_fused_x3.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
// the three size helpers of the library (mlp_fused.hip), restated for this stand-alone tool
extern "C" {
size_t rg_frag_elems(int rows, int cols) { return (size_t)((rows + 127) / 128 * 128) * (size_t)((cols + 31) / 32 * 32); }
size_t rg_sign_bytes(int rows, int cols) { return rg_frag_elems(rows, cols) / 8; }
size_t rg_wfrag_elems(int out_features, int in_features) { return (size_t)((out_features + 31) / 32) * (size_t)((in_features + 15) / 16) * 512; }
}

static void fill_random_bf16(void* d, size_t n, float scale) {
  std::vector<unsigned short> h(n);
  for (auto& v : h) {
    float f = ((rand() & 0xffff) / 32768.0f - 1.0f) * scale;
    unsigned u; memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

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
    hipMalloc(&w, 2 * n * 2); fill_random_bf16(w, 2 * n, 0.08f);
    a.wfrag[l] = (const bf16_t*)w; a.wfrag_lo[l] = (long)n;
    float* b; hipMalloc((void**)&b, dims[l + 1] * 4); hipMemset(b, 0, dims[l + 1] * 4);
    a.bias[l] = b;
    if (save) {
      const size_t fe = rg_frag_elems(B, dims[l]);
      void* f; hipMalloc(&f, 2 * fe * 2); a.act_frag[l] = (bf16_t*)f; a.act_lo[l] = (long)fe;
      if (l >= 1) { void* sg; hipMalloc(&sg, rg_sign_bytes(B, dims[l])); a.act_sign[l] = (unsigned*)sg; }
    }
  }
  float *x, *out;
  hipMalloc((void**)&x, (size_t)B * 128 * 4);
  {
    std::vector<float> h((size_t)B * 128);
    for (auto& v : h) v = (rand() & 0xffff) / 32768.0f - 1.0f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  hipMalloc((void**)&out, (size_t)B * 16 * 4);
  a.x = x; a.ldx = 128; a.x_is_f32 = 1; a.out32 = out; a.ldo = 16; a.pitch = 520; a.save = save;
  const int n_wg = argc > 2 ? atoi(argv[2]) : B / X3_BM, NPH = 16, NWV = FB_NW;  // argv[2]: fewer workgroups (clock / power experiments)
  unsigned long long* stamps;
  hipMalloc((void**)&stamps, (size_t)n_wg * NWV * NPH * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
  const size_t lds = (size_t)2 * X3_BM * 520 * 2;
  auto kern = mlp_fwd_x3_kernel<512 / (32 * FB_NW), FB_NW, 520>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) kern<<<n_wg, FB_NW * 64, lds>>>(a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) kern<<<n_wg, FB_NW * 64, lds>>>(a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("x3 forward NW=%d RING=%d save=%d: %.2f us/launch (stamps on), err=%d\n", FB_NW, RG_X3_RING, save, ms * 1e3 / 20, (int)hipGetLastError());
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
  printf("avg s_memtime ticks per wave: %.0f (MFMA-ideal cycles per 64-row workgroup: %.0f)\n", span / nw,
         3.0 * 2.0 * 64 * (128.0 * 512 + 2.0 * 512 * 512 + 512.0 * 16) / 4096.0);
  for (int p = 1; p <= 14; ++p) printf("  %-22s %9.0f ticks  %5.1f %%\n", names[p], tot[p] / nw, 100.0 * tot[p] / span);
  printf("per-wave L1 main loop / pack / barrier wait (ticks), averaged over workgroups:\n");
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

// Phase timing of the GROUPED fused forward kernel (round 6; measurement tool, not product code): the product kernel with its
// RG_STAMP hooks as s_memtime stamps.  C3 shapes: B = 65536 rows of 128 bf16 state features through a random row map, 3 x 512
// trunk, 16 groups x 200 outputs (group g = rows [4096 g, 4096 g + 4096) of the grouped space: no boundary tiles), fp32 output
// rows scattered back through the map (argv[1] = 1) or written in grouped order (0).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../reagent_amd/csrc -I../../include [-DRG_GROUPED_WHOLE=0] grouped_phases.hip -o grouped_phases
#include <hip/hip_runtime.h>
__device__ unsigned long long* g_stamps;
#define RG_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if ((threadIdx.x & 63) == 0)                                                                         \
      g_stamps[((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 24 + (slot)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#include "../../reagent_amd/csrc/mlp_fused.hip"
namespace rg {
int x3_forward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
int x3_backward_launch(const rg_mlp_desc*, MlpArgs&, hipStream_t) { return RG_EINVAL; }
void grouped_bias_reduce_launch(const float*, const int*, int, int, float*, int, hipStream_t) {}
}
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

int main(int argc, char** argv) {
  using namespace rg;
  const int scatter = argc > 1 ? atoi(argv[1]) : 1;
  const int B = 65536, G = 16, NQ = 200, dims[5] = {128, 512, 512, 512, NQ};
  MlpArgs a{};
  a.n_layers = 4; a.batch = B;
  for (int i = 0; i < 5; ++i) a.dims[i] = dims[i];
  const size_t gstride = rg_wfrag_elems(NQ, 512);
  for (int l = 0; l < 4; ++l) {
    a.acts[l] = l < 3 ? ACT_RELU : ACT_LINEAR;
    void* w; const size_t n = l < 3 ? rg_wfrag_elems(dims[l + 1], dims[l]) : gstride * G;
    hipMalloc(&w, n * 2); hipMemset(w, 0x3c, n * 2);
    a.wfrag[l] = (const bf16_t*)w;
    float* b; hipMalloc((void**)&b, (l < 3 ? dims[l + 1] : NQ * G) * 4); hipMemset(b, 0, (l < 3 ? dims[l + 1] : NQ * G) * 4);
    a.bias[l] = b;
  }
  std::vector<int> rowmap(B), key(B / 128), rb(G + 1);
  for (int i = 0; i < B; ++i) rowmap[i] = i;
  std::mt19937 rng(1);
  std::shuffle(rowmap.begin(), rowmap.end(), rng);
  for (int t = 0; t < B / 128; ++t) key[t] = t / (B / 128 / G);
  for (int g = 0; g <= G; ++g) rb[g] = g * (B / G);
  int *d_map, *d_key, *d_rb;
  hipMalloc((void**)&d_map, B * 4); hipMemcpy(d_map, rowmap.data(), B * 4, hipMemcpyHostToDevice);
  hipMalloc((void**)&d_key, key.size() * 4); hipMemcpy(d_key, key.data(), key.size() * 4, hipMemcpyHostToDevice);
  hipMalloc((void**)&d_rb, (G + 1) * 4); hipMemcpy(d_rb, rb.data(), (G + 1) * 4, hipMemcpyHostToDevice);
  bf16_t* x; float* out;
  hipMalloc((void**)&x, (size_t)B * 128 * 2); hipMemset(x, 0x3c, (size_t)B * 128 * 2);
  hipMalloc((void**)&out, (size_t)B * NQ * 4);
  a.x = x; a.ldx = 128; a.x_is_f32 = 0; a.out32 = out; a.ldo = NQ; a.pitch = 520; a.save = 0;
  a.rowmap = d_map; a.tile_key = d_key; a.row_begin = d_rb; a.n_groups = G; a.group_stride = (long)gstride; a.out_scatter = scatter;
  a.stage_out = 1; a.out_lds = 0;
  const int n_wg = B / 128, NPH = 24, NWV = FB_NW;
  unsigned long long* stamps;
  hipMalloc((void**)&stamps, (size_t)n_wg * NWV * NPH * 8);
  hipMemset(stamps, 0, (size_t)n_wg * NWV * NPH * 8);
  hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps));
  const size_t lds = (size_t)128 * 520 * 2 + (size_t)32 * (7 * 32 + 4) * 4;
  auto kern = mlp_fwd_grouped_kernel<512 / (32 * FB_NW), FB_NW, 520>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) kern<<<n_wg, FB_NW * 64, lds>>>(a);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 20; ++r) kern<<<n_wg, FB_NW * 64, lds>>>(a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("grouped forward WHOLE=%d RING=%d scatter=%d: %.2f us/launch (stamps on), err=%d\n", RG_GROUPED_WHOLE, RG_GROUPED_RING, scatter, ms * 1e3 / 20,
         (int)hipGetLastError());
  std::vector<unsigned long long> h((size_t)n_wg * NWV * NPH);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  // stamps: 0 start, 1 input tile, 2 + 4 l main loop, 3 + 4 l pack, 4 + 4 l barrier wait, 5 + 4 l LDS store (l = 0..2), 14 end of the output layer;
  // whole-tile path: 16 after its K loop, 17 after the barrier, 18 after staging + barrier, 19 after the row stores were issued
  const char* names[15] = {"", "x tile (row map)", "L0 mainloop(K=128)", "L0 pack", "L0 barrier wait", "L0 LDS store+barrier", "L1 mainloop(K=512)", "L1 pack",
                           "L1 barrier wait", "L1 LDS store+barrier", "L2 mainloop(K=512)", "L2 pack", "L2 barrier wait", "L2 LDS store+barrier", "grouped output layer"};
  double tot[15] = {0}, span = 0, o[4] = {0, 0, 0, 0};
  for (int g = 0; g < n_wg; ++g)
    for (int w = 0; w < NWV; ++w) {
      const unsigned long long* s = &h[((size_t)g * NWV + w) * NPH];
      for (int p = 1; p <= 14; ++p) tot[p] += (double)(s[p] - s[p - 1]);
      span += (double)(s[14] - s[0]);
      if (s[16]) { o[0] += (double)(s[16] - s[13]); o[1] += (double)(s[17] - s[16]); o[2] += (double)(s[18] - s[17]); o[3] += (double)(s[19] - s[18]); }
    }
  const double nw = (double)n_wg * NWV;
  printf("avg s_memtime ticks per wave: %.0f (100 MHz ticks -> %.1f us per workgroup)\n", span / nw, span / nw / 100.0);
  for (int p = 1; p <= 14; ++p) printf("  %-24s %9.0f ticks  %5.1f %%\n", names[p], tot[p] / nw, 100.0 * tot[p] / span);
  printf("  whole-tile path: K loop %.0f | barrier wait %.0f | staging + barrier %.0f | row stores issued %.0f ticks\n", o[0] / nw, o[1] / nw, o[2] / nw, o[3] / nw);
  for (int w = 0; w < NWV; ++w) {
    double k = 0;
    for (int g = 0; g < n_wg; ++g) { const unsigned long long* s = &h[((size_t)g * NWV + w) * NPH]; k += s[16] ? (double)(s[16] - s[13]) : 0; }
    printf("  wave %d: output K loop %8.0f\n", w, k / n_wg);
  }
  return 0;
}

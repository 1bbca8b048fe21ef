// TEST INFRASTRUCTURE ONLY — never shipped, never loaded by the reagent_amd package.
//
// A drop-in replacement for reagent_amd/csrc/rg_platform.h that lets the *unmodified* kernel
// sources be compiled for the x86 host (amdclang++ -x c++) and executed by a tiny SIMT
// interpreter: every GPU thread of a workgroup is a ucontext fiber, `__syncthreads()` and the
// wave64 collectives (MFMA, shuffles) are rendezvous points.  It exists so that `pytest -m "not
// gpu"` can check the kernels' index arithmetic / fragment layouts / epilogues against the oracle in
// a container that has no GPU.  It models the MFMA fragment layouts documented in the product
// header; tests/test_gpu_mfma_layout.py checks those same layouts on real hardware.
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <functional>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct __attribute__((aligned(16))) uint4 {
  unsigned x, y, z, w;
};
struct __attribute__((aligned(8))) uint2 {
  unsigned x, y;
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
// device query of the weight gradient's launch plan: the interpreter poses as a 256-CU chip
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
  memset(p, v, n);
  return 0;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) {
  memcpy(d, s, n);
  return 0;
}
#define hipMemcpyDeviceToDevice 3
#define hipMemcpyDefault 4

namespace emu {
struct ThreadCtx {
  dim3 tid, bid, bdim, gdim;
  int linear_tid;
};
ThreadCtx& cur();
void launch(dim3 grid, dim3 block, const std::function<void()>& body, size_t dyn_lds_bytes = 0);
char* dyn_lds();
void sync_threads();
// wave collective: every live lane of the wave deposits `bytes` bytes; returns pointer to a
// 64-slot array (slot stride = bytes) holding all lanes' values, valid until the next collective.
const void* wave_gather(const void* mine, size_t bytes);
}  // namespace emu

#define threadIdx (::emu::cur().tid)
#define blockIdx (::emu::cur().bid)
#define blockDim (::emu::cur().bdim)
#define gridDim (::emu::cur().gdim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __restrict__
#define __launch_bounds__(...)
static inline void __syncthreads() { ::emu::sync_threads(); }

namespace rg {

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

static inline float bf16_to_f32(bf16_t v) {
  uint32_t u = ((uint32_t)v) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline bf16_t f32_to_bf16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  uint32_t r = 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)((u + r) >> 16);
}

static inline int lane_id() { return ::emu::cur().linear_tid & 63; }
static inline int opaque(int x) { return x; }
static inline void sched_fence() {}
// LDS-DMA: the interpreter copies synchronously (lane-linear destination, like the hardware)
static inline void global_load_lds_b128(const void* gsrc, const void* lds_wave_base) {
  memcpy((char*)lds_wave_base + lane_id() * 16, gsrc, 16);
}
static inline void global_load_lds_b128_cached(const void* gsrc, const void* lds_wave_base) {
  memcpy((char*)lds_wave_base + lane_id() * 16, gsrc, 16);
}
#define RG_WAIT_VMCNT(n) ((void)0)
static inline void raw_barrier() { __syncthreads(); }
#define RG_SETPRIO(n) ((void)0)
#define RG_SCHED_MFMA(n) ((void)0)
#define RG_SCHED_DS_READ(n) ((void)0)
#define RG_SCHED_VMEM_READ(n) ((void)0)
#define RG_SCHED_VALU(n) ((void)0)
#define RG_SCHED_SALU(n) ((void)0)
static inline unsigned pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}

template <typename T>
static inline T gather_from(T v, int src_lane) {
  const T* all = (const T*)::emu::wave_gather(&v, sizeof(T));
  return all[src_lane & 63];
}
// individually rounded fp32 operations (HIP intrinsics; the host target has no fused multiply-add to contract into)
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
template <typename T> static inline void stream_store(T v, T* p) { *p = v; }
template <typename T> static inline T stream_load(const T* p) { return *p; }
static inline void pin_packed(const unsigned (&)[8]) {}
static inline float shfl_xor(float v, int mask) { return gather_from(v, lane_id() ^ mask); }
static inline int shfl_xor(int v, int mask) { return gather_from(v, lane_id() ^ mask); }
static inline float swap_adjacent_lanes(float v) { return gather_from(v, lane_id() ^ 1); }
static inline unsigned swap_adjacent_lanes(unsigned v) { return (unsigned)gather_from((int)v, lane_id() ^ 1); }
static inline unsigned perm_bytes(unsigned hi, unsigned lo, unsigned sel) {
  const unsigned long long src = ((unsigned long long)hi << 32) | lo;
  unsigned out = 0;
  for (int i = 0; i < 4; ++i) out |= (unsigned)((src >> (8 * ((sel >> (8 * i)) & 7))) & 0xff) << (8 * i);
  return out;
}
static inline int wave_uniform(int x) { return x; }
// fibers are cooperative (one runs at a time), so an atomic is a plain read-modify-write
static inline int atomicMax(int* addr, int v) { const int old = *addr; if (v > old) *addr = v; return old; }
// lanes are fibers here: a wave-level gather is the point where all lanes of the wave have run up to
static inline void wave_lds_sync() { int x = 0; (void)::emu::wave_gather(&x, sizeof(x)); }
static inline int atomicAdd(int* addr, int v) { const int old = *addr; *addr = old + v; return old; }
static inline float shfl_down(float v, int d) {
  int l = lane_id();
  return gather_from(v, (l + d < 64) ? l + d : l);
}
template <int MASK> static inline float shfl_xor_c(float v) { return gather_from(v, lane_id() ^ MASK); }
static inline float med3(float a, float b, float c) {
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return c < lo ? lo : (c > hi ? hi : c);
}
// the device's DPP prefix sum, step by step in the same order (row_shr 1 / 2 / 4 / 8 inside rows of 16, row_bcast:15, row_bcast:31)
static inline double wave_inclusive_sum_f64(double v) {
  const int l = lane_id();
  for (int sh = 1; sh <= 8; sh <<= 1) {
    const double* all = (const double*)::emu::wave_gather(&v, sizeof(double));
    const double src = (l & 15) >= sh ? all[l - sh] : 0.0;
    v += src;
  }
  {
    const double* all = (const double*)::emu::wave_gather(&v, sizeof(double));
    const int row = l >> 4;
    v += (row == 1 || row == 3) ? all[(row - 1) * 16 + 15] : 0.0;
  }
  {
    const double* all = (const double*)::emu::wave_gather(&v, sizeof(double));
    v += (l >= 32) ? all[31] : 0.0;
  }
  return v;
}
template <int SRC> static inline double read_lane_f64(double x) { return gather_from(x, SRC); }
static inline unsigned long long wave_ballot(bool p) {
  const int mine = p ? 1 : 0;
  const int* all = (const int*)::emu::wave_gather(&mine, sizeof(int));
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l) m |= (unsigned long long)(all[l] & 1) << l;
  return m;
}
static inline float shfl_idx(float v, int src) { return gather_from(v, src); }
static inline int shfl_idx(int v, int src) { return gather_from(v, src); }

static inline f32x16 mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
  struct AB {
    u16x8 a, b;
  } mine{a, b};
  const AB* all = (const AB*)::emu::wave_gather(&mine, sizeof(AB));
  const int l = lane_id();
  const int j = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 16; ++k) {
      const float av = bf16_to_f32(all[i + 32 * (k >> 3)].a[k & 7]);
      const float bv = bf16_to_f32(all[j + 32 * (k >> 3)].b[k & 7]);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  return d;
}
static inline f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
  struct AB {
    float a, b;
  } mine{a, b};
  const AB* all = (const AB*)::emu::wave_gather(&mine, sizeof(AB));
  const int l = lane_id();
  const int j = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = d[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(all[i + 32 * k].a, all[j + 32 * k].b, acc);
    d[r] = acc;
  }
  return d;
}

}  // namespace rg

#define RG_LAUNCH(kernel, grid, block, stream, ...) \
  ::emu::launch(grid, block, [&]() { kernel(__VA_ARGS__); })
#define RG_LAUNCH_BOUNDS(t, w)
#define RG_DYN_LDS(name) char* name = ::emu::dyn_lds()
#define RG_LAUNCH_DYN(kernel, grid, block, lds_bytes, stream, ...) \
  ::emu::launch(grid, block, [&]() { kernel(__VA_ARGS__); }, lds_bytes)
#define RG_ALLOW_LDS(kernel, bytes) (void)0

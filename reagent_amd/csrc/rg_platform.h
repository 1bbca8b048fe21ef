// rg_platform.h — gfx950 (CDNA4) device primitives used by every kernel in this library.
//
// Everything hardware-specific that the kernels touch goes through the thin `rg::` wrappers
// declared here: MFMA issue, wave64 cross-lane moves, bf16 conversion and the launch macro.
// Keeping them in one header means the fragment layouts below are stated exactly once.
//
// Fragment layouts (wave64; `l` = lane id), from /opt/skills/guides/cdna_hip_programming.md §3:
//   mfma_f32_32x32x16_bf16 : A[i][k]  lane l holds i = l&31, k = (l>>5)*8 + e   (e = 0..7)
//                            B[k][j]  lane l holds j = l&31, k = (l>>5)*8 + e
//                            D[i][j]  lane l, reg r: j = l&31, i = (r&3) + 8*(r>>2) + 4*(l>>5)
//   mfma_f32_32x32x2f32    : A[i][k]  lane l holds i = l&31, k = l>>5 (one f32)
//                            B[k][j]  lane l holds j = l&31, k = l>>5 ; D as above
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rg {

typedef unsigned short bf16_t;  // raw bf16 bits in memory
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_hw;

__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(u16x8 a, u16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_hw, a),
                                                 __builtin_bit_cast(bf16x8_hw, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_32x32x2_f32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __builtin_bit_cast(float, ((unsigned)v) << 16);
}
// round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (__bf16)f);
}

__device__ __forceinline__ float shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int shfl_xor(int v, int mask) { return __shfl_xor(v, mask, 64); }
// value of lane^1 through DPP quad_perm [1,0,3,2]: a VALU move, no LDS crossbar round trip
__device__ __forceinline__ unsigned swap_adjacent_lanes(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ float swap_adjacent_lanes(float v) {
  return __builtin_bit_cast(float, swap_adjacent_lanes(__builtin_bit_cast(unsigned, v)));
}
// v_perm_b32: result byte i = byte (sel >> 8i & 7) of the 8-byte value {hi, lo}
__device__ __forceinline__ unsigned perm_bytes(unsigned hi, unsigned lo, unsigned sel) {
  return __builtin_amdgcn_perm(hi, lo, sel);
}
// a value the program knows to be the same in every lane of the wave -> SGPR (scalar address math)
__device__ __forceinline__ int wave_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
// Inclusive prefix sum of a double over the 64 lanes by DPP moves (no LDS-pipe permutes): Kogge-Stone inside each row of 16
// lanes (row_shr 1, 2, 4, 8: a source lane outside the row reads as 0), then lane 15 of rows 0 / 2 into rows 1 / 3
// (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31).  A double moves as its two dwords.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_move_f64(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_inclusive_sum_f64(double v) {
  v += dpp_move_f64<0x111, 0xf>(v);  // row_shr:1
  v += dpp_move_f64<0x112, 0xf>(v);  // row_shr:2
  v += dpp_move_f64<0x114, 0xf>(v);  // row_shr:4
  v += dpp_move_f64<0x118, 0xf>(v);  // row_shr:8
  v += dpp_move_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_move_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
  return v;
}
// lane `src`'s value (src a compile-time lane index), through a scalar register
template <int SRC> __device__ __forceinline__ double read_lane_f64(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_readlane((int)b, SRC), hi = __builtin_amdgcn_readlane((int)(b >> 32), SRC);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
// bit l = lane l's predicate (all 64 lanes of the wave take part)
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ float shfl_idx(float v, int src) { return __shfl(v, src, 64); }
__device__ __forceinline__ int shfl_idx(int v, int src) { return __shfl(v, src, 64); }

// value of lane ^ MASK for a compile-time MASK: a DPP quad permute (1, 2: a VALU move the consumer can absorb), a
// ds_swizzle in bit mode (4, 8, 16: the LDS crossbar without the address VGPR ds_bpermute needs), ds_bpermute for 32
template <int MASK> __device__ __forceinline__ float shfl_xor_c(float v) {
  const int x = __builtin_bit_cast(int, v);
  int r;
  if constexpr (MASK == 1) r = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
  else if constexpr (MASK == 2) r = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
  else if constexpr (MASK < 32) r = __builtin_amdgcn_ds_swizzle(x, (MASK << 10) | 0x1F);      // and 0x1f, or 0, xor MASK
  else r = __shfl_xor(x, MASK, 64);
  return __builtin_bit_cast(float, r);
}
// median of three (v_med3_f32): med3(a, b, -inf) = min(a, b), med3(a, b, +inf) = max(a, b) for NaN-free inputs — a
// compare-exchange whose direction is DATA (the third operand), not control flow
__device__ __forceinline__ float med3(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// LDS hand-off between the lanes of ONE wave: the wave's earlier ds_writes are complete and visible to its later
// ds_reads (the lanes run in lockstep and LDS operations of a wave retire in order; this pins the compiler and
// drains the counter).  No s_barrier: other waves are not involved.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value the optimiser must treat as unknown here: keeps address arithmetic that depends on it from
// being hoisted out of an enclosing loop and parked in VGPRs across an MFMA main loop
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// store / load with the non-temporal hint: write-once streams that the writer does not read back, read-once streams —
// they should pass by the caches instead of evicting what the MFMA kernels keep re-reading (weights, inputs)
template <typename T> __device__ __forceinline__ void stream_store(T v, T* p) { __builtin_nontemporal_store(v, p); }
template <typename T> __device__ __forceinline__ T stream_load(const T* p) { return __builtin_nontemporal_load(p); }
// "these eight registers are needed now": an empty asm that consumes them (forces a computation to be finished here)
__device__ __forceinline__ void pin_packed(const unsigned (&P)[8]) {
  asm volatile("" ::"v"(P[0]), "v"(P[1]), "v"(P[2]), "v"(P[3]), "v"(P[4]), "v"(P[5]), "v"(P[6]), "v"(P[7]));
}
// scheduling fence: nothing is moved across it (pins "issue the loads, then the MFMA block")
__device__ __forceinline__ void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// ---- LDS-DMA (global_load_lds_dwordx4): 16 bytes per lane straight from global memory into LDS at
// wave-uniform base + lane*16, no VGPR round trip.  Issued from inline asm so that hipcc neither counts
// it nor drains it at the next barrier (its own bookkeeping would put s_waitcnt vmcnt(0) there and
// collapse a multi-stage ring to one stage in flight); the caller pairs it with RG_WAIT_VMCNT(n) and
// raw_barrier().  `lds_wave_base` must be the same in every lane of the wave.
__device__ __forceinline__ void global_load_lds_b128(const void* gsrc, const void* lds_wave_base) {
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_wave_base);
  unsigned keep;
  // `nt`: the operand stream of the weight gradient (420 MB per launch, each byte used once per reader) must not push
  // the weights and the next launches' inputs out of L2 / the memory-side cache (same-box A/B of the C2 step: -1.7 %,
  // the gain showing in the forward and backward launches that follow)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
// the same without the streaming hint: data every workgroup re-reads (a thin output layer's weights)
__device__ __forceinline__ void global_load_lds_b128_cached(const void* gsrc, const void* lds_wave_base) {
  const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(size_t)(__attribute__((address_space(3))) const char*)lds_wave_base);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}
// wait until at most n vector-memory operations of this wave (LDS-DMA included) are outstanding
#define RG_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// s_barrier without the waits __syncthreads() attaches
__device__ __forceinline__ void raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// wave priority for the SIMD's issue arbitration (0..3)
#define RG_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// scheduling-group hints (instruction classes of __builtin_amdgcn_sched_group_barrier)
#define RG_SCHED_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define RG_SCHED_DS_READ(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define RG_SCHED_VMEM_READ(n) __builtin_amdgcn_sched_group_barrier(0x020, n, 0)
#define RG_SCHED_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
#define RG_SCHED_SALU(n) __builtin_amdgcn_sched_group_barrier(0x004, n, 0)

// two floats -> packed bf16x2 (lo in bits 0..15); lowers to one v_cvt_pk_bf16_f32
// one v_cvt_pk_bf16_f32 (a <2 x float> -> <2 x bfloat> truncation; written with scalar casts and
// shifts the vectoriser picks its own pairing and repairs it with v_and/v_lshl/v_or_sdwa)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw;
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_hw));
}

}  // namespace rg

#define RG_LAUNCH(kernel, grid, block, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__)
#define RG_LAUNCH_BOUNDS(t, w) __launch_bounds__(t, w)
// dynamic LDS (kernels that need more than the 64 KB static limit; gfx950 has 160 KB per CU)
#define RG_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define RG_LAUNCH_DYN(kernel, grid, block, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(kernel, grid, block, lds_bytes, stream, __VA_ARGS__)
#define RG_ALLOW_LDS(kernel, bytes) \
  (void)hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

// Register-resident head phase helpers shared by ppo_grad_fast_kernel (ph_ppo_fast.hip) and the tower kernels of the
// ModularAlgorithm path (ph_modular.hip): four lanes per row, each owning 16 hidden units, quad-DPP reductions.
#pragma once
#include "ph_launch.h"

namespace ph {

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float dpp_quad(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the four lanes of a quad; every lane of the quad ends with the bitwise-identical result
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_quad<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp_quad<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}

// hidden units owned by lane q of a row quad, m = 0..15: base(q) + constant(m), so the LDS offsets fold into the
// instructions, and the H2 / dZ2 accesses of a wave (8 rows x 4 quads per 32 lanes) fall on distinct banks.
__device__ __forceinline__ int head_unit(int q, int m) { return 8 * q + (m & 7) + 32 * (m >> 3); }
// act_W row j sits at hw[8*(j + (j>>3))]: the one-row skew per 8 rows puts the four rows a wave reads at a time (j = 8q + c)
// on different banks while keeping 16-byte alignment
__device__ __forceinline__ int head_row(int j) { return 8 * (j + (j >> 3)); }
constexpr int HW_FLOATS = 8 * (HID + HID / 8);
// head_row(head_unit(q, m)) with the lane part and the constant part apart: j = 8q + (m & 7) + 32 (m >> 3) has j >> 3 = q + 4 (m >> 3)
// (0 <= q < 4), so the row sits at 72 q + [8 (m & 7) + 288 (m >> 3)] -- one per-lane base, sixteen immediate offsets.  Written through
// head_row the compiler does not see that and rebuilds every row's address (a multiply-add and a shift-add per row and pass).
__device__ __forceinline__ int head_row_base(int q) { return 72 * q; }
constexpr int head_row_imm(int m) { return 8 * (m & 7) + 288 * (m >> 3); }

// Calls f(m, w0, w1) for the 16 head-weight rows of lane q, four rows per group, the next group's ds_read_b128s issued
// before the current group's arithmetic (the scheduler otherwise emits read-wait-use per row: 16 LDS round trips).
// NK: logits actually used -- with NK <= 4 the second half of every row is never read.
template <int NK = 8, class F>
__device__ __forceinline__ void for_head_rows(const float* hw, int q, F&& f) {
  constexpr int G = 2;  // rows per group
  float4 wa[2][G], wb[2][G];
  const float* hwq = hw + head_row_base(q);
#pragma unroll
  for (int i = 0; i < G; ++i) {
    const float4* w = reinterpret_cast<const float4*>(hwq + head_row_imm(i));
    wa[0][i] = w[0];
    if constexpr (NK > 4) wb[0][i] = w[1];
    else wb[0][i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int g = 0; g < 16 / G; ++g) {
    if (g + 1 < 16 / G) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        const float4* w = reinterpret_cast<const float4*>(hwq + head_row_imm(G * (g + 1) + i));
        wa[(g + 1) & 1][i] = w[0];
        if constexpr (NK > 4) wb[(g + 1) & 1][i] = w[1];
        else wb[(g + 1) & 1][i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G; ++i) f(G * g + i, wa[g & 1][i], wb[g & 1][i]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// sum of 16 LDS values p[i*stride], all reads issued before the adds (fixed tree order)
__device__ __forceinline__ float lds_sum16(const float* p, int stride) {
  float t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = p[i * stride];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int w = 8; w > 0; w >>= 1) {
#pragma unroll
    for (int i = 0; i < w; ++i) t[i] += t[i + w];
  }
  return t[0];
}

}  // namespace ph

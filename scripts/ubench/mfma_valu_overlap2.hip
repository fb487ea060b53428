// Follow-up to mfma_valu_overlap.hip: is the VALU wave starved by ARBITRATION (age / priority) or by a shared datapath?
//  variants: which role is the older wave; s_setprio on the VALU wave; bf16 MFMA instead of f32 MFMA; gaps in the MFMA chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0 = f32 32x32x2 (dependent chain), 1 = bf16 32x32x16 (dependent chain), 2 = f32 with 4 independent accumulators
__device__ __forceinline__ float mfma_role(int iters, int lane) {
  if constexpr (KIND == 0) {
    f32x16 acc = {0};
    float a = 0.5f + lane, b = 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    return acc[0] + acc[7];
  } else if constexpr (KIND == 1) {
    f32x16 acc = {0};
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3f80 + lane); b[j] = 0x3e80; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    return acc[0] + acc[7];
  } else {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = 0.5f + lane, b = 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
      }
    }
    return c0[0] + c1[1] + c2[2] + c3[3];
  }
}
__device__ __forceinline__ float fma_role(int iters, int lane) {
  float v[16], out = 0.f;
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = 0.1f * j + lane;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64; ++r) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_fmaf(v[j], 0.999f, 0.001f);
    }
  }
#pragma unroll
  for (int j = 0; j < 16; ++j) out += v[j];
  return out;
}
__device__ __forceinline__ float lds_role(int iters, int lane, int wave, float* lds) {   // LDS only: no VALU in the loop body
  float out = 0.f;
  for (int i = 0; i < iters; ++i) {
    float h[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) h[j] = lds[(j + 16 * (wave & 3)) * 65 + lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) lds[4160 + (j + 16 * (wave & 3)) * 65 + lane] = h[j];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out += h[3];
  }
  return out;
}

// mfma_first: waves 0-3 run the MFMA role (older), else waves 4-7 do.  valu: 1 = fma, 2 = LDS only.  prio: s_setprio of the VALU wave
template <int KIND>
__global__ __launch_bounds__(512, 2) void k(int mfma_on, int valu, int mfma_first, int prio, int iters, long long* cyc, float* sink) {
  __shared__ float lds[2 * 64 * 65];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 2 * 64 * 65; i += 512) lds[i] = 0.001f * (float)(i & 255);
  __syncthreads();
  const bool is_mfma = (wave >> 2) == (mfma_first ? 0 : 1);
  float out = 0.f;
  long long t0 = clock64();
  if (is_mfma) {
    if (mfma_on) out = mfma_role<KIND>(iters, lane);
  } else {
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    if (valu == 1) out = fma_role(iters, lane);
    if (valu == 2) out = lds_role(iters, lane, wave, lds);
  }
  long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (out == 12345.678f) sink[tid] = out;
}

template <int KIND>
void run(const char* name, int mfma_on, int valu, int mfma_first, int prio, long long* cyc, float* sink) {
  const int iters = 64;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mfma_on, valu, mfma_first, prio, iters, cyc, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int w = 0; w < 256; ++w)
    for (int i = 0; i < 4; ++i) {
      a += h[w * 8 + (mfma_first ? i : 4 + i)];
      b += h[w * 8 + (mfma_first ? 4 + i : i)];
    }
  printf("%-26s mfma=%d other=%s mfma_wave_is_%s prio(other)=%d | per iteration: mfma wave %6.0f, other wave %6.0f cycles\n", name, mfma_on,
         valu == 1 ? "1024fma" : (valu == 2 ? "ldsonly" : "-"), mfma_first ? "older  " : "younger", prio, a / 1024 / iters, b / 1024 / iters);
}

int main() {
  long long* cyc;
  float* sink;
  hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  hipMalloc(&sink, 512 * sizeof(float));
  run<0>("f32 32x32x2 x32 dep", 0, 1, 1, 0, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 0, 2, 1, 0, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 1, 1, 0, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 1, 0, 0, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 1, 1, 1, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 1, 1, 3, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 1, 0, 3, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 2, 1, 0, cyc, sink);
  run<0>("f32 32x32x2 x32 dep", 1, 2, 0, 0, cyc, sink);
  run<2>("f32 32x32x2 x32 4 indep", 1, 0, 1, 0, cyc, sink);
  run<2>("f32 32x32x2 x32 4 indep", 1, 1, 1, 0, cyc, sink);
  run<2>("f32 32x32x2 x32 4 indep", 1, 1, 1, 3, cyc, sink);
  run<1>("bf16 32x32x16 x64 dep", 1, 0, 1, 0, cyc, sink);
  run<1>("bf16 32x32x16 x64 dep", 1, 1, 1, 0, cyc, sink);
  run<1>("bf16 32x32x16 x64 dep", 1, 1, 0, 0, cyc, sink);
  run<1>("bf16 32x32x16 x64 dep", 1, 1, 1, 3, cyc, sink);
  run<1>("bf16 32x32x16 x64 dep", 1, 2, 1, 0, cyc, sink);
  return 0;
}

// v_mfma_f32_16x16x32_bf16 issue cost on one wave: dependent chain on ONE accumulator vs 2 / 4 interleaved accumulators, alone
// and with a second wave on the same SIMD doing pure VALU work (does the bf16 matrix pipe run beside the vector ALU?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(512) void chain(float* out, long long* cyc, int valu_waves) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v0 = threadIdx.x * 0.5f, v1 = 1.0001f, v2 = 0.3f, v3 = 0.7f;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {   // waves 0..3: one per SIMD, MFMA
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
      for (int u = 0; u < 24 / NACC; ++u)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  } else if (wave < 4 + valu_waves) {   // waves 4..7: the second wave of each SIMD, VALU only (4 independent fma chains)
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
      for (int u = 0; u < 24; ++u) {
        v0 = __builtin_fmaf(v0, v1, v2); v2 = __builtin_fmaf(v2, v1, v3); v3 = __builtin_fmaf(v3, v1, v0); v1 = __builtin_fmaf(v1, 0.999f, 1e-6f);
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = v0 + v1 + v2 + v3;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}

template <int NACC>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8);
  for (int vw : {0, 4}) {
    chain<NACC><<<1, 512>>>(out, cyc, vw);
    chain<NACC><<<1, 512>>>(out, cyc, vw);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%s, %d VALU waves beside: MFMA wave %.1f cycles per MFMA (6144 MFMAs)", name, vw, (double)h[0] / 6144);
    if (vw) printf(" | VALU wave %.1f cycles per v_fma (24576)", (double)h[4] / 24576);
    printf("\n");
  }
}
int main() {
  run<1>("1 accumulator (dependent chain)");
  run<2>("2 accumulators interleaved");
  run<4>("4 accumulators interleaved");
  return 0;
}

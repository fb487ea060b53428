// In-wave fillers between v_mfma_f32_16x16x32_bf16 on gfx950: what does a VALU instruction placed between two MFMAs of the SAME
// wave cost, when the two MFMAs are on the SAME accumulator (a six-term product is a dependent chain) and when consecutive MFMAs
// alternate between 2 / 4 accumulators?  One or two waves per SIMD (256 / 512 threads, one workgroup on one CU).
// Every instruction of the timed loop is an asm volatile statement: the order written is the order issued.  build: hipcc --offload-arch=gfx950 -O3 -o mfma16_inwave mfma16_inwave.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { V_FMA = 0, V_PKFMA = 1, V_DOT2C = 2, V_EXP = 3, V_CVT = 4 };

template <int KIND>
__device__ __forceinline__ void filler(float (&v)[8], f32x2 (&w)[4], int j, unsigned sel) {
  float& x = v[j & 7];
  if constexpr (KIND == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(0.999f), "v"(0.001f));
  else if constexpr (KIND == V_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[j & 3]) : "v"(w[(j + 1) & 3]), "v"(w[(j + 2) & 3]));
  else if constexpr (KIND == V_DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(v[(j + 3) & 7]), "v"(sel));
  else if constexpr (KIND == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(v[(j + 3) & 7]), "v"(v[(j + 5) & 7]));
}

// everything in the timed loop is asm volatile: the order written here is the order issued
template <int NACC, int NV, int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, unsigned sel) {
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v[8];
  f32x2 w[4];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.01f * (threadIdx.x + j);
  for (int j = 0; j < 4; ++j) w[j] = (f32x2){0.9f + 0.01f * j, 0.8f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int u = 0; u < 24; ++u) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[u % NACC]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j) filler<KIND>(v, w, u * NV + j, sel);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int j = 0; j < 4; ++j) s += w[j].x + w[j].y;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}

template <int NACC, int NV, int KIND>
void run(const char* kind) {
  static float* out = nullptr;
  static long long* cyc = nullptr;
  if (!out) { hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 8 * 8); }
  for (int threads : {256, 512}) {
    k<NACC, NV, KIND><<<1, threads>>>(out, cyc, 0x0000BF80u);
    k<NACC, NV, KIND><<<1, threads>>>(out, cyc, 0x0000BF80u);
    long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%d acc, %d x %-7s per MFMA, %d wave/SIMD: %6.1f cycles per MFMA (wave 0)%s", NACC, NV, kind, threads / 256, (double)h[0] / 6144,
           threads == 256 ? " | " : "\n");
  }
}
template <int NACC, int KIND>
void sweep(const char* kind) {
  run<NACC, 1, KIND>(kind);
  run<NACC, 2, KIND>(kind);
  run<NACC, 3, KIND>(kind);
  run<NACC, 4, KIND>(kind);
  run<NACC, 6, KIND>(kind);
}
int main() {
  run<1, 0, V_FMA>("none");
  run<2, 0, V_FMA>("none");
  run<4, 0, V_FMA>("none");
  sweep<1, V_FMA>("fma");
  sweep<2, V_FMA>("fma");
  sweep<4, V_FMA>("fma");
  sweep<1, V_PKFMA>("pk_fma");
  sweep<4, V_PKFMA>("pk_fma");
  sweep<1, V_DOT2C>("dot2c");
  sweep<4, V_DOT2C>("dot2c");
  sweep<1, V_EXP>("exp");
  sweep<4, V_EXP>("exp");
  sweep<1, V_CVT>("cvt_pk");
  sweep<4, V_CVT>("cvt_pk");
  return 0;
}

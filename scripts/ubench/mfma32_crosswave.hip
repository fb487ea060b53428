// mfma16_crosswave.hip with the 32x32x16 bf16 instruction (twice the work per instruction: 8 passes instead of 4): waves 0-3 (one per
// SIMD) issue back-to-back v_mfma_f32_32x32x16_bf16 (4 accumulators of 16 registers), waves 4-7 a pure stream of ONE kind of VALU
// instruction.  Question: is what a matrix instruction takes of the SIMD's vector issue port a cost per INSTRUCTION (then half as many,
// twice as large instructions would free the port) or per PASS (then the tile shape does not matter)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { V_FMA = 0, V_PKFMA = 1, V_DOT2C = 2, V_EXP = 3, V_CVT = 4, V_PKADD = 5, V_PKMUL = 6 };

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, unsigned sel, int mfma_waves, int valu_waves) {
  const int wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * (threadIdx.x - e)); }
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float v[8];
  f32x2 w[4];
  for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.01f * (threadIdx.x + j);
  for (int j = 0; j < 4; ++j) w[j] = (f32x2){0.9f + 0.01f * j, 0.8f};
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (wave < 4) {
    if (mfma_waves) {
#pragma unroll 1
      for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u % 4]) : "v"(a), "v"(b));
      }
    }
  } else if (valu_waves) {
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
      for (int j = 0; j < 48; ++j) {
        float& x = v[j & 7];
        if constexpr (KIND == V_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(0.999f), "v"(0.001f));
        else if constexpr (KIND == V_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[j & 3]) : "v"(w[(j + 1) & 3]), "v"(w[(j + 2) & 3]));
        else if constexpr (KIND == V_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(w[j & 3]) : "v"(w[(j + 1) & 3]));
        else if constexpr (KIND == V_PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[j & 3]) : "v"(w[(j + 1) & 3]));
        else if constexpr (KIND == V_DOT2C) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x) : "v"(v[(j + 3) & 7]), "v"(sel));
        else if constexpr (KIND == V_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x) : "v"(v[(j + 3) & 7]), "v"(v[(j + 5) & 7]));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) s += v[j];
  for (int j = 0; j < 4; ++j) s += w[j].x + w[j].y;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[wave] = t1 - t0;
}
template <int KIND>
void run(const char* kind) {
  static float* out = nullptr;
  static long long* cyc = nullptr;
  if (!out) { (void)hipMalloc(&out, 512 * 4); (void)hipMalloc(&cyc, 8 * 8); }
  long long h[8];
  k<KIND><<<1, 512>>>(out, cyc, 0x0000BF80u, 0, 1);
  k<KIND><<<1, 512>>>(out, cyc, 0x0000BF80u, 0, 1);
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  const double alone = (double)h[4] / (256 * 48);
  k<KIND><<<1, 512>>>(out, cyc, 0x0000BF80u, 1, 1);
  k<KIND><<<1, 512>>>(out, cyc, 0x0000BF80u, 1, 1);
  (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-8s alone %5.1f cycles each | beside an MFMA wave: %5.1f cycles each, the MFMA wave %5.1f cycles per 32x32x16 MFMA (= two 16x16x32's work)\n", kind, alone,
         (double)h[4] / (256 * 48), (double)h[0] / 3072);
}
int main() {
  run<V_FMA>("fma");
  run<V_PKFMA>("pk_fma");
  run<V_PKADD>("pk_add");
  run<V_PKMUL>("pk_mul");
  run<V_DOT2C>("dot2c");
  run<V_EXP>("exp");
  run<V_CVT>("cvt_pk");
  return 0;
}

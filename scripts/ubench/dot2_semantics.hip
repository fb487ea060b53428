// what v_dot2c_f32_bf16 computes on gfx950, on known inputs (printed): hipcc --offload-arch=gfx950 -O3 -o /tmp/d2 scripts/ubench/dot2_semantics.hip && /tmp/d2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__device__ inline unsigned cvt_pk_bf16(float a, float b) {
  unsigned r;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ inline float dot2c_bf16(float acc, unsigned a, unsigned b) {
  asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
  return acc;
}
__global__ void k(const float* x, float* out, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  unsigned lo = 0x0000BF80u, hi = 0xBF800000u;
  asm volatile("" : "+v"(lo), "+v"(hi));
  const float x0 = x[2 * i], x1 = x[2 * i + 1];
  const unsigned h = cvt_pk_bf16(x0, x1);
  out[8 * i + 0] = __uint_as_float(h << 16);                 // bf16(x0) widened
  out[8 * i + 1] = __uint_as_float(h & 0xffff0000u);         // bf16(x1) widened
  out[8 * i + 2] = x0 - __uint_as_float(h << 16);            // the subtract form
  out[8 * i + 3] = x1 - __uint_as_float(h & 0xffff0000u);
  out[8 * i + 4] = dot2c_bf16(x0, h, lo);                    // the dot2c form
  out[8 * i + 5] = dot2c_bf16(x1, h, hi);
  out[8 * i + 6] = dot2c_bf16(0.0f, h, lo);                  // -bf16(x0) alone
  out[8 * i + 7] = dot2c_bf16(x0, 0u, lo);                   // x0 + 0
}
int main() {
  const float xs[] = {1.2345678f, -0.4336030f, 1.0f, 3.0f, 1.9254440f, 1e-3f, 257.0f, -1.0000001f, 0.5f + 1.0f / 512, 1.0f + 1.0f / 256,
                      1.0f + 1.0f / 65536, 100.125f};
  const int n = sizeof(xs) / 8;
  float *dx, *dout;
  hipMalloc(&dx, sizeof(xs));
  hipMalloc(&dout, n * 32);
  hipMemcpy(dx, xs, sizeof(xs), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, n);
  float o[8 * 16];
  hipMemcpy(o, dout, n * 32, hipMemcpyDeviceToHost);
  auto hex = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
  for (int i = 0; i < n; ++i) {
    printf("x = (%.9g, %.9g)  bf16 = (%.9g, %.9g)\n   sub : %.9g [%08x]  %.9g [%08x]\n   dot2: %.9g [%08x]  %.9g [%08x]\n   0 - h0 via dot2c: %.9g [%08x];  x0 + 0 via dot2c: %.9g [%08x]\n",
           xs[2 * i], xs[2 * i + 1], o[8 * i], o[8 * i + 1], o[8 * i + 2], hex(o[8 * i + 2]), o[8 * i + 3], hex(o[8 * i + 3]),
           o[8 * i + 4], hex(o[8 * i + 4]), o[8 * i + 5], hex(o[8 * i + 5]), o[8 * i + 6], hex(o[8 * i + 6]), o[8 * i + 7], hex(o[8 * i + 7]));
  }
  return 0;
}

// Does a single busy workgroup (255 of 256 CUs idle) run at full shader clock?  clock64() ticks at the shader clock,
// wall_clock64() at a constant 100 MHz: their ratio over a fixed amount of dependent VALU work is the effective clock.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int iters, long long* out, float* sink) {
  float v = threadIdx.x;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 64; ++j) v = __builtin_fmaf(v, 0.999f, 0.001f);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
  if (v == 123.f) sink[0] = v;
}
int main() {
  long long* out; float* sink; long long h[2];
  hipMalloc(&out, 16); hipMalloc(&sink, 4);
  for (int blocks : {1, 256, 1024}) for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 20000, out, sink);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%4d workgroup(s): %lld shader cycles in %lld ticks of 100 MHz = %.1f us -> %.2f GHz; %.2f cycles per dependent v_fma\n", blocks,
           h[0], h[1], h[1] / 100.0, h[0] / (h[1] * 10.0), (double)h[0] / (20000.0 * 64));
  }
  return 0;
}

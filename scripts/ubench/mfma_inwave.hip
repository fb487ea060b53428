// In-wave overlap: how many independent VALU / LDS instructions of the SAME wave hide under its own dependent f32 MFMA chain?
// One or two waves per SIMD (256- or 512-thread workgroups, one per CU), every wave runs the same mixed stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NL, bool TRANS>
__global__ __launch_bounds__(512, 2) void k(int iters, long long* cyc, float* sink) {
  __shared__ float lds[2 * 64 * 65];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 2 * 64 * 65; i += blockDim.x) lds[i] = 0.001f * (float)(i & 255);
  __syncthreads();
  f32x16 acc = {0};
  float a = 0.5f + lane, b = 0.25f;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = 0.1f * j + lane;
  float l[8] = {0};
  const float* lp = lds + (wave & 3) * 16 * 65 + lane;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        if (TRANS && (j & 3) == 3) v[j & 15] = __builtin_amdgcn_exp2f(v[j & 15]);
        else v[j & 15] = __builtin_fmaf(v[j & 15], 0.999f, 0.001f);
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) l[j & 7] += lp[((u * NL + j) & 15) * 65];
    }
  }
  long long t1 = clock64();
  float out = acc[0] + acc[7];
#pragma unroll
  for (int j = 0; j < 16; ++j) out += v[j];
#pragma unroll
  for (int j = 0; j < 8; ++j) out += l[j];
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (out == 12345.678f) sink[tid] = out;
}

template <int NV, int NL, bool TRANS>
void run(int threads, long long* cyc, float* sink) {
  const int iters = 64;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NV, NL, TRANS>), dim3(256), dim3(threads), 0, 0, iters, cyc, sink);
  hipDeviceSynchronize();
  std::vector<long long> h(256 * 8);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double a = 0;
  const int nw = threads / 64;
  for (int w = 0; w < 256; ++w)
    for (int i = 0; i < nw; ++i) a += h[w * 8 + i];
  printf("%d wave(s)/SIMD: 32 dependent f32 MFMAs, per MFMA %2d VALU%s + %d ds_read_b32 -> %6.0f cycles per 32 MFMAs (%.1f per MFMA)\n", nw / 4, NV,
         TRANS ? " (1/4 v_exp)" : "", NL, a / 256 / nw / iters, a / 256 / nw / iters / 32);
}

int main() {
  long long* cyc;
  float* sink;
  hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  hipMalloc(&sink, 512 * sizeof(float));
  for (int threads : {256, 512}) {
    run<0, 0, false>(threads, cyc, sink);
    run<4, 0, false>(threads, cyc, sink);
    run<8, 0, false>(threads, cyc, sink);
    run<12, 0, false>(threads, cyc, sink);
    run<16, 0, false>(threads, cyc, sink);
    run<24, 0, false>(threads, cyc, sink);
    run<32, 0, false>(threads, cyc, sink);
    run<8, 0, true>(threads, cyc, sink);
    run<16, 0, true>(threads, cyc, sink);
    run<0, 2, false>(threads, cyc, sink);
    run<0, 4, false>(threads, cyc, sink);
    run<8, 2, false>(threads, cyc, sink);
    run<12, 4, false>(threads, cyc, sink);
  }
  return 0;
}

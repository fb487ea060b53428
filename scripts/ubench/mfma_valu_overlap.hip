// Microbenchmark: do an f32-MFMA wave and a VALU / LDS wave on the SAME SIMD overlap on gfx950?
// 512-thread workgroup, one per CU: waves 0-3 (role A) and waves 4-7 (role B) land pairwise on the four SIMDs.
// Each role runs its loop alone and beside the other role; cycles from s_memtime per role.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { A_NONE = 0, A_MFMA32 = 1, A_MFMA32_LDS = 2, A_MFMA16 = 3 };
enum { B_NONE = 0, B_FMA = 1, B_TANH = 2, B_LDSRW = 3, B_MFMA32 = 4 };

__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = __builtin_fabsf(x);
  const float x2 = x * x;
  float p = 62.0f / 2835.0f;
  p = __builtin_fmaf(p, x2, -17.0f / 315.0f);
  p = __builtin_fmaf(p, x2, 2.0f / 15.0f);
  p = __builtin_fmaf(p, x2, -1.0f / 3.0f);
  p = __builtin_fmaf(p * x2, x, x);
  const float e = __builtin_amdgcn_exp2f(ax * (2.0f * 1.44269504088896340736f));
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  return ax < 0.3f ? p : __builtin_copysignf(t, x);
}

__global__ __launch_bounds__(512, 2) void k(int ra, int rb, int iters, long long* cyc, float* sink) {
  __shared__ float lds[2][64 * 65];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, role = wave >> 2;
  for (int i = tid; i < 2 * 64 * 65; i += 512) (&lds[0][0])[i] = 0.001f * (float)(i & 255);
  __syncthreads();
  float out = 0.f;
  long long t0 = clock64();
  if (role == 0) {
    if (ra == A_MFMA32) {
      f32x16 acc = {0};
      float a = 0.5f + lane, b = 0.25f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      out = acc[0] + acc[7];
    } else if (ra == A_MFMA32_LDS) {   // operands from LDS like tile_mma: 64 ds_read_b32 per 32 MFMAs, prefetched 4 ahead
      f32x16 acc = {0};
      const float* ap = &lds[0][(lane & 31) * 65 + (lane >> 5)];
      const float* bp = &lds[1][(lane >> 5) * 65 + (lane & 31)];
      for (int i = 0; i < iters; ++i) {
        float a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0[u] = ap[2 * u]; b0[u] = bp[2 * u * 65]; }
#pragma unroll
        for (int s = 0; s < 64; s += 8) {
          if (s + 8 < 64) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { a1[u] = ap[s + 8 + 2 * u]; b1[u] = bp[(s + 8 + 2 * u) * 65]; }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; ++u) { a0[u] = a1[u]; b0[u] = b1[u]; }
        }
      }
      out = acc[0] + acc[7];
    } else if (ra == A_MFMA16) {   // 16x16x4, two independent accumulators: same FLOPs as 32 x 32x32x2 -> 128 instr
      f32x4 c0 = {0}, c1 = {0};
      float a = 0.5f + lane, b = 0.25f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, c1, 0, 0, 0);
        }
      }
      out = c0[0] + c1[1];
    }
  } else {
    if (rb == B_FMA) {   // 16 independent chains x 64 FMAs per iteration = 1024 VALU ops
      float v[16];
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
    } else if (rb == B_TANH) {   // 16 tanh + 16 ds_write_b32 per iteration (one layer epilogue)
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = 0.01f * j + 0.001f * lane;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          v[j] = fast_tanh(v[j] + 0.3f);
          lds[1][(j + 16 * (wave & 3)) * 65 + lane] = v[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) out += v[j];
    } else if (rb == B_LDSRW) {   // 16 x (ds_read_b32, fma, ds_write_b32) + 16 ds_write (dZ1-in-place + X commit shape)
      for (int i = 0; i < iters; ++i) {
        float h[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) h[j] = lds[1][(j + 16 * (wave & 3)) * 65 + lane];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          lds[1][(j + 16 * (wave & 3)) * 65 + lane] = 0.5f * (1.0f - h[j] * h[j]);
          lds[0][(j + 16 * (wave & 3)) * 65 + lane] = h[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        out += h[3];
      }
    } else if (rb == B_MFMA32) {
      f32x16 acc = {0};
      float a = 0.5f + lane, b = 0.25f;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      out = acc[0] + acc[7];
    }
  }
  long long t1 = clock64();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  if (out == 12345.678f) sink[tid] = out;
}

int main() {
  long long* cyc;
  float* sink;
  hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  hipMalloc(&sink, 512 * sizeof(float));
  const char* an[] = {"-", "mfma32x32x2 x32 (reg operands)", "mfma32x32x2 x32 (LDS operands)", "mfma16x16x4 x128"};
  const char* bn[] = {"-", "1024 fma", "16 tanh + 16 ds_write", "16 ds_read + 32 ds_write", "mfma32x32x2 x32"};
  const int iters = 64;
  int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 1}, {1, 2}, {1, 3}, {2, 1}, {2, 2}, {2, 3}, {3, 1}, {3, 2},
                     {1, 4}, {2, 4}};
  for (auto& c : combos) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], iters, cyc, sink);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, c[0], c[1], iters, cyc, sink);
    hipDeviceSynchronize();
    std::vector<long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int w = 0; w < 256; ++w) {
      for (int i = 0; i < 4; ++i) { a += h[w * 8 + i]; b += h[w * 8 + 4 + i]; }
    }
    printf("A: %-32s B: %-26s | per iteration: A %7.0f cycles, B %7.0f cycles\n", an[c[0]], bn[c[1]], a / 1024 / iters, b / 1024 / iters);
  }
  return 0;
}

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 llvm_bf16x4_t;
#define LDSA __attribute__((address_space(3)))

__global__ void probe_tr(uint16_t* out, int stride_bytes) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  // lane t of a 16-lane group: block row t>>2, 8-byte piece t&3; groups at +4 rows
  const int t = l & 15, g = l >> 4;
  const int byte = (g * 4 + (t >> 2)) * stride_bytes + (t & 3) * 8;
  auto p = reinterpret_cast<LDSA llvm_bf16x4_t*>((LDSA char*)lds + byte);
  llvm_bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
  uint16_t u[4];
  __builtin_memcpy(u, &v, 8);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = u[j];
}

__device__ inline void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  float r1 = x - (float)h;
  m = (__bf16)r1;
  float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// C[64][64] = A[64][64] * B[64][64] (B given as Bt[n][k]); one wave computes a 16x16 block; 16 waves
__global__ void gemm_split(const float* A, const float* Bt, float* C, int terms) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mb = w >> 2, nb = w & 3;
  const int i = l & 15, kg = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int c = 0; c < 2; ++c) {
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
      const int k = c * 32 + kg * 8 + e;
      __bf16 h, m, lo;
      split3(A[(mb * 16 + i) * 64 + k], h, m, lo);
      a[0][e] = h; a[1][e] = m; a[2][e] = lo;
      split3(Bt[(nb * 16 + i) * 64 + k], h, m, lo);
      b[0][e] = h; b[1][e] = m; b[2][e] = lo;
    }
    // small terms first
    if (terms >= 9) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[2], acc, 0, 0, 0);
    if (terms >= 9) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[2], acc, 0, 0, 0);
    if (terms >= 9) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[1], acc, 0, 0, 0);
    if (terms >= 6) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
    if (terms >= 6) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
    if (terms >= 6) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
    if (terms >= 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
    if (terms >= 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; ++r) C[(mb * 16 + 4 * kg + r) * 64 + nb * 16 + i] = acc[r];
}
__global__ void gemm_f32mfma(const float* A, const float* Bt, float* C) {
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int mb = w >> 2, nb = w & 3;
  const int i = l & 15, kg = l >> 4;
  f32x4 acc = {0, 0, 0, 0};
  for (int k = 0; k < 64; k += 4)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(mb * 16 + i) * 64 + k + kg], Bt[(nb * 16 + i) * 64 + k + kg], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[(mb * 16 + 4 * kg + r) * 64 + nb * 16 + i] = acc[r];
}

int main() {
  uint16_t* d; hipMalloc(&d, 512);
  for (int stride : {32, 64, 136}) {
    probe_tr<<<1, 64>>>(d, stride);
    uint16_t h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("tr probe, row stride %d bytes (element index = byte/2): lane: 4 values\n", stride);
    for (int l = 0; l < 64; ++l) { printf("%2d:[%4d %4d %4d %4d] ", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
  }
  std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> A(4096), Bt(4096), C(4096);
  for (int scen = 0; scen < 2; ++scen) {
    for (auto& v : A) v = scen == 0 ? nd(rng) : std::tanh(nd(rng));
    for (auto& v : Bt) v = scen == 0 ? nd(rng) : 0.2f * nd(rng);
    float *dA, *dB, *dC; hipMalloc(&dA, 16384); hipMalloc(&dB, 16384); hipMalloc(&dC, 16384);
    hipMemcpy(dA, A.data(), 16384, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), 16384, hipMemcpyHostToDevice);
    std::vector<double> ref(4096); std::vector<float> f32c(4096);
    double scale = 0;
    for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) {
      double s = 0, sa = 0; float f = 0.f;
      for (int k = 0; k < 64; ++k) { s += (double)A[i*64+k] * Bt[j*64+k]; sa += std::fabs((double)A[i*64+k] * Bt[j*64+k]); f = std::fmaf(A[i*64+k], Bt[j*64+k], f); }
      ref[i*64+j] = s; f32c[i*64+j] = f; scale += sa;
    }
    scale /= 4096;
    auto report = [&](const char* name, const float* c) {
      double mx = 0, rms = 0;
      for (int q = 0; q < 4096; ++q) { double e = std::fabs(c[q] - ref[q]); mx = std::fmax(mx, e); rms += e * e; }
      printf("scen %d %-18s max err %.3e rms %.3e  (mean sum|a*b| %.3e -> max/scale %.3e)\n", scen, name, mx, std::sqrt(rms / 4096), scale, mx / scale);
    };
    report("host fmaf chain", f32c.data());
    gemm_f32mfma<<<1, 1024>>>(dA, dB, dC); hipMemcpy(C.data(), dC, 16384, hipMemcpyDeviceToHost); report("f32 mfma 16x16x4", C.data());
    for (int terms : {1, 3, 6, 9}) {
      gemm_split<<<1, 1024>>>(dA, dB, dC, terms); hipMemcpy(C.data(), dC, 16384, hipMemcpyDeviceToHost);
      char nm[32]; snprintf(nm, 32, "split bf16 %d-term", terms); report(nm, C.data());
    }
  }
  return 0;
}

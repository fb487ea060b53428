// Is the residual of the three-plane split computable with v_dot2c_f32_bf16 instead of (expand bf16 -> f32, v_sub_f32)?
//   r = x - (float)bf16(x)   vs   r' = dot2c(acc = x, {h, h'}, {-1, 0})
// Bitwise comparison of all three planes over 2^26 values per scale (normal, tiny, huge, tanh outputs, exact bf16s, denormal residues).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_dot2_probe scripts/ubench/split_dot2_probe.hip && /tmp/split_dot2_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ inline void split_ref(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const __bf16 hb = (__bf16)x;
  const float r1 = x - (float)hb;
  const __bf16 mb = (__bf16)r1;
  const float r2 = r1 - (float)mb;
  const __bf16 lb = (__bf16)r2;
  h = __builtin_bit_cast(unsigned short, hb);
  m = __builtin_bit_cast(unsigned short, mb);
  l = __builtin_bit_cast(unsigned short, lb);
}
// two values at a time: one v_cvt_pk_bf16_f32 per plane, one v_dot2c per residual
__device__ inline void split_dot2(float x0, float x1, bf16x2& h, bf16x2& m, bf16x2& l) {
  bf16x2 lo, hi;
  lo[0] = (__bf16)-1.0f; lo[1] = (__bf16)0.0f;
  hi[0] = (__bf16)0.0f;  hi[1] = (__bf16)-1.0f;
  h[0] = (__bf16)x0; h[1] = (__bf16)x1;
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(h, lo, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(h, hi, x1, false);
  m[0] = (__bf16)r0; m[1] = (__bf16)r1;
  const float s0 = __builtin_amdgcn_fdot2_f32_bf16(m, lo, r0, false), s1 = __builtin_amdgcn_fdot2_f32_bf16(m, hi, r1, false);
  l[0] = (__bf16)s0; l[1] = (__bf16)s1;
}

__device__ inline uint32_t mix(uint32_t z) {
  z ^= z >> 16; z *= 0x7feb352du; z ^= z >> 15; z *= 0x846ca68bu; z ^= z >> 16;
  return z;
}

__global__ void probe(int mode, unsigned long long* bad, float* first_bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float x[2];
  for (int k = 0; k < 2; ++k) {
    const uint32_t u = mix(2 * i + k + 0x9e3779b9u * (mode + 1));
    float v;
    if (mode == 0) v = __uint_as_float((u & 0x807fffffu) | ((120u + (u >> 23 & 7u)) << 23));          // |x| in [2^-7, 2)
    else if (mode == 1) v = __uint_as_float(u & 0xbfffffffu);                                         // any exponent below 2 (incl. denormals)
    else if (mode == 2) v = __uint_as_float((u & 0x807fffffu) | ((200u + (u >> 23 & 31u)) << 23));     // huge
    else if (mode == 3) v = tanhf(__uint_as_float((u & 0x807fffffu) | ((122u + (u >> 23 & 7u)) << 23)));   // tanh outputs
    else if (mode == 4) v = __uint_as_float(u & 0xffff0000u & 0xbfffffffu);                            // exact bf16 values
    else v = __uint_as_float((u & 0x807fffffu) | ((1u + (u >> 23 & 31u)) << 23));                      // tiny: residues go denormal
    if (!(v == v) || __builtin_isinf(v)) v = 1.0f;
    x[k] = v;
  }
  bf16x2 h, m, l;
  split_dot2(x[0], x[1], h, m, l);
  for (int k = 0; k < 2; ++k) {
    unsigned short rh, rm, rl;
    split_ref(x[k], rh, rm, rl);
    const bool same = rh == __builtin_bit_cast(unsigned short, h[k]) && rm == __builtin_bit_cast(unsigned short, m[k]) &&
                      rl == __builtin_bit_cast(unsigned short, l[k]);
    if (!same && atomicAdd(bad, 1ull) == 0ull) *first_bad = x[k];
  }
}

int main() {
  unsigned long long* bad;
  float* fb;
  hipMalloc(&bad, 8);
  hipMalloc(&fb, 4);
  const char* names[6] = {"[2^-7, 2)", "any exponent < 2", "huge", "tanh outputs", "exact bf16", "tiny (denormal residues)"};
  int rc = 0;
  for (int mode = 0; mode < 6; ++mode) {
    hipMemset(bad, 0, 8);
    hipMemset(fb, 0, 4);
    hipLaunchKernelGGL(probe, dim3(1 << 17), dim3(256), 0, 0, mode, bad, fb);
    unsigned long long n = 0;
    float f = 0;
    hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&f, fb, 4, hipMemcpyDeviceToHost);
    printf("mode %d %-26s: %llu of %u values differ from the v_sub_f32 split (first: %.9g)\n", mode, names[mode], n, 1u << 26, f);
    if (n && mode != 5 && mode != 1) rc = 1;
  }
  return rc;
}

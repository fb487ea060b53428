// Three-plane bf16 representation of float32 operands (ppo_grad_split_kernel, ph_ppo_split.hip) and the pre-split weight
// fragment image the optimizer step keeps up to date for it.
#pragma once
#include "ph_launch.h"

namespace ph {

// x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): both residues are exact in float32, and three 8-bit
// significands (round-to-nearest, signed residues) carry all 24 bits of x
__device__ __forceinline__ void split1(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  l = (__bf16)r2;
}

// Weight fragment image: the MFMA B fragments of every wave of ppo_grad_split_kernel, already split --
//   [net 2][wave 4][set 3: W1 by (feature, unit) | W2 by (input, unit) | W2 by (unit, output)][chunk 2][plane 3][lane 64][8] bf16
// so a wave's prologue is eighteen coalesced 16-byte loads and no arithmetic.  A parameter backs at most two elements
// (W2 sits in two sets); `map` (P x 2 ints, -1 = none) holds the element index in plane 0, planes are WIMG_PLANE apart.
// Elements no parameter backs (features >= F of W1) stay zero.  Maintained by ppo_adam_kernel after every step; rebuilt from the
// parameters at the start of every train() / gradient call (weight_image_kernel).
// the image positions (p0, p1 -- either may be -1) of a parameter already in hand
__device__ __forceinline__ void wimage_put_at(unsigned short* image, int p0, int p1, float x) {
  if (p0 < 0 && p1 < 0) return;
  __bf16 h, m, l;
  split1(x, h, m, l);
  const unsigned short hb = __builtin_bit_cast(unsigned short, h), mb = __builtin_bit_cast(unsigned short, m),
                       lb = __builtin_bit_cast(unsigned short, l);
  if (p0 >= 0) {
    image[p0] = hb;
    image[p0 + WIMG_PLANE] = mb;
    image[p0 + 2 * WIMG_PLANE] = lb;
  }
  if (p1 >= 0) {
    image[p1] = hb;
    image[p1 + WIMG_PLANE] = mb;
    image[p1 + 2 * WIMG_PLANE] = lb;
  }
}
__device__ __forceinline__ void wimage_put(unsigned short* image, const int* map, int p, float x) {
  wimage_put_at(image, map[2 * p], map[2 * p + 1], x);
}

}  // namespace ph

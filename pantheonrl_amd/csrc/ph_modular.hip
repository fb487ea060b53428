// ModularAlgorithm / ModularPolicy on the device (reference pantheonrl/algos/modular/policies.py:243-395 and
// pantheonrl/algos/modular/learn.py:221-351).
//
// ModularPolicy is a DAG of 64-64 tanh towers: the main network's policy and value towers read the observation; every
// partner module owns a policy tower and a value tower that BOTH read the main POLICY latent (policies.py:254,281); logits and
// values are sums of the main head and the partner head (policies.py:286,325-328).  One minibatch of ModularAlgorithm.train
// (learn.py:244-326) therefore decomposes into
//   forward :  main {pi, vf} towers  ->  every module's pi tower (the marginal regulariser needs all partners' logits,
//              learn.py:306) and the trained partner's vf tower, all on the main policy latent
//   loss    :  one lane per row: clipped surrogate + value + entropy terms on the composed heads, the regulariser
//              mean_rows sum_a | softmax(z_main) - mean_j softmax(z_main + z_j) |  (learn.py:311-315), dL/d(every head output)
//   backward:  every module's pi tower (and the trained partner's vf tower) with dL/dX = the latent's gradient written out,
//              then the main towers, the policy one with those latent gradients added to its own head's
//   step    :  the towers' gradient slabs summed in a fixed order, ONE global-norm clip over the main network and every
//              module, Adam with torch's per-parameter step counts (a module's value side only starts counting when its
//              partner is first trained -- optimizer.zero_grad() of torch 1.13 leaves zero gradients behind, not None).
// The tower kernel below is ppo_grad_fast_kernel's tile walk (same 32x32 MFMA tiles on LDS operands, same register-order
// gradient slabs, same register-resident four-lanes-per-row head phase) with the loss taken out and two things added: the head
// gradient comes from HBM (written by the loss kernel) plus up to two external dL/dH2 terms, and dL/dX = dZ1 W1^T is one more
// product.  LDS budget and buffer rotation are the fast kernel's (72.6 KB, two workgroups per CU; X and W1 are fetched a second
// time per backward tile under the tile's products); it has no cross-tile prefetch.
#include "ph_head.h"
#include "ph_rowtail.h"

namespace ph {

// X tile of a tower: rows of the rollout buffer gathered through the minibatch order (main towers; Box or one-hot
// observations of at most 64 features) or rows gi of a dense [nb][64] matrix (module towers reading the main policy latent).
// Box rows: issue (16 loads per thread, no waits) and commit (LDS stores) are separate so that the second staging of a
// backward tile (X returns to the buffer H2 occupied) rides under the tile's MFMA products.
struct TowerX {
  float v[16];
  __device__ __forceinline__ void issue(const int* rowphys, const TowerArgs& a, int tid) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i, r = e >> 6, c = e & 63;
      const int p = rowphys[r];
      v[i] = (p >= 0 && c < a.F) ? a.x[(size_t)p * a.x_ld + c] : 0.f;
    }
  }
  __device__ __forceinline__ void commit(float* bufX, int tid) const {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int e = tid + 256 * i;
      bufX[(e >> 6) * LDH + (e & 63)] = v[i];
    }
  }
};
// Discrete / MultiDiscrete observations: one-hot features (SB3 preprocess_obs), out-of-range values clamped.  All threads call
// it (one workgroup barrier inside); the caller barriers afterwards.
__device__ __forceinline__ void tower_stage_onehot(float* bufX, const int* rowphys, const TowerArgs& a, int tid) {
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + 256 * i;
    bufX[(e >> 6) * LDH + (e & 63)] = 0.f;
  }
  __syncthreads();
  for (int e = tid; e < 64 * a.x_ld; e += 256) {
    const int r = e / a.x_ld, comp = e - r * a.x_ld;
    const int p = rowphys[r];
    if (p < 0) continue;
    const int lo = a.obs_off[comp], n = a.obs_off[comp + 1] - lo;
    int v = (int)a.x[(size_t)p * a.x_ld + comp];
    v = v < 0 ? 0 : (v >= n ? n - 1 : v);
    if (lo + v < 64) bufX[r * LDH + lo + v] = 1.f;
  }
}

// LDS: bufA / bufB / bufC / W2 (4 x [64][65] f32) + head weights + per-row head gradients = 72.6 KB -> two workgroups per CU,
// the fast gradient kernel's budget: X -> H2 -> X share bufA, H1 -> dZ1 bufB, W1 -> dZ2 -> W1 bufC.
template <bool VALU, bool BWD>
__global__ __launch_bounds__(256, 2) void tower_kernel(TowerLaunch L) {
  if (L.stop_flag && *L.stop_flag) return;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 64;
  const TowerArgs& a = L.t[blockIdx.y];
  constexpr bool bwd = BWD;
  float* bufA = smem;                 // [R][LDH]   X -> H2 -> X
  float* bufB = bufA + R * LDH;       // [R][LDH]   H1 -> dZ1
  float* bufC = bufB + R * LDH;       // [64][LDH]  W1 -> dZ2 -> W1
  float* w2s = bufC + HID * LDH;      // [64][LDH]
  float* hw = w2s + HID * LDH;        // policy: head weights as [64][8] (skewed rows, columns >= L zero) | value: [64]
  float* dzs = hw + HW_FLOATS;        // policy: dL/dlogits [R][8] | value: dL/dv [R]
  float* b1s = dzs + R * 8;           // [64]
  float* b2s = b1s + HID;             // [64]
  float* hbs = b2s + HID;             // head bias [8] | [1]
  int* rowphys = (int*)(hbs + 16);    // [R] source row of X, -1 = padding
  int* rowgi = rowphys + R;           // [R] minibatch row, -1 = padding

  const bool pol = a.head == 1;
  const bool onehot = a.obs_off != nullptr;
  const int nk = a.L;

  // ---- weights that stay for the whole launch (W1 too in forward mode: nothing overwrites bufC there) ----
  {
    const int tid = threadIdx.x;
    WStage<256> w2r, w1r;
    w2r.issue(a.W2, 0, HID);
    if (!bwd) w1r.issue(a.W1, 0, a.F);
    float bias1 = 0.f, bias2 = 0.f, hv0 = 0.f, hv1 = 0.f, hb = 0.f;
    if (tid < HID) {
      bias1 = a.b1[tid];
      bias2 = a.b2[tid];
    }
    if (pol) {
      const int j0 = tid >> 3, k = tid & 7;
      if (k < nk) {
        hv0 = a.hW[j0 * nk + k];
        hv1 = a.hW[(j0 + 32) * nk + k];
      }
      if (tid < 8 && tid < nk) hb = a.hb[tid];
    } else {
      if (tid < HID) hv0 = a.hW[tid];
      if (tid == 0) hb = a.hb[0];
    }
    w2r.commit(w2s);
    if (!bwd) w1r.commit(bufC);
    if (tid < HID) {
      b1s[tid] = bias1;
      b2s[tid] = bias2;
    }
    if (pol) {
      hw[head_row(tid >> 3) + (tid & 7)] = hv0;
      hw[head_row((tid >> 3) + 32) + (tid & 7)] = hv1;
      if (tid < 8) hbs[tid] = hb;
    } else {
      if (tid < HID) hw[tid] = hv0;
      if (tid == 0) hbs[0] = hb;
    }
  }

  f32x16 gW1 = {0}, gW2 = {0};
  float gh0 = 0.f, gh1 = 0.f, gb1 = 0.f, gb2 = 0.f, ghb = 0.f;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    // thread coordinates re-derived per tile from an opaque copy of threadIdx.x (keeps the compiler from pinning hundreds of
    // loop-invariant LDS addresses in VGPRs across the tile loop: 509 -> < 256 registers)
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63, wave = tid >> 6;
    const int mt = wave >> 1, nt = wave & 1, li = lane & 31, lh = lane >> 5;

    // ---- rows of this tile, X, W1 (bufC is overwritten by dZ2 in every backward tile), the head gradient ----
    __syncthreads();   // the previous tile is done with every buffer
    if (tid < R) {
      const int gi = tile * R + tid;
      int p = -1;
      if (gi < a.nb) p = a.idx ? env_major_to_phys(a.idx[gi], a.T, a.E) : gi;
      rowphys[tid] = p;
      rowgi[tid] = gi < a.nb ? gi : -1;
    }
    WStage<256> w1r;
    if (bwd) w1r.issue(a.W1, 0, a.F, tid);
    __syncthreads();
    TowerX xr;
    if (onehot) {
      tower_stage_onehot(bufA, rowphys, a, tid);
    } else {
      xr.issue(rowphys, a, tid);
      xr.commit(bufA, tid);
    }
    if (bwd) {
      if (pol) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int e = tid + 256 * i, r = e >> 3;
          dzs[e] = rowgi[r] >= 0 ? a.dhead[(size_t)rowgi[r] * 8 + (e & 7)] : 0.f;
        }
      } else if (tid < R) {
        dzs[tid] = rowgi[tid] >= 0 ? a.dhead[rowgi[tid]] : 0.f;
      }
      w1r.commit(bufC, tid);
    }
    lds_barrier();

    // ---- S1: H1 = tanh(X W1 + b1) -> bufB ----
    {
      f32x16 acc = {0};
      acc = tile_mma<false, false, VALU>(bufA, LDH, bufC, LDH, mt * 32, nt * 32, 0, HID, acc, lane);
      const int col = nt * 32 + li;
      const float bb = b1s[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(mt * 32 + drow(r, lh)) * LDH + col] = fast_tanh(acc[r] + bb);
    }
    lds_barrier();
    // ---- S2: H2 = tanh(H1 W2 + b2) -> bufA (X is dead) ----
    {
      f32x16 acc = {0};
      acc = tile_mma<false, false, VALU>(bufB, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, acc, lane);
      const int col = nt * 32 + li;
      const float bb = b2s[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) bufA[(mt * 32 + drow(r, lh)) * LDH + col] = fast_tanh(acc[r] + bb);
    }
    lds_barrier();

    const int r = tid >> 2, q = tid & 3;
    const int gi = rowgi[r];
    float h[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) h[m] = bufA[r * LDH + head_unit(q, m)];

    if constexpr (!bwd) {
      // ---- forward: head output (logits incl. bias / value incl. bias), latent ----
      if (pol) {
        float z[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = 0.f;
        for_head_rows(hw, q, [&](int m, const float4& w0, const float4& w1) {
          z[0] = __builtin_fmaf(h[m], w0.x, z[0]);
          z[1] = __builtin_fmaf(h[m], w0.y, z[1]);
          z[2] = __builtin_fmaf(h[m], w0.z, z[2]);
          z[3] = __builtin_fmaf(h[m], w0.w, z[3]);
          z[4] = __builtin_fmaf(h[m], w1.x, z[4]);
          z[5] = __builtin_fmaf(h[m], w1.y, z[5]);
          z[6] = __builtin_fmaf(h[m], w1.z, z[6]);
          z[7] = __builtin_fmaf(h[m], w1.w, z[7]);
        });
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = (k < nk) ? quad_sum(z[k]) + hbs[k] : 0.f;
        if (q == 0 && gi >= 0 && a.head_out) {
          float4* o = reinterpret_cast<float4*>(a.head_out + (size_t)gi * 8);
          o[0] = make_float4(z[0], z[1], z[2], z[3]);
          o[1] = make_float4(z[4], z[5], z[6], z[7]);
        }
      } else {
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) v = __builtin_fmaf(h[m], hw[head_unit(q, m)], v);
        v = quad_sum(v) + hbs[0];
        if (q == 0 && gi >= 0 && a.head_out) a.head_out[gi] = v;
      }
      if (a.h2_out) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = tid + 256 * i, rr = e >> 6, c = e & 63;
          if (rowgi[rr] >= 0) a.h2_out[(size_t)rowgi[rr] * 64 + c] = bufA[rr * LDH + c];
        }
      }
      continue;
    } else {

    // ---- SH': dH2 = dhead . headW^T (+ external terms) ; dZ2 = dH2 * (1 - H2^2) -> bufC (W1 is dead) ----
    {
      float ext[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) ext[m] = 0.f;
      if (gi >= 0) {
        if (a.ext0) {
#pragma unroll
          for (int m = 0; m < 16; ++m) ext[m] = a.ext0[(size_t)gi * 64 + head_unit(q, m)];
        }
        if (a.ext1) {
#pragma unroll
          for (int m = 0; m < 16; ++m) ext[m] += a.ext1[(size_t)gi * 64 + head_unit(q, m)];
        }
      }
      if (pol) {
        float dz[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dz[k] = dzs[r * 8 + k];
        for_head_rows(hw, q, [&](int m, const float4& w0, const float4& w1) {
          float d = dz[0] * w0.x;
          d = __builtin_fmaf(dz[1], w0.y, d);
          d = __builtin_fmaf(dz[2], w0.z, d);
          d = __builtin_fmaf(dz[3], w0.w, d);
          d = __builtin_fmaf(dz[4], w1.x, d);
          d = __builtin_fmaf(dz[5], w1.y, d);
          d = __builtin_fmaf(dz[6], w1.z, d);
          d = __builtin_fmaf(dz[7], w1.w, d);
          bufC[r * LDH + head_unit(q, m)] = (d + ext[m]) * (1.0f - h[m] * h[m]);
        });
      } else {
        const float dv = dzs[r];
#pragma unroll
        for (int m = 0; m < 16; ++m)
          bufC[r * LDH + head_unit(q, m)] = (dv * hw[head_unit(q, m)] + ext[m]) * (1.0f - h[m] * h[m]);
      }
    }
    lds_barrier();

    // ---- S6a: d b2, head gradients, dW2 += H1^T dZ2, dH1 = dZ2 W2^T; X and W1 are fetched again underneath ----
    f32x16 dh1 = {0};
    if (!onehot) xr.issue(rowphys, a, tid);      // back into bufA for dW1 (committed once H2 is consumed)
    w1r.issue(a.W1, 0, a.F, tid);                 // back into bufC for dL/dX (committed once dZ2 is consumed)
    {
      gb2 += lds_sum16(bufC + wave * 16 * LDH + lane, LDH);
      if (pol) {
        const float* hp = bufA + lane;
        const float* dp = dzs + 2 * wave;
#pragma unroll 1
        for (int r0 = 0; r0 < R; r0 += 8, hp += 8 * LDH, dp += 8 * 8) {   // a real loop: 8 rows of reads, then their FMAs
          float hv[8];
          float2 d[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            hv[i] = hp[i * LDH];
            d[i] = *reinterpret_cast<const float2*>(dp + i * 8);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            gh0 = __builtin_fmaf(hv[i], d[i].x, gh0);
            gh1 = __builtin_fmaf(hv[i], d[i].y, gh1);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < 8) ghb += lds_sum16(dzs + wave * 16 * 8 + lane, 8);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float dv = dzs[wave * 16 + i];
          gh0 = __builtin_fmaf(bufA[(wave * 16 + i) * LDH + lane], dv, gh0);
          ghb += dv;
        }
      }
      gW2 = tile_mma<true, false, VALU>(bufB, LDH, bufC, LDH, mt * 32, nt * 32, 0, R, gW2, lane);
      dh1 = tile_mma<false, true, VALU>(bufC, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, dh1, lane);
    }
    lds_barrier();
    // ---- S6b: dZ1 = dH1 * (1 - H1^2) in place; X back into bufA; W1 back into bufC ----
    {
      float hv[16];
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) hv[rr] = bufB[(mt * 32 + drow(rr, lh)) * LDH + nt * 32 + li];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) bufB[(mt * 32 + drow(rr, lh)) * LDH + nt * 32 + li] = dh1[rr] * (1.0f - hv[rr] * hv[rr]);
    }
    if (onehot) tower_stage_onehot(bufA, rowphys, a, tid);
    else xr.commit(bufA, tid);
    w1r.commit(bufC, tid);
    lds_barrier();
    // ---- S7: d b1, dW1 += X^T dZ1, dL/dX = dZ1 W1^T ----
    gb1 += lds_sum16(bufB + wave * 16 * LDH + lane, LDH);
    gW1 = tile_mma<true, false, VALU>(bufA, LDH, bufB, LDH, mt * 32, nt * 32, 0, R, gW1, lane);
    if (a.dx_out) {
      f32x16 dx = {0};
      dx = tile_mma<false, true, VALU>(bufB, LDH, bufC, LDH, mt * 32, nt * 32, 0, HID, dx, lane);
      // rows of a tile are consecutive minibatch rows (tile * 64 + row): one base pointer, constant row offsets
      float* p0 = a.dx_out + ((size_t)tile * R + mt * 32 + 4 * lh) * 64 + nt * 32 + li;
      const int row_lim = a.nb - (tile * R + mt * 32 + 4 * lh);    // rows below this offset exist
      if (a.dx_accumulate) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int ro = (rr & 3) + 8 * (rr >> 2);
          if (ro < row_lim) p0[ro * 64] += dx[rr];
        }
      } else {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
          const int ro = (rr & 3) + 8 * (rr >> 2);
          if (ro < row_lim) p0[ro * 64] = dx[rr];
        }
      }
    }
    }
  }
  if constexpr (!bwd) return;

  // ---- epilogue: accumulators -> this tower's slab (the fast gradient kernel's register order), fixed-order cross-wave sums ----
  __syncthreads();
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* rslab = a.slab + (size_t)blockIdx.x * a.slab_stride;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int o = ((wave * 4 + r4) * 64 + lane) * 4;
      *reinterpret_cast<float4*>(rslab + RS_W2 + o) = make_float4(gW2[4 * r4], gW2[4 * r4 + 1], gW2[4 * r4 + 2], gW2[4 * r4 + 3]);
      *reinterpret_cast<float4*>(rslab + RS_W1 + o) = make_float4(gW1[4 * r4], gW1[4 * r4 + 1], gW1[4 * r4 + 2], gW1[4 * r4 + 3]);
    }
    if (pol) *reinterpret_cast<float2*>(rslab + RS_HW + lane * 8 + 2 * wave) = make_float2(gh0, gh1);
    float* part = bufA;  // [4][4 waves][64]
    part[(0 * 4 + wave) * 64 + lane] = gb1;
    part[(1 * 4 + wave) * 64 + lane] = gb2;
    part[(2 * 4 + wave) * 64 + lane] = gh0;
    part[(3 * 4 + wave) * 64 + lane] = ghb;
    __syncthreads();
    auto wsum = [&](int which, int i) {
      return ((part[(which * 4 + 0) * 64 + i] + part[(which * 4 + 1) * 64 + i]) + part[(which * 4 + 2) * 64 + i]) +
             part[(which * 4 + 3) * 64 + i];
    };
    if (tid < HID) {
      rslab[RS_B1 + tid] = wsum(0, tid);
      rslab[RS_B2 + tid] = wsum(1, tid);
      if (!pol) rslab[RS_HW + tid] = wsum(2, tid);
    }
    if (pol && tid < 8) rslab[RS_HB + tid] = wsum(3, tid);
    if (!pol && tid == 0) rslab[RS_HB] = wsum(3, 0);
  }
}

size_t tower_lds_bytes() { return sizeof(float) * (size_t)(4 * 64 * LDH + HW_FLOATS + 64 * 8 + 2 * HID + 16 + 2 * 64); }

template <bool VALU, bool BWD>
static hipError_t launch_tower_inst(const TowerLaunch& L, int nwg, int n_towers, hipStream_t s) {
  const size_t lds = tower_lds_bytes();
  static bool allowed[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (!allowed[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)tower_kernel<VALU, BWD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed[dev] = true;
  }
  hipLaunchKernelGGL((tower_kernel<VALU, BWD>), dim3(nwg, n_towers), dim3(256), lds, s, L);
  return hipGetLastError();
}
hipError_t launch_tower(const TowerLaunch& L, int nwg, int n_towers, int gemm_mode, hipStream_t s) {
  if (L.mode != 0)
    return gemm_mode == 1 ? launch_tower_inst<true, true>(L, nwg, n_towers, s) : launch_tower_inst<false, true>(L, nwg, n_towers, s);
  return gemm_mode == 1 ? launch_tower_inst<true, false>(L, nwg, n_towers, s) : launch_tower_inst<false, false>(L, nwg, n_towers, s);
}

// slab position -> parameter index (-1 = padding) of ONE tower's register-order slab (RS_NET floats): the single-net form of
// grad_slab_map with the tower's own offsets into the flat parameter vector
void tower_slab_map(int F, int L, int head, int oW1, int oB1, int oW2, int oB2, int oHW, int oHB, int* m) {
  for (int i = 0; i < RS_NET; ++i) m[i] = -1;
  for (int s = 0; s < HID * HID; ++s) {
    const int wave = s >> 10, r4 = (s >> 8) & 3, lane = (s >> 2) & 63, j = s & 3;
    const int r = 4 * r4 + j, mt = wave >> 1, nt = wave & 1;
    const int k = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = nt * 32 + (lane & 31);
    m[RS_W2 + s] = oW2 + k * HID + col;
    if (k < F) m[RS_W1 + s] = oW1 + k * HID + col;
  }
  for (int i = 0; i < HID; ++i) {
    m[RS_B1 + i] = oB1 + i;
    m[RS_B2 + i] = oB2 + i;
  }
  if (head == 1) {
    for (int j = 0; j < HID; ++j)
      for (int k = 0; k < L && k < 8; ++k) m[RS_HW + j * 8 + k] = oHW + j * L + k;
    for (int k = 0; k < L && k < 8; ++k) m[RS_HB + k] = oHB + k;
  } else {
    for (int j = 0; j < HID; ++j) m[RS_HW + j] = oHW + j;
    m[RS_HB] = oHB;
  }
}

// ---- the loss of one minibatch on the composed heads (learn.py:244-318), one lane per row ---------------------------------
__device__ __forceinline__ void softmax8(const float (&z)[8], int nk, float (&p)[8], float& lse) {
  float mx = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 8; ++k) mx = (k < nk) ? fmaxf(mx, z[k]) : mx;
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    p[k] = (k < nk) ? fast_exp(z[k] - mx) : 0.f;
    se += p[k];
  }
  lse = mx + fast_log(se);
  const float inv = __builtin_amdgcn_rcpf(se);
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] *= inv;
}

__global__ __launch_bounds__(256) void modular_loss_kernel(ModLossArgs a) {
  if (a.stop_flag && *a.stop_flag) return;
  __shared__ float red[4][NSTATP];
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  const int nk = a.L;
  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;
  if (gi < a.nb) {
    const int phys = env_major_to_phys(a.idx[gi], a.T, a.E);
    float zm[8], zk[8];
    {
      const float4* p = reinterpret_cast<const float4*>(a.zm + (size_t)gi * 8);
      const float4 x0 = p[0], x1 = p[1];
      zm[0] = x0.x; zm[1] = x0.y; zm[2] = x0.z; zm[3] = x0.w; zm[4] = x1.x; zm[5] = x1.y; zm[6] = x1.z; zm[7] = x1.w;
      const float4* pk = reinterpret_cast<const float4*>(a.zmod + ((size_t)a.k_mod * a.nb + gi) * 8);
      const float4 y0 = pk[0], y1 = pk[1];
      zk[0] = y0.x; zk[1] = y0.y; zk[2] = y0.z; zk[3] = y0.w; zk[4] = y1.x; zk[5] = y1.y; zk[6] = y1.z; zk[7] = y1.w;
    }
    // ---- PPO terms on the composed head: logits = main + partner, or the partner's alone with nomain (policies.py:325-328)
    float z[8], pr[8], lse;
#pragma unroll
    for (int k = 0; k < 8; ++k) z[k] = a.nomain ? zk[k] : zm[k] + zk[k];
    softmax8(z, nk, pr, lse);
    int act = (int)a.rb_act[phys];
    act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
    float ent = 0.f, zact = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      ent -= (k < nk) ? pr[k] * (z[k] - lse) : 0.f;
      zact = (k == act) ? z[k] : zact;
    }
    const float logp = zact - lse;
    // advantages are ALWAYS normalised (learn.py:260-261: no normalize_advantage switch, no len > 1 guard)
    const float adv = (a.rb_adv[phys] - a.advstats[0]) / (a.advstats[1] + 1e-8f);
    const float oldlp = a.rb_logp[phys];
    const float lr = logp - oldlp;
    const float ratio = fast_exp(lr);
    const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
    const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
    const float pl1 = adv * ratio, pl2 = adv * rc;
    const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
    const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);   // torch.min / clamp backward
    const float inv_nb = 1.0f / (float)a.nb;
    const float g_lp = -inv_nb * adv * ratio * gate;
    const float g_en = -a.ent_coef * inv_nb;
    float dzp[8];   // dL_ppo / d composed logits
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dlogp = ((k == act) ? 1.f : 0.f) - pr[k];
      const float dent = -pr[k] * ((z[k] - lse) + ent);
      dzp[k] = (k < nk) ? g_lp * dlogp + g_en * dent : 0.f;
    }
    st[0] = -fminf(pl1, pl2);
    st[2] = -ent;
    st[3] = (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
    st[4] = oldlp - logp;                       // learn.py:327: plain mean(old_log_prob - log_prob)
    // ---- value term on the summed value (policies.py:286) ----
    const float v = a.vm[gi] + a.vk[gi];
    const float retn = a.rb_ret[phys], oldv = a.rb_val[phys];
    float vp = v, pass = 1.f;
    if (a.clip_vf >= 0.f) {
      const float dlt = v - oldv;
      pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
      vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
    }
    const float err = vp - retn;
    st[1] = err * err;
    a.dv[gi] = a.vf_coef * 2.0f * err * inv_nb * pass;
    // ---- marginal regulariser (learn.py:298-318): | softmax(z_main) - mean_j softmax(z_main + z_j) | summed over actions.
    // Modules shared by several partners (baseline) enter with their multiplicity / num_partners.
    float pm[8], lsem;
    softmax8(zm, nk, pm, lsem);
    float pc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) pc[k] = 0.f;
    for (int m = 0; m < a.n_mod; ++m) {
      const float4* pz = reinterpret_cast<const float4*>(a.zmod + ((size_t)m * a.nb + gi) * 8);
      const float4 y0 = pz[0], y1 = pz[1];
      float zc[8] = {zm[0] + y0.x, zm[1] + y0.y, zm[2] + y0.z, zm[3] + y0.w, zm[4] + y1.x, zm[5] + y1.y, zm[6] + y1.z, zm[7] + y1.w};
      float c[8], l2;
      softmax8(zc, nk, c, l2);
#pragma unroll
      for (int k = 0; k < 8; ++k) pc[k] = __builtin_fmaf(a.weight[m], c[k], pc[k]);
    }
    float sg[8], reg = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float d = pm[k] - pc[k];
      sg[k] = (k < nk) ? ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) : 0.f;   // d|x|/dx, 0 at 0 (torch.abs backward)
      reg += (k < nk) ? fabsf(d) : 0.f;
    }
    st[5] = reg;
    const float g_reg = a.reg_coef * inv_nb;
    // through softmax(z_main): p_b (s_b - sum_a s_a p_a)
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot = __builtin_fmaf(sg[k], pm[k], dot);
    float dzm[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) dzm[k] = (a.nomain ? 0.f : dzp[k]) + g_reg * pm[k] * (sg[k] - dot);
    for (int m = 0; m < a.n_mod; ++m) {
      const float4* pz = reinterpret_cast<const float4*>(a.zmod + ((size_t)m * a.nb + gi) * 8);
      const float4 y0 = pz[0], y1 = pz[1];
      float zc[8] = {zm[0] + y0.x, zm[1] + y0.y, zm[2] + y0.z, zm[3] + y0.w, zm[4] + y1.x, zm[5] + y1.y, zm[6] + y1.z, zm[7] + y1.w};
      float c[8], l2;
      softmax8(zc, nk, c, l2);
      float dc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) dc = __builtin_fmaf(sg[k], c[k], dc);
      float dm[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float g = -g_reg * a.weight[m] * c[k] * (sg[k] - dc);   // flows to z_main and to z_m alike (their sum is the argument)
        dzm[k] += g;
        dm[k] = g + ((m == a.k_mod) ? dzp[k] : 0.f);
      }
      float4* o = reinterpret_cast<float4*>(a.dzmod + ((size_t)m * a.nb + gi) * 8);
      o[0] = make_float4(dm[0], dm[1], dm[2], dm[3]);
      o[1] = make_float4(dm[4], dm[5], dm[6], dm[7]);
    }
    float4* o = reinterpret_cast<float4*>(a.dzm + (size_t)gi * 8);
    o[0] = make_float4(dzm[0], dzm[1], dzm[2], dzm[3]);
    o[1] = make_float4(dzm[4], dzm[5], dzm[6], dzm[7]);
  }
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) {
    float v = st[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    st[k] = v;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) red[wave][k] = st[k];
  }
  __syncthreads();
  if (threadIdx.x < NSTATP)
    a.statpart[(size_t)blockIdx.x * NSTATP + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}
hipError_t launch_modular_loss(const ModLossArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(modular_loss_kernel, dim3((a.nb + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---- statistics of the minibatch, optimizer step counter, first-use step of the trained module's value side ---------------
__global__ __launch_bounds__(256) void modular_finalize_kernel(ModFinalizeArgs a) {
  if (a.stop_flag && *a.stop_flag) {
    if (threadIdx.x < PH_NSTAT && a.stats_out) a.stats_out[threadIdx.x] = 0.f;
    return;
  }
  __shared__ float part[32][NSTATP];
  __shared__ float means[NSTATP];
  const int tid = threadIdx.x;
  const int kst = tid & (NSTATP - 1), j = tid >> 3;
  float v = 0.f;
  for (int w = j; w < a.nstatpart; w += 32) v += a.statpart[(size_t)w * NSTATP + kst];
  part[j][kst] = v;
  __syncthreads();
  if (tid < NSTATP) {
    float t = 0.f;
    for (int i = 0; i < 32; ++i) t += part[i][tid];
    means[tid] = t / (float)a.nb;
  }
  __syncthreads();
  if (tid == 0) {
    const int before = *a.step;
    *a.step = before + 1;
    if (a.mod_first[a.k_mod] < 0) a.mod_first[a.k_mod] = before;   // this module's value side takes part from this step on
    *a.kl_sum += means[4];
    if (a.stats_out) {
      a.stats_out[0] = means[0];
      a.stats_out[1] = means[1];
      a.stats_out[2] = means[2];
      a.stats_out[3] = means[3];
      a.stats_out[4] = means[4];
      a.stats_out[5] = means[0] + a.ent_coef * means[2] + a.vf_coef * means[1] + a.reg_coef * means[5];
      a.stats_out[6] = 0.f;          // gradient norm: modular_adam_kernel
      a.stats_out[7] = means[5];     // marginal regularisation loss
    }
  }
}
hipError_t launch_modular_finalize(const ModFinalizeArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(modular_finalize_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

// the target-KL test of learn.py:332-334: after a whole epoch, on the mean of that epoch's per-minibatch KLs
__global__ void modular_epoch_end_kernel(float* kl_sum, int n_mb, float target_kl, int* stop_flag) {
  if (*stop_flag == 0 && target_kl >= 0.f && (*kl_sum / (float)n_mb) > 1.5f * target_kl) *stop_flag = 1;
  *kl_sum = 0.f;
}
hipError_t launch_modular_epoch_end(float* kl_sum, int n_mb, float target_kl, int* stop_flag, hipStream_t s) {
  hipLaunchKernelGGL(modular_epoch_end_kernel, dim3(1), dim3(1), 0, s, kl_sum, n_mb, target_kl, stop_flag);
  return hipGetLastError();
}

// ---- clip_grad_norm_ + Adam over the main network and every module, torch's per-parameter step counts ---------------------
// A parameter takes part in a step iff it has a gradient tensor.  torch 1.13's optimizer.zero_grad() zeroes gradients in place
// (set_to_none = False), so once a parameter has received one it keeps taking part -- with g = 0 when the loss does not reach it
// (its moments decay, its step count advances).  The regulariser reaches the main network and every module's policy side from
// the first minibatch; a module's VALUE side (value tower + value head) is reached only while its partner is trained: it joins
// at the step recorded in mod_first[module] and counts its own steps from there.
__global__ __launch_bounds__(256) void modular_adam_kernel(ModAdamArgs a) {
  if (a.stop_flag && *a.stop_flag) return;
  __shared__ float sh[4];
  __shared__ float coef_s;
  __shared__ float ss_s[PH_MOD_MAX + 1], bc2s_s[PH_MOD_MAX + 1];
  const int tid = threadIdx.x;
  float q = 0.f;
  for (int k = tid; k < a.nblk; k += blockDim.x) q += a.blocksq[k];
  for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
  if ((tid & 63) == 0) sh[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    const float total = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
    const float cc = a.max_norm / (total + 1e-6f);
    coef_s = cc < 1.0f ? cc : 1.0f;
    if (blockIdx.x == 0 && a.stats_out) a.stats_out[6] = total;
  }
  if (tid <= a.n_mod) {   // slot 0: parameters that take part from the first step; slot 1 + m: module m's value side
    const int first = tid == 0 ? 0 : a.mod_first[tid - 1];
    const double t = (double)(*a.step - (first < 0 ? 0 : first));
    const double bc1 = 1.0 - pow((double)a.beta1, t), bc2 = 1.0 - pow((double)a.beta2, t);
    ss_s[tid] = (float)((double)a.lr / bc1);
    bc2s_s[tid] = (float)sqrt(bc2);
  }
  __syncthreads();
  const int p = blockIdx.x * blockDim.x + tid;
  if (p >= a.P) return;
  int slot = 0;
  bool live = true, reached = true;
  for (int s = 0; s < a.n_seg; ++s) {
    if (p >= a.seg_lo[s] && p < a.seg_hi[s]) {
      const int m = a.seg_mod[s];
      slot = 1 + m;
      live = a.mod_first[m] >= 0;
      reached = m == a.k_mod;
    }
  }
  if (!live) return;                                   // no gradient tensor yet: the optimizer skips the parameter
  const float g = reached ? a.grad[p] * coef_s : 0.f;  // (zeros contribute nothing to the norm either)
  const float m1 = a.m[p] + (g - a.m[p]) * (1.0f - a.beta1);
  const float v = a.v[p] * a.beta2 + (1.0f - a.beta2) * g * g;
  const float denom = sqrtf(v) / bc2s_s[slot] + a.eps;
  a.m[p] = m1;
  a.v[p] = v;
  a.params[p] = a.params[p] - ss_s[slot] * (m1 / denom);
}
hipError_t launch_modular_adam(const ModAdamArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(modular_adam_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---- ModularPolicy.forward (policies.py:271-288) after the towers: composed logits and value of one row, sampling, fused add --
__global__ __launch_bounds__(256) void modular_act_kernel(FwdArgs a, const float* zm, const float* zk, const float* vm,
                                                          const float* vk, int nomain, float* logits_main, float* logits_partner) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const NetDims& nd = a.nd;
  if (g < a.n) {
    float z[8];
    const float4* p = reinterpret_cast<const float4*>(zm + (size_t)g * 8);
    const float4* pk = reinterpret_cast<const float4*>(zk + (size_t)g * 8);
    const float4 x0 = p[0], x1 = p[1], y0 = pk[0], y1 = pk[1];
    const float m8[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    const float k8[8] = {y0.x, y0.y, y0.z, y0.w, y1.x, y1.y, y1.z, y1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      z[k] = nomain ? k8[k] : m8[k] + k8[k];
      if (k < nd.L) {
        if (logits_main) logits_main[(size_t)g * nd.L + k] = m8[k];
        if (logits_partner) logits_partner[(size_t)g * nd.L + k] = k8[k];
      }
    }
    discrete8_row_tail(a, nd, g, z, fwd_counter(a));   // mask offset (policies.py:330-333), sampling, log-prob, buffer row
    value_row_tail(a, g, vm[g] + vk[g]);               // policies.py:286
  }
  if (a.rb_obs) {   // RolloutBuffer.add copies the observation
    const size_t total = (size_t)a.n * nd.D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
      a.rb_obs[e] = a.obs[e];
  }
}
hipError_t launch_modular_act(const FwdArgs& a, const float* zm, const float* zk, const float* vm, const float* vk, int nomain,
                              float* logits_main, float* logits_partner, hipStream_t s) {
  hipLaunchKernelGGL(modular_act_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a, zm, zk, vm, vk, nomain, logits_main,
                     logits_partner);
  return hipGetLastError();
}

}  // namespace ph

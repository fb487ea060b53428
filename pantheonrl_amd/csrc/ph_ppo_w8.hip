// ppo_grad_w8_kernel: the phased PPO minibatch gradient (SB3 PPO.train() inner loop, pantheonrl/common/agents.py:155;
// arithmetic from SURVEY.md A.3 and the in-tree copy pantheonrl/algos/adap/adap_learn.py:253-344) with EIGHT waves per
// workgroup -- four per SIMD at two workgroups per CU -- for Box observations with F <= 64 features and a Discrete head
// with <= 8 logits.
//
// PMC on the four-wave kernel (ppo_grad_fast_kernel) shows waves waiting on an instruction dependency for half of their
// cycles with the matrix pipe a third busy: two waves per SIMD do not cover the LDS / MFMA latencies between barriers.
// This variant keeps that kernel's phases, LDS footprint (~71 KB -> two workgroups per CU) and prefetch scheme but splits
// every 64x64 product into sixteen 16x16 v_mfma_f32_16x16x4_f32 tiles, two per wave (sharing the A operand), so that a
// workgroup has 8 waves and a lane needs <= 128 registers.  LDS matrices are XOR-swizzled [64][64] (ph_device.h), the head
// runs with eight lanes per row, weight-gradient tiles (two 16x16 tiles of dW1 and of dW2 per wave) stay in registers
// across a workgroup's row tiles.
#include "ph_launch.h"

namespace ph {

__device__ __forceinline__ void w8_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CTRL>
__device__ __forceinline__ float w8_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// sum over the eight lanes of an aligned octet; all eight end with the identical bits
__device__ __forceinline__ float oct_sum(float v) {
  v += w8_dpp<0xB1>(v);            // quad_perm [1,0,3,2]
  v += w8_dpp<0x4E>(v);            // quad_perm [2,3,0,1]
  v += __shfl_xor(v, 4, 64);       // the other quad of the octet
  return v;
}

// acc[j] += sum_s A(s) x B(s, j), j = 0, 1, over 16 k-steps; operands fetched two steps ahead of their MFMAs
template <bool VALU, class FA, class FB>
__device__ __forceinline__ void mma_1x2(f32x4 (&acc)[2], FA&& fa, FB&& fb, int lane) {
  float av[3], bv[3][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    av[s] = fa(s);
    bv[s][0] = fb(s, 0);
    bv[s][1] = fb(s, 1);
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    if (s + 2 < 16) {
      av[(s + 2) % 3] = fa(s + 2);
      bv[(s + 2) % 3][0] = fb(s + 2, 0);
      bv[(s + 2) % 3][1] = fb(s + 2, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    acc[0] = mma16<VALU>(av[s % 3], bv[s % 3][0], acc[0], lane);
    acc[1] = mma16<VALU>(av[s % 3], bv[s % 3][1], acc[1], lane);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct W8Meta {
  int phys;
  float adv, old, act;
};

constexpr int W8_NT = 512, W8_R = 64;
constexpr int W8_LDS_FLOATS = 4 * 4096 + 64 * 8 + 64 * 8 + 64 + 64 + 16 + 3 * 64 + 64 + 64 * 4;

template <bool VALU>
__device__ __forceinline__ void w8_body(const GradArgs& a) {
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = W8_R;
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  float* bufA = smem;               // [64][64] swizzled   X -> H2 -> X
  float* bufB = bufA + 4096;        // H1 -> dZ1
  float* bufC = bufB + 4096;        // W1 -> dZ2 -> W1
  float* w2s = bufC + 4096;         // W2
  float* hw = w2s + 4096;           // policy: act_W [64][8] (columns >= L zero) | value: val_W [64]
  float* dzs = hw + 64 * 8;         // policy: dL/dlogits [R][8] | value: dL/dv [R]
  float* b1s = dzs + 64 * 8;        // [64]
  float* b2s = b1s + 64;            // [64]
  float* hbs = b2s + 64;            // act_b [8] (-3e38 beyond L) | val_b
  float* radv = hbs + 16;           // [R] normalised advantage | returns
  float* rold = radv + R;           // [R] old log-prob | old values
  float* ract = rold + R;           // [R] action index
  int* rowphys = (int*)(ract + R);  // [R] physical buffer row, -1 = padding
  float* stat = (float*)(rowphys + R);   // [R][4] per-row running sums of the minibatch statistics (kept out of the registers)

  const int net = blockIdx.y;
  const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
  const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
  float* slab = a.slabs + (size_t)blockIdx.x * lay.P;
  const float inv_nb = 1.0f / (float)a.nb;
  const int nk = nd.L;
  const uint64_t perm_key = a.idx ? 0ull : epoch_key(a.perm_seed + (a.epoch ? *a.epoch : 0ull), a.perm_epoch);
  const bool norm = net == 0 && a.norm_adv && a.nb > 1;
  const float adv_mean = norm ? a.advstats[0] : 0.f;
  const float adv_den = norm ? a.advstats[1] + 1e-8f : 1.f;

  // rows of a tile: lane i < 8 of wave w serves row w + 8i (a wave-instruction of the observation gather is one row)
  auto row_index = [&](int tile, int wave, int lane) -> int {
    const int gi = tile * R + wave + 8 * lane;
    if (lane >= 8 || gi >= a.nb) return -1;
    return a.idx ? a.idx[gi] : (int)feistel_perm((uint32_t)(a.mb_start + gi), a.perm_n, a.perm_hb, perm_key);
  };
  auto row_scalars = [&](int n) -> W8Meta {
    W8Meta m;
    m.phys = -1;
    m.adv = m.old = m.act = 0.f;
    if (n >= 0) {
      m.phys = env_major_to_phys(n, a.T, a.E);
      if (net == 0) {
        m.adv = a.rb_adv[m.phys];
        m.old = a.rb_logp[m.phys];
        m.act = a.rb_act[m.phys];
      } else {
        m.adv = a.rb_ret[m.phys];
        m.old = a.rb_val[m.phys];
      }
    }
    return m;
  };
  auto load_x = [&](int physv, float (&xr)[8], int lane) {   // raw loads; masking happens at commit
    const int f = lane < nd.F ? lane : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int p = __builtin_amdgcn_readlane(physv, i);
      xr[i] = a.rb_obs[(size_t)(p < 0 ? 0 : p) * nd.D + f];
    }
  };
  // 64x64 weight block -> swizzled LDS: thread t carries float4 q = t and q = t + 512 of the 1024
  auto load_w = [&](const float* W, int nrows, float4 (&wr)[2], int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + W8_NT * i, k = q >> 4, c4 = (q & 15) << 2;
      wr[i] = (k < nrows) ? *reinterpret_cast<const float4*>(W + (size_t)k * HID + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_w = [&](float* dst, const float4 (&wr)[2], int tid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + W8_NT * i, k = q >> 4, c4 = (q & 15) << 2, sw = swz(k);   // bit 0 of sw is clear
      *reinterpret_cast<float2*>(dst + k * 64 + (c4 ^ sw)) = make_float2(wr[i].x, wr[i].y);
      *reinterpret_cast<float2*>(dst + k * 64 + ((c4 + 2) ^ sw)) = make_float2(wr[i].z, wr[i].w);
    }
  };

  // ---- prologue ----
  W8Meta meta;
  float xr[8];
  {
    float4 w1r[2];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    meta = row_scalars(row_index(blockIdx.x, wave, lane));
    load_x(meta.phys, xr, lane);
    float4 w2r[2];
    load_w(a.params + oW1, nd.F, w1r, tid);
    load_w(a.params + oW2, HID, w2r, tid);
    float bias1 = 0.f, bias2 = 0.f, hv = 0.f, hb = 0.f;
    if (tid < HID) {
      bias1 = a.params[oB1 + tid];
      bias2 = a.params[oB2 + tid];
    }
    if (net == 0) {
      const int j = tid >> 3, k = tid & 7;
      if (k < nk) hv = a.params[lay.act_W + j * nk + k];
      // padded logits get a -3e38 "bias": they drop out of softmax, entropy and every gradient with no special cases
      if (tid < 8) hb = (tid < nk) ? a.params[lay.act_b + tid] : -3.0e38f;
    } else {
      if (tid < HID) hv = a.params[lay.val_W + tid];
      if (tid == 0) hb = a.params[lay.val_b];
    }
    store_w(bufC, w1r, tid);
    store_w(w2s, w2r, tid);
    if (tid < HID) {
      b1s[tid] = bias1;
      b2s[tid] = bias2;
    }
    if (net == 0) {
      hw[tid] = hv;
      if (tid < 8) hbs[tid] = hb;
    } else {
      if (tid < HID) hw[tid] = hv;
      if (tid == 0) hbs[0] = hb;
    }
  }

  f32x4 G1[2], G2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    G1[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    G2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gh = 0.f;               // policy: d act_W[j = tid & 63][k = wave] (complete) | value: d val_W[j] partial of 8 rows
  float gb1 = 0.f, gb2 = 0.f;   // bias-gradient partials: column tid & 63, rows 8*wave .. 8*wave+7
  float ghb = 0.f;              // policy: d act_b[lane] partial (lane < 8) | value: d val_b partial
  if (threadIdx.x < R * 4) stat[threadIdx.x] = 0.f;   // policy: loss, -entropy, clip count, KL sums | value: squared error

  bool first = true;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, first = false) {
    // lane coordinates re-derived per tile from an opaque copy (stops the hoisting of loop-invariant LDS addresses)
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tm = wave >> 1, tn0 = 2 * (wave & 1);   // this wave's tiles: row block tm, column tiles tn0, tn0 + 1
    const bool has_next = tile + (int)gridDim.x < a.ntiles;
    // operand addressing (ph_ppo_rp.hip): B-type (row 4s+g, col 16t+c) = ((bq ^ 16t) ^ KS(s)) + 256 s; A-type (row R, col 4s+g)
    // = (R*64 + (g ^ swz(R))) ^ 4s
    const int bq = g * 64 + (c ^ (2 * g) ^ (16 * (g & 1)));
    const int aqm = (16 * tm + c) * 64 + (g ^ swz(16 * tm + c));
#define W8_KS(s) ((((8 * (s)) & 62) ^ (16 * ((s) & 1))))
#define W8_B(buf, s, t) (buf)[((bqp ^ (16 * (t))) ^ W8_KS(s)) + 256 * (s)]
#define W8_A(buf, s) (buf)[aqp ^ (4 * (s))]
    // every phase works on its own opaque copies of the two bases: otherwise the operand addresses of all phases are
    // computed once per tile and stay live across it (tens of VGPRs)
#define W8_PHASE_BASES() int bqp = bq, aqp = aqm; asm volatile("" : "+v"(bqp), "+v"(aqp))

    // ---- T0: this tile's rows land in LDS ----
    if (lane < 8) {
      const int row = wave + 8 * lane;
      rowphys[row] = meta.phys;
      radv[row] = (norm && meta.phys >= 0) ? (meta.adv - adv_mean) / adv_den : meta.adv;
      rold[row] = meta.old;
      ract[row] = meta.act;
    }
    {
      const bool fok = lane < nd.F;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int p = __builtin_amdgcn_readlane(meta.phys, i);
        bufA[sidx(wave + 8 * i, lane)] = (p >= 0 && fok) ? xr[i] : 0.f;
      }
    }
    w8_barrier();
    if (first) PH_STAMP(a.prof, 1);
    const int n_next = has_next ? row_index(tile + gridDim.x, wave, lane) : -1;

    // ---- S1: H1 = tanh(X W1 + b1) -> bufB ----
    f32x4 acc[2];
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      W8_PHASE_BASES();
      mma_1x2<VALU>(acc, [&](int s) { return W8_A(bufA, s); }, [&](int s, int j) { return W8_B(bufC, s, tn0 + j); }, lane);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 16 * (tn0 + j) + c;
      const float b = b1s[col];
#pragma unroll
      for (int r = 0; r < 4; ++r) bufB[sidx(16 * tm + 4 * g + r, col)] = fast_tanh(acc[j][r] + b);
    }
    w8_barrier();
    if (first) PH_STAMP(a.prof, 2);

    // ---- S2: H2 = tanh(H1 W2 + b2) -> bufC (W1 is dead after layer 1; X stays in bufA for the whole tile) ----
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      W8_PHASE_BASES();
      mma_1x2<VALU>(acc, [&](int s) { return W8_A(bufB, s); }, [&](int s, int j) { return W8_B(w2s, s, tn0 + j); }, lane);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = 16 * (tn0 + j) + c;
      const float b = b2s[col];
#pragma unroll
      for (int r = 0; r < 4; ++r) bufC[sidx(16 * tm + 4 * g + r, col)] = fast_tanh(acc[j][r] + b);
    }
    w8_barrier();
    if (first) PH_STAMP(a.prof, 3);

    // ---- SH: head forward, loss, dL/dhead; dZ2 = dH2 * (1 - H2^2) stays in registers until the head-weight gradient has
    //      read H2 (it replaces H2 in bufC two barriers later); eight lanes per row, units q + 8m ----
    float dz2[8];
    {
      const int r = tid >> 3, q = tid & 7;
      const bool valid = rowphys[r] >= 0;
      float h[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) h[m] = bufC[sidx(r, q + 8 * m)];
      if (net == 0) {
        float z[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) z[k] = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if ((m & 1) == 0) __builtin_amdgcn_sched_barrier(0);   // two weight rows (16 VGPRs) in flight at a time
          const float4* w = reinterpret_cast<const float4*>(hw + (q + 8 * m) * 8);
          const float4 w0 = w[0], w1 = w[1];
          z[0] = __builtin_fmaf(h[m], w0.x, z[0]);
          z[1] = __builtin_fmaf(h[m], w0.y, z[1]);
          z[2] = __builtin_fmaf(h[m], w0.z, z[2]);
          z[3] = __builtin_fmaf(h[m], w0.w, z[3]);
          z[4] = __builtin_fmaf(h[m], w1.x, z[4]);
          z[5] = __builtin_fmaf(h[m], w1.y, z[5]);
          z[6] = __builtin_fmaf(h[m], w1.z, z[6]);
          z[7] = __builtin_fmaf(h[m], w1.w, z[7]);
        }
        float pr[8];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          z[k] = oct_sum(z[k]) + hbs[k];
          mx = fmaxf(mx, z[k]);
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          pr[k] = fast_exp(z[k] - mx);
          se += pr[k];
        }
        const float lse = mx + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
        int act = (int)ract[r];
        act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
        float ent = 0.f, zact = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          pr[k] *= inv;
          ent -= pr[k] * (z[k] - lse);
          zact = (k == act) ? z[k] : zact;
        }
        const float logp = zact - lse;
        const float adv = radv[r];
        const float lr = logp - rold[r];
        const float ratio = fast_exp(lr);
        const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float pl1 = adv * ratio, pl2 = adv * rc;
        // torch.min backward: the smaller branch gets the gradient, ties split 1/2 + 1/2; clamp passes it iff lo <= ratio <= hi
        const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
        const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
        const float live = valid ? 1.f : 0.f;
        const float g_lp = -inv_nb * adv * ratio * gate * live;
        const float g_en = -a.ent_coef * inv_nb * live;
        if (valid && q == 0) {
          stat[r * 4 + 0] += -fminf(pl1, pl2);
          stat[r * 4 + 1] += -ent;
          stat[r * 4 + 2] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          stat[r * 4 + 3] += (ratio - 1.0f) - lr;
        }
        float (&dz)[8] = pr;   // dL/dlogit k replaces the probability it is computed from
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float dlogp = ((k == act) ? 1.f : 0.f) - pr[k];
          const float dent = -pr[k] * ((z[k] - lse) + ent);
          dz[k] = g_lp * dlogp + g_en * dent;
        }
        if (q == 0) {
          float4* o = reinterpret_cast<float4*>(dzs + r * 8);
          o[0] = make_float4(dz[0], dz[1], dz[2], dz[3]);
          o[1] = make_float4(dz[4], dz[5], dz[6], dz[7]);
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          if ((m & 1) == 0) __builtin_amdgcn_sched_barrier(0);
          const float4* w = reinterpret_cast<const float4*>(hw + (q + 8 * m) * 8);
          const float4 w0 = w[0], w1 = w[1];
          float d = dz[0] * w0.x;
          d = __builtin_fmaf(dz[1], w0.y, d);
          d = __builtin_fmaf(dz[2], w0.z, d);
          d = __builtin_fmaf(dz[3], w0.w, d);
          d = __builtin_fmaf(dz[4], w1.x, d);
          d = __builtin_fmaf(dz[5], w1.y, d);
          d = __builtin_fmaf(dz[6], w1.z, d);
          d = __builtin_fmaf(dz[7], w1.w, d);
          dz2[m] = d * (1.0f - h[m] * h[m]);
        }
      } else {
        float wv[8];
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          wv[m] = hw[q + 8 * m];
          v = __builtin_fmaf(h[m], wv[m], v);
        }
        v = oct_sum(v) + hbs[0];
        const float retn = radv[r], oldv = rold[r];
        float vp = v, pass = 1.f;
        if (a.clip_vf >= 0.f) {
          const float dlt = v - oldv;
          pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
          vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
        }
        const float err = vp - retn;
        const float dv = valid ? a.vf_coef * 2.0f * err * inv_nb * pass : 0.f;
        if (valid && q == 0) stat[r * 4] += err * err;
        if (q == 0) dzs[r] = dv;
#pragma unroll
        for (int m = 0; m < 8; ++m) dz2[m] = dv * wv[m] * (1.0f - h[m] * h[m]);
      }
    }
    w8_barrier();
    if (first) PH_STAMP(a.prof, 4);

    // ---- SD: head-weight gradients from H2 (bufC) and dL/dhead (dzs) ----
    {
      const int col = tid & 63, part = tid >> 6;            // part == wave
      if (net == 0) {
        // d act_W[col][k = wave] over all 64 rows, 8 rows of operands in flight
#pragma unroll 1
        for (int r0 = 0; r0 < R; r0 += 8) {
          float hv[8], dv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            hv[i] = bufC[sidx(r0 + i, col)];
            dv[i] = dzs[(r0 + i) * 8 + part];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i) gh = __builtin_fmaf(hv[i], dv[i], gh);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < 8) {
#pragma unroll
          for (int i = 0; i < 8; ++i) ghb += dzs[(8 * part + i) * 8 + lane];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dv = dzs[8 * part + i];
          gh = __builtin_fmaf(bufC[sidx(8 * part + i, col)], dv, gh);
          ghb += dv;
        }
      }
    }
    w8_barrier();
    // ---- SW: dZ2 replaces H2 ----
    {
      const int r = tid >> 3, q = tid & 7;
#pragma unroll
      for (int m = 0; m < 8; ++m) bufC[sidx(r, q + 8 * m)] = dz2[m];
    }
    w8_barrier();

    // ---- S6a: d b2 ; dW2 += H1^T dZ2 ; dH1 = dZ2 W2^T ----
    {
      const int col = tid & 63, part = tid >> 6;
      float t2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t2[i] = bufC[sidx(8 * part + i, col)];
      gb2 += ((t2[0] + t2[1]) + (t2[2] + t2[3])) + ((t2[4] + t2[5]) + (t2[6] + t2[7]));
    }
    {
      W8_PHASE_BASES();
      mma_1x2<VALU>(G2, [&](int s) { return W8_B(bufB, s, tm); }, [&](int s, int j) { return W8_B(bufC, s, tn0 + j); }, lane);
    }
    if (first) PH_STAMP(a.prof, 5);
    acc[0] = acc[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      W8_PHASE_BASES();
      int wq[2];   // B[k = j][n = k'] = W2[k'][j]: row 16t + c of W2s, column 4s + g -- an A-type access
#pragma unroll
      for (int j = 0; j < 2; ++j) wq[j] = (16 * (tn0 + j) + c) * 64 + (g ^ swz(16 * (tn0 + j) + c));
      mma_1x2<VALU>(acc, [&](int s) { return W8_A(bufC, s); }, [&](int s, int j) { return w2s[wq[j] ^ (4 * s)]; }, lane);
    }
    w8_barrier();
    if (first) PH_STAMP(a.prof, 6);

    // ---- S6b: dZ1 = dH1 * (1 - H1^2) in place ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = sidx(16 * tm + 4 * g + r, 16 * (tn0 + j) + c);
        const float hv = bufB[o];
        bufB[o] = acc[j][r] * (1.0f - hv * hv);
      }
    }
    w8_barrier();
    if (first) PH_STAMP(a.prof, 7);

    // ---- S7: dW1 += X^T dZ1 ; d b1.  The next tile's rows are gathered underneath. ----
    float4 w1r[2];
    if (has_next) {
      meta = row_scalars(n_next);               // next tile's row scalars and rows, committed at its T0
      load_w(a.params + oW1, nd.F, w1r, tid);   // refill of bufC (dZ2 is consumed): lands under the dW1 MFMAs
      load_x(meta.phys, xr, lane);
    }
    {
      const int col = tid & 63, part = tid >> 6;
      float t1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t1[i] = bufB[sidx(8 * part + i, col)];
      gb1 += ((t1[0] + t1[1]) + (t1[2] + t1[3])) + ((t1[4] + t1[5]) + (t1[6] + t1[7]));
    }
    {
      W8_PHASE_BASES();
      mma_1x2<VALU>(G1, [&](int s) { return W8_B(bufA, s, tm); }, [&](int s, int j) { return W8_B(bufB, s, tn0 + j); }, lane);
    }
    if (has_next) store_w(bufC, w1r, tid);
    w8_barrier();  // bufA / bufB / row scalars are free for the next tile
    if (first) PH_STAMP(a.prof, 8);
  }
  PH_STAMP(a.prof, 12);

  // ---- epilogue: owned tiles -> slab; per-wave partials folded across the 8 waves in a fixed order ----
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4, tm = wave >> 1, tn0 = 2 * (wave & 1);
    // weight-gradient tiles -> LDS (row-major [64][64] over the dead bufB / bufC), then row-contiguous copies to the slab:
    // coalesced stores and no per-element 64-bit address arithmetic in registers
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int e = (16 * tm + 4 * g + r) * 64 + 16 * (tn0 + j) + c;
        bufB[e] = G2[j][r];
        bufC[e] = G1[j][r];
      }
    }
    w8_barrier();
#pragma unroll 4
    for (int e = tid; e < 64 * 64; e += W8_NT) {
      slab[oW2 + e] = bufB[e];
      if ((e >> 6) < nd.F) slab[oW1 + e] = bufC[e];
    }
    if (net == 0 && wave < nk) slab[lay.act_W + lane * nk + wave] = gh;   // d act_W[j = lane][k = wave], summed over all rows
    // per-row statistic sums -> wave 0 folds them (the partial record uses the slots of the four-wave kernels: 0 policy
    // loss, 1 value loss, 2 -entropy, 3 clip count, 4 KL)
    float st[NSTATP] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (wave == 0) {
      const float s0 = stat[lane * 4 + 0], s1 = stat[lane * 4 + 1], s2 = stat[lane * 4 + 2], s3 = stat[lane * 4 + 3];
      if (net == 0) {
        st[0] = s0; st[2] = s1; st[3] = s2; st[4] = s3;
      } else {
        st[1] = s0;
      }
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        float v = st[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        st[k] = v;
      }
    }
    float* part = bufA;  // [5][8 waves][64]
    part[(0 * 8 + wave) * 64 + lane] = gb1;
    part[(1 * 8 + wave) * 64 + lane] = gb2;
    part[(2 * 8 + wave) * 64 + lane] = gh;    // value net: d val_W partials
    part[(3 * 8 + wave) * 64 + lane] = ghb;
    {
      float sv = 0.f;
#pragma unroll
      for (int k = 0; k < NSTATP; ++k) {
        const float tot = __shfl(st[k], 0, 64);   // outside the select: a cross-lane read needs its source lane active
        sv = (lane == k) ? tot : sv;
      }
      part[(4 * 8 + wave) * 64 + lane] = sv;
    }
    w8_barrier();
    auto wsum = [&](int which, int idx) {
      float v = part[(which * 8 + 0) * 64 + idx];
#pragma unroll
      for (int w = 1; w < 8; ++w) v += part[(which * 8 + w) * 64 + idx];
      return v;
    };
    if (tid < HID) {
      slab[oB1 + tid] = wsum(0, tid);
      slab[oB2 + tid] = wsum(1, tid);
      if (net == 1) slab[lay.val_W + tid] = wsum(2, tid);
    }
    if (net == 0 && tid < nk) slab[lay.act_b + tid] = wsum(3, tid);
    if (net == 1 && tid == 0) slab[lay.val_b] = wsum(3, 0);
    if (tid < NSTATP) a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + tid] = wsum(4, tid);
  }
  PH_STAMP(a.prof, 13);
}

template <bool VALU>
__global__ __launch_bounds__(W8_NT, 4) void ppo_grad_w8_kernel(GradArgs a) {
  if (*a.stop_flag) return;
  w8_body<VALU>(a);
}

bool grad_w8_eligible(const NetDims& nd) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_GRAD_W8");
    enabled = (e && e[0] == '1') ? 1 : 0;
  }
  return enabled && nd.obs_kind == PH_SPACE_BOX && nd.nchunk == 1 && nd.A == 1 && nd.L <= 8;
}

template <bool VALU>
static hipError_t launch_w8_variant(const GradArgs& a, int nwg, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)W8_LDS_FLOATS;
  static bool allowed = false;  // > 64 KiB of dynamic LDS is opt-in, once per kernel (kept out of graph capture)
  if (!allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_w8_kernel<VALU>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  hipLaunchKernelGGL((ppo_grad_w8_kernel<VALU>), dim3(nwg, 2), dim3(W8_NT), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_ppo_grad_w8(const GradArgs& a, int nwg, int gemm_mode, hipStream_t s) {
  return gemm_mode != 0 ? launch_w8_variant<true>(a, nwg, s) : launch_w8_variant<false>(a, nwg, s);
}

}  // namespace ph

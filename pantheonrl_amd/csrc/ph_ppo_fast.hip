// ppo_grad_fast_kernel: the PPO minibatch gradient (SB3 PPO.train() inner loop, pantheonrl/common/agents.py:155; arithmetic
// from SURVEY.md A.3 and the in-tree copy pantheonrl/algos/adap/adap_learn.py:253-344) for the shape class every
// BASELINE config but Liar's Dice falls in: one feature chunk (F <= 64) and a Discrete action head with <= 8 logits.
//
// Same decomposition as ppo_grad_kernel (grid (nWG, 2 nets), 64-row tiles, every 64x64 product as 32x32
// v_mfma_f32_32x32x2_f32 tiles on LDS operands, one per wave) with three structural differences:
//   * the head (logits / value, loss, dL/dlogits, dH2, dZ2) is ONE register-resident VALU phase -- four lanes per row,
//     each owning 16 hidden units, quad-DPP reductions -- instead of four barrier-separated phases around 32-wide padded
//     MFMA tiles of a 6-wide head;
//   * weight- and bias-gradient accumulators live in registers across the row tiles of a workgroup and are stored to
//     the workgroup's slab once (no slab read-modify-write per tile, no store drain at barriers), in the accumulators' own
//     register order: 16-byte stores, 1 KB contiguous per wave instruction; the reduce kernel maps slab positions to
//     parameter indices through a per-spec table (grad_slab_map);
//   * barriers wait for LDS only (s_waitcnt lgkmcnt(0); s_barrier), so global loads issued in one phase -- the next
//     tile's gathered rows, its per-row scalars, the W1 refill -- stay in flight across phases and are consumed later.
// LDS: bufA/bufB/bufC/W2 (4 x [64][65] f32) + head weights + per-row scalars = 72.9 KB -> two workgroups per CU.
#include "ph_head.h"

// Issue-slot cuts of round 3 (f32 MFMA and VALU share the SIMD's lanes, DESIGN.md 3.1: every VALU / LDS instruction removed
// from the tile walk is time).  Each can be switched off at build time for same-box A/B (scripts/build_variants.sh):
//   PH_FAST_NK      the policy head's loops run over the L logits that exist (template parameter), not over 8 padded slots
//   PH_FAST_GHROWS  d act_W accumulated per wave over ITS 16 rows for all logits (48 LDS reads + 6*16 FMAs per tile) instead of
//                   two logit columns over all 64 rows (128 reads + 128 FMAs), cross-wave sum once in the epilogue
//   PH_FAST_PHYS    the minibatch order arrives as physical buffer rows (adv_stats_kernel translates once per train()): no
//                   integer division per row in the tile walk
//   PH_FAST_FOLDB1  F < 64 Box observations: column 63 of X is 1 and row 63 of the staged W1 is b1, so the MFMA adds the bias
//                   (bitwise: fmaf(1, b, acc) = acc + b) and d b1 is row 63 of the dW1 accumulators
#ifndef PH_FAST_NK
#define PH_FAST_NK 1
#endif
#ifndef PH_FAST_GHROWS
#define PH_FAST_GHROWS 1
#endif
#ifndef PH_FAST_PHYS
#define PH_FAST_PHYS 1
#endif
#ifndef PH_FAST_FOLDB1
#define PH_FAST_FOLDB1 1
#endif

namespace ph {

struct RowMeta {
  int phys;
  float adv, old, act;
};

// Box observation tile in registers: lane = feature, register i = row wave + 4*i.  The row bases come from the lanes
// that computed them through v_readlane (scalar), so the gather is 16 back-to-back 256-byte row loads per wave.
struct XRegs {
  float v[16];
  // raw loads only: masking happens at commit, so nothing here waits for the data
  __device__ __forceinline__ void issue(int physv, const float* obs, const NetDims& nd, int lane) {
    const int f = lane < nd.F ? lane : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = __builtin_amdgcn_readlane(physv, i);
      v[i] = __builtin_nontemporal_load(obs + (size_t)(p < 0 ? 0 : p) * nd.D + f);   // read once per epoch and net
    }
  }
  // FOLD: column 63 of a live row is 1 (the bias row of the staged W1 multiplies it)
  template <bool FOLD>
  __device__ __forceinline__ void commit(float* dst, int physv, const NetDims& nd, int wave, int lane) const {
    const bool fok = lane < nd.F;
    const float pad = (FOLD && lane == 63) ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = __builtin_amdgcn_readlane(physv, i);
      dst[(wave + 4 * i) * LDH + lane] = (p >= 0) ? (fok ? v[i] : pad) : 0.f;
    }
  }
};

template <bool VALU, int NK, bool FOLD>
__global__ __launch_bounds__(256, 2) void ppo_grad_fast_kernel(GradArgs a) {
  if (*a.stop_flag) return;
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 64, NT = 256;
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  float* bufA = smem;               // [R][LDH]   X -> H2 -> X
  float* bufB = bufA + R * LDH;     // [R][LDH]   H1 -> dZ1
  float* bufC = bufB + R * LDH;     // [64][LDH]  W1 -> dZ2 -> W1
  float* w2s = bufC + HID * LDH;    // [64][LDH]
  float* hw = w2s + HID * LDH;      // policy: act_W as [64][8] (columns >= L zero) | value: val_W [64]
  float* dzs = hw + HW_FLOATS;       // policy: dL/dlogits [R][8] | value: dL/dv [R]
  float* b1s = dzs + R * 8;         // [64]
  float* b2s = b1s + HID;           // [64]
  float* hbs = b2s + HID;           // act_b [8] | val_b
  float* radv = hbs + 16;           // [R] normalised advantage | returns
  float* rold = radv + R;           // [R] old log-prob | old values
  float* ract = rold + R;           // [R] action index
  int* rowphys = (int*)(ract + R);  // [R] physical buffer row, -1 = padding

  const int net = blockIdx.y;
  const bool box = nd.obs_kind == PH_SPACE_BOX;
  const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
  const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
  const float inv_nb = 1.0f / (float)a.nb;
  const int nk = nd.L;
  const float* row63 = FOLD ? a.params + oB1 : nullptr;   // the staged W1's row 63 (FOLD: b1)

  // loop invariants that live in memory are read once here (inside the tile loop each would be a fresh dependent load
  // -- the asm barriers are memory clobbers -- and its wait would also drain the prefetches in flight)
  const uint64_t perm_key = a.idx ? 0ull : epoch_key(a.perm_seed + (a.epoch ? *a.epoch : 0ull), a.perm_epoch);
  const bool norm = net == 0 && a.norm_adv && a.nb > 1;
  const float adv_mean = norm ? a.advstats[0] : 0.f;
  const float adv_den = norm ? a.advstats[1] + 1e-8f : 1.f;

  // per-row gathers of one tile, lane i < 16 of wave w serving row w + 4*i, in two steps so that the index load of the
  // materialised minibatch order (ph_ppo_train) can be issued phases before the gathers that depend on it
  auto row_index = [&](int tile, int wave, int lane) -> int {
    const int gi = tile * R + wave + 4 * lane;
    if (lane >= 16 || gi >= a.nb) return -1;
#if PH_FAST_PHYS
    if (a.idx_phys) return a.idx_phys[gi];     // already a physical row
#endif
    return a.idx ? a.idx[gi] : (int)feistel_perm((uint32_t)(a.mb_start + gi), a.perm_n, a.perm_hb, perm_key);
  };
  auto row_scalars = [&](int n) -> RowMeta {
    RowMeta m;
    m.phys = -1;
    m.adv = m.old = m.act = 0.f;
    if (n >= 0) {
#if PH_FAST_PHYS
      m.phys = a.idx_phys ? n : env_major_to_phys(n, a.T, a.E);
#else
      m.phys = env_major_to_phys(n, a.T, a.E);
#endif
      if (net == 0) {
        m.adv = a.rb_adv[m.phys];   // normalised when it is committed to LDS (no wait on the gather here)
        m.old = a.rb_logp[m.phys];
        m.act = a.rb_act[m.phys];
      } else {
        m.adv = a.rb_ret[m.phys];
        m.old = a.rb_val[m.phys];
      }
    }
    return m;
  };

  // ---- prologue: every global load of the first tile is in flight before the first wait ----
  WStage<NT> w2r, w1r;
  XRegs xt;
  XStage<R, NT> xs;  // one-hot observation path (gathers at commit)
  RowMeta meta;
  {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    w1r.issue(a.params + oW1, 0, nd.F, -1, row63);
    w2r.issue(a.params + oW2, 0, HID);
    const int n0 = row_index(blockIdx.x, wave, lane);
    float bias1 = 0.f, bias2 = 0.f, hv0 = 0.f, hv1 = 0.f, hb = 0.f;
    if (tid < HID) {
      bias1 = a.params[oB1 + tid];
      bias2 = a.params[oB2 + tid];
    }
    if (net == 0) {
      const int j0 = tid >> 3, k = tid & 7;  // elements tid and tid + 256 of the [64][8] block
      if (k < nk) {
        hv0 = a.params[lay.act_W + j0 * nk + k];
        hv1 = a.params[lay.act_W + (j0 + 32) * nk + k];
      }
      // padded logits get a -3e38 "bias": they drop out of softmax, entropy and every gradient with no special cases
      if (tid < 8) hb = (tid < nk) ? a.params[lay.act_b + tid] : -3.0e38f;
    } else {
      if (tid < HID) hv0 = a.params[lay.val_W + tid];
      if (tid == 0) hb = a.params[lay.val_b];
    }
    meta = row_scalars(n0);
    if (box) xt.issue(meta.phys, a.rb_obs, nd, lane);
    w1r.commit(bufC);
    w2r.commit(w2s);
    if (tid < HID) {
      b1s[tid] = bias1;
      b2s[tid] = bias2;
    }
    if (net == 0) {
      hw[head_row(tid >> 3) + (tid & 7)] = hv0;
      hw[head_row((tid >> 3) + 32) + (tid & 7)] = hv1;
      if (tid < 8) hbs[tid] = hb;
    } else {
      if (tid < HID) hw[tid] = hv0;
      if (tid == 0) hbs[0] = hb;
    }
  }

  f32x16 gW1 = {0}, gW2 = {0};
  float gh0 = 0.f;              // value: d val_W[j] partial of this wave
#if !PH_FAST_GHROWS
  float gh1 = 0.f;              // policy: gh0 / gh1 = d act_W[j][2w], [j][2w+1] over all rows
#else
  float ghr[NK];                // policy: d act_W[lane][k] partial over THIS wave's rows
#pragma unroll
  for (int k = 0; k < NK; ++k) ghr[k] = 0.f;
#endif
  float gb1 = 0.f, gb2 = 0.f;   // bias-gradient partials of this wave's 16 rows (lane = hidden unit)
  float ghb = 0.f;              // policy: d act_b[lane] partial (lane < 8) | value: d val_b partial
  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;

  bool first = true;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, first = false) {
    // thread coordinates are re-derived per tile from an opaque copy of threadIdx.x (stops the compiler from pinning
    // hundreds of loop-invariant LDS addresses in VGPRs across the tile loop)
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63, wave = tid >> 6;
    const int mt = wave >> 1, nt = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const bool has_next = tile + (int)gridDim.x < a.ntiles;

    // ---- T0: this tile's rows (gathered during the previous tile / the prologue) land in LDS ----
    if (lane < 16) {
      const int row = wave + 4 * lane;
      rowphys[row] = meta.phys;
      radv[row] = (norm && meta.phys >= 0) ? (meta.adv - adv_mean) / adv_den : meta.adv;
      rold[row] = meta.old;
      ract[row] = meta.act;
    }
    if (box) xt.template commit<FOLD>(bufA, meta.phys, nd, wave, lane);
    lds_barrier();
    if (!box) {
      xs.commit(bufA, rowphys, a.rb_obs, nd, 0, tid);
      lds_barrier();
    }
    if (first) PH_STAMP(a.prof, 1);

    // ---- S1: H1 = tanh(X W1 + b1) -> bufB ----
    {
      f32x16 acc = {0};
      acc = tile_mma<false, false, VALU>(bufA, LDH, bufC, LDH, mt * 32, nt * 32, 0, HID, acc, lane);
      const int col = nt * 32 + li;
      const float bb = FOLD ? 0.f : b1s[col];   // once: read inside the loop it is re-fetched, and waited for, behind every store to bufB
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(mt * 32 + drow(r, lh)) * LDH + col] = fast_tanh(FOLD ? acc[r] : acc[r] + bb);
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 2);

    // ---- S2: H2 = tanh(H1 W2 + b2) -> bufA ----
    const int n_next = has_next ? row_index(tile + gridDim.x, wave, lane) : -1;   // consumed in S6a
    {
      f32x16 acc = {0};
      acc = tile_mma<false, false, VALU>(bufB, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, acc, lane);
      const int col = nt * 32 + li;
      const float bb = b2s[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) bufA[(mt * 32 + drow(r, lh)) * LDH + col] = fast_tanh(acc[r] + bb);
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 3);

    // ---- SH: head forward, loss, dL/dhead, dZ2 = dH2 * (1 - H2^2) -> bufC; four lanes per row ----
    {
      const int r = tid >> 2, q = tid & 3;
      const bool valid = rowphys[r] >= 0;
      float h[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) h[m] = bufA[r * LDH + head_unit(q, m)];
      if (net == 0) {
        float z[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) z[k] = 0.f;
        for_head_rows<NK>(hw, q, [&](int m, const float4& w0, const float4& w1) {
          const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int k = 0; k < NK; ++k) z[k] = __builtin_fmaf(h[m], wk[k], z[k]);
        });
        float pr[NK];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          z[k] = quad_sum(z[k]) + hbs[k];     // (slots >= L, if NK is the padded 8: bias -3e38, they drop out of everything)
          mx = fmaxf(mx, z[k]);
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          pr[k] = fast_exp(z[k] - mx);
          se += pr[k];
        }
        const float lse = mx + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
        int act = (int)ract[r];
        act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
        float ent = 0.f, zact = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
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
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
        }
        float dz[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dz[k] = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const float dlogp = ((k == act) ? 1.f : 0.f) - pr[k];
          const float dent = -pr[k] * ((z[k] - lse) + ent);
          dz[k] = g_lp * dlogp + g_en * dent;
        }
        if (q == 0) {
          float4* o = reinterpret_cast<float4*>(dzs + r * 8);
          o[0] = make_float4(dz[0], dz[1], dz[2], dz[3]);
          if constexpr (NK > 4) o[1] = make_float4(dz[4], dz[5], dz[6], dz[7]);
        }
        for_head_rows<NK>(hw, q, [&](int m, const float4& w0, const float4& w1) {
          const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
          float d = dz[0] * wk[0];
#pragma unroll
          for (int k = 1; k < NK; ++k) d = __builtin_fmaf(dz[k], wk[k], d);
          bufC[r * LDH + head_unit(q, m)] = d * (1.0f - h[m] * h[m]);
        });
      } else {
        float wv[16];
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          wv[m] = hw[head_unit(q, m)];
          v = __builtin_fmaf(h[m], wv[m], v);
        }
        v = quad_sum(v) + hbs[0];
        const float retn = radv[r], oldv = rold[r];
        float vp = v, pass = 1.f;
        if (a.clip_vf >= 0.f) {
          const float dlt = v - oldv;
          pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
          vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
        }
        const float err = vp - retn;
        const float dv = valid ? a.vf_coef * 2.0f * err * inv_nb * pass : 0.f;
        if (valid && q == 0) st[1] += err * err;
        if (q == 0) dzs[r] = dv;
#pragma unroll
        for (int m = 0; m < 16; ++m) bufC[r * LDH + head_unit(q, m)] = dv * wv[m] * (1.0f - h[m] * h[m]);
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 4);

    // ---- S6a: dW2 += H1^T dZ2 ; dH1 = dZ2 W2^T ; d b2, d head weights, d head bias (VALU, beside the MFMAs) ----
    RowMeta meta_next = meta;
    if (has_next) {
      meta_next = row_scalars(n_next);                        // next tile's row scalars, committed at its T0
      if (first) PH_STAMP(a.prof, 8);
      w1r.issue(a.params + oW1, 0, nd.F, tid, row63);         // refill of bufC, committed in S6b
    }
    if (first) PH_STAMP(a.prof, 9);
    f32x16 dh1 = {0};
    {
      gb2 += lds_sum16(bufC + wave * 16 * LDH + lane, LDH);
      if (net == 0) {
#if PH_FAST_GHROWS
        // d act_W[lane][k] += sum over this wave's 16 rows of H2[row][lane] * dz[row][k]: the dz reads are wave-uniform
        // (LDS broadcasts), four rows of reads in flight, then their FMAs
        const float* hp = bufA + wave * 16 * LDH + lane;
        const float* dp = dzs + wave * 16 * 8;
#pragma unroll 1
        for (int r0 = 0; r0 < 16; r0 += 4, hp += 4 * LDH, dp += 4 * 8) {
          float hv[4];
          float4 da[4], db[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            hv[i] = hp[i * LDH];
            da[i] = *reinterpret_cast<const float4*>(dp + i * 8);
            if constexpr (NK > 4) db[i] = *reinterpret_cast<const float4*>(dp + i * 8 + 4);
            else db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float dk[8] = {da[i].x, da[i].y, da[i].z, da[i].w, db[i].x, db[i].y, db[i].z, db[i].w};
#pragma unroll
            for (int k = 0; k < NK; ++k) ghr[k] = __builtin_fmaf(hv[i], dk[k], ghr[k]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (lane < NK) ghb += lds_sum16(dzs + wave * 16 * 8 + lane, 8);
#else
        const float* hp = bufA + lane;
        const float* dp = dzs + 2 * wave;
#pragma unroll 1
        for (int r0 = 0; r0 < R; r0 += 8, hp += 8 * LDH, dp += 8 * 8) {  // a real loop: 8 rows of reads, then their FMAs
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
#endif
      } else {
        float hv[16], dv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          hv[i] = bufA[(wave * 16 + i) * LDH + lane];
          dv[i] = dzs[wave * 16 + i];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          gh0 = __builtin_fmaf(hv[i], dv[i], gh0);
          ghb += dv[i];
        }
      }
      if (first) PH_STAMP(a.prof, 10);
      gW2 = tile_mma<true, false, VALU>(bufB, LDH, bufC, LDH, mt * 32, nt * 32, 0, R, gW2, lane);
      if (first) PH_STAMP(a.prof, 11);
      dh1 = tile_mma<false, true, VALU>(bufC, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, dh1, lane);
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 5);

    // ---- S6b: dZ1 = dH1 * (1 - H1^2) in place; X back into bufA; W1 back into bufC ----
    {
      float hv[16];   // all reads, then all writes: interleaved, every read waits behind the previous (possibly aliasing) store
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[r] = bufB[(mt * 32 + drow(r, lh)) * LDH + nt * 32 + li];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(mt * 32 + drow(r, lh)) * LDH + nt * 32 + li] = dh1[r] * (1.0f - hv[r] * hv[r]);
    }
    if (box) xt.template commit<FOLD>(bufA, meta.phys, nd, wave, lane);
    else xs.commit(bufA, rowphys, a.rb_obs, nd, 0, tid);
    if (has_next) w1r.commit(bufC, tid);
    lds_barrier();
    if (first) PH_STAMP(a.prof, 6);

    // ---- S7: dW1 += X^T dZ1 ; d b1.  The next tile's rows are gathered underneath. ----
    meta = meta_next;
    if (has_next && box) xt.issue(meta.phys, a.rb_obs, nd, lane);
    if constexpr (!FOLD) gb1 += lds_sum16(bufB + wave * 16 * LDH + lane, LDH);   // FOLD: d b1 is row 63 of gW1
    gW1 = tile_mma<true, false, VALU>(bufA, LDH, bufB, LDH, mt * 32, nt * 32, 0, R, gW1, lane);
    lds_barrier();  // bufA / bufB / row scalars are free for the next tile
    if (first) PH_STAMP(a.prof, 7);
  }
  PH_STAMP(a.prof, 12);

  // ---- epilogue: accumulators -> slab (once), cross-wave sums in a fixed order ----
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the workgroup's slab half in the accumulators' own register order: 16-byte stores, 1 KB contiguous per wave instruction
    float* rslab = a.slabs + ((size_t)blockIdx.x * 2 + net) * RS_NET;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const int o = ((wave * 4 + r4) * 64 + lane) * 4;
      *reinterpret_cast<float4*>(rslab + RS_W2 + o) = make_float4(gW2[4 * r4], gW2[4 * r4 + 1], gW2[4 * r4 + 2], gW2[4 * r4 + 3]);
      *reinterpret_cast<float4*>(rslab + RS_W1 + o) = make_float4(gW1[4 * r4], gW1[4 * r4 + 1], gW1[4 * r4 + 2], gW1[4 * r4 + 3]);
    }
#if !PH_FAST_GHROWS
    if (net == 0) *reinterpret_cast<float2*>(rslab + RS_HW + lane * 8 + 2 * wave) = make_float2(gh0, gh1);
#endif
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) {
      float v = st[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      st[k] = v;
    }
    float* part = bufA;  // [5 + NK][4 waves][64]
    part[(0 * 4 + wave) * 64 + lane] = gb1;
    part[(1 * 4 + wave) * 64 + lane] = gb2;
    part[(2 * 4 + wave) * 64 + lane] = gh0;   // value net: d val_W partials
    part[(3 * 4 + wave) * 64 + lane] = ghb;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NSTATP; ++k) part[(4 * 4 + wave) * 64 + k] = st[k];
    }
#if PH_FAST_GHROWS
    if (net == 0) {
#pragma unroll
      for (int k = 0; k < NK; ++k) part[((5 + k) * 4 + wave) * 64 + lane] = ghr[k];
    }
#endif
    lds_barrier();
    auto wsum = [&](int which, int idx) {
      return ((part[(which * 4 + 0) * 64 + idx] + part[(which * 4 + 1) * 64 + idx]) + part[(which * 4 + 2) * 64 + idx]) +
             part[(which * 4 + 3) * 64 + idx];
    };
    if (tid < HID) {
      if constexpr (!FOLD) rslab[RS_B1 + tid] = wsum(0, tid);
      rslab[RS_B2 + tid] = wsum(1, tid);
      if (net == 1) rslab[RS_HW + tid] = wsum(2, tid);
    }
#if PH_FAST_GHROWS
    if (net == 0) {   // d act_W[j][k], j = tid & 63, k = tid >> 6 (+ 4): fixed-order sum of the four waves' row partials
#pragma unroll
      for (int k0 = 0; k0 < NK; k0 += 4) {
        const int k = k0 + (tid >> 6), j = tid & 63;
        if (k < NK) rslab[RS_HW + j * 8 + k] = wsum(5 + k, j);
      }
    }
#endif
    if (net == 0 && tid < 8) rslab[RS_HB + tid] = wsum(3, tid);
    if (net == 1 && tid == 0) rslab[RS_HB] = wsum(3, 0);
    if (tid < NSTATP) a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + tid] = wsum(4, tid);
  }
  PH_STAMP(a.prof, 13);
}

static size_t grad_fast_lds_bytes() {
  return sizeof(float) * (size_t)(4 * 64 * LDH + HW_FLOATS + 64 * 8 + 2 * HID + 16 + 3 * 64 + 64);
}

bool grad_fast_eligible(const NetDims& nd) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_GRAD_FAST");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled && !nd.gauss && nd.nchunk == 1 && nd.A == 1 && nd.L <= 8;
}

// FOLD variant: Box observations with a free 64th feature column
bool grad_fast_fold(const NetDims& nd) { return PH_FAST_FOLDB1 && nd.obs_kind == PH_SPACE_BOX && nd.F < HID; }

template <bool VALU, int NK, bool FOLD>
static hipError_t launch_fast_inst(const GradArgs& a, int nwg, hipStream_t s) {
  const size_t lds = grad_fast_lds_bytes();
  static bool allowed_dev[64] = {false};  // > 64 KiB of dynamic LDS is opt-in per kernel and device (kept out of graph capture)
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& allowed = allowed_dev[(dev >= 0 && dev < 64) ? dev : 0];
  if (!allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_fast_kernel<VALU, NK, FOLD>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  hipLaunchKernelGGL((ppo_grad_fast_kernel<VALU, NK, FOLD>), dim3(nwg, 2), dim3(256), lds, s, a);
  return hipGetLastError();
}
template <bool VALU, int NK>
static hipError_t launch_fast_nk(const GradArgs& a, int nwg, hipStream_t s) {
  return grad_fast_fold(a.nd) ? launch_fast_inst<VALU, NK, true>(a, nwg, s) : launch_fast_inst<VALU, NK, false>(a, nwg, s);
}
template <bool VALU>
static hipError_t launch_fast_variant(const GradArgs& a, int nwg, hipStream_t s) {
#if PH_FAST_NK
  switch (a.nd.L) {     // the head's loops are unrolled over the logits that exist
    case 1: return launch_fast_nk<VALU, 1>(a, nwg, s);
    case 2: return launch_fast_nk<VALU, 2>(a, nwg, s);
    case 3: return launch_fast_nk<VALU, 3>(a, nwg, s);
    case 4: return launch_fast_nk<VALU, 4>(a, nwg, s);
    case 5: return launch_fast_nk<VALU, 5>(a, nwg, s);
    case 6: return launch_fast_nk<VALU, 6>(a, nwg, s);
    case 7: return launch_fast_nk<VALU, 7>(a, nwg, s);
    default: break;
  }
#endif
  return launch_fast_nk<VALU, 8>(a, nwg, s);
}

// slab position -> parameter index (-1 = padding) for both nets of one workgroup's register-order slab: [net][RS_NET]
void grad_slab_map(const ph_layout& lay, int* map, bool fold) {
  const int F = lay.F, L = lay.L;
  for (int net = 0; net < 2; ++net) {
    int* m = map + net * RS_NET;
    for (int i = 0; i < RS_NET; ++i) m[i] = -1;
    const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
    const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
    for (int s = 0; s < HID * HID; ++s) {   // position = ((wave * 4 + r4) * 64 + lane) * 4 + j holds accumulator r = 4 * r4 + j
      const int wave = s >> 10, r4 = (s >> 8) & 3, lane = (s >> 2) & 63, j = s & 3;
      const int r = 4 * r4 + j, mt = wave >> 1, nt = wave & 1;
      const int k = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = nt * 32 + (lane & 31);
      m[RS_W2 + s] = oW2 + k * HID + col;
      if (k < F) m[RS_W1 + s] = oW1 + k * HID + col;
      else if (fold && k == HID - 1) m[RS_W1 + s] = oB1 + col;   // the bias row of the staged W1 (grad_fast_fold)
    }
    for (int i = 0; i < HID; ++i) {
      if (!fold) m[RS_B1 + i] = oB1 + i;
      m[RS_B2 + i] = oB2 + i;
    }
    if (net == 0) {
      for (int j = 0; j < HID; ++j)
        for (int k = 0; k < L && k < 8; ++k) m[RS_HW + j * 8 + k] = lay.act_W + j * L + k;
      for (int k = 0; k < L && k < 8; ++k) m[RS_HB + k] = lay.act_b + k;
    } else {
      for (int j = 0; j < HID; ++j) m[RS_HW + j] = lay.val_W + j;
      m[RS_HB] = lay.val_b;
    }
  }
}

hipError_t launch_ppo_grad_fast(const GradArgs& a, int nwg, int gemm_mode, hipStream_t s) {
  return gemm_mode == 1 ? launch_fast_variant<true>(a, nwg, s) : launch_fast_variant<false>(a, nwg, s);
}

}  // namespace ph

// K4: ActorCriticPolicy.forward / evaluate_actions for the SB3 MlpPolicy (two separate 64-64 tanh MLPs),
// fused with categorical sampling / log-prob / entropy, the optional action-mask logit offset and the
// RolloutBuffer.add row write.   Reference call sites: pantheonrl/common/util.py:63-81,
// pantheonrl/common/agents.py:162,172-179; structure pantheonrl/algos/modular/policies.py:273-290,364-383.
//
// Grid: (row tiles of R rows, 2).  blockIdx.y = 0 -> policy net workgroup (logits, sampling, log-prob),
// 1 -> value net workgroup (values + the observation copy into the rollout buffer).  Each workgroup keeps the
// 64x64 weight blocks in LDS and runs every layer as 32x32 v_mfma_f32_32x32x2_f32 tiles, one tile per wave.
#include <cstring>
#include <initializer_list>
#include "ph_rowtail.h"
#include "ph_liar_group.h"

namespace ph {


// General action head of one row, one lane per row, the row's logits z[0..L) in LDS (modified in place): optional mask
// offset, logits output, then per action component sampling / argmax / given action, log-prob, entropy and the
// rollout-buffer writes (Discrete and MultiDiscrete; Discrete with <= 8 logits takes the register path)
__device__ __forceinline__ void general_row_tail(const FwdArgs& a, const NetDims& nd, int g, float* z, uint64_t ctr) {
  const bool small = !nd.gauss && nd.A == 1 && nd.L <= 8;
  if (a.mask && !small) {  // modular/policies.py:330-333 : logits - 30*(~mask)
    for (int k = 0; k < nd.L; ++k) z[k] = z[k] - 30.0f * (1.0f - (float)(a.mask[(size_t)g * nd.L + k] != 0));
  }
  if (a.logits && !small)
    for (int k = 0; k < nd.L; ++k) a.logits[(size_t)g * nd.L + k] = z[k];
  float logp = 0.f, ent = 0.f;
  if (nd.gauss) {
    // Box action space: SB3's DiagGaussianDistribution -- z[0..A) are the means, log_std[A] follows val_b in the parameter vector.
    // action = mean + exp(log_std) * eps; log-prob and entropy are the sums over the dimensions of Normal's.  `uniforms` carries
    // the STANDARD-NORMAL draws eps when given (teacher forcing), else Box-Muller over two Philox uniforms per dimension.
    const float* ls = a.params + nd.lay.val_b + 1;
    for (int c = 0; c < nd.A; ++c) {
      const float mu = z[c], lsd = ls[c];
      float act;
      if (a.given_actions) act = a.given_actions[(size_t)g * nd.A + c];
      else if (a.deterministic) act = mu;
      else {
        float eps;
        if (a.uniforms) eps = a.uniforms[(size_t)g * nd.A + c];
        else {
          const float u1 = philox_uniform(a.seed, ctr, (uint32_t)g, (uint32_t)c);
          const float u2 = philox_uniform(a.seed, ctr, (uint32_t)g, (uint32_t)(c + 128));
          eps = __builtin_sqrtf(-2.0f * fast_log(fmaxf(u1, 1.0e-30f))) * __builtin_cosf(6.28318530717958647692f * u2);
        }
        act = mu + fast_exp(lsd) * eps;
      }
      const float d = (act - mu) * fast_exp(-lsd);
      logp += (-0.5f * d * d - lsd) - 0.91893853320467274178f;   // - log sqrt(2 pi)
      ent += 1.41893853320467274178f + lsd;                       // 0.5 + 0.5 log(2 pi) + log_std
      if (a.act_f32) a.act_f32[(size_t)g * nd.A + c] = act;
      if (a.rb_act) {
        const long long ridx = rb_row(a, g);
        if (ridx >= 0) a.rb_act[(size_t)ridx * nd.A + c] = act;
      }
    }
  } else if (small) {
    // fast path (Discrete action space, <= 8 logits): the row lives in registers, one exp per logit
    float zr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) zr[k] = (k < nd.L) ? z[k] : 0.f;
    discrete8_row_tail(a, nd, g, zr, fwd_counter(a));
    return;
  } else
  for (int c = 0; c < nd.A; ++c) {
    const int lo = nd.act_off[c], nk = nd.act_off[c + 1] - lo;
    float m = z[lo];
    for (int k = 1; k < nk; ++k) m = fmaxf(m, z[lo + k]);
    float se = 0.f;
    for (int k = 0; k < nk; ++k) se += fast_exp(z[lo + k] - m);
    const float lse = m + fast_log(se);
    int act;
    if (a.given_actions) {
      act = (int)a.given_actions[(size_t)g * nd.A + c];
      act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
    } else if (a.deterministic) {
      act = 0;
      float best = z[lo];
      for (int k = 1; k < nk; ++k)
        if (z[lo + k] > best) { best = z[lo + k]; act = k; }
    } else {
      const float u = a.uniforms ? a.uniforms[(size_t)g * nd.A + c]
                                 : philox_uniform(a.seed, ctr, (uint32_t)g, (uint32_t)c);
      float cum = 0.f;
      act = 0;
      for (int k = 0; k < nk - 1; ++k) {  // inverse CDF: count prefix sums <= u
        cum += fast_exp(z[lo + k] - lse);
        act += (u >= cum) ? 1 : 0;
      }
    }
    float e = 0.f;
    for (int k = 0; k < nk; ++k) {
      const float lp = z[lo + k] - lse;
      e -= fast_exp(lp) * lp;
    }
    logp += z[lo + act] - lse;
    ent += e;
    if (a.act_i32) a.act_i32[(size_t)g * nd.A + c] = act;
    if (a.act_f32) a.act_f32[(size_t)g * nd.A + c] = (float)act;
    if (a.rb_act) {
      const long long ridx = rb_row(a, g);
      if (ridx >= 0) a.rb_act[(size_t)ridx * nd.A + c] = (float)act;
    }
  }
  if (a.logp) a.logp[g] = logp;
  if (a.entropy) a.entropy[g] = ent;
  if (a.rb_logp) {
    const long long ridx = rb_row(a, g);
    if (ridx >= 0) a.rb_logp[ridx] = logp;
  }
}

template <int R, int LP, bool VALU>
__device__ __forceinline__ void policy_fwd_body(const FwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = R * 4;
  const NetDims& nd = a.nd;
  constexpr int Lp = LP, LDO = LP + 1;   // compile-time so LDS offsets fold into the ds_read immediates
  float* bufA = smem;                    // [R][LDH]  X chunk, later H2
  float* bufB = bufA + R * LDH;          // [R][LDH]  H1
  float* w1s = bufB + R * LDH;           // [64][LDH] W1 chunk
  float* w2s = w1s + HID * LDH;          // [64][LDH]
  float* wos = w2s + HID * LDH;          // Wo [64][LDO]
  float* outs = wos + HID * LDO;         // OUT [R][LDO]
  float* b1s = outs + R * LDO;           // [64]
  float* b2s = b1s + HID;                // [64]
  float* bos = b2s + HID;                // [Lp] (policy) / val_W[64] (value)
  int* rowphys = (int*)(bos + 64);       // [R]

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int mt = wave >> 1, nt = wave & 1;
  const int net = blockIdx.y;
  const int row0 = blockIdx.x * R;
  const ph_layout& lay = nd.lay;
  const float* W1 = a.params + (net == 0 ? lay.pi_W1 : lay.vf_W1);
  const float* B1 = a.params + (net == 0 ? lay.pi_b1 : lay.vf_b1);
  const float* W2 = a.params + (net == 0 ? lay.pi_W2 : lay.vf_W2);
  const float* B2 = a.params + (net == 0 ? lay.pi_b2 : lay.vf_b2);

  PH_STAMP(a.prof, 0);
  // ---- every staging load of the kernel is issued here, back to back: one memory latency in total ----
  if (tid < R) rowphys[tid] = (row0 + tid < a.n) ? row0 + tid : -1;
  WStage<NT> w2r, w1r;
  WoStage<NT> wor;
  XStage<R, NT> xr;
  w2r.issue(W2, 0, HID);
  w1r.issue(W1, 0, nd.F);
  if (net == 0) wor.issue(a.params + lay.act_W, nd.L, Lp);
  float bias1 = 0.f, bias2 = 0.f, bias3 = 0.f;
  if (tid < HID) {
    bias1 = B1[tid];
    bias2 = B2[tid];
    bias3 = (net == 0) ? ((tid < nd.L) ? a.params[lay.act_b + tid] : 0.f) : a.params[lay.val_W + tid];
  }
  PH_STAMP(a.prof, 8);
  __syncthreads();  // rowphys visible
  PH_STAMP(a.prof, 9);
  xr.issue(rowphys, a.obs, nd, 0);
  w2r.commit(w2s);
  PH_STAMP(a.prof, 10);
  w1r.commit(w1s);
  if (net == 0) wor.commit(wos, Lp, LDO);
  if (tid < HID) {
    b1s[tid] = bias1;
    b2s[tid] = bias2;
    bos[tid] = bias3;
  }
  PH_STAMP(a.prof, 11);
  xr.commit(bufA, rowphys, a.obs, nd, 0);
  __syncthreads();
  PH_STAMP(a.prof, 1);

  // ---- layer 1: Z1 = X * W1 (feature chunks of 64 accumulated in the MFMA accumulator) ----
  f32x16 acc = {0};
  for (int c = 0; c < nd.nchunk; ++c) {
    if (c > 0) {
      __syncthreads();  // previous chunk consumed
      w1r.issue(W1, c * HID, nd.F);
      xr.issue(rowphys, a.obs, nd, c);
      w1r.commit(w1s);
      xr.commit(bufA, rowphys, a.obs, nd, c);
      __syncthreads();
    }
    PH_STAMP(a.prof, 2);
    acc = tile_mma<false, false, VALU>(bufA, LDH, w1s, LDH, mt * 32, nt * 32, 0, HID, acc);
  }
  PH_STAMP(a.prof, 3);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = mt * 32 + drow(r, lh), col = nt * 32 + li;
    bufB[row * LDH + col] = fast_tanh(acc[r] + b1s[col]);
  }
  __syncthreads();
  PH_STAMP(a.prof, 4);

  // ---- layer 2 ----
  f32x16 acc2 = {0};
  acc2 = tile_mma<false, false, VALU>(bufB, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, acc2);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = mt * 32 + drow(r, lh), col = nt * 32 + li;
    bufA[row * LDH + col] = fast_tanh(acc2[r] + b2s[col]);
  }
  __syncthreads();
  PH_STAMP(a.prof, 5);

  if (net == 1) {
    // ---- value head: v = H2 . val_W + val_b (VALU dot, one lane per row) ----
    if (tid < R && rowphys[tid] >= 0) {
      float v = 0.f;
      for (int j = 0; j < HID; ++j) v = __builtin_fmaf(bufA[tid * LDH + j], bos[j], v);
      value_row_tail(a, row0 + tid, v + a.params[lay.val_b]);
    }
    copy_obs_rows(a, row0, (a.n - row0 < R) ? a.n - row0 : R, nd.D);
    PH_STAMP(a.prof, 7);
    return;
  }

  // ---- policy head: logits = H2 * act_W + act_b  (tiles: R/32 x Lp/32) ----
  {
    const int ntn = Lp >> 5;
    if (wave < (R >> 5) * ntn) {
      const int hm = wave / ntn, hn = wave - hm * ntn;
      f32x16 acc3 = {0};
      acc3 = tile_mma<false, false, VALU>(bufA, LDH, wos, LDO, hm * 32, hn * 32, 0, HID, acc3);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = hm * 32 + drow(r, lh), col = hn * 32 + li;
        outs[row * LDO + col] = acc3[r] + bos[col];
      }
    }
  }
  __syncthreads();
  PH_STAMP(a.prof, 6);

  // ---- distribution: one lane per row ----
  if (tid < R && rowphys[tid] >= 0) general_row_tail(a, nd, row0 + tid, outs + tid * LDO, fwd_counter(a));
  PH_STAMP(a.prof, 7);
}


// ---- 16-row variant for the rollout step (single feature chunk, Discrete head with <= 8 logits) -------------------------
// The step forward of E = 1024 environments is a latency chain, not a throughput problem: 32-row workgroups leave 3/4 of the
// CUs idle and every layer costs a 32-MFMA dependent chain per wave.  Here a workgroup owns 16 rows and each of its four
// waves one 16-column output tile of a layer: 16 v_mfma_f32_16x16x4_f32 in two interleaved accumulator chains per
// layer, twice the workgroups, and the head runs as a register-resident VALU phase (four lanes per row) in one wave.
__device__ __forceinline__ void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// s_waitcnt vmcnt(0) as the BUILTIN (gfx9 encoding: vmcnt 0, expcnt 7, lgkmcnt 15): the wait-insertion pass accounts for it
__device__ __forceinline__ void vm_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }

template <int CTRL>
__device__ __forceinline__ float dpp_quad_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum_f(float v) {
  v += dpp_quad_f<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp_quad_f<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}

// sc != nullptr: the scripted rollout -- the weights are fetched once and the step below runs sc->n_steps times, each time on
// the next rows of the observation / reward / done sequences and of the rollout buffer (see ScriptedSteps)
// px_persistent (with px and sc): the exchange rollout -- step t pushes its actions as stamp-in-band words of step t and the
// value workgroups consume the words of step t - 1 (t >= 1), all inside the loop; the last step's reward, which the
// launch-by-launch walk credits after the rollout from the unpacked joint action, is credited at the end of the loop.
//
// Round 4: no weight ever sits in LDS.  A lane's B operands of the three products of a step -- W1[g + 4s][16 wave + c],
// W2[g + 4s][16 wave + c] and, in wave 0, the head's weights [g + 4s][c] -- do not depend on the step: they are fetched from the
// parameters once per launch into 48 registers (the scripted rollouts keep them for all n_steps steps: 32 LDS reads per layer and
// step gone; the single-step launch loses the 33 KB weight staging that was 40 % of its time).  The head is a third 16x16x4 MFMA
// product (z = H2 act_W, or H2 val_W in column 0) instead of 128 FMAs per lane over LDS-resident weights.  Round 6, fourth session:
// it is split over the four waves' K quarters (wave w: units 16 w .. 16 w + 15, four MFMAs) instead of running as a sixteen-deep
// chain in wave 0 beside three waiting waves; the four 16 x 8 partial tiles go through 2 KB of LDS and lane r < 16 of wave 0 adds
// them in one fixed order, (p0 + p1) + (p2 + p3), and owns row r for the distribution tail.  The head phase's barrier is also the
// point behind which nobody reads H2, so the barrier at the top of the next step is gone.  Every form of the kernel (per step,
// one launch, exchange; MFMA and VALU cross-check) shares the code, so they stay bitwise one another; per-step launches gain most
// (twelve fewer weight registers to fetch per launch as well): stepwise 92.9 -> 99.9 M agent-steps/s (profiles/r06_bd_*).
// LEAN: the launch was checked (fwd_args_lean) to use none of the optional paths of the argument record -- no masks, no teacher-forced
// uniforms or actions, no logits / entropy / float-action outputs, a rectangular buffer with the fused add, no joint-action reward, no
// host doorbell, no debug stamps.  The copy of the record the body works on then carries those fields as CONSTANTS: every
// `if (a.mask)` of the row tails folds away instead of being a basic block of its own behind a restored scalar register (the
// headline rollout's step held ~200 v_readlane restores and ~60 such blocks, 8 copies of the tail apart), and the record no longer
// competes for the 100 scalar registers.  Same arithmetic, same stores: bitwise the general form (the rollout tests run both).
template <bool KEEP_JOINT>
__device__ __forceinline__ void fwd_args_pin_lean(FwdArgs& a) {
  a.mask = nullptr;
  a.uniforms = nullptr;
  a.given_actions = nullptr;
  a.deterministic = 0;
  a.act_f32 = nullptr;
  a.entropy = nullptr;
  a.logits = nullptr;
#ifndef PH_LEAN_PROF   // (scripts/rollout_phase.py's view of the lean form: a -DPH_LEAN_PROF build keeps the stamp pointer)
  a.prof = nullptr;
#endif
  a.pos_env = nullptr;
  a.rec_mask = nullptr;
  if constexpr (!KEEP_JOINT) {   // (the exchange rollout's value rows read the joint action: LEAN = 2 keeps those fields)
    a.joint = nullptr;
    a.joint_ll = nullptr;
  }
  a.env_mask = nullptr;
  a.host_done = nullptr;
  __builtin_assume(a.rb_obs != nullptr);
  __builtin_assume(a.rb_act != nullptr);
  __builtin_assume(a.rb_rew != nullptr);
  __builtin_assume(a.rb_es != nullptr);
  __builtin_assume(a.rb_val != nullptr);
  __builtin_assume(a.rb_logp != nullptr);
  __builtin_assume(a.es_in != nullptr);
}
bool fwd_args_lean(const FwdArgs& a, bool joint_ok = false) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_ROLLOUT_LEAN");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
#ifdef PH_LEAN_PROF
  constexpr bool lean_prof = true;
#else
  constexpr bool lean_prof = false;
#endif
  return enabled && !a.mask && !a.uniforms && !a.given_actions && !a.deterministic && !a.act_f32 && !a.entropy && !a.logits && (lean_prof || !a.prof) &&
         !a.pos_env && !a.rec_mask && (joint_ok || (!a.joint && !a.joint_ll)) && !a.env_mask && !a.host_done && a.rb_obs && a.rb_act && a.rb_rew &&
         a.rb_es && a.rb_val && a.rb_logp && a.es_in;
}

// The head product of the 16-row forward split over the four waves' K quarters (1) or as one sixteen-MFMA chain in wave 0 (0: the
// form up to round 6's third session; A/B switch, scripts/build_variants.sh)
#ifndef PH_FWD16_HEAD_KSPLIT
#define PH_FWD16_HEAD_KSPLIT 1
#endif
constexpr int ZS_FLOATS = (PH_FWD16_HEAD_KSPLIT ? 4 : 1) * 16 * 8;   // the head's output tile(s) in LDS

template <bool VALU, int LEAN = 0>
__device__ __forceinline__ void policy_fwd16_body(const FwdArgs& a0, const ph_p2p* px = nullptr, int px_t = 0, int px_a_local = 0,
                                                  int agent = 0, const ScriptedSteps* sc = nullptr, int px_persistent = 0) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int R = 16, NT = 256;
  const NetDims& nd = a0.nd;
  FwdArgs a = a0;
  if constexpr (LEAN == 1) fwd_args_pin_lean<false>(a);
  if constexpr (LEAN == 2) fwd_args_pin_lean<true>(a);
#ifdef PH_LEAN_PROF
  long long* const prof0 = a0.prof;
#else
  long long* const prof0 = LEAN ? nullptr : a0.prof;
#endif
  float* xs = smem;                 // [16][LDH]  X, later H2
  float* hs = xs + R * LDH;         // [16][LDH]  H1
  float* zs = hs + R * LDH;         // [4][16][8] head output, one partial per wave (its quarter of the units; wave 0's carries the bias):
                                    //            policy logits | value in column 0
  int* rowphys = (int*)(zs + ZS_FLOATS);  // [16]
  float* upre = (float*)(rowphys + 16);   // [2][16] sampling uniforms of the rows, drawn a step ahead (scripted rollouts)
  unsigned long long** pxll = (unsigned long long**)(upre + 32);   // [PH_MAX_RANKS] every rank's receive area (exchange forms)

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int net = blockIdx.y;
  const int row0 = blockIdx.x * R;
  const ph_layout& lay = nd.lay;
  const float* W1 = a.params + (net == 0 ? lay.pi_W1 : lay.vf_W1);
  const float* B1 = a.params + (net == 0 ? lay.pi_b1 : lay.vf_b1);
  const float* W2 = a.params + (net == 0 ? lay.pi_W2 : lay.vf_W2);
  const float* B2 = a.params + (net == 0 ? lay.pi_b2 : lay.vf_b2);
  const int nk = nd.L;

  PH_STAMP(a.prof, 0);
  // every global load of the kernel is issued here; row indices are trivial (row0 + r), so X needs no metadata pass
  if (tid < R) rowphys[tid] = (row0 + tid < a.n) ? row0 + tid : -1;
  // The exchange descriptor lives in DEVICE memory (P2PStep): every `px->field` is a global load, and inside the step loop the
  // compiler cannot hoist them past the loop's own global stores -- the policy rows' push walked four dependent round trips per
  // step (world, count, rank, ll[p]).  Its scalars are read once here, the peers' area pointers staged in LDS.
  // (LEAN forms only: the general form of the exchange rollout is built for three waves per SIMD and has no register to spare)
  constexpr bool PXH = LEAN != 0;
  const int pxw_h = (PXH && px) ? px->world : 0, pxc_h = (PXH && px) ? px->count : 0, pxr_h = (PXH && px) ? px->rank : 0;
  const int pxT_h = (PXH && px) ? px->T : 1, pxslots_h = (PXH && px) ? px->ll_slots : 1;
  unsigned long long* const ll_self_h = (PXH && px) ? px->ll[pxr_h] : nullptr;
  if (PXH && px && tid < pxw_h && tid < PH_MAX_RANKS) pxll[tid] = px->ll[tid];
#define pxw (PXH ? pxw_h : px->world)
#define pxc (PXH ? pxc_h : px->count)
#define pxr (PXH ? pxr_h : px->rank)
#define pxT (PXH ? pxT_h : px->T)
#define pxslots (PXH ? pxslots_h : px->ll_slots)
#define ll_self (PXH ? ll_self_h : px->ll[px->rank])
#define PXLL(p) (PXH ? pxll[p] : px->ll[p])
  // B operands: element [g + 4s][this lane's column] of W1 and W2; of the head, the K quarter of this wave (PH_FWD16_HEAD_KSPLIT):
  // units 16 wave + g + 4j, j < 4 -- the head product is then four MFMAs in every wave instead of sixteen in wave 0 with three
  // waves waiting, its partial tiles meet in LDS (zs) and are added in ONE fixed order by the row's lane
#if PH_FWD16_HEAD_KSPLIT
  constexpr int NBH = 4;
#else
  constexpr int NBH = 16;
#endif
  float bw1[16], bw2[16], bwh[NBH];
  const int col = 16 * wave + c;
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    const int k = g + 4 * s;
    bw1[s] = (k < nd.F) ? W1[k * HID + col] : 0.f;
    bw2[s] = W2[k * HID + col];
  }
#pragma unroll
  for (int s = 0; s < NBH; ++s) {
    const int k = g + 4 * (PH_FWD16_HEAD_KSPLIT ? 4 * wave + s : s);
    bwh[s] = 0.f;
    if (PH_FWD16_HEAD_KSPLIT || wave == 0) {
      if (net == 0) {
        if (c < nk) bwh[s] = a.params[lay.act_W + k * nk + c];
      } else if (c == 0) {
        bwh[s] = a.params[lay.val_W + k];
      }
    }
  }
  const float bias1 = B1[col], bias2 = B2[col];
  float hbias = 0.f;
  if (wave == 0) {
    if (net == 0) hbias = (c < nk) ? a.params[lay.act_b + c] : 0.f;
    else hbias = (c == 0) ? a.params[lay.val_b] : 0.f;
  }
  XStage<R, NT> xr;
  lds_only_barrier();  // rowphys visible
  if (sc) a.obs = sc->obs_seq;
  xr.issue(rowphys, a.obs, nd, 0);
  xr.commit(xs, rowphys, a.obs, nd, 0);
  // Every load issued so far (weights, biases, the first rows) is waited for HERE, once, as an instruction the compiler's wait
  // insertion can see: otherwise the registers loaded ahead of the step loop stay "pending" at its header, and every step waits
  // `vmcnt(k)` inside layer 1 and `vmcnt(0)` before the bias adds -- which, with ONE counter for loads and stores on gfx950, drains
  // the previous step's STORES in the middle of the products (round 6, third session: profiles/r06_al_*)
  vm_drain();
  lds_only_barrier();
  PH_STAMP(a.prof, 1);

  const int n_steps = sc ? sc->n_steps : 1;
  unsigned long long epoch_hi = 0ull;
  if (sc) {   // the RNG epoch word is constant for the launch: one read instead of one per sampling tail
    epoch_hi = a0.epoch ? (unsigned long long)(*a0.epoch) << 32 : 0ull;
    a.epoch = nullptr;
    a.counter = a0.counter + epoch_hi;
  }
  const unsigned long long px_epoch = (px && px_persistent) ? *px->epoch : 0ull;   // constant for the launch as well
  // (scripted rollouts) the sampling uniform of step t + 1 is a function of (seed, counter + t + 1, row) alone: wave 1, idle
  // during the head phase of the policy workgroup, draws it a step ahead, so the ten Philox rounds leave the one lane per row
  // that walks softmax -> inverse CDF -> log-prob.  Same call, same value: the rollout stays bitwise the launch-by-launch walk.
  const bool draw_ahead = sc && net == 0 && !a.uniforms && !a.deterministic && !a.given_actions;   // (launch constants: a0's, or LEAN's)
  auto draw_uniforms = [&](int t1) {
    if (lane < R && row0 + lane < a0.n)
      upre[(t1 & 1) * R + lane] = philox_uniform(a0.seed, a0.counter + (unsigned long long)t1 + epoch_hi, (uint32_t)(row0 + lane), 0u);
  };
  if (draw_ahead && wave == 1) draw_uniforms(0);   // visible to wave 0 after the barriers of step 0's layers

  // one 16x16 output tile per wave: D[row 4g+r][this lane's column] = sum_k A[row][k] B[k][column]; two accumulator chains
  auto product = [&](const float* A, const float (&bw)[16]) -> f32x4 {
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + c * LDH + g;
    float av[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) av[s] = ap[4 * s];
    // all sixteen operand reads in flight before the first MFMA: left alone, the scheduler sinks each ds_read2 in front of the two
    // MFMAs that use it and recycles ONE register pair -- read, wait, two MFMAs, eight times per product: eight exposed LDS round
    // trips per layer of a step whose whole point is latency (round 6: one-launch rollout 0.333 -> 0.307 ms, profiles/r06_ah_*)
#ifndef PH_FWD16_NO_HOIST   // (A/B switch: scripts/build_variants.sh)
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      e = mma16<VALU>(av[s], bw[s], e, lane);
      o = mma16<VALU>(av[s + 1], bw[s + 1], o, lane);
    }
    return e + o;
  };

  for (int t = 0; t < n_steps; ++t) {
  // debug stamps of ONE step in the middle of a scripted rollout (slots 8..15: scripts/rollout_phase.py)
  long long* const pstep = (sc && t == 8) ? prof0 : nullptr;
  PH_STAMP(pstep, 8);
  if (t > 0) {   // (scripted rollout) the next step's argument record and observation rows; the weights stay where they are
    const size_t row = (size_t)t * a0.n;
    a.obs = sc->obs_seq + row * nd.D;
    a.counter = a0.counter + (unsigned long long)t + epoch_hi;
    a.rb_obs = a0.rb_obs + row * nd.D;
    a.rb_act = a0.rb_act + row * nd.A;
    a.rb_rew = a0.rb_rew + row;
    a.rb_es = a0.rb_es + row;
    a.rb_val = a0.rb_val + row;
    a.rb_logp = a0.rb_logp + row;
    a.es_in = sc->done_seq + (row - a0.n);          // Agent.update(reward, done) of the previous step:
    a.pending_reward = sc->rew_seq + (row - a0.n);  //   last_episode_starts = done, rewards[pos - 1] += reward
    a.prev_rew = a0.rb_rew + (row - a0.n);
    if constexpr (!LEAN) {
      a.mask = (sc->mask_seq && sc->mask_policy) ? sc->mask_seq + row * nd.L : nullptr;
      a.env_mask = (sc->mask_seq && sc->mask_env) ? sc->mask_seq + row * nd.L : nullptr;
    }
    if (px_persistent && a0.joint) {   // the joint action of step t - 1 as stamp-in-band words (value_row_tail polls them)
      a.joint = a0.joint;
      a.joint_ll = ll_self + (size_t)p2p_persistent_slot(px_epoch, pxT, t - 1) * pxw * pxc;
      a.ll_t = t - 1;
    }
    a.prof = nullptr;
#if !PH_FWD16_HEAD_KSPLIT
    lds_only_barrier();   // the previous step's head is done with xs (H2)
#endif                    // (K-split head: every wave read its part of H2 before the head phase's own barrier)
    xr.issue(rowphys, a.obs, nd, 0);
    xr.commit(xs, rowphys, a.obs, nd, 0);
    vm_drain();   // the rows are in (the commit waited for them): say so, or the observation copy at the end of the step waits
                  // `vmcnt(0)` for registers that arrived a microsecond ago and drains the row tail's stores instead
    lds_only_barrier();
  }
  PH_STAMP(pstep, 9);

  // value workgroup, rectangular rollout: what the row tail reads (previous done, pending reward, the reward row it adds to) is
  // fetched here, under the layers; RolloutBuffer.add's observation copy of Box rows goes out straight from the staged registers
  // (the row was read once: no second trip to the observation)
  const bool pre_ok = net == 1 && wave == 0 && lane < R && row0 + lane < a.n && !a.pos_env && a.rb_val;
  ValuePre vpre = {0.f, 0.f, 0.f, 0ull, 0ull, 0u, 0, 0};
  if (pre_ok) vpre = value_row_preload(a, row0 + lane);
  const bool copy_from_regs = net == 1 && a.rb_obs && !a.pos_env && nd.obs_kind == PH_SPACE_BOX;
  {
    const f32x4 z1 = product(xs, bw1);
#pragma unroll
    for (int r = 0; r < 4; ++r) hs[(4 * g + r) * LDH + col] = fast_tanh(z1[r] + bias1);
  }
  lds_only_barrier();
  PH_STAMP(a.prof, 3);
  PH_STAMP(pstep, 10);
  {
    const f32x4 z2 = product(hs, bw2);
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(4 * g + r) * LDH + col] = fast_tanh(z2[r] + bias2);   // X is dead: H2 over it
  }
  lds_only_barrier();
  PH_STAMP(a.prof, 5);
  PH_STAMP(pstep, 11);

  // ---- head: a third product (columns = logits, or the value in column 0) ----
#if PH_FWD16_HEAD_KSPLIT
  {   // every wave: its K quarter (four MFMAs, two chains), the partial tile to zs[wave]; wave 0's partial carries the bias
    if (wave == 0 && pre_ok) value_row_preload_words(a, row0 + lane, vpre);   // (exchange rollouts: the joint action's words, under the product)
    const float* ap = xs + c * LDH + g + 16 * wave;
    float av[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) av[j] = ap[4 * j];
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    e = mma16<VALU>(av[0], bwh[0], e, lane);
    o = mma16<VALU>(av[1], bwh[1], o, lane);
    e = mma16<VALU>(av[2], bwh[2], e, lane);
    o = mma16<VALU>(av[3], bwh[3], o, lane);
    const f32x4 zh = e + o;
    if (c < 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) zs[(wave * R + 4 * g + r) * 8 + c] = zh[r] + hbias;   // (hbias is 0 outside wave 0)
    }
  }
  lds_only_barrier();   // the four partial tiles are in; nobody reads xs (H2) behind this point
  PH_STAMP(pstep, 12);
  if (wave == 0) {
    PH_STAMP(pstep, 13);
    const int r = lane, grow = row0 + lane;   // lane r < 16 owns row r
    if (r < R && grow < a.n) {
      // the row's logits: (partial 0 + partial 1) + (partial 2 + partial 3), the same tree in every form of this kernel
      float z[8];
      {
        float4 q[4][2];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          q[w][0] = *reinterpret_cast<const float4*>(zs + (w * R + r) * 8);
          q[w][1] = *reinterpret_cast<const float4*>(zs + (w * R + r) * 8 + 4);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          z[4 * h + 0] = (q[0][h].x + q[1][h].x) + (q[2][h].x + q[3][h].x);
          z[4 * h + 1] = (q[0][h].y + q[1][h].y) + (q[2][h].y + q[3][h].y);
          z[4 * h + 2] = (q[0][h].z + q[1][h].z) + (q[2][h].z + q[3][h].z);
          z[4 * h + 3] = (q[0][h].w + q[1][h].w) + (q[2][h].w + q[3][h].w);
        }
      }
      if (net == 0) {
#else
  if (wave == 0) {
    if (pre_ok) value_row_preload_words(a, row0 + lane, vpre);   // (exchange rollouts: the joint action's words, under the head product)
    const f32x4 zh = product(xs, bwh);
    PH_STAMP(pstep, 12);
    if (c < 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) zs[(4 * g + r) * 8 + c] = zh[r] + hbias;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // zs is written and read by this wave only
    __builtin_amdgcn_wave_barrier();
    PH_STAMP(pstep, 13);
    const int r = lane, grow = row0 + lane;   // lane r < 16 owns row r
    if (r < R && grow < a.n) {
      if (net == 0) {
        const float4 z0 = *reinterpret_cast<const float4*>(zs + r * 8), z1 = *reinterpret_cast<const float4*>(zs + r * 8 + 4);
        float z[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#endif
        const float* up = draw_ahead ? upre + (t & 1) * R + r : nullptr;
        const uint64_t ctr = fwd_counter(a);
        int act;
        switch (nk) {   // the logit count as a compile-time constant: the tail's loops shrink to the logits that exist
          case 1: act = discrete8_row_tail<1>(a, nd, grow, z, ctr, up, pstep); break;
          case 2: act = discrete8_row_tail<2>(a, nd, grow, z, ctr, up, pstep); break;
          case 3: act = discrete8_row_tail<3>(a, nd, grow, z, ctr, up, pstep); break;
          case 4: act = discrete8_row_tail<4>(a, nd, grow, z, ctr, up, pstep); break;
          case 5: act = discrete8_row_tail<5>(a, nd, grow, z, ctr, up, pstep); break;
          case 6: act = discrete8_row_tail<6>(a, nd, grow, z, ctr, up, pstep); break;
          case 7: act = discrete8_row_tail<7>(a, nd, grow, z, ctr, up, pstep); break;
          default: act = discrete8_row_tail<8>(a, nd, grow, z, ctr, up, pstep); break;
        }
        if (px) {  // push: (stamp << 32 | action) as one 8-byte store into every rank's receive area, slot t mod ll_slots
          const int pt = px_t + t;
          const int slot = px_persistent ? p2p_persistent_slot(px_epoch, pxT, pt) : pt % pxslots;
          const size_t off = (size_t)slot * pxw * pxc + (size_t)(pxr * px_a_local + agent) * a.n + grow;
          const unsigned long long w =
              ((unsigned long long)p2p_stamp32(px_persistent ? px_epoch : *px->epoch, pxT, pt) << 32) | (unsigned long long)(unsigned)act;
          for (int p = 0; p < pxw; ++p)
            __hip_atomic_store(PXLL(p) + off, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      } else {
#if PH_FWD16_HEAD_KSPLIT
        const float v = z[0];
#else
        const float v = zs[r * 8];
#endif
        value_row_tail(a, grow, v, pre_ok, vpre);
        // (scripted rollout) the last step's own reward: the flush that precedes GAE on the launch-by-launch path
        if (sc && t == n_steps - 1) {
          float add = sc->rew_seq[(size_t)t * a0.n + grow];
          if (px_persistent && a0.joint) {   // ph_buffer_add_reward_joint of the launch-by-launch walk: base + bonus * [own == partner's]
            FwdArgs b = a;
            b.joint_ll = ll_self + (size_t)p2p_persistent_slot(px_epoch, pxT, t) * pxw * pxc;
            b.ll_t = t;
            int p = *a0.partner_seat;
            p = p < 0 ? 0 : (p >= a0.n_seats ? a0.n_seats - 1 : p);
            const int mine = ll_read(b, b.joint_ll + (size_t)a0.seat * a0.n + grow);
            const int theirs = ll_read(b, b.joint_ll + (size_t)p * a0.n + grow);
            add += joint_reward(mine, theirs, a0.bonus, a0.reward_rule);
          }
          a.rb_rew[grow] += add;
        }
      }
    }
  }
  PH_STAMP(pstep, 14);
  if (draw_ahead && wave == 1 && t + 1 < n_steps) draw_uniforms(t + 1);
  if (copy_from_regs) {   // stores only, at the end of the step: nothing of this step waits behind them
    const int kk = tid & 63;
#pragma unroll
    for (int i = 0; i < XStage<R, NT>::ITERS; ++i) {
      const int rr = (tid + NT * i) >> 6;
      if (kk < nd.D && row0 + rr < a.n) a.rb_obs[(size_t)(row0 + rr) * nd.D + kk] = xr.v[i];
    }
  } else if (net == 1) {
    copy_obs_rows(a, row0, (a.n - row0 < R) ? a.n - row0 : R, nd.D);
  }
  PH_STAMP(a.prof, 7);
  PH_STAMP(pstep, 15);
  }
}

#undef pxw
#undef pxc
#undef pxr
#undef pxT
#undef pxslots
#undef ll_self
#undef PXLL

template <bool VALU, bool LEAN = false>
__global__ __launch_bounds__(256) void policy_fwd16_kernel(FwdArgs a) {
  policy_fwd16_body<VALU, LEAN ? 1 : 0>(a);
  // ph_policy_act_host waits on two host words instead of on the stream: a stream wait goes through the runtime's completion
  // signal (interrupt or a polled signal, then its bookkeeping), the words arrive with the outputs.  Every lane's stores are
  // ordered before the word by its own system-scope fence; the barrier orders all lanes' fences before lane 0's store.
  if constexpr (!LEAN) {
    if (a.host_done) {
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(a.host_done + blockIdx.y, a.host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
template <bool VALU, bool LEAN>
__global__ __launch_bounds__(256) void policy_fwd16_rollout_kernel(FwdArgs a, ScriptedSteps sc) {
  policy_fwd16_body<VALU, LEAN ? 1 : 0>(a, nullptr, 0, 0, 0, &sc);
}
template <bool LEAN>
__global__ __launch_bounds__(256) void policy_fwd16_multi_kernel(FwdMulti m) {
  policy_fwd16_body<false, LEAN ? 2 : 0>(m.a[blockIdx.z], m.px.x, m.px.t, m.px.a_local, blockIdx.z);
}
// The N > 1 counterpart of policy_fwd16_rollout_kernel: every local agent's T steps in one launch, with what crosses ranks at
// every SimultaneousEnv step (multiagentenv.py:149-170: each seat's action to everyone who needs it) done in-kernel -- the
// policy workgroup of 16 environments stores each sampled action as one stamped 8-byte word into every rank's receive area and
// the value workgroup of the NEXT step polls exactly the two words it consumes.  Policy workgroups never wait, so no cycle of
// waits exists as long as every workgroup of the launch is resident (the launcher's caller checks the grid against the chip).
// Three waves per SIMD (168 VGPRs; eight registers -- pointer pairs of the row tails -- live in scratch): the runtime then admits three
// workgroups per CU, two after the hold-back (exchange_rollout_blocks_per_cu), i.e. 512 resident workgroups on the chip against the
// 256 the default N > 1 layout needs -- at two waves per SIMD (192 VGPRs) the margin was zero, and any shortfall drops every rank to
// one launch per step.  Same-box A/B of the two builds: 101.3 vs 102.0 M agent-steps/s in --mode fusedstep (within the spread);
// two ranks sharing ONE device keep the one-launch form: 68.8 M against 42.2 M (profiles/r05_r_exchange_launch_bounds_ab.txt).
// LEAN: every local agent's record passed fwd_args_lean(.., joint_ok) and no step carries action masks (the launcher checks):
// the default N > 1 layout.  Config 5 (masks) takes the general form.
template <bool LEAN>
__global__ __launch_bounds__(256, 3) void policy_fwd16_exchange_rollout_kernel(FwdMulti m, ScriptedMulti sm) {
  policy_fwd16_body<false, LEAN ? 2 : 0>(m.a[blockIdx.z], m.px.x, 0, m.px.a_local, blockIdx.z, &sm.sc[blockIdx.z], 1);
}

// index of the current device into the per-device "LDS opt-in done" tables of the launchers below
static int current_device_slot() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return (dev >= 0 && dev < 64) ? dev : 0;
}

static size_t fwd16_lds_bytes() { return sizeof(float) * (size_t)(2 * 16 * LDH + ZS_FLOATS + 16 + 32) + sizeof(void*) * PH_MAX_RANKS; }

bool fwd16_eligible(const NetDims& nd, int n) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_FWD16");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled && !nd.gauss && nd.nchunk == 1 && nd.A == 1 && nd.L <= 8 && n < 16384;
}

template <bool VALU>
static hipError_t launch_fwd16_variant(const FwdArgs& a, hipStream_t s) {
  static bool allowed_dev[64] = {false};  // dynamic LDS above 64 KiB is opt-in, per kernel and device
  const size_t lds = fwd16_lds_bytes();
  bool& allowed = allowed_dev[current_device_slot()];
  if (!allowed && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)policy_fwd16_kernel<VALU>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  if (!VALU && fwd_args_lean(a)) hipLaunchKernelGGL((policy_fwd16_kernel<false, true>), dim3((a.n + 15) / 16, 2), dim3(256), lds, s, a);
  else hipLaunchKernelGGL((policy_fwd16_kernel<VALU>), dim3((a.n + 15) / 16, 2), dim3(256), lds, s, a);
  return hipGetLastError();
}

// A scripted rollout is a latency chain per workgroup, and a launch of E / 16 x 2 workgroups (128 at the bench size) covers half
// the chip: when two learners' rollouts run side by side the dispatcher is free to put a workgroup of each on the SAME CU while
// other CUs stay empty, and the two chains then share SIMDs for the whole launch (measured: 390 us alone, 440-470 us side by side).
// Asking for more than half a CU's LDS makes a CU take ONE rollout workgroup, whoever launched it (same-box A/B 107.6 -> 109.7 M
// agent-steps/s, profiles/r04_g_ab_hoist_spread.txt; PH_ROLLOUT_SPREAD=0 switches it off; only
// for launches of at most #CUs / 2 workgroups, so that two of them still fit on the chip at once).
static size_t rollout_lds_bytes(int nwg_total) {
  static int spread = -1, num_cu[64] = {0};
  if (spread < 0) {
    const char* e = getenv("PH_ROLLOUT_SPREAD");
    spread = (e && e[0] == '0') ? 0 : 1;
  }
  const size_t lds = fwd16_lds_bytes();
  if (!spread) return lds;
  int& cu = num_cu[current_device_slot()];
  if (cu == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = -1;
  }
  return (cu > 0 && 2 * nwg_total <= cu) ? (size_t)81 * 1024 + 512 : lds;
}
hipError_t launch_policy_fwd16_rollout(const FwdArgs& a, const ScriptedSteps& sc, int gemm_mode, hipStream_t s) {
  dim3 grid((a.n + 15) / 16, 2), block(256);
  const size_t lds = rollout_lds_bytes((int)(grid.x * grid.y));
  const int variant = gemm_mode == 1 ? 1 : ((fwd_args_lean(a) && !sc.mask_seq) ? 2 : 0);   // 1: VALU cross-check, 2: lean, 0: general
  const void* fn = variant == 1 ? (const void*)policy_fwd16_rollout_kernel<true, false>
                                : (variant == 2 ? (const void*)policy_fwd16_rollout_kernel<false, true> : (const void*)policy_fwd16_rollout_kernel<false, false>);
  if (lds > 64 * 1024) {   // dynamic LDS above 64 KiB is opt-in, per kernel and device
    static bool allowed_dev[3][64] = {{false}};
    bool& allowed = allowed_dev[variant][current_device_slot()];
    if (!allowed) {
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      allowed = true;
    }
  }
  if (variant == 1) hipLaunchKernelGGL((policy_fwd16_rollout_kernel<true, false>), grid, block, lds, s, a, sc);
  else if (variant == 2) hipLaunchKernelGGL((policy_fwd16_rollout_kernel<false, true>), grid, block, lds, s, a, sc);
  else hipLaunchKernelGGL((policy_fwd16_rollout_kernel<false, false>), grid, block, lds, s, a, sc);
  return hipGetLastError();
}

hipError_t exchange_rollout_blocks_per_cu(int* blocks_out) {
  const size_t lds = fwd16_lds_bytes();
  int api = 0;
  // (the general form: the lean form of the same kernel needs no more registers)
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, (const void*)policy_fwd16_exchange_rollout_kernel<false>, 256, lds);
  if (e != hipSuccess) return e;
  const int by_lds = (int)((size_t)160 * 1024 / lds);
  int blocks = api < 8 ? api : 8;
  // The API's answer is exact where LDS bounds it; where registers bound it the hardware can admit one block fewer per CU than
  // the API says (SGPR allocation granule), and a polling grid sized by the optimistic figure never completes: hold one back.
  if (blocks > by_lds) blocks = by_lds;
  else if (blocks < by_lds && blocks > 1) blocks -= 1;
  *blocks_out = blocks < 1 ? 1 : blocks;
  return hipSuccess;
}

hipError_t launch_policy_fwd16_exchange_rollout(const FwdMulti& m, const ScriptedMulti& sm, int n_agents, hipStream_t s) {
  bool lean = true;
  for (int i = 0; i < n_agents; ++i) lean = lean && fwd_args_lean(m.a[i], true) && !sm.sc[i].mask_seq;
  const dim3 grid((m.a[0].n + 15) / 16, 2, n_agents);
  if (lean) hipLaunchKernelGGL(policy_fwd16_exchange_rollout_kernel<true>, grid, dim3(256), fwd16_lds_bytes(), s, m, sm);
  else hipLaunchKernelGGL(policy_fwd16_exchange_rollout_kernel<false>, grid, dim3(256), fwd16_lds_bytes(), s, m, sm);
  return hipGetLastError();
}

// Action head of one row on 32 lanes (lane k owns logit k <= 31; padding lanes are one-lane segments of their own): the
// per-component max / sum / CDF are segmented Hillis-Steele scans over the lanes (a lane takes the value `d` lanes below
// only while that lane is still inside its own component) built from DPP moves, so a MultiDiscrete head costs a few dozen
// VALU steps and three ds_bpermutes instead of ~4 LDS round trips per logit in a one-lane-per-row loop.  seg: lo / last lane of this lane's component and
// its index (-1 = padding).  Semantics are those of general_row_tail.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {   // lanes without a source read 0 (bound_ctrl)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// inclusive segmented scan over a 32-lane group with DPP only: row_shr 1/2/4/8 inside each 16-lane row, then row_bcast15
// hands lane 15's value to the upper row, which takes it while its component started at or below lane 15
template <bool MAX>
__device__ __forceinline__ float seg_scan32(float v, int k, int lo) {
  const int kr = k & 15;
#define PH_SCAN_STEP(D, CTRL)                                   \
  {                                                             \
    const float o = dpp_f<CTRL>(v);                             \
    const bool ok = kr >= (D) && k - (D) >= lo;                 \
    v = ok ? (MAX ? fmaxf(v, o) : v + o) : v;                   \
  }
  PH_SCAN_STEP(1, 0x111)
  PH_SCAN_STEP(2, 0x112)
  PH_SCAN_STEP(4, 0x114)
  PH_SCAN_STEP(8, 0x118)
#undef PH_SCAN_STEP
  const float o = dpp_f<0x142>(v);   // row_bcast15
  const bool ok = k >= 16 && lo <= 15;
  return ok ? (MAX ? fmaxf(v, o) : v + o) : v;
}
// sum over the 32-lane group, valid in its lanes 16..31
__device__ __forceinline__ float sum32_upper(float v) {
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x124>(v);   // row_ror 4
  v += dpp_f<0x128>(v);   // row_ror 8
  return v + dpp_f<0x142>(v);
}
// u_drawn: the lane's sampling uniform (head_draw32), drawn by the caller ahead of the logits
__device__ __forceinline__ bool head_draws(const FwdArgs& a) { return !a.given_actions && !a.deterministic && !a.uniforms; }
__device__ __forceinline__ float head_draw32(const FwdArgs& a, int g, bool row_ok, int comp, uint64_t ctr) {
  return (row_ok && comp >= 0) ? philox_uniform(a.seed, ctr, (uint32_t)g, (uint32_t)comp) : 0.f;
}
__device__ __forceinline__ void head_tail32(const FwdArgs& a, const NetDims& nd, int g, bool row_ok, long long ridx, float z,
                                            int k, int lo, int last, int comp, float u_drawn) {
  const bool own = row_ok && comp >= 0;     // this lane holds a real logit of a real row
  const int nk = last - lo + 1;
  if (own && a.mask) z = z - 30.0f * (1.0f - (float)(a.mask[(size_t)g * nd.L + k] != 0));  // modular/policies.py:330-333
  if (own && a.logits) a.logits[(size_t)g * nd.L + k] = z;
  const float M = __shfl(seg_scan32<true>(z, k, lo), last, 32);
  const float e = fast_exp(z - M);
  const float s = seg_scan32<false>(e, k, lo);
  const float S = __shfl(s, last, 32);
  const float lse = M + fast_log(S), inv = __builtin_amdgcn_rcpf(S);
  const int half = (threadIdx.x >> 5) & 1;  // which 32 lanes of the wave
  int act = 0;
  if (a.given_actions) {
    act = own ? (int)a.given_actions[(size_t)g * nd.A + comp] : 0;
    act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
  } else if (a.deterministic) {             // first index of the maximum
    const unsigned long long b = __ballot(z == M);
    const unsigned bits = ((unsigned)(b >> (32 * half)) >> lo) & (nk >= 32 ? 0xffffffffu : ((1u << nk) - 1u));
    act = bits ? __ffs(bits) - 1 : 0;
  } else {                                  // inverse CDF: count the prefix sums <= u among the first nk-1
    float u = 0.f;
    if (own) u = a.uniforms ? a.uniforms[(size_t)g * nd.A + comp] : u_drawn;
    const unsigned long long b = __ballot(own && k < last && u >= s * inv);
    const unsigned bits = ((unsigned)(b >> (32 * half)) >> lo) & (nk >= 32 ? 0xffffffffu : ((1u << nk) - 1u));
    act = __popc(bits);
  }
  const float zact = __shfl(z, lo + act, 32);
  const float lp = z - lse;
  const float ent = sum32_upper(own ? -(e * inv) * lp : 0.f);
  const float logp = sum32_upper((own && k == lo) ? zact - lse : 0.f);
  if (!row_ok) return;   // ridx: rollout-buffer row of g (rb_row), -1 = not recorded
  if (comp >= 0 && k == lo) {
    if (a.act_i32) a.act_i32[(size_t)g * nd.A + comp] = act;
    if (a.act_f32) a.act_f32[(size_t)g * nd.A + comp] = (float)act;
    if (a.rb_act && ridx >= 0) a.rb_act[(size_t)ridx * nd.A + comp] = (float)act;
  }
  if (k == 16) {   // the row totals live in the upper half of the group
    if (a.logp) a.logp[g] = logp;
    if (a.entropy) a.entropy[g] = ent;
    if (a.rb_logp && ridx >= 0) a.rb_logp[ridx] = logp;
  }
}

// ---- 16-row variant for one-hot observations (Discrete / MultiDiscrete spaces) with any action head ----------------------
// SB3 feeds the one-hot encoding through a dense first layer (F = 270 for Liar's Dice: 5 feature chunks, each a global ->
// LDS -> MFMA round trip in the general kernel).  A one-hot row times W1 is the sum of D rows of W1: sixteen lanes per
// observation row gather those rows (16 bytes per lane, all D loads in flight at once) and add them in component order,
// which is the dense layer's k-ordered accumulation with the zero terms left out.  Layer 2 and the head run as in the
// 16-row kernel above (one 16x16 output tile per wave); the head is the general Discrete / MultiDiscrete row tail.
// FUSED: called from a 512-thread workgroup whose lower half runs the policy net and whose upper half runs the value net of
// the same 16 rows (liar_rollout_kernel): tid / net / row0 / LDS base come from the caller, and the value half executes the
// barrier the policy half has around its logits so that both halves reach every workgroup barrier.
// RES (persistent rollouts): one net's weights of one agent staged into LDS ONCE per launch (stage_resident_net) instead of by every
// forward -- the layout the forward itself stages (W2 [64][LDH], b2, the head's bias), the head's weights with a narrower leading
// dimension (RES_LDO: only 32 logit columns exist), and b1, which the per-launch form reads from global memory.  Same values in the
// same operand positions: the forward's arithmetic does not change.
// a row of W1 that is all zeros: what the layer-1 gather reads where an observation row has no feature (see the body)
__device__ float ph_zero_row[HID];
constexpr int RES_LDO = 33;
constexpr int RES_NET_FLOATS = HID * LDH + HID * RES_LDO + HID + HID + 32;
struct ResidentNet {
  float* w2s;   // [64][LDH]
  float* wos;   // policy: act_W [64][RES_LDO], columns >= L zero | value: val_W [64]
  float* b1s;   // [64]
  float* b2s;   // [64]
  float* hbs;   // act_b [32] | val_b
};
__device__ __forceinline__ ResidentNet resident_net_at(float* base) {
  ResidentNet r;
  r.w2s = base;
  r.wos = r.w2s + HID * LDH;
  r.b1s = r.wos + HID * RES_LDO;
  r.b2s = r.b1s + HID;
  r.hbs = r.b2s + HID;
  return r;
}
// scratch of one half (one net) of a fused forward in RES form: xs, hs, outs, feat, aoff, seg, ridxs (see the body)
constexpr int RES_SCRATCH_FLOATS = 2 * 16 * LDH + 16 * 33 + 2 * 16 * 64 + 40 + 96 + 2 * 16 + 8;

template <bool VALU, bool FUSED = false, bool RES = false>
__device__ __forceinline__ void policy_fwd16h_body(const FwdArgs& a, int row0_in = 0, int net_in = 0, int tid_in = 0,
                                                   float* smem_in = nullptr, int row_end_in = 0,
                                                   const ResidentNet resv = ResidentNet{nullptr, nullptr, nullptr, nullptr, nullptr},
                                                   const int* ooff = nullptr) {
  const ResidentNet* const res = &resv;   // (by value: a pointer to a caller's record would pin it in scratch memory)
  extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
  float* smem = FUSED ? smem_in : smem_dyn;
  constexpr int R = 16, NT = 256, LDO = 33, FS = 64;
  constexpr int LDW = RES ? RES_LDO : LDH;   // leading dimension of the head's weights
  const NetDims& nd = a.nd;
  float* xs = smem;                 // [16][LDH]  H2
  float* hs = xs + R * LDH;         // [16][LDH]  H1
  float* w2s = RES ? res->w2s : hs + R * LDH;        // [64][LDH]
  float* wos = RES ? res->wos : w2s + HID * LDH;     // policy: act_W [64][LDW], columns >= L zero | value: val_W [64]
  float* outs = RES ? hs + R * LDH : wos + HID * LDH;    // [16][LDO] logits
  float* b2s = RES ? res->b2s : outs + R * LDO;      // [64]
  float* hbs = RES ? res->hbs : b2s + HID;           // act_b [32] | val_b
  // [16][FS] per (row, component): the hot row of W1, or ph_zero_row where there is none (padding rows, components >= D)
  const float** feat = (const float**)(RES ? outs + R * LDO : hbs + 32);
  int* aoff = (int*)(feat + R * FS);   // [40] prefix sums of the action nvec (A + 1 entries)
  int* seg = aoff + 40;             // [3][32] per logit: first / last lane of its component, component index (RES: filled once per launch)
  long long* ridxs = (long long*)(seg + 96);          // [16] rollout-buffer row of each observation row (-1 = not recorded)

  const int tid = FUSED ? tid_in : (int)threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
  const int net = FUSED ? net_in : (int)blockIdx.y;
  const int row0 = FUSED ? row0_in : (int)blockIdx.x * R;
  // rows [row0, n_end) are this workgroup's; the persistent rollout may own fewer than 16 tables per workgroup (a.n stays the
  // row stride of the ragged rollout-buffer addressing)
  const int n_end = FUSED ? row_end_in : a.n;
  const ph_layout& lay = nd.lay;
  const float* W1 = a.params + (net == 0 ? lay.pi_W1 : lay.vf_W1);
  const float* B1 = a.params + (net == 0 ? lay.pi_b1 : lay.vf_b1);
  const float* W2 = a.params + (net == 0 ? lay.pi_W2 : lay.vf_W2);
  const float* B2 = a.params + (net == 0 ? lay.pi_b2 : lay.vf_b2);
  const int D = nd.D;

  PH_STAMP(a.prof, 0);
  // Every global load that does not depend on the observations is issued here, back to back (a kernel starts with cold
  // caches: each dependent round trip costs ~1 us at this occupancy).  The observation -> feature-row loads go first.
  // (Staging the whole of W1 -- 69 KB per net for Liar's Dice -- into LDS so that the gather stays on the CU measured
  // slower: +3.4 k cycles of staging against -1.7 k in the gather.)
  // (unconditional loads at clamped positions, then selects: a load under `if (comp < D && row < n_end)` is a basic block of its
  // own, and the four of a lane then go out one after the other, each behind the wait for the one before)
  const float* fv[R * FS / NT];
  {
    const int* off = RES ? ooff : nd.obs_off;   // RES: the prefix sums sit in LDS for the launch
    int lo[R * FS / NT], hi[R * FS / NT];
    float xo[R * FS / NT];
#pragma unroll
    for (int i = 0; i < R * FS / NT; ++i) {
      const int r = 4 * wave + i, comp = lane, row = row0 + r;   // the rows this wave gathers below
      const bool ok = comp < D && row < n_end;
      const int cs = ok ? comp : 0, rs = ok ? row : row0;
      lo[i] = off[cs];
      hi[i] = off[cs + 1];
      xo[i] = a.obs[(size_t)rs * D + cs];
    }
#pragma unroll
    for (int i = 0; i < R * FS / NT; ++i) {
      const int r = 4 * wave + i, comp = lane, row = row0 + r;   // the rows this wave gathers below
      const int nn = hi[i] - lo[i];
      int x = (int)xo[i];
      x = x < 0 ? 0 : (x >= nn ? nn - 1 : x);
      fv[i] = (comp < D && row < n_end) ? W1 + (size_t)(lo[i] + x) * HID : ph_zero_row;
    }
  }
  long long ridxv = -1;
  if (net == 0 && tid < R && row0 + tid < n_end && (a.rb_act || a.rb_logp)) ridxv = rb_row(a, row0 + tid);
  WStage<NT> w2r;
  const int gr = tid >> 4, gl = tid & 15;   // gather: row gr, hidden units 4*gl .. 4*gl+3
  float b1v[4];
  float bias2 = 0.f, hb = 0.f, hv[8];
  int aoffv = 0;
  if constexpr (!RES) {
    w2r.issue(W2, 0, HID, tid);
#pragma unroll
    for (int i = 0; i < 4; ++i) b1v[i] = B1[4 * gl + i];
    if (tid < HID) bias2 = B2[tid];
    if (net == 0) {
      if (tid <= nd.A) aoffv = nd.act_off[tid];
#pragma unroll
      for (int i = 0; i < 8; ++i) {   // act_W [64][L] -> [64][32], zero padded
        const int e = tid + NT * i, j = e >> 5, k = e & 31;
        hv[i] = (k < nd.L) ? a.params[lay.act_W + j * nd.L + k] : 0.f;
      }
      if (tid < 32) hb = (tid < nd.L) ? a.params[lay.act_b + tid] : 0.f;
    } else {
      hv[0] = (tid < HID) ? a.params[lay.val_W + tid] : 0.f;
      if (tid == 0) hb = a.params[lay.val_b];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) b1v[i] = res->b1s[4 * gl + i];
  }
  // commits, in issue order
#pragma unroll
  for (int i = 0; i < R * FS / NT; ++i) feat[(4 * wave + i) * FS + lane] = fv[i];
  if (net == 0 && tid < R) ridxs[tid] = ridxv;
  // a wave gathers the four rows whose positions it has just written: no workgroup barrier, its own LDS writes only
  // (ridxs is read behind three more barriers)
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  PH_STAMP(a.prof, 1);

  // ---- layer 1: gather-sum of W1 rows in component order; everything staged for the later layers is committed while the
  // gather loads are in flight ----
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float u_tail[2] = {0.f, 0.f};   // sampling uniforms of this lane's two tail rows
  const float* const* fr = feat + gr * FS;
  // Every position of feat is a readable row (ph_zero_row where the observation has no feature there), so the gathers of a wave
  // are straight-line code: no per-component exec mask, no zero-filled registers behind a skipped load, nothing for the adds to
  // skip (x + 0.f = x: the accumulator starts at +0 and is never -0).  What remains conditional is scalar: a wave none of whose
  // four rows is live gathers nothing, and components go in groups of eight up to D.  (Written as `f >= 0 ? load : 0` per
  // component, every load is a basic block with its own feat read in front: 32 gathers one LDS latency apart, 4.2 k of this
  // phase's 5.2 k cycles in the rollout kernel; the zero-fills of the masked form were another ~260 v_mov.)
  const bool wave_live = row0 + 4 * __builtin_amdgcn_readfirstlane(wave) < n_end;
  float4 w[32];
  typedef const f32x4 __attribute__((address_space(1))) * global_row4;   // (a pointer read from LDS would load through `flat`)
  auto gather8 = [&](float4* dst, int c0) {   // components c0 .. c0 + 7 of this lane's row
    ulonglong2 q[4];
    const ulonglong2* fr2 = reinterpret_cast<const ulonglong2*>(fr + c0);
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = fr2[u];
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(q[u].x), "+v"(q[u].y));   // one batch of reads, then the gathers
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const f32x4 v = *(global_row4)(((u & 1) ? q[u >> 1].y : q[u >> 1].x) + 16ull * (unsigned)gl);
      dst[u] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  if (wave_live) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (8 * q < D) gather8(w + 8 * q, 8 * q);
  }
  if constexpr (!RES) {
    w2r.commit(w2s, tid);
    if (tid < HID) b2s[tid] = bias2;
    if (net == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = tid + NT * i;
        wos[(e >> 5) * LDH + (e & 31)] = hv[i];
      }
      if (tid < 32) hbs[tid] = hb;
      if (tid <= nd.A) aoff[tid] = aoffv;
    } else {
      if (tid < HID) wos[tid] = hv[0];
      if (tid == 0) hbs[0] = hb;
    }
  }
  {
    // The sampling uniforms of the row tails (Philox: a chain of ~100 dependent instructions that needs no logit) are drawn here,
    // under the gathers' latency.  RES: the logit -> component table is in LDS for the launch; otherwise from the prefix sums.
    if (net == 0 && head_draws(a)) {
      const int k = tid & 31;
      int comp = -1;
      if constexpr (RES) {
        comp = seg[64 + k];
      } else {
        for (int cc = 0; cc < nd.A; ++cc)
          if (k >= nd.act_off[cc] && k < nd.act_off[cc + 1]) comp = cc;
      }
      const uint64_t ctr = fwd_counter(a);
      const int r0 = row0 + (tid >> 5);
      u_tail[0] = head_draw32(a, r0, r0 < n_end, comp, ctr);
      u_tail[1] = (row0 + 8 < n_end) ? head_draw32(a, r0 + 8, r0 + 8 < n_end, comp, ctr) : 0.f;
    }
    if (wave_live) {
      // (w[0] opaque: otherwise "0 + w[0]" moves into the block of the first gathers, with a full wait behind them; the uniforms
      // ride along so that they are computed in front of that wait)
      asm volatile("" : "+v"(w[0].x), "+v"(w[0].y), "+v"(w[0].z), "+v"(w[0].w), "+v"(u_tail[0]), "+v"(u_tail[1]));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (8 * q < D) {
#pragma unroll
          for (int u = 8 * q; u < 8 * q + 8; ++u) {
            acc.x += w[u].x;
            acc.y += w[u].y;
            acc.z += w[u].z;
            acc.w += w[u].w;
          }
        }
      for (int c0 = 32; c0 < D; c0 += 32) {   // more than 32 components: further batches
        float4 w2[32];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c0 + 8 * q < D) gather8(w2 + 8 * q, c0 + 8 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (c0 + 8 * q < D) {
#pragma unroll
            for (int u = 8 * q; u < 8 * q + 8; ++u) {
              acc.x += w2[u].x;
              acc.y += w2[u].y;
              acc.z += w2[u].z;
              acc.w += w2[u].w;
            }
          }
      }
    }
    float* h = hs + gr * LDH + 4 * gl;
    h[0] = fast_tanh(acc.x + b1v[0]);
    h[1] = fast_tanh(acc.y + b1v[1]);
    h[2] = fast_tanh(acc.z + b1v[2]);
    h[3] = fast_tanh(acc.w + b1v[3]);
  }
  lds_only_barrier();
  PH_STAMP(a.prof, 3);
  if (!RES && net == 0 && tid < 32) {   // component of logit `tid` (read by the head after two more barriers)
    int lo = tid, last = tid, comp = -1;
    for (int cc = 0; cc < nd.A; ++cc) {
      const int l0 = aoff[cc], l1 = aoff[cc + 1];
      if (tid >= l0 && tid < l1) {
        lo = l0;
        last = l1 - 1;
        comp = cc;
      }
    }
    seg[tid] = lo;
    seg[32 + tid] = last;
    seg[64 + tid] = comp;
  }

  // one 16x16 output tile per wave: D[row 4g+r][col col0 + c] = sum_k A[row][k] W[k][col]; two accumulator chains
  auto layer = [&](const float* A, const float* W, int col0, int ldb) -> f32x4 {
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    const float* ap = A + c * LDH + g;
    const float* bp = W + g * ldb + col0 + c;
    float av[16], bv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      av[s] = ap[4 * s];
      bv[s] = bp[4 * s * ldb];
    }
    // keep all 32 operand reads ahead of the products: left to itself the scheduler (in the 512-thread rollout kernel) emits
    // read - wait - MFMA sixteen times, one LDS latency per product
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 16; s += 2) {
      e = mma16<VALU>(av[s], bv[s], e, lane);
      o = mma16<VALU>(av[s + 1], bv[s + 1], o, lane);
    }
    return e + o;
  };
  {
    const f32x4 z2 = layer(hs, w2s, 16 * wave, LDH);
    const float b = b2s[16 * wave + c];
#pragma unroll
    for (int r = 0; r < 4; ++r) xs[(4 * g + r) * LDH + 16 * wave + c] = fast_tanh(z2[r] + b);
  }
  lds_only_barrier();
  PH_STAMP(a.prof, 5);

  if (net == 0) {
    // ---- policy head: logits [16][32] as two 16x16 tiles (waves 0, 1), then one lane per row ----
    if (wave < 2) {
      const f32x4 z3 = layer(xs, wos, 16 * wave, LDW);
      const float b = hbs[16 * wave + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) outs[(4 * g + r) * LDO + 16 * wave + c] = z3[r] + b;
    }
    lds_only_barrier();
    PH_STAMP(a.prof, 6);
    {
      const int k = tid & 31, lo = seg[k], last = seg[32 + k], comp = seg[64 + k];
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {   // 32 lanes per row, 8 rows per pass
        if (pass == 1 && row0 + 8 >= n_end) break;   // no live row in the second pass (workgroup-uniform)
        const int r = pass * 8 + (tid >> 5);
        head_tail32(a, nd, row0 + r, row0 + r < n_end, ridxs[r], outs[r * LDO + k], k, lo, last, comp, u_tail[pass]);
      }
    }
  } else {
    if constexpr (FUSED) lds_only_barrier();   // the policy half's barrier after its logits
    // ---- value head: wave 0, four lanes per row, quad-DPP reduction; every wave copies observations ----
    if (wave == 0) {
      const int r = lane >> 2, q = lane & 3;
      float v = 0.f, hx[16], hw[16];
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        const int j = 8 * q + (m & 7) + 32 * (m >> 3);
        hx[m] = xs[r * LDH + j];
        hw[m] = wos[j];
      }
      __builtin_amdgcn_sched_barrier(0);   // the 32 reads in flight together, then the (ordered) sum
#pragma unroll
      for (int m = 0; m < 16; ++m) v = __builtin_fmaf(hx[m], hw[m], v);
      v = quad_sum_f(v) + hbs[0];
      if (q == 0 && row0 + r < n_end) value_row_tail(a, row0 + r, v);
    }
    copy_obs_rows(a, row0, (n_end - row0 < R) ? n_end - row0 : R, nd.D, tid, NT);
  }
  PH_STAMP(a.prof, 7);
}

template <bool VALU>
__global__ __launch_bounds__(256) void policy_fwd16h_kernel(FwdArgs a) {
  policy_fwd16h_body<VALU>(a);
}

static size_t fwd16h_lds_bytes() {
  return sizeof(float) * (size_t)(2 * 16 * LDH + 2 * HID * LDH + 16 * 33 + HID + 32 + 2 * 16 * 64 + 40 + 96 + 2 * 16);
}

// one net (policy: net 0, value: net 1) of `params` into a resident set, by the 256 lanes of the half that runs that net -- the very
// loads and LDS positions the per-launch forward stages (policy_fwd16h_body), with the head's weights at leading dimension RES_LDO
__device__ __forceinline__ void stage_resident_net(const ResidentNet& rn, const float* params, const NetDims& nd, int net, int tid) {
  constexpr int NT = 256;
  const ph_layout& lay = nd.lay;
  WStage<NT> w2r;
  w2r.issue(params + (net == 0 ? lay.pi_W2 : lay.vf_W2), 0, HID, tid);
  w2r.commit(rn.w2s, tid);
  if (tid < HID) {
    rn.b1s[tid] = params[(net == 0 ? lay.pi_b1 : lay.vf_b1) + tid];
    rn.b2s[tid] = params[(net == 0 ? lay.pi_b2 : lay.vf_b2) + tid];
  }
  if (net == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // act_W [64][L] -> [64][32], zero padded
      const int e = tid + NT * i, j = e >> 5, k = e & 31;
      rn.wos[j * RES_LDO + k] = (k < nd.L) ? params[lay.act_W + j * nd.L + k] : 0.f;
    }
    if (tid < 32) rn.hbs[tid] = (tid < nd.L) ? params[lay.act_b + tid] : 0.f;
  } else {
    if (tid < HID) rn.wos[tid] = params[lay.val_W + tid];
    if (tid == 0) rn.hbs[0] = params[lay.val_b];
  }
}

// ---- the one-hot forward of at most four rows: one net on ONE wave, no workgroup barrier inside -------------------------------
// policy_fwd16h_body pads its rows to a 16-row MFMA tile and spreads a layer over four waves, with a workgroup barrier behind every
// phase; with one to four live rows (the persistent rollout of a 256-table game: ONE table per workgroup) that is a chain of
// five barriers and 15 padding rows per product.  Here lane j owns hidden unit / logit j of every row: layer 1 is the same
// component-ordered sum of W1 rows (one coalesced 256-byte load per component), layers 2 / 3 are fmaf chains over k with the
// activations broadcast from LDS.  v_mfma_f32_16x16x4_f32 IS a k-ordered fmaf chain (test_mfma_and_valu_tiles_agree_bitwise), and
// the chains here are split exactly like the body's two accumulators (k blocks 0, 2, 4, .. and 1, 3, 5, ..; then their sum), so
// every logit and value is bitwise what the 16-row body computes for that row.  The row tails are the body's (head_tail32,
// value_row_tail).  NR: rows held (1, 2 or 4); rows >= n_end are padding.
template <int NR>
__device__ __forceinline__ void policy_fwd_rows_wave(const FwdArgs& a, int net, int row0, int n_end, int lane, float* smem,
                                                     const ResidentNet rn, const int* ooff) {
  constexpr int R = 16, LDO = 33, FS = 64, TP = (NR + 1) / 2;   // the scratch layout is the body's (RES form)
  constexpr int LDR = 68;   // row stride of this function's own H1 / H2 rows (inside the body's [16][LDH] areas): 16-byte aligned rows
  const NetDims& nd = a.nd;
  float* xs = smem;                 // [16][LDH]  H2 (rows < NR)
  float* hs = xs + R * LDH;         // [16][LDH]  H1
  float* outs = hs + R * LDH;
  const float** feat = (const float**)(outs + R * LDO);
  int* aoff = (int*)(feat + R * FS);
  int* seg = aoff + 40;
  long long* ridxs = (long long*)(seg + 96);
  const ph_layout& lay = nd.lay;
  const float* W1 = a.params + (net == 0 ? lay.pi_W1 : lay.vf_W1);
  const int D = nd.D;
  auto wave_sync = [] {
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  };
  auto hpos = [](int k) { return 8 * (k >> 3) + 2 * (k & 3) + ((k >> 2) & 1); };   // see dense() below

  PH_STAMP(a.prof, 0);
  // ---- hot row of W1 per (row, component): lane = component ----
  {
    const float* fv[NR];
    int lo[NR], hi[NR];
    float xo[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const bool ok = lane < D && row0 + r < n_end;
      const int cs = ok ? lane : 0, rs = ok ? row0 + r : row0;
      lo[r] = ooff[cs];
      hi[r] = ooff[cs + 1];
      xo[r] = a.obs[(size_t)rs * D + cs];
    }
    long long ridxv = -1;
    if (net == 0 && lane < NR && row0 + lane < n_end && (a.rb_act || a.rb_logp)) ridxv = rb_row(a, row0 + lane);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int nn = hi[r] - lo[r];
      int x = (int)xo[r];
      x = x < 0 ? 0 : (x >= nn ? nn - 1 : x);
      fv[r] = (lane < D && row0 + r < n_end) ? W1 + (size_t)(lo[r] + x) * HID : ph_zero_row;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) feat[r * FS + lane] = fv[r];
    if (net == 0 && lane < NR) ridxs[lane] = ridxv;
  }
  wave_sync();
  PH_STAMP(a.prof, 1);

  // ---- layer 1: lane j sums element j of the rows' hot W1 rows, in component order ----
  typedef const float __attribute__((address_space(1))) * global_f32;
  float acc[NR];
  float u_tail[TP];
#pragma unroll
  for (int t = 0; t < TP; ++t) u_tail[t] = 0.f;
  {
    float w[NR][32];
    auto gather8 = [&](float* dst, int r, int c0) {   // components c0 .. c0 + 7 of row r (the pointers are wave-uniform: broadcast reads)
      ulonglong2 q[4];
      const ulonglong2* fr2 = reinterpret_cast<const ulonglong2*>(feat + r * FS + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = fr2[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(q[u].x), "+v"(q[u].y));
#pragma unroll
      for (int u = 0; u < 8; ++u) dst[u] = *(global_f32)(((u & 1) ? q[u >> 1].y : q[u >> 1].x) + 4ull * (unsigned)lane);
    };
#pragma unroll
    for (int r = 0; r < NR; ++r)
      if (row0 + r < n_end) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (8 * q < D) gather8(w[r] + 8 * q, r, 8 * q);
      }
    if (net == 0 && head_draws(a)) {   // the tails' sampling uniforms, under the gathers' latency
      const int k = lane & 31, comp = seg[64 + k];
      const uint64_t ctr = fwd_counter(a);
#pragma unroll
      for (int t = 0; t < TP; ++t) {
        const int r = (lane >> 5) + 2 * t;
        u_tail[t] = head_draw32(a, row0 + r, r < NR && row0 + r < n_end, comp, ctr);
      }
    }
    float b1 = rn.b1s[lane];
    asm volatile("" : "+v"(b1));
#pragma unroll
    for (int t = 0; t < TP; ++t) asm volatile("" : "+v"(u_tail[t]));   // drawn in front of the wait for the gathers
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      acc[r] = 0.f;
      if (row0 + r < n_end) {
        asm volatile("" : "+v"(w[r][0]));   // (otherwise "0 + w[r][0]" moves into the block of the first gathers, with a full wait there)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (8 * q < D) {
#pragma unroll
            for (int u = 8 * q; u < 8 * q + 8; ++u) acc[r] += w[r][u];
          }
        for (int c0 = 32; c0 < D; c0 += 32) {   // more than 32 components: further batches
          float w2[32];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (c0 + 8 * q < D) gather8(w2 + 8 * q, r, c0 + 8 * q);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (c0 + 8 * q < D) {
#pragma unroll
              for (int u = 8 * q; u < 8 * q + 8; ++u) acc[r] += w2[u];
            }
        }
      }
      hs[r * LDR + hpos(lane)] = fast_tanh(acc[r] + b1);
    }
  }
  wave_sync();
  PH_STAMP(a.prof, 3);

  // out[r][col] = e + o: e takes the k blocks 0, 2, .. of four, o the blocks 1, 3, .. -- the body's two MFMA accumulator chains,
  // here the two halves of one packed fmaf chain.  A (this function's own H1 / H2 rows) is stored with unit k at position hpos(k), so
  // that units k and k + 4 of a block pair are neighbours: a 16-byte broadcast read is two ready-made operand pairs.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto dense = [&](const float* A, const int* arow, int nrows, const float* W, int ldw, int col, float* out) {
    constexpr int MR = TP > NR ? TP : NR;
    f32x2 eo[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) eo[r] = f32x2{0.f, 0.f};
#pragma unroll
    for (int P = 0; P < 8; ++P) {
      f32x2 wv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wv[i] = f32x2{W[(8 * P + i) * ldw + col], W[(8 * P + 4 + i) * ldw + col]};
#pragma unroll
      for (int r = 0; r < MR; ++r) {
        if (r >= nrows) continue;   // (compile-time)
        const float4 ha = *reinterpret_cast<const float4*>(A + arow[r] * LDR + 8 * P);
        const float4 hb = *reinterpret_cast<const float4*>(A + arow[r] * LDR + 8 * P + 4);
        eo[r] = __builtin_elementwise_fma(f32x2{ha.x, ha.y}, wv[0], eo[r]);
        eo[r] = __builtin_elementwise_fma(f32x2{ha.z, ha.w}, wv[1], eo[r]);
        eo[r] = __builtin_elementwise_fma(f32x2{hb.x, hb.y}, wv[2], eo[r]);
        eo[r] = __builtin_elementwise_fma(f32x2{hb.z, hb.w}, wv[3], eo[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < MR; ++r)
      if (r < nrows) out[r] = eo[r][0] + eo[r][1];
  };
  {
    int rows[NR];
    float z2[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) rows[r] = r;
    dense(hs, rows, NR, rn.w2s, LDH, lane, z2);
    const float b2 = rn.b2s[lane];
#pragma unroll
    for (int r = 0; r < NR; ++r) xs[r * LDR + hpos(lane)] = fast_tanh(z2[r] + b2);
  }
  wave_sync();
  PH_STAMP(a.prof, 5);

  if (net == 0) {
    // ---- policy head: the 32 lanes of a half wave hold the 32 logits of a row -- head_tail32's layout: rows (lane >> 5) + 2 t ----
    const int k = lane & 31, lo = seg[k], last = seg[32 + k], comp = seg[64 + k];
    int rows[TP];
    float z3[TP];
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int r = (lane >> 5) + 2 * t;
      rows[t] = r < NR ? r : 0;
    }
    dense(xs, rows, TP, rn.wos, RES_LDO, k, z3);
    const float hb = rn.hbs[k];
    PH_STAMP(a.prof, 6);
#pragma unroll
    for (int t = 0; t < TP; ++t) {
      const int r = (lane >> 5) + 2 * t;
      if (2 * t >= NR || row0 + 2 * t >= n_end) break;   // no live row in this pass (wave-uniform)
      head_tail32(a, nd, row0 + r, r < NR && row0 + r < n_end, ridxs[rows[t]], z3[t] + hb, k, lo, last, comp, u_tail[t]);
    }
  } else {
    // ---- value head: four lanes per row, quad-DPP reduction (the body's); then the observation rows of the buffer ----
    const int r = lane >> 2, q = lane & 3;
    float v = 0.f, hx[16], hw[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const int j = 8 * q + (m & 7) + 32 * (m >> 3);
      hx[m] = xs[(r < NR ? r : 0) * LDR + hpos(j)];
      hw[m] = rn.wos[j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 16; ++m) v = __builtin_fmaf(hx[m], hw[m], v);
    v = quad_sum_f(v) + rn.hbs[0];
    if (q == 0 && r < NR && row0 + r < n_end) value_row_tail(a, row0 + r, v);
    copy_obs_rows(a, row0, (n_end - row0 < NR) ? n_end - row0 : NR, nd.D, lane, 64);
  }
  PH_STAMP(a.prof, 7);
}

// ---- persistent Liar's Dice self-play rollout ------------------------------------------------------------------------------------
// n_steps vectorised MultiAgentEnv.step calls of ph_liar_selfplay_step in ONE launch: tables are independent, so one 512-thread
// workgroup owns up to 16 tables (launch_liar_rollout: as few as spreads them over every CU) for the whole rollout -- ego forward -> move -> partner reply -> move / credit / re-deal -> partner
// opening -> move -- with workgroup barriers where the launch-by-launch walk has kernel boundaries.  With more than four tables per
// workgroup its lower half runs the policy net of the acting agent, its upper half the value net (policy_fwd16h_body<.., FUSED>); with
// up to four, each net runs on one wave (policy_fwd_rows_wave: the same rows bit for bit).  A table's book-keeping runs on the 32 lanes
// of a half wave (ph_liar_group.h: the rules of ph_liar.h's per-step lane functions, restated for that layout), so every number that is
// read again (game state, observations, both buffers, book-keeping, cached values) is bitwise what 6 x n_steps launches produce; what
// disappears is their ~1 us per dependent cold-cache round trip, the launch boundaries, and the partner forwards whose outputs
// would be scratch (see the loop).
struct LiarRolloutArgs {
  ph_liar_selfplay s;
  FwdArgs ego, reply, opening;        // the three forwards of a step; ego.rb_* point at row ego_pos0
  int n_steps;
  unsigned long long counter0;        // step t uses counter0 + t (ego forward, dice), 2c and 2c + 1 (partner forwards)
  const unsigned long long* epoch;
  float* alt_rewards;
  int alt_T;
  float* ego_rew_row0;                // ego rewards row of step 0
  int no_skip;                        // PH_LIAR_SKIP=0: run every partner forward, needed or not (A/B switch)
};

static size_t fwd16h_lds_bytes();

// The (at most 16) tables a workgroup owns keep their whole state in LDS for the rollout: game state, the three observation arrays, the
// action / reward / flag scratch of the step and the partner's per-table book-keeping are mirrored in at the start, the
// argument records get pointers REBASED into the mirror (mirror - row0 * stride, so that the unchanged code indexes them with the
// global table number through generic addressing), and everything is written back at the end.  What still goes to HBM per step
// is what must: the rollout-buffer rows, the partner's late rewards and the value / log-prob outputs.
// (LiarMirror: ph_liar_group.h)
template <typename T>
__device__ __forceinline__ T* rebase(T* mirror, int row0, int per) { return mirror - (size_t)row0 * per; }

// NR: 0 = the 16-row forward (policy_fwd16h_body on the two halves of the workgroup); 1 / 2 / 4 = at most that many tables per
// workgroup, each net's forward on one wave (policy_fwd_rows_wave)
template <int NR>
__global__ __launch_bounds__(512) void liar_rollout_kernel(LiarRolloutArgs r, int half_floats, int rpw) {
  extern __shared__ __attribute__((aligned(16))) float smem_roll[];
  const int tid512 = threadIdx.x, half = tid512 >> 8, tid = tid512 & 255;
  // per half (net): the forward's scratch, then the weights of that net for the ego and for the partner, resident for the launch
  float* sm = smem_roll + (size_t)half * half_floats;
  const ResidentNet res_ego = resident_net_at(sm + RES_SCRATCH_FLOATS), res_alt = resident_net_at(sm + RES_SCRATCH_FLOATS + RES_NET_FLOATS);
  const int row0 = blockIdx.x * rpw;            // rpw <= 16 tables per workgroup (the forward's tile is 16 rows, the rest padding)
  const ph_liar_selfplay& g = r.s;              // the global arrays
  const int nrow = (g.n - row0 < rpw) ? g.n - row0 : rpw;

  // ---- mirror in ----
  LiarMirror m;
  {
    int* ip = (int*)(smem_roll + 2 * (size_t)half_floats) + 80;   // 16-byte aligned: half_floats is a multiple of 4; 80 ints: the obs prefix sums
    m.hands = ip;
    m.history = m.hands + 16 * 12;
    m.nmoves = m.history + 16 * 24;
    m.alt_pos = m.nmoves + 16;
    m.ego_act = m.alt_pos + 16;
    m.alt_act = m.ego_act + 32;
    float* fp = (float*)(m.alt_act + 32);
    m.obs_ego = fp;
    m.obs_alt = m.obs_ego + 16 * 30;
    m.obs_next = m.obs_alt + 16 * 30;
    m.rew1 = m.obs_next + 16 * 30;
    m.rew2 = m.rew1 + 32;
    m.es_alt = m.rew2 + 32;
    m.es_ego = m.es_alt + 16;
    m.u8 = (unsigned char*)(m.es_ego + 16);
  }
  auto copy_in = [&](auto* dst, const auto* src, int per) {
    for (int i = tid512; i < nrow * per; i += 512) dst[i] = src[(size_t)row0 * per + i];
  };
  copy_in(m.hands, g.hands, 12);
  copy_in(m.history, g.history, 24);
  copy_in(m.nmoves, g.nmoves, 1);
  copy_in(m.alt_pos, g.alt_pos, 1);
  copy_in(m.ego_act, g.ego_actions, 2);
  copy_in(m.alt_act, g.alt_actions, 2);
  copy_in(m.obs_ego, g.obs_ego, 30);
  copy_in(m.obs_alt, g.obs_alt, 30);
  copy_in(m.obs_next, g.obs_next, 30);
  copy_in(m.rew1, g.rew1, 2);
  copy_in(m.rew2, g.rew2, 2);
  copy_in(m.es_alt, g.es_alt, 1);
  copy_in(m.es_ego, g.ego_episode_start, 1);
  unsigned char* gu8[12] = {g.ego_first, g.alt_boundary, g.alt_term, g.alt_open, g.alt_acted, g.done1, g.done2, g.running, g.can,
                            g.alt_opens, g.ego_opens, g.done};
#pragma unroll
  for (int k = 0; k < 12; ++k)
    if (tid512 < nrow) m.u8[16 * k + tid512] = gu8[k][row0 + tid512];

  // ---- the description and the argument records, rebased into the mirror ----
  ph_liar_selfplay s = g;
  s.hands = rebase(m.hands, row0, 12);
  s.history = rebase(m.history, row0, 24);
  s.nmoves = rebase(m.nmoves, row0, 1);
  s.alt_pos = rebase(m.alt_pos, row0, 1);
  s.ego_actions = rebase(m.ego_act, row0, 2);
  s.alt_actions = rebase(m.alt_act, row0, 2);
  s.obs_ego = rebase(m.obs_ego, row0, 30);
  s.obs_alt = rebase(m.obs_alt, row0, 30);
  s.obs_next = rebase(m.obs_next, row0, 30);
  s.rew1 = rebase(m.rew1, row0, 2);
  s.rew2 = rebase(m.rew2, row0, 2);
  s.es_alt = rebase(m.es_alt, row0, 1);
  s.ego_episode_start = rebase(m.es_ego, row0, 1);
  s.ego_first = rebase(m.u8 + 0 * 16, row0, 1);
  s.alt_boundary = rebase(m.u8 + 1 * 16, row0, 1);
  s.alt_term = rebase(m.u8 + 2 * 16, row0, 1);
  s.alt_open = rebase(m.u8 + 3 * 16, row0, 1);
  s.alt_acted = rebase(m.u8 + 4 * 16, row0, 1);
  s.done1 = rebase(m.u8 + 5 * 16, row0, 1);
  s.done2 = rebase(m.u8 + 6 * 16, row0, 1);
  s.running = rebase(m.u8 + 7 * 16, row0, 1);
  s.can = rebase(m.u8 + 8 * 16, row0, 1);
  s.alt_opens = rebase(m.u8 + 9 * 16, row0, 1);
  s.ego_opens = rebase(m.u8 + 10 * 16, row0, 1);
  s.done = rebase(m.u8 + 11 * 16, row0, 1);
  __syncthreads();

  // ---- what every forward of the launch reads and no step changes: both agents' weights of this half's net, the observation
  //      and action prefix sums, the logit -> component table of the head (policy half) ----
  int* const ooff = (int*)(smem_roll + 2 * (size_t)half_floats);     // [D + 1 <= 65]
  {
    const NetDims& nd = r.ego.nd;
    stage_resident_net(res_ego, r.ego.params, nd, half, tid);
    stage_resident_net(res_alt, r.reply.params, nd, half, tid);
    if (tid512 <= nd.D) ooff[tid512] = nd.obs_off[tid512];
    if (half == 0) {   // aoff / seg of the policy half's scratch (policy_fwd16h_body's layout)
      int* aoff = (int*)(sm + 2 * 16 * LDH + 16 * 33) + 2 * 16 * 64;
      int* seg = aoff + 40;
      if (tid < 32) {
        int lo = tid, last = tid, comp = -1;
        for (int cc = 0; cc < nd.A; ++cc) {
          const int l0 = nd.act_off[cc], l1 = nd.act_off[cc + 1];
          if (tid >= l0 && tid < l1) {
            lo = l0;
            last = l1 - 1;
            comp = cc;
          }
        }
        seg[tid] = lo;
        seg[32 + tid] = last;
        seg[64 + tid] = comp;
      }
      if (tid <= nd.A) aoff[tid] = nd.act_off[tid];
    }
  }
  __syncthreads();

  // book-keeping: table i of the workgroup on the 32 lanes of group i (ph_liar_group.h)
  const int grp = tid512 >> 5, gl = tid512 & 31;
  const bool keeper = grp < nrow;
  LiarGroupCtx gc;
  gc.n = g.n;
  gc.alt_rewards = r.alt_rewards;
  gc.alt_T = r.alt_T;
  gc.episodes = g.episodes;
  gc.dice_seed = g.dice_seed;
  gc.probegostart = g.probegostart;
  // one inlined copy of the forward body: the three forwards of a step are a loop whose argument record is selected with
  // scalar selects (three inlined copies cost 1.4 KB of scratch per lane and 649 spilled SGPRs)
  // the RNG epoch is constant for the launch: read once here instead of by every forward's sampling tail and every re-deal
  const unsigned long long epoch_hi = r.epoch ? (unsigned long long)(*r.epoch) << 32 : 0ull;
  for (int ph = 0; ph < 3 * r.n_steps; ++ph) {
    const int t = ph / 3, f = ph - 3 * t;
    const unsigned long long counter = r.counter0 + (unsigned long long)t;
    const size_t row = (size_t)t * g.n;
    long long* prof = (t == 2) ? r.ego.prof : nullptr;   // debug stamps of the third step (scripts/liar_rollout_profile.py)
    if (f == 0) PH_STAMP(prof, 8);
    if (ph == 0) PH_STAMP(r.ego.prof, 15);   // start of step 0: (slot 8 - slot 15) / 2 = a step without stamps
    // A partner forward none of this workgroup's tables asks for is skipped: the reply where every game ended with the ego's
    // move (running = 0), the opening where no fresh game starts with the partner (alt_opens = 0).  For such tables the
    // launch-by-launch walk computes a forward whose only outputs are scratch (alt_actions / the log-prob cache of a table
    // that does not move; nothing is recorded, no random stream advances -- Philox is keyed by (counter, table)), so every
    // defined number stays what it was.  With one table per workgroup ~9 of 10 opening forwards go away, and with them the
    // opening's book-keeping pass wherever no table starts a fresh game at all.
    bool need = true;
    if (f != 0 && !r.no_skip) {
      const unsigned char* fl = m.u8 + (f == 1 ? 7 : 9) * 16;   // running | alt_opens, written before the last barrier
      int any = 0, fresh = 0;
      for (int i = 0; i < nrow; ++i) {
        any |= fl[i];
        fresh |= m.u8[10 * 16 + i];                             // ego_opens
      }
      need = __builtin_amdgcn_readfirstlane(any) != 0;
      if (f == 2 && !need && __builtin_amdgcn_readfirstlane(fresh) == 0) {   // liar_sp_after_opening_lane would return at once
        PH_STAMP(prof, 13);
        PH_STAMP(prof, 14);
        continue;
      }
    }
    if (need) {
      // The three argument records share everything but the fields below (ph_liar_selfplay_rollout builds them: same spec, same
      // table count; launch_liar_rollout checks what this relies on).  Selecting whole records -- three kernarg reads of ~1 KB and a
      // few hundred scalar selects per forward -- cost 2.2 k cycles of a 29.7 k-cycle step.
      FwdArgs a;
      __builtin_memset(&a, 0, sizeof(a));   // every field not set below IS zero in the three records (checked at the launch): the
      a.nd = r.ego.nd;                      // tails' optional paths (masks, given actions, logits out, ...) fold away
      a.n = r.ego.n;
      const bool ego_f = f == 0;
      a.params = ego_f ? r.ego.params : r.reply.params;
      a.seed = ego_f ? r.ego.seed : r.reply.seed;
      a.values = ego_f ? r.ego.values : r.reply.values;
      a.logp = ego_f ? r.ego.logp : r.reply.logp;
      a.rb_obs = ego_f ? r.ego.rb_obs + row * r.ego.nd.D : r.reply.rb_obs;
      a.rb_act = ego_f ? r.ego.rb_act + row * r.ego.nd.A : r.reply.rb_act;
      a.rb_rew = ego_f ? r.ego.rb_rew + row : r.reply.rb_rew;
      a.rb_es = ego_f ? r.ego.rb_es + row : r.reply.rb_es;
      a.rb_val = ego_f ? r.ego.rb_val + row : r.reply.rb_val;
      a.rb_logp = ego_f ? r.ego.rb_logp + row : r.reply.rb_logp;
      a.rb_T = ego_f ? r.ego.rb_T : r.reply.rb_T;
      a.counter = (ego_f ? counter : 2ull * counter + (unsigned long long)(f - 1)) + epoch_hi;
      a.epoch = nullptr;
      a.prof = ego_f ? prof : nullptr;   // the body's stamps: the ego forward of the stamped step
      a.obs = ego_f ? s.obs_ego : ((f == 1) ? s.obs_next : s.obs_alt);
      a.es_in = ego_f ? s.ego_episode_start : s.es_alt;
      a.act_i32 = ego_f ? s.ego_actions : s.alt_actions;
      a.pos_env = ego_f ? nullptr : s.alt_pos;
      a.rec_mask = ego_f ? nullptr : s.can;
      if constexpr (NR == 0) {
        ResidentNet rn;   // the acting agent's set, pointer by pointer (scalar selects)
        rn.w2s = f == 0 ? res_ego.w2s : res_alt.w2s;
        rn.wos = f == 0 ? res_ego.wos : res_alt.wos;
        rn.b1s = f == 0 ? res_ego.b1s : res_alt.b1s;
        rn.b2s = f == 0 ? res_ego.b2s : res_alt.b2s;
        rn.hbs = f == 0 ? res_ego.hbs : res_alt.hbs;
        // (the lane number opaque per forward: what the body derives from it -- a few dozen LDS addresses -- is then computed where
        // it is used instead of once in front of the rollout loop and held, spilled, through every phase of every step)
        int tid_f = tid;
        asm volatile("" : "+v"(tid_f));
        policy_fwd16h_body<false, true, true>(a, row0, half, tid_f, sm, row0 + nrow, rn, ooff);
      } else {
        const int wv = __builtin_amdgcn_readfirstlane(tid512 >> 6);   // waves 0 / 1: the policy / the value net; the others wait
        if (wv < 2) {
          float* smn = smem_roll + (size_t)wv * half_floats;
          const ResidentNet rn = resident_net_at(smn + RES_SCRATCH_FLOATS + (f == 0 ? 0 : RES_NET_FLOATS));
          int lane_f = tid512 & 63;
          asm volatile("" : "+v"(lane_f));
          policy_fwd_rows_wave<NR>(a, wv, row0, row0 + nrow, lane_f, smn, rn, ooff);
        }
      }
      __syncthreads();
    }
    PH_STAMP(prof, 9 + 2 * f);
    if (keeper) {
      // the group / lane numbers are made opaque per pass: otherwise the per-lane addresses of the book-keeping (a dozen into the
      // mirror) are hoisted out of the rollout loop and held -- spilled -- across the forwards
      int gi = grp, gk = gl;
      asm volatile("" : "+v"(gi), "+v"(gk));
      if (f == 0) liar_grp_after_ego(m, gc, gi, row0 + gi, gk);
      else if (f == 1) liar_grp_after_reply(m, gc, gi, row0 + gi, gk, r.ego_rew_row0 + row, counter + epoch_hi);
      else liar_grp_after_opening(m, gi, gk);
    }
    __syncthreads();
    PH_STAMP(prof, 10 + 2 * f);
  }

  PH_STAMP(r.ego.prof, 2);   // end of the last step: (slot 2 - slot 15) / n_steps = the average step, stamped step included
  // ---- mirror out ----
  auto copy_out = [&](auto* dst, const auto* src, int per) {
    for (int i = tid512; i < nrow * per; i += 512) dst[(size_t)row0 * per + i] = src[i];
  };
  copy_out(g.hands, m.hands, 12);
  copy_out(g.history, m.history, 24);
  copy_out(g.nmoves, m.nmoves, 1);
  copy_out(g.alt_pos, m.alt_pos, 1);
  copy_out(g.ego_actions, m.ego_act, 2);
  copy_out(g.alt_actions, m.alt_act, 2);
  copy_out(g.obs_ego, m.obs_ego, 30);
  copy_out(g.obs_alt, m.obs_alt, 30);
  copy_out(g.obs_next, m.obs_next, 30);
  copy_out(g.rew1, m.rew1, 2);
  copy_out(g.rew2, m.rew2, 2);
  copy_out(g.es_alt, m.es_alt, 1);
  copy_out(g.ego_episode_start, m.es_ego, 1);
  // (the lane's index is made opaque so that the twelve flag addresses of the mirror-in are recomputed here, not kept across
  // the rollout)
  int lane_row = row0 + tid512;
  asm volatile("" : "+v"(lane_row));
#pragma unroll
  for (int k = 0; k < 12; ++k)
    if (tid512 < nrow) gu8[k][lane_row] = m.u8[16 * k + tid512];
}

static bool fwd16h_eligible(const NetDims& nd, int n);
bool liar_rollout_eligible(const NetDims& nd, int n) { return fwd16h_eligible(nd, n); }

hipError_t launch_liar_rollout(const ph_liar_selfplay& s, const FwdArgs& ego, const FwdArgs& reply, const FwdArgs& opening,
                               int n_steps, unsigned long long counter0, const unsigned long long* epoch, float* ego_rew_row0,
                               hipStream_t st) {
  LiarRolloutArgs r;
  r.s = s;
  r.ego = ego;
  r.reply = reply;
  r.opening = opening;
  r.n_steps = n_steps;
  r.counter0 = counter0;
  r.epoch = epoch;
  r.alt_rewards = s.alt_rb->rewards;
  r.alt_T = s.alt_rb->T;
  r.ego_rew_row0 = ego_rew_row0;
  if (reply.params != opening.params) return hipErrorInvalidValue;   // the partner's two forwards share one resident weight set
  {   // the kernel patches one record per forward: everything outside the patched fields must agree
    // (byte copies: struct assignment need not carry padding bytes, and the records are compared as bytes below)
    FwdArgs e, p, o;
    std::memcpy(&e, &ego, sizeof(FwdArgs));
    std::memcpy(&p, &reply, sizeof(FwdArgs));
    std::memcpy(&o, &opening, sizeof(FwdArgs));
    for (FwdArgs* x : {&e, &p, &o}) {
      x->params = nullptr; x->seed = 0; x->values = nullptr; x->logp = nullptr;
      x->rb_obs = nullptr; x->rb_act = nullptr; x->rb_rew = nullptr; x->rb_es = nullptr; x->rb_val = nullptr; x->rb_logp = nullptr;
      x->rb_T = 0; x->counter = 0; x->epoch = nullptr; x->prof = nullptr;
      x->obs = nullptr; x->es_in = nullptr; x->act_i32 = nullptr; x->pos_env = nullptr; x->rec_mask = nullptr;
    }
    if (std::memcmp(&e, &p, sizeof(FwdArgs)) != 0 || std::memcmp(&p, &o, sizeof(FwdArgs)) != 0) return hipErrorInvalidValue;
    FwdArgs z;   // ... and be zero apart from the spec and the table count (the kernel builds its record from exactly these)
    std::memset(&z, 0, sizeof(z));
    std::memcpy(&z.nd, &ego.nd, sizeof(z.nd));
    z.n = ego.n;
    if (std::memcmp(&e, &z, sizeof(FwdArgs)) != 0) return hipErrorInvalidValue;
    if (reply.seed != opening.seed || reply.values != opening.values || reply.logp != opening.logp || reply.rb_obs != opening.rb_obs ||
        reply.rb_act != opening.rb_act || reply.rb_rew != opening.rb_rew || reply.rb_es != opening.rb_es || reply.rb_val != opening.rb_val ||
        reply.rb_logp != opening.rb_logp || reply.rb_T != opening.rb_T)
      return hipErrorInvalidValue;
  }
  const size_t half = (sizeof(float) * (size_t)(RES_SCRATCH_FLOATS + 2 * RES_NET_FLOATS) + 15) & ~(size_t)15;
  const size_t lds = 2 * half + 80 * sizeof(int) + ((LIAR_MIRROR_BYTES + 15) & ~15);
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  // Tables per workgroup.  The forward's tile is 16 rows, but its one-hot first layer gathers D rows of W1 (256 B each) per
  // table, forward and net through ONE CU's L2 port, and a 256-table game on a 256-CU part leaves 240 CUs idle at 16 tables
  // per workgroup: spread the tables over as many CUs as there are (3.58 -> 2.99 ms per 128-step rollout of 256 tables;
  // 8 / 4 / 2 tables per workgroup: 3.32 / 3.12 / 3.04 ms).  Rows are independent in every phase, so the numbers do not change.
  static int cus[64] = {0};
  // read per launch (two getenv calls against a 2 ms kernel) so that one test process can walk the settings
  const char* e_rpw = getenv("PH_LIAR_RPW");
  const int v_rpw = e_rpw ? atoi(e_rpw) : 0;
  const int forced = (v_rpw >= 1 && v_rpw <= 16) ? v_rpw : 0;
  const char* e_skip = getenv("PH_LIAR_SKIP");
  r.no_skip = (e_skip && e_skip[0] == '0') ? 1 : 0;
  if (cus[dev] == 0) {
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    cus[dev] = n_cu;
  }
  int rpw = forced ? forced : (s.n + cus[dev] - 1) / cus[dev];
  rpw = rpw < 1 ? 1 : (rpw > 16 ? 16 : rpw);
  // the forward's form: one wave per net for up to four tables per workgroup, the 16-row tile above that (bitwise the same rows)
  const char* e_form = getenv("PH_LIAR_WAVE_FORWARD");
  const bool wave_form = !(e_form && e_form[0] == '0');
  const int form = !wave_form ? 0 : (rpw == 1 ? 1 : (rpw == 2 ? 2 : (rpw <= 4 ? 3 : 0)));
  static bool allowed[64][4] = {};
  const void* fn = form == 1 ? (const void*)liar_rollout_kernel<1> : form == 2 ? (const void*)liar_rollout_kernel<2>
                   : form == 3 ? (const void*)liar_rollout_kernel<4> : (const void*)liar_rollout_kernel<0>;
  if (!allowed[dev][form]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed[dev][form] = true;
  }
  const dim3 grid((s.n + rpw - 1) / rpw), block(512);
  const int hf = (int)(half / sizeof(float));
  switch (form) {
    case 1: hipLaunchKernelGGL(liar_rollout_kernel<1>, grid, block, lds, st, r, hf, rpw); break;
    case 2: hipLaunchKernelGGL(liar_rollout_kernel<2>, grid, block, lds, st, r, hf, rpw); break;
    case 3: hipLaunchKernelGGL(liar_rollout_kernel<4>, grid, block, lds, st, r, hf, rpw); break;
    default: hipLaunchKernelGGL(liar_rollout_kernel<0>, grid, block, lds, st, r, hf, rpw); break;
  }
  return hipGetLastError();
}

// one-hot observations of at most 64 components, at most 32 logits (any number of action components)
static bool fwd16h_eligible(const NetDims& nd, int n) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_FWD16H");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled && !nd.gauss && nd.obs_kind != PH_SPACE_BOX && nd.obs_off && nd.D <= 64 && nd.Lp == 32 && n < 16384;
}

template <bool VALU>
static hipError_t launch_fwd16h_variant(const FwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((policy_fwd16h_kernel<VALU>), dim3((a.n + 15) / 16, 2), dim3(256), fwd16h_lds_bytes(), s, a);
  return hipGetLastError();
}

template <int R, int LP, bool VALU>
__global__ __launch_bounds__(R * 4) void policy_fwd_kernel(FwdArgs a) {
  policy_fwd_body<R, LP, VALU>(a);
}

// the same step for several local agents in ONE launch: blockIdx.z selects the agent's argument record (agent-per-GPU
// self-play hosts two learners per GPU; their forwards are independent, 64-workgroup, latency-bound launches)
template <int R, int LP>
__global__ __launch_bounds__(R * 4) void policy_fwd_multi_kernel(FwdMulti m) {
  policy_fwd_body<R, LP, false>(m.a[blockIdx.z]);
}



size_t fwd_lds_bytes(int R, int Lp) {
  const int LDO = Lp + 1;
  return sizeof(float) * (size_t)(2 * R * LDH + 2 * HID * LDH + (HID + R) * LDO + 3 * 64 + R);
}

template <int R, int LP, bool VALU>
static hipError_t launch_fwd_variant(const FwdArgs& a, hipStream_t s) {
  dim3 grid((a.n + R - 1) / R, 2), block(R * 4);
  const size_t lds = fwd_lds_bytes(R, LP);
  static size_t allowed_dev[64] = {0};  // dynamic LDS above 64 KiB is opt-in, per kernel and device (hipFuncSetAttribute)
  size_t& allowed = allowed_dev[current_device_slot()];
  if (allowed == 0) allowed = 64 * 1024;
  if (lds > allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)policy_fwd_kernel<R, LP, VALU>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed = lds;
  }
  hipLaunchKernelGGL((policy_fwd_kernel<R, LP, VALU>), grid, block, lds, s, a);
  return hipGetLastError();
}

template <int R, int LP>
static hipError_t launch_fwd_multi_variant(const FwdMulti& m, int n_agents, hipStream_t s) {
  dim3 grid((m.a[0].n + R - 1) / R, 2, n_agents), block(R * 4);
  const size_t lds = fwd_lds_bytes(R, LP);
  static size_t allowed_dev[64] = {0};
  size_t& allowed = allowed_dev[current_device_slot()];
  if (allowed == 0) allowed = 64 * 1024;
  if (lds > allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)policy_fwd_multi_kernel<R, LP>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed = lds;
  }
  hipLaunchKernelGGL((policy_fwd_multi_kernel<R, LP>), grid, block, lds, s, m);
  return hipGetLastError();
}

// all records must share n and the padded logit count (checked by the ABI layer)
hipError_t launch_policy_fwd_multi(const FwdMulti& m, int n_agents, hipStream_t s) {
  if (fwd16_eligible(m.a[0].nd, m.a[0].n)) {
    bool lean = true;
    for (int i = 0; i < n_agents; ++i) lean = lean && fwd_args_lean(m.a[i], true);
    const dim3 grid((m.a[0].n + 15) / 16, 2, n_agents);
    if (lean) hipLaunchKernelGGL(policy_fwd16_multi_kernel<true>, grid, dim3(256), fwd16_lds_bytes(), s, m);
    else hipLaunchKernelGGL(policy_fwd16_multi_kernel<false>, grid, dim3(256), fwd16_lds_bytes(), s, m);
    return hipGetLastError();
  }
  const bool big = m.a[0].n >= 16384;
  const bool lp64 = m.a[0].nd.Lp == 64;
  if (big) return lp64 ? launch_fwd_multi_variant<64, 64>(m, n_agents, s) : launch_fwd_multi_variant<64, 32>(m, n_agents, s);
  return lp64 ? launch_fwd_multi_variant<32, 64>(m, n_agents, s) : launch_fwd_multi_variant<32, 32>(m, n_agents, s);
}

hipError_t launch_policy_fwd(const FwdArgs& a, int gemm_mode, hipStream_t s) {
  // R = 32 rows per workgroup keeps >= 2*n/32 workgroups in flight for the small-E rollout step
  const bool big = (gemm_mode != 1 && a.n >= 16384);
  const bool lp64 = a.nd.Lp == 64;
  if (a.nd.Lp != 32 && a.nd.Lp != 64) return hipErrorInvalidValue;
  if (fwd16_eligible(a.nd, a.n)) return gemm_mode == 1 ? launch_fwd16_variant<true>(a, s) : launch_fwd16_variant<false>(a, s);
  if (fwd16h_eligible(a.nd, a.n)) return gemm_mode == 1 ? launch_fwd16h_variant<true>(a, s) : launch_fwd16h_variant<false>(a, s);
  if (gemm_mode == 1) return lp64 ? launch_fwd_variant<32, 64, true>(a, s) : launch_fwd_variant<32, 32, true>(a, s);
  if (big) return lp64 ? launch_fwd_variant<64, 64, false>(a, s) : launch_fwd_variant<64, 32, false>(a, s);
  return lp64 ? launch_fwd_variant<32, 64, false>(a, s) : launch_fwd_variant<32, 32, false>(a, s);
}

// ---- env-side illegal action fix-up (pettingzoo.py:81-82): integer, bit-exact ----
__global__ void fix_illegal_kernel(int* actions, const unsigned char* mask, int n, int L) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  const unsigned char* m = mask + (size_t)g * L;
  int act = actions[g];
  if (act < 0 || act >= L || !m[act]) {
    int first = 0;
    for (int k = L - 1; k >= 0; --k)
      if (m[k]) first = k;
    actions[g] = first;
  }
}
hipError_t launch_fix_illegal(int* actions, const unsigned char* mask, int n, int L, hipStream_t s) {
  hipLaunchKernelGGL(fix_illegal_kernel, dim3((n + 255) / 256), dim3(256), 0, s, actions, mask, n, L);
  return hipGetLastError();
}

}  // namespace ph

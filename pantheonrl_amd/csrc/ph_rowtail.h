// Per-row tails of the policy forward shared by the forward kernels (ph_policy.hip) and the ModularPolicy action kernel
// (ph_modular.hip): rollout-buffer row addressing, the stamp-in-band word poll, the value-side row write with the late reward,
// the observation copy of RolloutBuffer.add, and the Discrete (<= 8 logits) distribution tail.
#pragma once
#include "ph_launch.h"

namespace ph {

// Row of the rollout buffer that env g's transition goes to, or -1 when it is not recorded.  Rectangular mode: the
// caller pre-offset the rb_* pointers to row `pos`, so the index is g.  Ragged mode (turn-based games: every env has
// its own write position, SURVEY.md 8e "per-env pos"): rb_* are the array bases and the row is pos_env[g].
__device__ __forceinline__ long long rb_row(const FwdArgs& a, int g) {
  if (!a.pos_env) return g;
  const int p = a.pos_env[g];
  if (!a.rec_mask[g] || p >= a.rb_T) return -1;
  return (long long)p * a.n + g;
}

// one stamp-in-band word of the fused peer-to-peer exchange: spin (bounded) until it carries the expected step's stamp
__device__ __forceinline__ int ll_read(const FwdArgs& a, const unsigned long long* word) {
  const unsigned want = p2p_stamp32(*a.ll_epoch, a.ll_T, a.ll_t);
  const long long t0 = wall_clock64();
  int polls = 0;
  while (true) {
    const unsigned long long v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(v >> 32) == want) return (int)(unsigned)v;
    if ((unsigned long long)(wall_clock64() - t0) > a.ll_timeout) {
      p2p_note_timeout(a.ll_error, 1, a.ll_t, a.joint_ll ? (unsigned long long)(word - a.joint_ll) : 0ull, want, v);
      return (int)(unsigned)v;
    }
    poll_backoff(polls);
  }
}

// both words of a joint action at once: the two reads travel together (one behind the other each was a round trip of its own)
__device__ __forceinline__ void ll_read2(const FwdArgs& a, const unsigned long long* w0, const unsigned long long* w1, int& v0, int& v1) {
  const unsigned want = p2p_stamp32(*a.ll_epoch, a.ll_T, a.ll_t);
  const long long t0 = wall_clock64();
  int polls = 0;
  while (true) {
    const unsigned long long x0 = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long x1 = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    v0 = (int)(unsigned)x0;
    v1 = (int)(unsigned)x1;
    const bool ok0 = (unsigned)(x0 >> 32) == want, ok1 = (unsigned)(x1 >> 32) == want;
    if (ok0 && ok1) return;
    if ((unsigned long long)(wall_clock64() - t0) > a.ll_timeout) {
      const unsigned long long* w = ok0 ? w1 : w0;
      p2p_note_timeout(a.ll_error, 1, a.ll_t, a.joint_ll ? (unsigned long long)(w - a.joint_ll) : 0ull, want, ok0 ? x1 : x0);
      return;
    }
    poll_backoff(polls);
  }
}

// What the value-net tail of row g READS from global memory (rectangular rollouts): fetched at the top of a step, under the
// layers, so that the tail is stores only -- three dependent round trips less at the end of every step of a rollout
// ... and, for the exchange rollouts, room for the two stamped words of the previous step's joint action (own seat, partner seat),
// which value_row_preload_words requests ahead of the tail.  A word that is not there yet is polled in the tail (both together).
struct ValuePre {
  float es, pend, prev;
  unsigned long long w_mine, w_theirs;
  unsigned want;
  int partner, has_words;
};
__device__ __forceinline__ ValuePre value_row_preload(const FwdArgs& a, int g) {
  ValuePre p;
  p.es = a.es_in ? a.es_in[g] : 0.f;
  p.pend = a.pending_reward ? a.pending_reward[g] : 0.f;
  p.prev = (a.pending_reward && a.prev_rew) ? a.prev_rew[g] : 0.f;
  p.w_mine = p.w_theirs = 0ull;
  p.want = 0u;
  p.partner = 0;
  p.has_words = 0;
  return p;
}

// the two words of the previous step's joint action, requested ahead of the tail (callers: just before the head product -- by then
// the words were pushed two layers ago, and the read completes under the product)
__device__ __forceinline__ void value_row_preload_words(const FwdArgs& a, int g, ValuePre& p) {
  if (a.pending_reward && a.joint && a.joint_ll) {
    int q = *a.partner_seat;
    q = q < 0 ? 0 : (q >= a.n_seats ? a.n_seats - 1 : q);
    p.partner = q;
    p.want = p2p_stamp32(*a.ll_epoch, a.ll_T, a.ll_t);
    p.w_mine = __hip_atomic_load(a.joint_ll + (size_t)a.seat * a.n + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    p.w_theirs = __hip_atomic_load(a.joint_ll + (size_t)q * a.n + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    p.has_words = 1;
  }
}

// value-net tail of one row: cache V, write the rollout-buffer row scalars, fold the previous step's late reward in
// (has_pre: `pre` holds the row's inputs, fetched ahead by value_row_preload -- rectangular rollouts only)
__device__ __forceinline__ void value_row_tail(const FwdArgs& a, int g, float v, bool has_pre = false,
                                               ValuePre pre = ValuePre{0.f, 0.f, 0.f, 0ull, 0ull, 0u, 0, 0}) {
  const long long ridx = a.rb_val ? rb_row(a, g) : -1;
  if (a.values && (!a.pos_env || ridx >= 0)) a.values[g] = v;  // ragged: V of the last RECORDED action is cached
  if (ridx >= 0) {
    a.rb_val[ridx] = v;
    a.rb_rew[ridx] = 0.f;
    a.rb_es[ridx] = has_pre ? pre.es : a.es_in[g];
    if (a.pending_reward) {
      float add = has_pre ? pre.pend : a.pending_reward[g];
      if (a.joint) {  // shared coordination term of the synthetic SimultaneousEnv transition (joint action)
        int p = *a.partner_seat;
        p = p < 0 ? 0 : (p >= a.n_seats ? a.n_seats - 1 : p);
        if (a.joint_ll) {
          int mine, theirs;
          if (has_pre && pre.has_words && pre.partner == p && (unsigned)(pre.w_mine >> 32) == pre.want &&
              (unsigned)(pre.w_theirs >> 32) == pre.want) {   // both arrived with the early read
            mine = (int)(unsigned)pre.w_mine;
            theirs = (int)(unsigned)pre.w_theirs;
          } else {
            ll_read2(a, a.joint_ll + (size_t)a.seat * a.n + g, a.joint_ll + (size_t)p * a.n + g, mine, theirs);
          }
          add += joint_reward(mine, theirs, a.bonus, a.reward_rule);
        } else {
          add += joint_reward(a.joint[(size_t)a.seat * a.n + g], a.joint[(size_t)p * a.n + g], a.bonus, a.reward_rule);
        }
      }
      if (has_pre) a.prev_rew[g] = pre.prev + add;
      else a.prev_rew[g] += add;
    }
  }
}

// RolloutBuffer.add copies the observation (agents.py:172-173): rows [row0, row0 + nrow) by the whole workgroup
__device__ __forceinline__ void copy_obs_rows(const FwdArgs& a, int row0, int nrow, int D, int tid = -1, int nt = 0) {
  if (!a.rb_obs) return;
  if (tid < 0) {
    tid = threadIdx.x;
    nt = blockDim.x;
  }
  if (!a.pos_env) {
    const size_t off = (size_t)row0 * D;
    for (int e = tid; e < nrow * D; e += nt) a.rb_obs[off + e] = a.obs[off + e];
  } else {
    for (int e = tid; e < nrow * D; e += nt) {
      const int r = e / D, d = e - r * D;
      const long long ridx = rb_row(a, row0 + r);
      if (ridx >= 0) a.rb_obs[(size_t)ridx * D + d] = a.obs[(size_t)(row0 + r) * D + d];
    }
  }
}

// Philox counter of a forward: the caller's step counter in the low word, the RNG epoch (a device word the iteration graphs
// advance between replays) in the high one.  A launch that knows the epoch (the persistent Liar's Dice rollout reads it once)
// folds it into a.counter and passes no pointer, which removes a dependent round trip to L2 from every sampling tail.
// (Reading the word at the top of the single-step kernels instead measured slower: 6.5 -> 7.0 us per policy_fwd16 launch.)
__device__ __forceinline__ uint64_t fwd_counter(const FwdArgs& a) {
  return a.counter + (a.epoch ? (uint64_t)(*a.epoch) << 32 : 0ull);
}

// Discrete action space with <= 8 logits, one lane per row, the row's logits in registers (z[k >= L] ignored): optional
// mask offset, logits output, sampling / argmax / given action, log-prob, entropy and the rollout-buffer writes
// u_pre: this row's sampling uniform when the caller drew it ahead of time (the same Philox call, off the critical lane)
// NK: 8 = the logit count is a run-time value (nd.L, padding logits carry -3e38 and add exact zeros); 1..7 = it is NK, known at
// compile time -- the loops shrink to the logits that exist.  Bitwise the same results: the generic form only ever adds +0.0 and
// never selects a padding logit.
// -DPH_TAIL_STAMPS (scripts/rollout_phase.py's finer view): shader clock at three points inside the tail, slots 2 / 4 / 6 of
// the workgroup's stamp record, for the step the caller passes a buffer for
#if defined(PH_TAIL_STAMPS)
#define PH_TAIL_STAMP(slot) PH_STAMP(tail_dbg, slot)
#else
#define PH_TAIL_STAMP(slot) do { } while (0)
#endif
template <int NK = 8>
__device__ __forceinline__ int discrete8_row_tail(const FwdArgs& a, const NetDims& nd, int g, float (&zr)[8], uint64_t ctr,
                                                  const float* u_pre = nullptr, long long* tail_dbg = nullptr) {
  const int nk = NK < 8 ? NK : nd.L;
  if (a.mask) {  // modular/policies.py:330-333 : logits - 30*(~mask)
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nk) zr[k] = zr[k] - 30.0f * (1.0f - (float)(a.mask[(size_t)g * nk + k] != 0));
  }
  if (a.logits) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < nk) a.logits[(size_t)g * nk + k] = zr[k];
  }
  float pr[8];
  float m = -3.0e38f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (NK < 8 && k >= NK) continue;   // (compile-time: the loop is unrolled)
    zr[k] = (k < nk) ? zr[k] : -3.0e38f;
    m = fmaxf(m, zr[k]);
  }
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pr[k] = 0.f;
    if (NK < 8 && k >= NK) continue;
    pr[k] = (k < nk) ? fast_exp(zr[k] - m) : 0.f;
    se += pr[k];
  }
  const float lse = m + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
  PH_TAIL_STAMP(2);
  int act = 0;
  if (a.given_actions) {
    act = (int)a.given_actions[g];
    act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
  } else if (a.deterministic) {
    float best = zr[0];
#pragma unroll
    for (int k = 1; k < 8; ++k)
      if (k < nk && zr[k] > best) { best = zr[k]; act = k; }
  } else {
    const float u = a.uniforms ? a.uniforms[g] : (u_pre ? *u_pre : philox_uniform(a.seed, ctr, (uint32_t)g, 0u));
    float cum = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {  // inverse CDF: count prefix sums <= u
      if (NK < 8 && k >= NK - 1) continue;
      cum += pr[k] * inv;
      act += (k < nk - 1 && u >= cum) ? 1 : 0;
    }
  }
  PH_TAIL_STAMP(4);
  float zact = 0.f, ent = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (NK < 8 && k >= NK) continue;
    const float lp = zr[k] - lse;
    ent -= (k < nk) ? pr[k] * inv * lp : 0.f;
    zact = (k == act) ? zr[k] : zact;
  }
  const float logp = zact - lse;
  // What the ENVIRONMENT consumes: with fix_illegal an illegal sample becomes the first legal index (pettingzoo.py:81-82 does
  // this inside the env, before base_env.step); the agent is not told, so its buffer row and log-prob keep the sampled action.
  int env_act = act;
  if (a.env_mask && a.env_mask[(size_t)g * nk + act] == 0) {
    env_act = 0;
#pragma unroll
    for (int k = 7; k >= 0; --k)
      if (k < nk && a.env_mask[(size_t)g * nk + k] != 0) env_act = k;
  }
  PH_TAIL_STAMP(6);
  if (a.act_i32) a.act_i32[g] = env_act;
  if (a.act_f32) a.act_f32[g] = (float)act;
  if (a.logp) a.logp[g] = logp;
  if (a.entropy) a.entropy[g] = ent;
  if (a.rb_act || a.rb_logp) {
    const long long ridx = rb_row(a, g);
    if (ridx >= 0) {
      if (a.rb_act) a.rb_act[ridx] = (float)act;
      if (a.rb_logp) a.rb_logp[ridx] = logp;
    }
  }
  return env_act;
}

}  // namespace ph

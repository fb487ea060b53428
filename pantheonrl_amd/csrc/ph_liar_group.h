// Liar's Dice self-play book-keeping with a table spread over 32 lanes -- the form the persistent rollout kernel runs
// (ph_policy.hip: liar_rollout_kernel).  ph_liar.h keeps a table in one lane's registers, which is right for the per-step
// kernels (thousands of tables, one launch each) and wrong inside the persistent kernel: there a workgroup owns <= 16 tables,
// with 256 tables one, and its 511 other lanes wait at a barrier while one lane walks ~300-550 dependent instructions per
// pass (7-9 cycles each: 2.1 k / 4.9 k / 1.4 k cycles of a 39 k-cycle step, profiles/r05_ab_liar_rollout_phase.txt).
// Here a table is a half wave: lane l holds history word l (l < 24) / hand word l (l < 12) / observation element l (l < 30),
// the history shift and an observation row are one LDS store each, the four Philox blocks of a re-deal run in four lanes.
// The state is the workgroup's LDS mirror (LiarMirror), indexed by the local table number i; e = the global table number
// (random streams, rollout-buffer columns).  Every rule is the one of ph_liar.h (same reference lines), and the two forms are
// held bitwise equal by the persistent-vs-stepwise tests (tests/test_gpu_parity.py: every array of the self-play state, both
// rollout buffers, after whole rollouts).
#pragma once
#include "ph_liar.h"

namespace ph {

// The tables a workgroup owns, in LDS (filled and written back by liar_rollout_kernel)
struct LiarMirror {
  int* hands;      // [16][12]
  int* history;    // [16][24]
  int* nmoves;     // [16]
  int* alt_pos;    // [16]
  int* ego_act;    // [16][2]
  int* alt_act;    // [16][2]
  float* obs_ego;  // [16][30]
  float* obs_alt;
  float* obs_next;
  float* rew1;     // [16][2]
  float* rew2;
  float* es_alt;   // [16]
  float* es_ego;   // [16]
  unsigned char* u8;   // [12][16], rows:
  enum Flag { EGO_FIRST = 0, ALT_BOUNDARY, ALT_TERM, ALT_OPEN, ALT_ACTED, DONE1, DONE2, RUNNING, CAN, ALT_OPENS, EGO_OPENS, DONE };
  __device__ __forceinline__ unsigned char& flag(Flag k, int i) const { return u8[16 * (int)k + i]; }
};
constexpr int LIAR_MIRROR_BYTES = 16 * (12 + 24 + 1 + 1 + 2 + 2) * 4 + 16 * (3 * 30 + 2 + 2 + 1 + 1) * 4 + 12 * 16;

// what a pass needs of the launch besides the mirror
struct LiarGroupCtx {
  int n;                           // tables of the whole game (column stride of the rollout buffers)
  float* alt_rewards;              // partner buffer rewards [alt_T][n]
  int alt_T;
  unsigned long long* episodes;
  uint64_t dice_seed;
  float probegostart;
};

// LDS operations of a wave execute in issue order, so a lane's read behind another lane's write (same wave) sees it; this keeps
// the compiler from moving them across and waits for the data of the reads before
__device__ __forceinline__ void liar_grp_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// LiarEnv.getObs (liar.py:53-56) of table i for one player into row i of `obs`: lane l < 30 writes element l
__device__ __forceinline__ void liar_grp_write_obs(const LiarMirror& m, int i, int l, bool ego, int nm, float* obs) {
  if (l >= 30) return;
  float v;
  if (l < 6) {
    v = (float)m.hands[12 * i + (ego ? 0 : 6) + l];
  } else {
    const int j = l - 6;
    v = ((j >> 1) < nm) ? (float)m.history[24 * i + j] : ((j & 1) ? 0.f : (float)LD_SIDES);   // the null move [6, 0]
  }
  obs[30 * i + l] = v;
}

// One move in table i (liar_move's rules: sanitize_action liar.py:58-67, eval_bluff :69-75, step :77-102) proposed in act[2i..]
// by `ego`; writes rew / the done flag / the observation of the OTHER player into obs_next and, where obs2 is given and the game
// goes on (or obs2_always), the same row into obs2.  nm_out: moves on the table after.
// Everything the pass needs of the table is read at its top, in one LDS round trip: lane l takes the history words the shift
// (l - 2, l) and its observation element (l - 8, l - 6: the shifted history seen from element l) are made of, so neither waits
// for the shifted history to be stored.
__device__ __forceinline__ LiarOutcome liar_grp_move(const LiarMirror& m, int i, int l, const int* act, bool ego, float* rew,
                                                     LiarMirror::Flag done_flag, int& nm_out, float* obs2 = nullptr,
                                                     bool obs2_always = false) {
  const int* hist = m.history + 24 * i;
  const int nm = m.nmoves[i];
  int a0 = act[2 * i], a1 = act[2 * i + 1];
  const int h0 = hist[0], h1 = hist[1];
  const int j = l - 6;                                       // history word of observation element l
  const int w_m2_ld = hist[(l >= 2 && l < 24) ? l - 2 : 0], w_0_ld = hist[l < 24 ? l : 0];
  const int o_m2_ld = hist[(j >= 2 && j < 24) ? j - 2 : 0], o_0_ld = hist[(j >= 0 && j < 24) ? j : 0];
  int hand = m.hands[12 * i + (ego ? 6 : 0) + (l < 6 ? l : 0)];   // the other player's hand (getObs(not isego))
  int w_m2 = w_m2_ld, w_0 = w_0_ld, o_m2 = o_m2_ld, o_0 = o_0_ld;
  // Other lanes overwrite the words this lane just read (lane l - 2 stores hist[l - 2] below; obs_next / obs2 may alias what a
  // caller reads next).  Per thread those addresses do not alias, so nothing but this keeps the loads above the stores: the values
  // are made opaque (they must be in registers here) and no memory access may move across.
  asm volatile("" : "+v"(w_m2), "+v"(w_0), "+v"(o_m2), "+v"(o_0), "+v"(hand) : : "memory");
  bool call = false;
  if (nm > 0) {
    if (a1 <= h1 || a0 == LD_SIDES) call = true;
  } else if (a0 == LD_SIDES) {
    a0 = 0;
    a1 = 0;
  }
  if (!call && a0 == LD_SIDES && a1 == 2 * LD_DICE - 1) call = true;  // the literal "bluff!" move
  LiarOutcome o{0.f, 0.f, false};
  const bool shifted = !call && nm < LD_MAXMOVES;
  if (call) {
    bool bluff = false;
    if (nm > 0) {
      const int have = (h0 >= 0 && h0 < LD_SIDES) ? m.hands[12 * i + h0] + m.hands[12 * i + 6 + h0] : 0;
      bluff = h1 > have - 1;
    }
    o.r_ego = (bluff == ego) ? 1.f : -1.f;
    o.r_alt = -o.r_ego;
    o.done = true;
  } else if (shifted) {
    // newest first: word l takes word l - 2 where that is a recorded move
    if (l < 24) m.history[24 * i + l] = (l < 2) ? (l == 0 ? a0 : a1) : ((l - 2 < 2 * nm) ? w_m2 : w_0);
    if (l == 0) m.nmoves[i] = nm + 1;
  }
  nm_out = shifted ? nm + 1 : nm;
  if (l < 30) {   // LiarEnv.getObs (liar.py:53-56) of the table after the move
    float v;
    if (l < 6) {
      v = (float)hand;
    } else {
      const int word = shifted ? ((j < 2) ? (j == 0 ? a0 : a1) : ((j - 2 < 2 * nm) ? o_m2 : o_0)) : o_0;
      v = ((j >> 1) < nm_out) ? (float)word : ((j & 1) ? 0.f : (float)LD_SIDES);   // the null move [6, 0]
    }
    m.obs_next[30 * i + l] = v;
    if (obs2 && (obs2_always || !o.done)) obs2[30 * i + l] = v;
  }
  if (l == 0) {
    rew[2 * i] = o.r_ego;
    rew[2 * i + 1] = o.r_alt;
    m.flag(done_flag, i) = o.done ? 1 : 0;
  }
  return o;
}

// LiarEnv.multi_reset of table i (liar_deal's draws: die d = word d % 4 of Philox block d / 4 keyed (seed, counter, e), the
// first mover from word 0 of block 100): lanes 0..2 draw the dice blocks, lane 3 the first mover.  Returns ego_first.
__device__ __forceinline__ bool liar_grp_deal(const LiarMirror& m, const LiarGroupCtx& c, int i, int e, int l, uint64_t counter) {
  const int b = l & 3;
  float u4[4];
  philox_uniform4(c.dice_seed, counter, (uint32_t)e, b == 3 ? 100u : (uint32_t)b, u4);
  int side[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int s = (int)(u4[w] * LD_SIDES);
    side[w] = s >= LD_SIDES ? LD_SIDES - 1 : s;
  }
  const int first = __shfl((u4[0] < c.probegostart) ? 1 : 0, 3, 32);
  // lane l < 12 counts side l % 6 among the six dice of player l / 6
  const int k = l % 6, p = l / 6;
  int cnt = 0;
#pragma unroll
  for (int d = 0; d < 2 * LD_DICE; ++d) {
    const int sd = __shfl(side[d & 3], d >> 2, 32);
    cnt += ((d < LD_DICE ? 0 : 1) == p && sd == k) ? 1 : 0;
  }
  if (l < 12) m.hands[12 * i + l] = cnt;
  if (l < 24) m.history[24 * i + l] = 0;
  if (l == 0) {
    m.nmoves[i] = 0;
    m.flag(LiarMirror::EGO_FIRST, i) = first ? 1 : 0;
  }
  liar_grp_sync();
  return first != 0;
}

// the partner's rollout cursor of table i (LiarSeat), read by every lane of the group, written by lane 0
__device__ __forceinline__ LiarSeat liar_grp_seat(const LiarMirror& m, int i) {
  LiarSeat q;
  q.pos = m.alt_pos[i];
  q.boundary = m.flag(LiarMirror::ALT_BOUNDARY, i) != 0;
  q.open = m.flag(LiarMirror::ALT_OPEN, i) != 0;
  q.acted = m.flag(LiarMirror::ALT_ACTED, i) != 0;
  return q;
}
__device__ __forceinline__ void liar_grp_credit(const LiarMirror& m, const LiarGroupCtx& c, LiarSeat& q, int i, int e, int l, float r,
                                                bool done, bool credited) {   // liar_sp_credit
  const bool mm = credited && q.open;
  if (l == 0) {
    if (mm && q.pos >= 1 && q.pos <= c.alt_T) liar_add_f32(c.alt_rewards + (size_t)(q.pos - 1) * c.n + e, r);
    if (done) m.flag(LiarMirror::ALT_BOUNDARY, i) = 1;
    if (mm && done) m.flag(LiarMirror::ALT_TERM, i) = 1;
  }
  if (done) q.boundary = true;
}
__device__ __forceinline__ void liar_grp_prepare(const LiarMirror& m, const LiarGroupCtx& c, const LiarSeat& q, int i, int l,
                                                 bool requested) {   // liar_sp_prepare
  if (l == 0) {
    m.flag(LiarMirror::CAN, i) = (requested && q.pos < c.alt_T) ? 1 : 0;
    m.es_alt[i] = q.boundary ? 1.f : 0.f;
  }
}
__device__ __forceinline__ void liar_grp_commit(const LiarMirror& m, LiarSeat& q, int i, int l, bool can) {   // liar_sp_commit
  if (can) {
    q.pos += 1;
    q.boundary = false;
  }
  q.open = can;
  q.acted = true;
  if (l == 0) {
    if (can) {
      m.alt_pos[i] = q.pos;
      m.flag(LiarMirror::ALT_BOUNDARY, i) = 0;
      m.flag(LiarMirror::ALT_TERM, i) = 0;
    }
    m.flag(LiarMirror::ALT_OPEN, i) = can ? 1 : 0;
    m.flag(LiarMirror::ALT_ACTED, i) = 1;
  }
}

// liar_sp_after_ego_lane
__device__ __forceinline__ void liar_grp_after_ego(const LiarMirror& m, const LiarGroupCtx& c, int i, int e, int l) {
  LiarSeat q = liar_grp_seat(m, i);
  int nm;
  const LiarOutcome o1 = liar_grp_move(m, i, l, m.ego_act, true, m.rew1, LiarMirror::DONE1, nm);
  liar_grp_credit(m, c, q, i, e, l, o1.r_alt, o1.done, q.acted);
  if (l == 0) m.flag(LiarMirror::RUNNING, i) = o1.done ? 0 : 1;
  liar_grp_prepare(m, c, q, i, l, !o1.done);
}
// liar_sp_after_reply_lane (never deal-only inside the rollout)
__device__ __forceinline__ void liar_grp_after_reply(const LiarMirror& m, const LiarGroupCtx& c, int i, int e, int l, float* ego_rew_row,
                                                     uint64_t counter) {
  LiarSeat q = liar_grp_seat(m, i);
  bool ego_first = m.flag(LiarMirror::EGO_FIRST, i) != 0;
  const bool run = m.flag(LiarMirror::RUNNING, i) != 0, can = m.flag(LiarMirror::CAN, i) != 0;
  const bool d1 = m.flag(LiarMirror::DONE1, i) != 0;
  const float r1_ego = m.rew1[2 * i];
  LiarOutcome o2{0.f, 0.f, false};
  int nm = 0;
  if (run) {
    liar_grp_commit(m, q, i, l, can);
    o2 = liar_grp_move(m, i, l, m.alt_act, false, m.rew2, LiarMirror::DONE2, nm, m.obs_ego);   // obs_ego = obs_next where the game goes on
  }
  const bool d2 = run && o2.done;
  liar_grp_credit(m, c, q, i, e, l, o2.r_alt, d2, run);
  const bool done = d1 || d2;
  if (l == 0) {
    liar_add_f32(ego_rew_row + e, r1_ego + (run ? o2.r_ego : 0.f));   // both transitions of the step (agents.py:44-47)
    m.es_ego[i] = done ? 1.f : 0.f;
    m.flag(LiarMirror::DONE, i) = done ? 1 : 0;
    if (done) atomicAdd(c.episodes, 1ull);
  }
  if (done) {
    liar_grp_sync();   // (a table that ended with the reply: its last observation was read from the hands the deal replaces)
    ego_first = liar_grp_deal(m, c, i, e, l, counter);
    if (l == 0) m.flag(LiarMirror::ALT_ACTED, i) = 0;
  }
  const bool alt_opens = done && !ego_first;
  if (l == 0) {
    m.flag(LiarMirror::ALT_OPENS, i) = alt_opens ? 1 : 0;
    m.flag(LiarMirror::EGO_OPENS, i) = (done && ego_first) ? 1 : 0;
  }
  liar_grp_prepare(m, c, q, i, l, alt_opens);
  if (alt_opens) liar_grp_write_obs(m, i, l, false, 0, m.obs_alt);
}
// liar_sp_after_opening_lane
__device__ __forceinline__ void liar_grp_after_opening(const LiarMirror& m, int i, int l) {
  const bool alt_opens = m.flag(LiarMirror::ALT_OPENS, i) != 0, ego_opens = m.flag(LiarMirror::EGO_OPENS, i) != 0;
  if (!alt_opens && !ego_opens) return;
  if (alt_opens) {
    LiarSeat q = liar_grp_seat(m, i);
    int nm;
    liar_grp_commit(m, q, i, l, m.flag(LiarMirror::CAN, i) != 0);
    (void)liar_grp_move(m, i, l, m.alt_act, false, m.rew2, LiarMirror::DONE2, nm, m.obs_ego, true);   // the ego's view after the opening move
  } else {
    liar_grp_write_obs(m, i, l, true, m.nmoves[i], m.obs_ego);   // of the fresh deal
  }
}

}  // namespace ph

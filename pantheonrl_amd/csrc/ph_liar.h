// Liar's Dice on the device: a table's state, the game rules (pantheonrl/envs/liargym/liar.py:53-102) and the per-table
// book-keeping of the vectorised self-play step as lane functions (one lane per table): what the per-step kernels run
// (ph_envs.hip).  The persistent rollout kernel (ph_policy.hip: liar_rollout_kernel) runs the same rules with a table spread over
// 32 lanes (ph_liar_group.h); tests hold the two forms bitwise equal.
#pragma once
#include "ph_launch.h"

namespace ph {

constexpr int LD_SIDES = 6, LD_DICE = 6, LD_MAXMOVES = 12;

// A table's state lives in registers while a lane works on it: 16-byte loads / stores of the (12) hand and (24) history rows,
// every index a compile-time constant (a loop of dependent global loads and stores costs a memory latency per iteration).
struct LiarTable {
  int hand[12];   // ego histogram (6) then partner histogram (6)
  int hist[24];   // moves newest first (side, count-1)
  int nm;
};
__device__ __forceinline__ void liar_load(LiarTable& t, int e, const int* hands, const int* history, const int* nmoves) {
  const int4* hp = reinterpret_cast<const int4*>(hands + (size_t)e * 12);
  const int4* qp = reinterpret_cast<const int4*>(history + (size_t)e * 24);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int4 v = hp[i];
    t.hand[4 * i] = v.x; t.hand[4 * i + 1] = v.y; t.hand[4 * i + 2] = v.z; t.hand[4 * i + 3] = v.w;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int4 v = qp[i];
    t.hist[4 * i] = v.x; t.hist[4 * i + 1] = v.y; t.hist[4 * i + 2] = v.z; t.hist[4 * i + 3] = v.w;
  }
  t.nm = nmoves[e];
}
__device__ __forceinline__ void liar_store_history(const LiarTable& t, int e, int* history, int* nmoves) {
  int4* qp = reinterpret_cast<int4*>(history + (size_t)e * 24);
#pragma unroll
  for (int i = 0; i < 6; ++i) qp[i] = make_int4(t.hist[4 * i], t.hist[4 * i + 1], t.hist[4 * i + 2], t.hist[4 * i + 3]);
  nmoves[e] = t.nm;
}
__device__ __forceinline__ void liar_store_hands(const LiarTable& t, int e, int* hands) {
  int4* hp = reinterpret_cast<int4*>(hands + (size_t)e * 12);
#pragma unroll
  for (int i = 0; i < 3; ++i) hp[i] = make_int4(t.hand[4 * i], t.hand[4 * i + 1], t.hand[4 * i + 2], t.hand[4 * i + 3]);
}

// LiarEnv.getObs (liar.py:53-56): a player's hand + the history padded with the null move [6, 0]; o = 30 floats, 8-byte aligned
__device__ __forceinline__ void liar_write_obs(const LiarTable& t, bool ego, float* o) {
  float2* o2 = reinterpret_cast<float2*>(o);
#pragma unroll
  for (int k = 0; k < 3; ++k)
    o2[k] = make_float2((float)(ego ? t.hand[2 * k] : t.hand[6 + 2 * k]), (float)(ego ? t.hand[2 * k + 1] : t.hand[7 + 2 * k]));
#pragma unroll
  for (int m = 0; m < LD_MAXMOVES; ++m)
    o2[3 + m] = make_float2((float)(m < t.nm ? t.hist[2 * m] : LD_SIDES), (float)(m < t.nm ? t.hist[2 * m + 1] : 0));
}
// One move of Liar's Dice in table e (state in t, written back when it changes).
//   actions (n, 2)  int32 : raw (side, count-1) proposed by whoever moves; `ego` says who that is
// Outputs: obs_next (n, 30) f32 = observation of the OTHER player (liar.py:53-56), rew (n, 2) f32 (ego, partner),
//          done (n) u8.  History / nmoves are updated in place.  What was written to rew / done is also returned: a caller
//          that goes on with it must not read it back (a load behind a store waits for the store's acknowledgement).
struct LiarOutcome {
  float r_ego, r_alt;
  bool done;
};
__device__ __forceinline__ LiarOutcome liar_move(LiarTable& t, int e, int* history, int* nmoves, const int* actions, bool ego,
                                          float* obs_next, float* rew, unsigned char* done) {
  const int nm = t.nm;
  const int2 act = *reinterpret_cast<const int2*>(actions + 2 * (size_t)e);
  int a0 = act.x, a1 = act.y;
  // sanitize_action (liar.py:58-67)
  bool call = false;
  if (nm > 0) {
    if (a1 <= t.hist[1] || a0 == LD_SIDES) call = true;
  } else if (a0 == LD_SIDES) {
    a0 = 0;
    a1 = 0;
  }
  if (!call && a0 == LD_SIDES && a1 == 2 * LD_DICE - 1) call = true;  // the literal "bluff!" move
  float r_ego = 0.f, r_alt = 0.f;
  unsigned char d = 0;
  if (call) {
    bool bluff = false;  // eval_bluff (liar.py:69-75)
    if (nm > 0) {
      const int side = t.hist[0];
      int have = 0;
#pragma unroll
      for (int k = 0; k < LD_SIDES; ++k) have += (k == side) ? t.hand[k] + t.hand[6 + k] : 0;
      bluff = t.hist[1] > have - 1;
    }
    const bool ego_wins = (bluff == ego);
    r_ego = ego_wins ? 1.f : -1.f;
    r_alt = -r_ego;
    d = 1;
  } else if (nm < LD_MAXMOVES) {
#pragma unroll
    for (int k = 21; k >= 0; --k) t.hist[k + 2] = (k < 2 * nm) ? t.hist[k] : t.hist[k + 2];
    t.hist[0] = a0;
    t.hist[1] = a1;
    t.nm = nm + 1;
    liar_store_history(t, e, history, nmoves);
  }
  liar_write_obs(t, !ego, obs_next + (size_t)e * 30);  // getObs(not isego)
  *reinterpret_cast<float2*>(rew + 2 * (size_t)e) = make_float2(r_ego, r_alt);
  done[e] = d;
  return LiarOutcome{r_ego, r_alt, d != 0};
}

// LiarEnv.multi_reset of table e: N_DICE dice per player from Philox4x32-10 (one 24-bit draw per die, like the reference's
// randint per die: die d is word d%4 of Philox block d/4 keyed (seed, counter, e)), empty history, first mover ~
// Bernoulli(probegostart) from word 0 of block 100
__device__ __forceinline__ bool liar_deal(LiarTable& t, int e, int* hands, int* history, int* nmoves, unsigned char* ego_first,
                                          uint64_t seed, uint64_t counter, float probegostart) {
#pragma unroll
  for (int k = 0; k < 12; ++k) t.hand[k] = 0;
#pragma unroll
  for (int blk = 0; blk < 2 * LD_DICE / 4; ++blk) {
    float u4[4];
    philox_uniform4(seed, counter, (uint32_t)e, (uint32_t)blk, u4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int die = 4 * blk + i;
      int side = (int)(u4[i] * LD_SIDES);
      side = side >= LD_SIDES ? LD_SIDES - 1 : side;
#pragma unroll
      for (int k = 0; k < LD_SIDES; ++k) t.hand[(die < LD_DICE ? 0 : 6) + k] += (k == side) ? 1 : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < 24; ++k) t.hist[k] = 0;
  t.nm = 0;
  liar_store_hands(t, e, hands);
  liar_store_history(t, e, history, nmoves);
  const bool first = philox_uniform(seed, counter, (uint32_t)e, 100u) < probegostart;
  ego_first[e] = first ? 1 : 0;
  return first;   // = ego_first[e]
}

// ---- vectorised Liar's Dice self-play: the step loop's book-keeping, one lane per table ------------------------------------
// (MultiAgentEnv._update_players / _get_actions, multiagentenv.py:149-170, and OnPolicyAgent.update, agents.py:186-203,
// applied to n tables; the partner's rollout rows are ragged: table e writes row alt_pos[e]).  A table's state is touched
// by its own lane only, so everything between two policy forwards is ONE launch: a vectorised step is
//   ego forward | after_ego | partner forward | after_reply | partner forward (openers) | after_opening
//
// Each pass reads what it needs of the table's book-keeping ONCE, at its top (the loads are independent of each other: one
// round trip), carries it in registers (LiarSeat) and stores what changed: on this target loads and stores return in issue
// order, so a read-back of a flag behind a store -- or behind the late-reward atomic to HBM -- waits for that store's
// acknowledgement, and a pass written as "store the flag, read the flag" pays one such wait per flag.
__device__ __forceinline__ void liar_add_f32(float* p, float v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v);
}
// the partner's side of table e: where its next rollout row goes and the state of the reward window of its last recorded row
struct LiarSeat {
  int pos;         // alt_pos: rows recorded in this table's column so far
  bool boundary;   // alt_boundary: a game ended since the last recorded row (its episode_start)
  bool open;       // alt_open: the last forward was recorded, later rewards belong to it
  bool acted;      // alt_acted: the partner has moved in the current game
};
__device__ __forceinline__ LiarSeat liar_seat_load(const ph_liar_selfplay& s, int e) {
  LiarSeat q;
  q.pos = s.alt_pos[e];
  q.boundary = s.alt_boundary[e] != 0;
  q.open = s.alt_open[e] != 0;
  q.acted = s.alt_acted[e] != 0;
  return q;
}
// a reward r / an episode end that follows the partner's last forward (credited: that forward happened in this game)
__device__ __forceinline__ void liar_sp_credit(const ph_liar_selfplay& s, LiarSeat& q, float* alt_rewards, int alt_T, int e, float r,
                                               bool done, bool credited) {
  const bool m = credited && q.open;
  // no-return float atomic: the address is this table's alone, so the sum is the plain "+=" -- but the lane does not wait a
  // round trip to HBM for the old value in the middle of its book-keeping
  if (m && q.pos >= 1 && q.pos <= alt_T) liar_add_f32(alt_rewards + (size_t)(q.pos - 1) * s.n + e, r);
  if (done) {
    s.alt_boundary[e] = 1;
    q.boundary = true;
  }
  if (m && done) s.alt_term[e] = 1;
}
// what the partner's next forward records: a row where it is asked to move and its column still has room
__device__ __forceinline__ void liar_sp_prepare(const ph_liar_selfplay& s, const LiarSeat& q, int alt_T, int e, bool requested) {
  s.can[e] = (requested && q.pos < alt_T) ? 1 : 0;
  s.es_alt[e] = q.boundary ? 1.f : 0.f;
}
// after a partner forward (can: it recorded a row): advance the recorded column, open / close the reward window, mark the
// partner as having acted
__device__ __forceinline__ void liar_sp_commit(const ph_liar_selfplay& s, LiarSeat& q, int e, bool can) {
  if (can) {
    q.pos += 1;
    s.alt_pos[e] = q.pos;
    s.alt_boundary[e] = 0;
    q.boundary = false;
    s.alt_term[e] = 0;
  }
  q.open = can;            // a full column: a later reward belongs to a row that was not recorded
  s.alt_open[e] = can ? 1 : 0;
  q.acted = true;
  s.alt_acted[e] = 1;
}

// the ego has moved (its forward wrote ego_actions): play the move, credit the partner where it already acted this game,
// find the tables that go on and prepare the partner's reply there
__device__ __forceinline__ void liar_sp_after_ego_lane(const ph_liar_selfplay& s, int e, float* alt_rewards, int alt_T) {
  LiarTable t;
  liar_load(t, e, s.hands, s.history, s.nmoves);
  LiarSeat q = liar_seat_load(s, e);
  const LiarOutcome o1 = liar_move(t, e, s.history, s.nmoves, s.ego_actions, true, s.obs_next, s.rew1, s.done1);
  liar_sp_credit(s, q, alt_rewards, alt_T, e, o1.r_alt, o1.done, q.acted);
  s.running[e] = o1.done ? 0 : 1;
  liar_sp_prepare(s, q, alt_T, e, !o1.done);
}
// the partner has replied where the game went on: play that move, credit both, the ego's reward row / episode flags /
// next observation; then (also the whole of a deal-only call) re-deal the finished tables, find who opens the new games
// and prepare the partner's opening forward
__device__ __forceinline__ void liar_sp_after_reply_lane(const ph_liar_selfplay& s, int e, float* alt_rewards, int alt_T,
                                                         float* ego_rew_row, uint64_t counter, const unsigned long long* epoch,
                                                         int deal_only) {
  LiarTable t;
  liar_load(t, e, s.hands, s.history, s.nmoves);
  LiarSeat q = liar_seat_load(s, e);
  bool ego_first = s.ego_first[e] != 0;    // of the game in progress; a re-deal below replaces it
  bool fresh = s.done[e] != 0;             // a deal-only call: the caller's flags
  if (!deal_only) {
    const bool run = s.running[e] != 0, can = s.can[e] != 0, d1 = s.done1[e] != 0;
    const float r1_ego = s.rew1[2 * e];
    LiarOutcome o2{0.f, 0.f, false};
    if (run) {
      liar_sp_commit(s, q, e, can);
      o2 = liar_move(t, e, s.history, s.nmoves, s.alt_actions, false, s.obs_next, s.rew2, s.done2);
    }
    const bool d2 = run && o2.done;
    liar_sp_credit(s, q, alt_rewards, alt_T, e, o2.r_alt, d2, run);
    const bool done = d1 || d2;
    liar_add_f32(ego_rew_row + e, r1_ego + (run ? o2.r_ego : 0.f));   // both transitions of the step (agents.py:44-47)
    s.ego_episode_start[e] = done ? 1.f : 0.f;
    if (run && !d2) liar_write_obs(t, true, s.obs_ego + (size_t)e * 30);   // = obs_next of the move just played
    s.done[e] = done ? 1 : 0;
    if (done) atomicAdd(s.episodes, 1ull);
    fresh = done;
  }
  if (fresh) {
    ego_first = liar_deal(t, e, s.hands, s.history, s.nmoves, s.ego_first, s.dice_seed,
                          counter + (epoch ? (uint64_t)(*epoch) << 32 : 0ull), s.probegostart);
    s.alt_acted[e] = 0;
  }
  s.alt_opens[e] = (fresh && !ego_first) ? 1 : 0;
  s.ego_opens[e] = (fresh && ego_first) ? 1 : 0;
  liar_sp_prepare(s, q, alt_T, e, fresh && !ego_first);
  if (fresh && !ego_first) liar_write_obs(t, false, s.obs_alt + (size_t)e * 30);
}
// the partner has opened the new games it starts: play that move; the ego's observation of every fresh table
__device__ __forceinline__ void liar_sp_after_opening_lane(const ph_liar_selfplay& s, int e) {
  const bool alt_opens = s.alt_opens[e] != 0, ego_opens = s.ego_opens[e] != 0;
  if (!alt_opens && !ego_opens) return;
  LiarTable t;
  liar_load(t, e, s.hands, s.history, s.nmoves);
  if (alt_opens) {
    LiarSeat q = liar_seat_load(s, e);
    liar_sp_commit(s, q, e, s.can[e] != 0);
    (void)liar_move(t, e, s.history, s.nmoves, s.alt_actions, false, s.obs_next, s.rew2, s.done2);
  }
  liar_write_obs(t, true, s.obs_ego + (size_t)e * 30);   // after the partner's opening move, or of the fresh deal
}

}  // namespace ph

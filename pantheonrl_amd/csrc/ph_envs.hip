// Vectorised integer rules of the two games whose rules live entirely in the reference tree:
// rock-paper-scissors (pantheonrl/envs/rpsgym/rps.py:41-45) and Liar's Dice (pantheonrl/envs/liargym/liar.py:53-83).
// One lane per environment; everything is integer arithmetic and must be bit-exact with the Python games.
#include "ph_launch.h"

namespace ph {

// ego payoff: (ego - alt + 3) % 3 mapped {0: 0, 1: +1, 2: -1}; zero-sum
__global__ void rps_step_kernel(const int* __restrict__ ego_act, const int* __restrict__ alt_act,
                                float* __restrict__ ego_rew, float* __restrict__ alt_rew, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int o = (ego_act[e] - alt_act[e] + 3) % 3;
  o = (o == 2) ? -1 : o;
  ego_rew[e] = (float)o;
  alt_rew[e] = (float)(-o);
}
hipError_t launch_rps_step(const int* ego_act, const int* alt_act, float* ego_rew, float* alt_rew, int n, hipStream_t s) {
  hipLaunchKernelGGL(rps_step_kernel, dim3((n + 255) / 256), dim3(256), 0, s, ego_act, alt_act, ego_rew, alt_rew, n);
  return hipGetLastError();
}

// Synthetic 2-player SimultaneousEnv transition of the one-agent-per-GPU round-robin layout (BASELINE config 4; reference
// multiagentenv.py:149-243 with E environments): environment e is currently partnered with partner partnerid[e].  From the
// all-gathered actions (row 0 = ego, row 1 + k = partner k) it takes the action of e's partner, pays the shared reward
// base[e] + bonus * [ego action == partner action], and, where the episode ends, advances e's partner id round-robin -- the
// next reset's resample_round_robin (multiagentenv.py:118-125,224) of that environment alone.  The header columns of the next
// step's routing block [partner id | the reward and done flag just produced] are written for the partner ranks.
__global__ void roundrobin_env_step_kernel(const int* __restrict__ joint, int* __restrict__ partnerid,
                                           const float* __restrict__ base, const float* __restrict__ done,
                                           float* __restrict__ reward_out, int* __restrict__ alt_action_out,
                                           float* __restrict__ next_block, int block_ld, int n_partners, float bonus, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int pid = partnerid[e];
  pid = pid < 0 ? 0 : (pid >= n_partners ? n_partners - 1 : pid);
  const int a_ego = joint[e], a_alt = joint[(size_t)(1 + pid) * n + e];
  const float r = base[e] + ((a_ego == a_alt) ? bonus : 0.f);
  reward_out[e] = r;
  if (alt_action_out) alt_action_out[e] = a_alt;
  const float d = done[e];
  const int next = (d != 0.f) ? (pid + 1) % n_partners : pid;
  partnerid[e] = next;
  if (next_block) {
    float* row = next_block + (size_t)e * block_ld;
    row[0] = (float)next;
    row[1] = r;
    row[2] = d;
  }
}
hipError_t launch_roundrobin_env_step(const int* joint, int* partnerid, const float* base, const float* done, float* reward_out,
                                      int* alt_action_out, float* next_block, int block_ld, int n_partners, float bonus, int n,
                                      hipStream_t s) {
  hipLaunchKernelGGL(roundrobin_env_step_kernel, dim3((n + 255) / 256), dim3(256), 0, s, joint, partnerid, base, done,
                     reward_out, alt_action_out, next_block, block_ld, n_partners, bonus, n);
  return hipGetLastError();
}

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
//          done (n) u8.  History / nmoves are updated in place.
__device__ __forceinline__ void liar_move(LiarTable& t, int e, int* history, int* nmoves, const int* actions, bool ego,
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
}

// LiarEnv.multi_reset of table e: N_DICE dice per player from Philox4x32-10 (one 24-bit draw per die, like the reference's
// randint per die: die d is word d%4 of Philox block d/4 keyed (seed, counter, e)), empty history, first mover ~
// Bernoulli(probegostart) from word 0 of block 100
__device__ __forceinline__ void liar_deal(LiarTable& t, int e, int* hands, int* history, int* nmoves, unsigned char* ego_first,
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
  ego_first[e] = philox_uniform(seed, counter, (uint32_t)e, 100u) < probegostart ? 1 : 0;
}

__global__ void liar_step_kernel(const int* hands, int* history, int* nmoves, const int* actions,
                                 const unsigned char* __restrict__ is_ego, const unsigned char* __restrict__ active,
                                 float* obs_next, float* rew, unsigned char* done, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (active && !active[e]) return;
  LiarTable t;
  liar_load(t, e, hands, history, nmoves);
  liar_move(t, e, history, nmoves, actions, is_ego[e] != 0, obs_next, rew, done);
}
hipError_t launch_liar_step(const int* hands, int* history, int* nmoves, const int* actions, const unsigned char* is_ego,
                            const unsigned char* active, float* obs_next, float* rew, unsigned char* done, int n,
                            hipStream_t s) {
  hipLaunchKernelGGL(liar_step_kernel, dim3((n + 255) / 256), dim3(256), 0, s, hands, history, nmoves, actions, is_ego,
                     active, obs_next, rew, done, n);
  return hipGetLastError();
}

// LiarEnv.getObs(isego) without a move: the observation of the requested player in every active env
__global__ void liar_obs_kernel(const int* __restrict__ hands, const int* __restrict__ history,
                                const int* __restrict__ nmoves, const unsigned char* __restrict__ is_ego,
                                const unsigned char* __restrict__ active, float* __restrict__ obs_out, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (active && !active[e]) return;
  LiarTable t;
  liar_load(t, e, hands, history, nmoves);
  liar_write_obs(t, is_ego[e] != 0, obs_out + (size_t)e * 30);
}
hipError_t launch_liar_obs(const int* hands, const int* history, const int* nmoves, const unsigned char* is_ego,
                           const unsigned char* active, float* obs_out, int n, hipStream_t s) {
  hipLaunchKernelGGL(liar_obs_kernel, dim3((n + 255) / 256), dim3(256), 0, s, hands, history, nmoves, is_ego, active,
                     obs_out, n);
  return hipGetLastError();
}

__global__ void liar_reset_kernel(int* __restrict__ hands, int* __restrict__ history, int* __restrict__ nmoves,
                                  const unsigned char* __restrict__ reset_mask, unsigned char* __restrict__ ego_first,
                                  uint64_t seed, uint64_t counter, const unsigned long long* __restrict__ epoch,
                                  float probegostart, int n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  if (reset_mask && !reset_mask[e]) return;
  LiarTable t;
  liar_deal(t, e, hands, history, nmoves, ego_first, seed, counter + (epoch ? (uint64_t)(*epoch) << 32 : 0ull), probegostart);
}
hipError_t launch_liar_reset(int* hands, int* history, int* nmoves, const unsigned char* reset_mask,
                             unsigned char* ego_first, unsigned long long seed, unsigned long long counter,
                             const unsigned long long* epoch, float probegostart, int n, hipStream_t s) {
  hipLaunchKernelGGL(liar_reset_kernel, dim3((n + 255) / 256), dim3(256), 0, s, hands, history, nmoves, reset_mask,
                     ego_first, seed, counter, epoch, probegostart, n);
  return hipGetLastError();
}

// ---- vectorised Liar's Dice self-play: the step loop's book-keeping, one lane per table ------------------------------------
// (MultiAgentEnv._update_players / _get_actions, multiagentenv.py:149-170, and OnPolicyAgent.update, agents.py:186-203,
// applied to n tables; the partner's rollout rows are ragged: table e writes row alt_pos[e]).  A table's state is touched
// by its own lane only, so everything between two policy forwards is ONE launch: a vectorised step is
//   ego forward | after_ego | partner forward | after_reply | partner forward (openers) | after_opening
__device__ __forceinline__ void liar_sp_credit(const ph_liar_selfplay& s, float* alt_rewards, int alt_T, int e, float r, bool done,
                                               bool credited) {
  const bool m = credited && s.alt_open[e];
  if (m) {
    const int p = s.alt_pos[e];
    if (p >= 1 && p <= alt_T) alt_rewards[(size_t)(p - 1) * s.n + e] += r;
  }
  if (done) s.alt_boundary[e] = 1;
  if (m && done) s.alt_term[e] = 1;
}
// what the partner's next forward records: a row where it is asked to move and its column still has room
__device__ __forceinline__ void liar_sp_prepare(const ph_liar_selfplay& s, int alt_T, int e, bool requested) {
  s.can[e] = (requested && s.alt_pos[e] < alt_T) ? 1 : 0;
  s.es_alt[e] = s.alt_boundary[e] ? 1.f : 0.f;
}
// after a partner forward: advance the recorded column, open / close the reward window, mark the partner as having acted
__device__ __forceinline__ void liar_sp_commit(const ph_liar_selfplay& s, int e) {
  if (s.can[e]) {
    s.alt_pos[e] += 1;
    s.alt_boundary[e] = 0;
    s.alt_term[e] = 0;
    s.alt_open[e] = 1;
  } else {
    s.alt_open[e] = 0;     // the column is full: a later reward belongs to a row that was not recorded
  }
  s.alt_acted[e] = 1;
}

// the ego has moved (its forward wrote ego_actions): play the move, credit the partner where it already acted this game,
// find the tables that go on and prepare the partner's reply there
__global__ void liar_sp_after_ego_kernel(ph_liar_selfplay s, float* alt_rewards, int alt_T) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s.n) return;
  LiarTable t;
  liar_load(t, e, s.hands, s.history, s.nmoves);
  liar_move(t, e, s.history, s.nmoves, s.ego_actions, true, s.obs_next, s.rew1, s.done1);
  const bool d1 = s.done1[e] != 0;
  liar_sp_credit(s, alt_rewards, alt_T, e, s.rew1[2 * e + 1], d1, s.alt_acted[e] != 0);
  s.running[e] = d1 ? 0 : 1;
  liar_sp_prepare(s, alt_T, e, !d1);
}
// the partner has replied where the game went on: play that move, credit both, the ego's reward row / episode flags /
// next observation; then (also the whole of a deal-only call) re-deal the finished tables, find who opens the new games
// and prepare the partner's opening forward
__global__ void liar_sp_after_reply_kernel(ph_liar_selfplay s, float* alt_rewards, int alt_T, float* ego_rew_row,
                                           uint64_t counter, const unsigned long long* __restrict__ epoch, int deal_only) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s.n) return;
  LiarTable t;
  liar_load(t, e, s.hands, s.history, s.nmoves);
  if (!deal_only) {
    const bool run = s.running[e] != 0;
    if (run) {
      liar_sp_commit(s, e);
      liar_move(t, e, s.history, s.nmoves, s.alt_actions, false, s.obs_next, s.rew2, s.done2);
    }
    const bool d2 = run && s.done2[e] != 0;
    liar_sp_credit(s, alt_rewards, alt_T, e, s.rew2[2 * e + 1], d2, run);
    const bool done = s.done1[e] != 0 || d2;
    ego_rew_row[e] += s.rew1[2 * e] + (run ? s.rew2[2 * e] : 0.f);    // both transitions of the step (agents.py:44-47)
    s.ego_episode_start[e] = done ? 1.f : 0.f;
    if (run && !d2) liar_write_obs(t, true, s.obs_ego + (size_t)e * 30);   // = obs_next of the move just played
    s.done[e] = done ? 1 : 0;
    if (done) {
      s.alt_acted[e] = 0;
      atomicAdd(s.episodes, 1ull);
    }
  }
  const bool fresh = s.done[e] != 0;
  if (fresh)
    liar_deal(t, e, s.hands, s.history, s.nmoves, s.ego_first, s.dice_seed, counter + (epoch ? (uint64_t)(*epoch) << 32 : 0ull),
              s.probegostart);
  const bool ego_first = s.ego_first[e] != 0;
  s.alt_opens[e] = (fresh && !ego_first) ? 1 : 0;
  s.ego_opens[e] = (fresh && ego_first) ? 1 : 0;
  if (fresh) s.alt_acted[e] = 0;
  liar_sp_prepare(s, alt_T, e, fresh && !ego_first);
  if (fresh && !ego_first) liar_write_obs(t, false, s.obs_alt + (size_t)e * 30);
}
// the partner has opened the new games it starts: play that move; the ego's observation of every fresh table
__global__ void liar_sp_after_opening_kernel(ph_liar_selfplay s) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s.n) return;
  const bool alt_opens = s.alt_opens[e] != 0, ego_opens = s.ego_opens[e] != 0;
  if (!alt_opens && !ego_opens) return;
  LiarTable t;
  liar_load(t, e, s.hands, s.history, s.nmoves);
  if (alt_opens) {
    liar_sp_commit(s, e);
    liar_move(t, e, s.history, s.nmoves, s.alt_actions, false, s.obs_next, s.rew2, s.done2);
  }
  liar_write_obs(t, true, s.obs_ego + (size_t)e * 30);   // after the partner's opening move, or of the fresh deal
}
#define PH_SP_GRID(s) dim3(((s).n + 255) / 256), dim3(256)
hipError_t launch_liar_sp_after_ego(const ph_liar_selfplay& s, hipStream_t st) {
  hipLaunchKernelGGL(liar_sp_after_ego_kernel, PH_SP_GRID(s), 0, st, s, s.alt_rb->rewards, s.alt_rb->T);
  return hipGetLastError();
}
hipError_t launch_liar_sp_after_reply(const ph_liar_selfplay& s, float* ego_rew_row, unsigned long long counter,
                                      const unsigned long long* epoch, int deal_only, hipStream_t st) {
  hipLaunchKernelGGL(liar_sp_after_reply_kernel, PH_SP_GRID(s), 0, st, s, s.alt_rb->rewards, s.alt_rb->T, ego_rew_row,
                     (uint64_t)counter, epoch, deal_only);
  return hipGetLastError();
}
hipError_t launch_liar_sp_after_opening(const ph_liar_selfplay& s, hipStream_t st) {
  hipLaunchKernelGGL(liar_sp_after_opening_kernel, PH_SP_GRID(s), 0, st, s);
  return hipGetLastError();
}

// ---- peer-to-peer action exchange over xGMI (include/pantheon_hip.h: ph_p2p) --------------------------------------------
// push: one workgroup.  Every peer's receive slot gets this rank's `count` actions with plain (uncached, fine-grained
// memory) stores; after a workgroup barrier one lane fences at system scope and publishes the step stamp to every peer --
// the data stores of all lanes are ordered before the stamp store by barrier + release fence.
__global__ __launch_bounds__(1024) void p2p_push_kernel(ph_p2p x, const int* __restrict__ local, int t) {
  const int par = t & 1;
  const size_t off = (size_t)x.rank * x.count;
  for (int i = threadIdx.x; i < x.count; i += blockDim.x) {
    const int v = local[i];
    for (int p = 0; p < x.world; ++p) __builtin_nontemporal_store(v, x.joint[par][p] + off + i);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x < x.world) {
    const unsigned long long stamp = (*x.epoch) * (unsigned long long)x.T + (unsigned long long)t + 1ull;
    __threadfence_system();
    __hip_atomic_store(x.flags[threadIdx.x] + x.rank, stamp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_p2p_push(const ph_p2p& x, const int* local, int t, hipStream_t s) {
  hipLaunchKernelGGL(p2p_push_kernel, dim3(1), dim3(1024), 0, s, x, local, t);
  return hipGetLastError();
}

// wait: one wave, lane = source rank; polls this rank's own stamp array until every source has published step t (or
// the bound expires: then the error word is bumped and the kernel returns -- a lost peer must not hang the device)
__global__ __launch_bounds__(64) void p2p_wait_kernel(ph_p2p x, int t) {
  const int src = threadIdx.x;
  if (src >= x.world) return;
  const unsigned long long want = (*x.epoch) * (unsigned long long)x.T + (unsigned long long)t + 1ull;
  const unsigned long long* flag = x.flags[x.rank] + src;
  const long long t0 = wall_clock64();
  bool ok = false;
  while (true) {
    if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= want) { ok = true; break; }
    if ((unsigned long long)(wall_clock64() - t0) > x.timeout_cycles) break;
    __builtin_amdgcn_s_sleep(8);
  }
  if (!ok) atomicAdd(x.error, 1ull);
  __threadfence_system();
}
hipError_t launch_p2p_wait(const ph_p2p& x, int t, hipStream_t s) {
  hipLaunchKernelGGL(p2p_wait_kernel, dim3(1), dim3(64), 0, s, x, t);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void p2p_ll_push_kernel(ph_p2p x, const int* __restrict__ local, int t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= x.count) return;
  const unsigned long long w = ((unsigned long long)p2p_stamp32(*x.epoch, x.T, t) << 32) | (unsigned long long)(unsigned)local[i];
  for (int p = 0; p < x.world; ++p)
    __hip_atomic_store(x.ll[p] + (size_t)(t % x.ll_slots) * x.world * x.count + (size_t)x.rank * x.count + i, w, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
hipError_t launch_p2p_ll_push(const ph_p2p& x, const int* local, int t, hipStream_t s) {
  hipLaunchKernelGGL(p2p_ll_push_kernel, dim3((x.count + 255) / 256), dim3(256), 0, s, x, local, t);
  return hipGetLastError();
}

// words of step t (stamp-in-band area) -> this rank's plain int32 receive slot of parity t & 1, for ordinary consumers
__global__ __launch_bounds__(256) void p2p_ll_unpack_kernel(ph_p2p x, int t) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= x.world * x.count) return;
  const unsigned want = p2p_stamp32(*x.epoch, x.T, t);
  const unsigned long long* word = x.ll[x.rank] + (size_t)(t % x.ll_slots) * x.world * x.count + i;
  const long long t0 = wall_clock64();
  unsigned long long v;
  while (true) {
    v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(v >> 32) == want) break;
    if ((unsigned long long)(wall_clock64() - t0) > x.timeout_cycles) {
      atomicAdd(x.error, 1ull);
      break;
    }
    __builtin_amdgcn_s_sleep(2);
  }
  x.joint[t & 1][x.rank][i] = (int)(unsigned)v;
}
hipError_t launch_p2p_ll_unpack(const ph_p2p& x, int t, hipStream_t s) {
  const int n = x.world * x.count;
  hipLaunchKernelGGL(p2p_ll_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, t);
  return hipGetLastError();
}

// HistoryQueue for n envs: one lane per (env, feature) walks its column of frames from the oldest to the newest
__global__ void framestack_push_kernel(float* __restrict__ stack, const float* __restrict__ obs,
                                       const unsigned char* __restrict__ reset_mask,
                                       const float* __restrict__ default_obs, int n, int D, int nf) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (size_t)n * D) return;
  const int e = (int)(g / D), d = (int)(g - (size_t)e * D);
  float* col = stack + (size_t)e * nf * D + d;
  const bool reset = reset_mask && reset_mask[e];
  const float fill = default_obs ? default_obs[d] : 0.f;
  for (int f = nf - 1; f >= 1; --f) col[(size_t)f * D] = reset ? fill : col[(size_t)(f - 1) * D];
  col[0] = obs[g];
}
hipError_t launch_framestack_push(float* stack, const float* obs, const unsigned char* reset_mask,
                                  const float* default_obs, int n, int D, int nf, hipStream_t s) {
  const size_t total = (size_t)n * D;
  hipLaunchKernelGGL(framestack_push_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, stack, obs,
                     reset_mask, default_obs, n, D, nf);
  return hipGetLastError();
}

}  // namespace ph

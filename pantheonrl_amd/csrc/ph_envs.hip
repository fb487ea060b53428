// Vectorised integer rules of the two games whose rules live entirely in the reference tree:
// rock-paper-scissors (pantheonrl/envs/rpsgym/rps.py:41-45) and Liar's Dice (pantheonrl/envs/liargym/liar.py:53-83).
// One lane per environment; everything is integer arithmetic and must be bit-exact with the Python games.
#include "ph_launch.h"
#include "ph_liar.h"

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

__global__ void liar_sp_after_ego_kernel(ph_liar_selfplay s, float* alt_rewards, int alt_T) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s.n) return;
  liar_sp_after_ego_lane(s, e, alt_rewards, alt_T);
}
__global__ void liar_sp_after_reply_kernel(ph_liar_selfplay s, float* alt_rewards, int alt_T, float* ego_rew_row,
                                           uint64_t counter, const unsigned long long* __restrict__ epoch, int deal_only) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s.n) return;
  liar_sp_after_reply_lane(s, e, alt_rewards, alt_T, ego_rew_row, counter, epoch, deal_only);
}
__global__ void liar_sp_after_opening_kernel(ph_liar_selfplay s) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= s.n) return;
  liar_sp_after_opening_lane(s, e);
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
  unsigned long long seen = 0ull;
  int polls = 0;
  while (true) {
    seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (seen >= want) { ok = true; break; }
    if ((unsigned long long)(wall_clock64() - t0) > x.timeout_cycles) break;
    poll_backoff(polls);
  }
  if (!ok) p2p_note_timeout(x.error, 3, t, (unsigned long long)src, want, seen);
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
__global__ __launch_bounds__(256) void p2p_ll_unpack_kernel(ph_p2p x, int t, int slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= x.world * x.count) return;
  const unsigned want = p2p_stamp32(*x.epoch, x.T, t);
  if (slot == -2) slot = p2p_persistent_slot(*x.epoch, x.T, t);   // the persistent rollout's slot of step t
  else if (slot < 0) slot = t % x.ll_slots;
  const unsigned long long* word = x.ll[x.rank] + (size_t)slot * x.world * x.count + i;
  const long long t0 = wall_clock64();
  unsigned long long v;
  int polls = 0;
  while (true) {
    v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((unsigned)(v >> 32) == want) break;
    if ((unsigned long long)(wall_clock64() - t0) > x.timeout_cycles) {
      p2p_note_timeout(x.error, 2, t, (unsigned long long)i, want, v);
      break;
    }
    poll_backoff(polls);
  }
  x.joint[t & 1][x.rank][i] = (int)(unsigned)v;
}
hipError_t launch_p2p_ll_unpack(const ph_p2p& x, int t, hipStream_t s, int slot) {
  const int n = x.world * x.count;
  hipLaunchKernelGGL(p2p_ll_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, t, slot);
  return hipGetLastError();
}

// ---- BASELINE config 4 engine-side (include/pantheon_hip.h: ph_rr_link) -------------------------------------------------------
// One agent per GPU, ego on rank 0 against K round-robin partners: what crosses ranks per environment step is the routing block
// (rank 0 -> partners) and every partner's actions (-> rank 0).  Both travel as direct stores into IPC-mapped fine-grained
// receive areas followed by a monotonic stamp (iteration * T + t + 1); consumers poll the stamp (bounded) inside the kernel
// that needs the data.  A step is strictly ping-pong between rank 0 and each partner, so two slots (step parity) suffice.
__device__ __forceinline__ bool rr_wait(const unsigned long long* flag, unsigned long long want, unsigned long long timeout,
                                        unsigned long long* error) {
  const long long t0 = wall_clock64();
  int polls = 0;
  while (true) {
    const unsigned long long seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (seen >= want) return true;
    if ((unsigned long long)(wall_clock64() - t0) > timeout) {
      p2p_note_timeout(error, 4, 0, 0ull, want, seen);
      return false;
    }
    poll_backoff(polls);
  }
}

// rank 0: routing block of step t -> every partner's slot, then the stamp.  The interleaved block [id | reward | done | obs] is
// split on the way: a partner's slot holds the header rows (n, 4) first and the observations (n, D) contiguously behind them,
// so that the partner's forward kernel reads its observations straight from the slot.  Grid (RR_SEND_SPLIT, K): the
// workgroups of a partner count themselves in after a system-scope fence; the last one publishes the stamp.
__global__ __launch_bounds__(256) void rr_send_block_kernel(RRSend a) {
  const int k = blockIdx.y, D = a.block_ld - 3;
  float* hdr = a.dst[k];
  float* obs = hdr + (size_t)a.n * 4;
  const int rows_per = (a.n + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rows_per, r1 = (r0 + rows_per < a.n) ? r0 + rows_per : a.n;
  for (int i = r0 * a.block_ld + threadIdx.x; i < r1 * a.block_ld; i += blockDim.x) {
    const int r = i / a.block_ld, c = i - r * a.block_ld;
    const float v = a.src[i];
    if (c < 3) hdr[r * 4 + c] = v;
    else obs[(size_t)r * D + (c - 3)] = v;
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int old = atomicAdd(a.arrive + k, 1u);
    if ((old + 1u) % gridDim.x == 0u) {
      __threadfence_system();
      __hip_atomic_store(a.stamp[k], a.want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}
hipError_t launch_rr_send_block(const RRSend& a, int n_partners, hipStream_t s) {
  hipLaunchKernelGGL(rr_send_block_kernel, dim3(RR_SEND_SPLIT, n_partners), dim3(256), 0, s, a);
  return hipGetLastError();
}

// rank 0: wait for every partner's actions of this step, then the transition (roundrobin_env_step_kernel's arithmetic)
__global__ __launch_bounds__(256) void rr_env_step_kernel(RREnvStep a) {
  if (threadIdx.x < a.n_partners) (void)rr_wait(a.stamps + 1 + threadIdx.x, a.want, a.timeout, a.error);   // a timeout bumps *error
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.n) return;
  int pid = a.partnerid[e];
  pid = pid < 0 ? 0 : (pid >= a.n_partners ? a.n_partners - 1 : pid);
  if (a.partner_trace) a.partner_trace[e] = pid;
  const int a_ego = a.joint[e];
  const int a_alt = __hip_atomic_load(a.joint + (size_t)(1 + pid) * a.n + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const float r = a.base[e] + ((a_ego == a_alt) ? a.bonus : 0.f);
  a.reward_out[e] = r;
  if (a.alt_action_out) a.alt_action_out[e] = a_alt;
  const float d = a.done[e];
  const int next = (d != 0.f) ? (pid + 1) % a.n_partners : pid;
  a.partnerid[e] = next;
  float* row = a.next_block + (size_t)e * a.block_ld;
  row[0] = (float)next;
  row[1] = r;
  row[2] = d;
}
hipError_t launch_rr_env_step(const RREnvStep& a, hipStream_t s) {
  hipLaunchKernelGGL(rr_env_step_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// partner k, before its forward: wait for the routing block; Agent.update of the previous step where this partner acted
// (agents.py:186-203: reward into the open row, episode boundary, terminal flag); which environments it acts in now, whether
// their column has room, the episode_start of the row; the partner-seat observation compacted for the forward kernel
__global__ __launch_bounds__(256) void rr_partner_pre_kernel(RRPartnerStep a) {
  if (threadIdx.x == 0) (void)rr_wait(a.block_stamp, a.want, a.timeout, a.error);   // a timeout bumps *error
  __syncthreads();
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < a.n) {
    const float4 h = *reinterpret_cast<const float4*>(a.block + (size_t)e * 4);   // header rows (n, 4) lead the slot
    const float pid = h.x, r = h.y, d = h.z;
    const bool m = a.prev_mask[e] != 0, open = a.open[e] != 0;
    const int p = a.pos[e];
    if (m && open && p >= 1 && p <= a.T) a.rewards[(size_t)(p - 1) * a.n + e] += r;
    const bool dd = m && d != 0.f;           // the done flag counts only where this partner acted (blk[:, 2] * mask)
    bool boundary = (a.boundary[e] != 0) || dd;
    bool term = (a.term[e] != 0) || (m && open && dd);
    const bool mask = pid == (float)a.k;
    const bool room = p < a.T;
    const bool can = mask && room, blocked = mask && !room;
    a.es[e] = boundary ? 1.f : 0.f;
    a.can[e] = can ? 1 : 0;
    // RaggedVecOnPolicyAgent.get_action's book-keeping after the forward (the forward reads pos / can / es only)
    a.boundary[e] = can ? 0 : (boundary ? 1 : 0);
    a.term[e] = can ? 0 : (term ? 1 : 0);
    a.open[e] = can ? 1 : (blocked ? 0 : (open ? 1 : 0));
    a.prev_mask[e] = mask ? 1 : 0;
  }
}
hipError_t launch_rr_partner_pre(const RRPartnerStep& a, hipStream_t s) {
  hipLaunchKernelGGL(rr_partner_pre_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// partner k, after its forward: advance the write rows of the columns that recorded, send the actions to rank 0, stamp
__global__ __launch_bounds__(1024) void rr_partner_post_kernel(RRPartnerStep a) {
  for (int e = threadIdx.x; e < a.n; e += blockDim.x) {
    if (a.can[e] && a.pos[e] < a.T) a.pos[e] += 1;
    __builtin_nontemporal_store(a.actions[e], a.act_dst + e);
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(a.act_stamp, a.want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
hipError_t launch_rr_partner_post(const RRPartnerStep& a, hipStream_t s) {
  hipLaunchKernelGGL(rr_partner_post_kernel, dim3(1), dim3(1024), 0, s, a);
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

// Launch-argument records and launcher prototypes shared between the kernel translation units and ph_abi.hip.
#pragma once
#include "ph_device.h"

namespace ph {

// stamp of step t of the iteration whose epoch word reads `epoch` (peer-to-peer exchange; 32 bits travel in-band)
__host__ __device__ inline unsigned p2p_stamp32(unsigned long long epoch, int T, int t) {
  return (unsigned)(epoch * (unsigned long long)T + (unsigned long long)t + 1ull);
}

// A bounded in-kernel wait that expired: count it in error[0]; the FIRST one of the area's lifetime also leaves a record of what
// was waited for -- error[1] = kind | step << 8 | index << 32 (kind: 1 stamp-in-band word polled by a forward's value tail,
// 2 the same word polled by the unpack kernel, 3 a stamp flag of the push / wait pair, 4 a round-robin block or action stamp;
// index: word offset inside the step's slot = seat * n + row, or the source rank), error[2] = the stamp wanted, error[3] = the
// word seen last.  The areas reserve 64 bytes (eight words) for this.
__device__ __forceinline__ void p2p_note_timeout(unsigned long long* error, int kind, int t, unsigned long long index,
                                                 unsigned long long want, unsigned long long seen) {
  if (atomicAdd(error, 1ull) == 0ull) {
    error[1] = (unsigned long long)(unsigned)kind | ((unsigned long long)(unsigned)(t & 0xffffff) << 8) | (index << 32);
    error[2] = want;
    error[3] = seen;
  }
}

// Back-off of a bounded in-kernel wait for a word another process (or block) writes.  The first polls are ~60 ns apart (a healthy
// peer answers within a few); then the wave backs off, up to ~4 us between polls: a waiter that shares its device with the rank it
// waits for (test boxes: N ranks on one GPU) must not take the issue slots and the fabric bandwidth the peer's stores need.
__device__ __forceinline__ void poll_backoff(int& polls) {
  if (polls < 16) __builtin_amdgcn_s_sleep(2);
  else if (polls < 64) __builtin_amdgcn_s_sleep(16);
  else __builtin_amdgcn_s_sleep(127);
  ++polls;
}

// NOTE for new fields: liar_rollout_kernel (ph_policy.hip) rebuilds its three records from {nd, n} plus the fields
// launch_liar_rollout lists as patched, and the launcher refuses records whose other bytes are not zero -- a field added here must
// either stay zero on that path or join both lists.
struct FwdArgs {
  NetDims nd;
  const float* params;
  const float* obs;  // (n, D)
  int n;
  const unsigned char* mask;   // (n, L) or null
  const float* uniforms;       // (n, A) or null
  const float* given_actions;  // (n, A) or null
  uint64_t seed, counter;
  const unsigned long long* epoch;  // device RNG epoch word or null
  int deterministic;
  int* act_i32;
  float* act_f32;
  float* values;
  float* logp;
  float* entropy;
  float* logits;
  // fused RolloutBuffer.add (pointers already offset to row `pos`), all null when not fused
  float* rb_obs;
  float* rb_act;
  float* rb_rew;
  float* rb_es;
  float* rb_val;
  float* rb_logp;
  const float* es_in;
  long long* prof;              // debug: per-workgroup phase timestamps (clock64), or null
  float* prev_rew;              // rewards row pos-1, or null
  const float* pending_reward;  // (E) added to prev_rew (Agent.update folded into the next step's launch)
  const int* pos_env;           // ragged mode: per-env write row (device), rb_* are array bases; null = rectangular
  const unsigned char* rec_mask;  // ragged mode: which envs record this action
  int rb_T;
  const int* joint;             // (n_seats, n) all-gathered actions of the previous step, or null
  const unsigned long long* joint_ll;  // the same as stamp-in-band words (fused peer-to-peer step), or null
  const unsigned long long* ll_epoch;  // stamp of those words = p2p_stamp32(*ll_epoch, ll_T, ll_t)
  int ll_T, ll_t;
  unsigned long long ll_timeout;       // bound of one poll in wall_clock64 ticks
  unsigned long long* ll_error;        // timed-out polls
  int n_seats, seat;
  const int* partner_seat;      // device int
  float bonus;
  int reward_rule;              // how the joint action enters the reward (ph_rowtail.h joint_reward): 0 match bonus, 1 rock-paper-scissors
  const unsigned char* env_mask;  // (n, L) or null: act_i32 (what the environment / the exchange consumes) receives the
                                  // ENV-side fix-up of an illegal sample -- the first legal index, pettingzoo.py:81-82 -- while
                                  // the rollout buffer keeps the sampled action, as the reference's agent does.  Independent of
                                  // `mask` (the policy-side logit offset of ModularPolicy): OnPolicyAgent hands a plain PPO
                                  // policy obs.obs only (agents.py:162), so its samples are unmasked and the env repairs them
  unsigned int* host_done;      // ph_policy_act_host: [2] words in coherent host memory, or null; the policy / value workgroup of a
  unsigned int host_seq;        // ONE-tile launch stores host_seq into word blockIdx.y after its last output (system scope)
};

// T steps of the 16-row forward against a scripted environment inside one launch (policy_fwd16_rollout_kernel): the per-step
// argument record is derived from the step-0 record `a` -- observation / reward / done rows t of the three sequences, rollout
// buffer rows pos0 + t, Philox counter counter0 + t
struct ScriptedSteps {
  int n_steps;
  const float* obs_seq;    // (T, n, D)
  const float* rew_seq;    // (T, n)
  const float* done_seq;   // (T, n)
  const unsigned char* mask_seq;   // (T, n, L) action masks of every step, or null
  int mask_policy, mask_env;       // which sides see them: FwdArgs.mask and / or FwdArgs.env_mask
};

constexpr int MAX_LOCAL_AGENTS = 4;
// peer-to-peer exchange fused into the step launch (policy_fwd16_multi_kernel): the policy workgroups store every row's sampled
// action, stamp in-band, straight into every rank's receive area; consumers poll the words they need.  x.world == 0: off.
struct P2PStep {
  const ph_p2p* x;   // DEVICE copy of the descriptor (a by-value copy inside the kernel arguments, indexed dynamically,
                     // makes the compiler spill the whole argument block to scratch), or null
  int t;             // step index of this launch
  int a_local;       // local agents per rank (seat = rank * a_local + blockIdx.z)
  int persistent;    // 1: the launch walks all T steps (policy_fwd16_exchange_rollout_kernel); word slot of step t is
                     // p2p_persistent_slot(epoch, T, t)
};
// Word slot of step t in the persistent exchange rollout.  A launch's policy workgroups never wait, so they may finish all T
// pushes while value workgroups of the same launch are still consuming early steps; the next iteration of a faster PEER must
// then not overwrite words this rank has not read.  Iterations alternate between two halves of 2T slots: a peer can only
// reach iteration k + 2 (the half iteration k used) after this rank pushed the last step of iteration k + 1, i.e. after this
// rank's launch of iteration k has completed.
__host__ __device__ inline int p2p_persistent_slot(unsigned long long epoch, int T, int t) { return (int)(epoch & 1ull) * T + t; }
struct FwdMulti {
  FwdArgs a[MAX_LOCAL_AGENTS];
  P2PStep px;
};
struct ScriptedMulti {
  ScriptedSteps sc[MAX_LOCAL_AGENTS];
};
bool fwd16_eligible(const NetDims& nd, int n);
hipError_t launch_policy_fwd16_rollout(const FwdArgs& a, const ScriptedSteps& sc, int gemm_mode, hipStream_t s);
// all T steps of every local agent of the symmetric agent-per-GPU layout in ONE launch, the per-step action hand-off done
// in-kernel over the stamp-in-band words (m.px.persistent = 1)
hipError_t launch_policy_fwd16_exchange_rollout(const FwdMulti& m, const ScriptedMulti& sm, int n_agents, hipStream_t s);
// workgroups of that kernel one CU keeps resident, as the runtime's occupancy query answers for this device (with a margin of one
// where the answer is not LDS-bound: MI355X_MICROARCH.md, "Residency and cooperative launch")
hipError_t exchange_rollout_blocks_per_cu(int* blocks_out);

struct GradArgs {
  NetDims nd;
  const float* params;
  // rollout buffer (time-major)
  const float* rb_obs;
  const float* rb_act;
  const float* rb_val;
  const float* rb_logp;
  const float* rb_adv;
  const float* rb_ret;
  int T, E;
  // minibatch rows: explicit env-major indices, or a Feistel permutation of [0, perm_n) starting at mb_start
  const int* idx;
  const int* idx_phys;     // the same rows already translated to physical buffer rows (adv_stats_kernel), or null
  uint32_t perm_n, perm_hb;
  uint64_t perm_seed;      // key = epoch_key(perm_seed + *epoch, perm_epoch)
  int perm_epoch;
  const unsigned long long* epoch;
  int mb_start;
  int nb;                  // rows in this minibatch
  const float* advstats;   // {mean, std} of this minibatch's advantages
  float clip, clip_vf, ent_coef, vf_coef;
  int norm_adv;
  float* slabs;            // [gridDim.x][P]
  float* statpart;         // [2*gridDim.x][NSTATP]
  const int* stop_flag;    // device flag set by the KL early stop
  long long* prof;         // debug: per-workgroup phase timestamps (clock64), or null
  int ntiles;
  const unsigned short* wimage;   // split kernel: the pre-split weight fragment image of `params` (ph_split.h)
  // split kernel: the gradient pack of this train() call (ph_split.h) -- the observation rows as bf16 planes, split ONCE per
  // call (they do not change across its epochs), and the per-row scalars of each net packed in minibatch order
  const uint4* ximg;       // [N + 1][3 planes][64] bf16 = 24 granules of 16 bytes per buffer row; row N is all zero (dead rows of a partial tile)
  int ximg_zero_row;       // N
  const uint4* rec_pi;     // this minibatch: {physical row, advantage, old log-prob, action} per position
  const uint4* rec_vf;     // this minibatch: {physical row, return, old value, -}
};

// What the joint action of a SimultaneousEnv step adds to an agent's reward (multiagentenv.py:395-409 hands every agent ITS reward):
// rule 0 = the synthetic driver's shared coordination term, bonus * [own action == partner's]; rule 1 = rock-paper-scissors,
// bonus * payoff with (own - partner's + 3) % 3 == 1 a win, == 2 a loss (rps.py:41-45: the ego's gain, the partner gets its negative)
__device__ __host__ inline float joint_reward(int mine, int theirs, float bonus, int rule) {
  if (rule == 1) {
    const int d = (mine - theirs + 3) % 3;
    return d == 1 ? bonus : (d == 2 ? -bonus : 0.f);
  }
  return mine == theirs ? bonus : 0.f;
}

__device__ __host__ inline uint64_t epoch_key(uint64_t seed, int epoch) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(epoch + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// env-major flat index n = e*T + t (SB3 swap_and_flatten) -> row of the time-major buffer
__device__ __forceinline__ int env_major_to_phys(int n, int T, int E) {
  const int e = n / T;
  return (n - e * T) * E + e;
}
// physical buffer row of minibatch element gi: explicit index array, or the keyed Feistel permutation of this epoch
__device__ __forceinline__ int minibatch_row(const GradArgs& a, int gi) {
  int n;
  if (a.idx) {
    n = a.idx[gi];
  } else {
    const uint64_t key = epoch_key(a.perm_seed + (a.epoch ? *a.epoch : 0ull), a.perm_epoch);
    n = (int)feistel_perm((uint32_t)(a.mb_start + gi), a.perm_n, a.perm_hb, key);
  }
  return env_major_to_phys(n, a.T, a.E);
}

#ifndef PH_ADV_SPLIT
#define PH_ADV_SPLIT 32
#endif
#ifndef PH_ADV_THREADS
#define PH_ADV_THREADS 256
#endif
constexpr int ADV_SPLIT = PH_ADV_SPLIT;      // workgroups per minibatch in the advantage-statistics pass (32 x 256 lanes: 1 280 workgroups
                                             // at the bench size spread evenly over the CUs; 8 x 1024 left a quarter of them with two: 21.3 -> 17.4 us)
constexpr int ADV_THREADS = PH_ADV_THREADS;  // lanes of one of them
struct AdvStatArgs {
  const float* rb_adv;
  int T, E;
  const int* perms;  // (n_epochs, N) or null
  uint32_t perm_n, perm_hb;
  uint64_t perm_seed;
  const unsigned long long* epoch;
  int N, batch, n_mb;  // minibatch k of epoch ep covers [k*batch, min(N,(k+1)*batch))
  float* out;          // [n_epochs*n_mb][2]
  double* partial;     // [n_epochs*n_mb][ADV_SPLIT][2] per-segment (sum, sum of squares)
  int* idx_out;        // (n_epochs, N) or null: the env-major index of every element, materialised for the grad launches
  int* phys_out;       // (n_epochs, N) or null: the physical buffer row of every element (t * E + e)
  // the split gradient kernel's row records in minibatch order (GradArgs.rec_pi / rec_vf), or null
  uint4* rec_pi_out = nullptr;   // (n_epochs, N) {phys, advantage, old log-prob, action}
  uint4* rec_vf_out = nullptr;   // (n_epochs, N) {phys, return, old value, 0}
  const float *rb_logp = nullptr, *rb_act = nullptr, *rb_ret = nullptr, *rb_val = nullptr;   // read only when the records are written (action length 1)
  const uint4* rowrec = nullptr; // [T*E][2] the same scalars packed by physical row (obs_planes_kernel), or null: gather from the arrays
  int* clear_flag = nullptr;     // set to 0 by the launch (the train() call's KL stop flag), or null
};

struct ReduceArgs {
  const float* slabs;
  int nslab, P;
  int slab_len;          // floats per slab (P for the canonical layout)
  const int* map;        // slab position -> parameter index (-1 = padding), or null = identity (canonical slabs)
  float* grad;
  float* blocksq;        // [gridDim.x]
  const float* statpart; // [nstatpart][NSTATP] per-workgroup partial sums of the grad kernel
  int nstatpart;
  float* stats_out;      // [PH_NSTAT] for this minibatch, or null
  int nb;
  float ent_coef, vf_coef, target_kl;
  int* stop_flag;
  int* step;             // optimizer step counter (incremented here when the step will be applied), may be null
  float* scalars;        // [4]: kl, applied, stop-requested
  // an additional loss term whose gradient arrives as n_extra slabs in canonical parameter order (ADAP's context loss)
  // Slab layout: the parameters [0, extra_cut) followed by the parameters [extra_lo, extra_hi) -- the policy network and the
  // action head; the value side receives nothing from the term.
  const float* extra = nullptr;       // [n_extra][extra_len], already scaled by the term's coefficient
  int n_extra = 0, extra_len = 0, extra_cut = 0, extra_lo = 0, extra_hi = 0;
  const float* extra_loss = nullptr;  // [n_extra] partial sums of the raw term
  float extra_norm = 0.f;             // raw term = extra_norm * sum(extra_loss)
  float extra_coef = 0.f;             // loss statistic += extra_coef * raw term
  float* extra_loss_out = nullptr;    // [1] raw term of this minibatch, or null
  int wide = 0;                       // 1: the learner has the device to itself -- 16-byte loads (more registers); same summation tree either way
};

struct AdamArgs {
  float* params;
  float* m;
  float* v;
  const float* grad;
  const float* blocksq;
  int nblk, P;
  const int* step;        // already incremented for this step
  const float* scalars;   // [1] = applied, [2] = stop requested
  int* stop_flag;
  float lr, beta1, beta2, eps, max_norm;
  float* stats_out;       // [PH_NSTAT] or null: writes grad_norm at [6]
  unsigned short* wimage; // the split gradient kernel's weight fragment image (ph_split.h) kept in step with params, or null
  const int* wimage_map;  // [P][2]
};

// reduce + clip + Adam of one minibatch as ONE launch (ph_step.h: step_body; ppo_step_kernel) for a learner that has its device
// to itself
struct StepArgs {
  ReduceArgs r;
  AdamArgs ad;
  unsigned long long* words = nullptr;   // [slab blocks + 1] stamped {tag, value} words
  unsigned int* gen = nullptr;           // launch generation (device word, advanced by block 0 at the end of every step)
  unsigned long long timeout = 0;        // wall_clock64 ticks: bound of every wait
  unsigned int* sweep_error = nullptr;   // count of waits that ran into the bound
};

size_t fwd_lds_bytes(int R, int Lp);
size_t grad_lds_bytes(int R, int Lp, int onehot_D, int nchunk);
hipError_t launch_policy_fwd(const FwdArgs& a, int gemm_mode, hipStream_t s);
hipError_t launch_policy_fwd_multi(const FwdMulti& m, int n_agents, hipStream_t s);
hipError_t launch_fix_illegal(int* actions, const unsigned char* mask, int n, int L, hipStream_t s);
hipError_t launch_gae(const float* rew, const float* val, const float* es, const float* lv, const float* dn, float* adv,
                      float* ret, int T, int E, double gamma, double lam, int mode, hipStream_t s);
hipError_t launch_buffer_add(float* d_obs, float* d_act, float* d_rew, float* d_es, float* d_val, float* d_lp,
                             const float* obs, const float* act, const float* es, const float* val, const float* lp,
                             int E, int D, int A, hipStream_t s);
hipError_t launch_buffer_compact(const ph_rollout& src, const ph_rollout& dst, const int* cols, int n, int D, int A,
                                 hipStream_t s);
hipError_t launch_reward_add_ragged(float* rewards, const int* pos_env, const float* reward, const unsigned char* mask,
                                    int T, int E, hipStream_t s);
hipError_t launch_ragged_advance(int* pos_env, const unsigned char* mask, int T, int E, hipStream_t s);
hipError_t launch_liar_obs(const int* hands, const int* history, const int* nmoves, const unsigned char* is_ego,
                           const unsigned char* active, float* obs_out, int n, hipStream_t s);
hipError_t launch_liar_reset(int* hands, int* history, int* nmoves, const unsigned char* reset_mask,
                             unsigned char* ego_first, unsigned long long seed, unsigned long long counter,
                             const unsigned long long* epoch, float probegostart, int n, hipStream_t s);
hipError_t launch_reward_add(float* rew_row, const float* reward, const unsigned char* env_mask, int E, hipStream_t s);
hipError_t launch_reward_add_const(float* rew_row, float reward, int E, hipStream_t s);
hipError_t launch_reward_add_joint(float* rew_row, const float* base, const int* joint, int E, int n_seats, int seat,
                                   const int* partner_seat, float bonus, int rule, hipStream_t s);
hipError_t launch_framestack_push(float* stack, const float* obs, const unsigned char* reset_mask,
                                  const float* default_obs, int n, int D, int nf, hipStream_t s);
hipError_t launch_roundrobin_env_step(const int* joint, int* partnerid, const float* base, const float* done, float* reward_out,
                                      int* alt_action_out, float* next_block, int block_ld, int n_partners, float bonus, int n,
                                      hipStream_t s);
hipError_t launch_rps_step(const int* ego_act, const int* alt_act, float* ego_rew, float* alt_rew, int n, hipStream_t s);
hipError_t launch_liar_step(const int* hands, int* history, int* nmoves, const int* actions, const unsigned char* is_ego,
                            const unsigned char* active, float* obs_next, float* rew, unsigned char* done, int n,
                            hipStream_t s);
hipError_t launch_ppo_grad(const GradArgs& a, int nwg, int gemm_mode, hipStream_t s);
// device-side book-keeping of the vectorised Liar's Dice self-play step (ph_envs.hip)
hipError_t launch_liar_sp_after_ego(const ph_liar_selfplay& s, hipStream_t st);
hipError_t launch_liar_sp_after_reply(const ph_liar_selfplay& s, float* ego_rew_row, unsigned long long counter,
                                      const unsigned long long* epoch, int deal_only, hipStream_t st);
hipError_t launch_liar_sp_after_opening(const ph_liar_selfplay& s, hipStream_t st);
// persistent Liar's Dice rollout (ph_policy.hip: liar_rollout_kernel)
bool liar_rollout_eligible(const NetDims& nd, int n);

// ---- ADAP's context loss (ph_adap.hip) ------------------------------------------------------------------------------
constexpr int ADAP_ROWS = 16;   // (state, context) rows of one workgroup
struct AdapArgs {
  NetDims nd;
  const float* params;
  const float* rb_obs;
  int T, E;
  const int* idx;          // env-major index of every minibatch element (nb)
  int nb;
  int ctx_size, n_ctx, n_states;   // n_states = min(num_state_samples, nb)
  int sampler;             // PH_CTX_*
  float coef;              // context_loss_coeff
  const int* state_idx;    // (n_states) positions in the minibatch, or null = head of a keyed permutation of [0, nb)
  const float* contexts;   // (n_ctx, ctx_size), or null = drawn in the kernel
  uint64_t seed;
  const unsigned long long* epoch;
  uint32_t mbi, nb_hb;     // minibatch number within the train() call; Feistel half width for nb
  float* extra;            // [gridDim.x][adap_slab_floats(lay)]
  float* loss_part;        // [gridDim.x]
  int* used_state_idx;     // (n_states) out or null
  float* used_contexts;    // (n_ctx, ctx_size) out or null
  const int* stop_flag;
  long long* prof;         // debug: per-workgroup phase timestamps (clock64), or null
};
// ---- AdapPolicyMult (ph_adapmult.hip): dense [rows][width] intermediates of ONE net at a time, sized for the largest row count ----
struct AmWork {
  float *x, *xa, *y, *h;        // [rows][64], [rows][64 C], [rows][64], [rows][64]
  float *z, *v;                 // logits [rows][L], value [rows]
  float *dz, *dv;               // dL/dlogits, dL/dvalue
  float *dzh, *dy, *dza;        // backward: [rows][64], [rows][64], [rows][64 C]
  float *xg;                    // the minibatch's rows [rows][D]
  float *act, *oldlp, *adv, *ret, *oldv;   // its per-row scalars
};
struct AmGather {
  const float *rb_obs, *rb_act, *rb_logp, *rb_adv, *rb_ret, *rb_val;
  const int* idx;
  int nb, T, E, D, norm_adv;
  const float* advstats;
  float *xg, *act, *oldlp, *adv, *ret, *oldv;
};
struct AmCtx {
  const float* rb_obs;
  const int* idx;
  int nb, T, E, D;
  int ctx_size, n_ctx, n_states, sampler;
  const int* state_idx;
  const float* contexts;
  uint64_t seed;
  const unsigned long long* epoch;
  uint32_t mbi, nb_hb;
  float* rows;                  // [n_states * n_ctx][D] out
  int* used_state_idx;
  float* used_contexts;
};
hipError_t am_forward_net(const ph_adapmult_layout& L, const float* params, int net, const float* X, int ldx, int rows,
                          const AmWork& w, hipStream_t s);
hipError_t am_backward_net(const ph_adapmult_layout& L, const float* params, int net, const float* X, int ldx, int rows,
                           const AmWork& w, float* slabs, int nslab, int slab_len, int head_w_off, int head_b_off, hipStream_t s);
hipError_t launch_am_act(const FwdArgs& a, const float* z, const float* v, hipStream_t s);
hipError_t launch_am_gather(const AmGather& g, hipStream_t s);
hipError_t launch_am_loss(const AmWork& w, int L, int nb, const ph_ppo_hyper& hp, float* statpart, int nslab, int net,
                          hipStream_t s);
hipError_t launch_am_ctx_rows(const AmCtx& a, hipStream_t s);
hipError_t launch_am_ctx_loss(const float* z, int L, int n_states, int Cs, float wgt, float* dz, float* loss_part, hipStream_t s);
int adap_workgroups(int n_ctx, int n_states);
int adap_slab_floats(const ph_layout& lay);
size_t adap_lds_bytes(const NetDims& nd, int n_ctx, int ctx_size);
hipError_t launch_adap_context(const AdapArgs& a, int nwg, hipStream_t s);
hipError_t launch_liar_rollout(const ph_liar_selfplay& s, const FwdArgs& ego, const FwdArgs& reply, const FwdArgs& opening,
                               int n_steps, unsigned long long counter0, const unsigned long long* epoch, float* ego_rew_row0,
                               hipStream_t st);
// peer-to-peer action exchange (ph_envs.hip)
hipError_t launch_p2p_push(const ph_p2p& x, const int* local, int t, hipStream_t s);
hipError_t launch_p2p_wait(const ph_p2p& x, int t, hipStream_t s);
hipError_t launch_p2p_ll_unpack(const ph_p2p& x, int t, hipStream_t s, int slot = -1 /* default: t mod ll_slots */);
hipError_t launch_p2p_ll_push(const ph_p2p& x, const int* local, int t, hipStream_t s);
// engine-side round-robin layout (ph_envs.hip)
constexpr int RR_SEND_SPLIT = 16;        // workgroups per partner in the block send
struct RRSend {
  const float* src;                      // this step's routing block, interleaved (n, block_ld)
  int n, block_ld;
  float* dst[PH_MAX_RANKS];              // [k] partner k's slot of this step's parity: header rows (n, 4), then obs (n, D)
  unsigned long long* stamp[PH_MAX_RANKS];   // [k] partner k's block stamp
  unsigned int* arrive;                  // [K] local arrival counters (monotonic)
  unsigned long long want;
};
struct RREnvStep {
  const unsigned long long* stamps;      // rank 0's stamp array: [1 + k] = partner k's action stamp
  unsigned long long want, timeout;
  unsigned long long* error;
  const int* joint;                      // (1 + K, n): row 0 the ego's actions, row 1 + k partner k's (this step's slot)
  int* partnerid;
  int* partner_trace;                    // (n) or null
  const float* base;
  const float* done;
  float* reward_out;
  int* alt_action_out;
  float* next_block;
  int block_ld, n_partners, n;
  float bonus;
};
struct RRPartnerStep {
  const unsigned long long* block_stamp;
  unsigned long long* act_stamp;         // on rank 0
  unsigned long long want, timeout;
  unsigned long long* error;
  const float* block;                    // this step's slot as received: header rows (n, 4), then the observations (n, D)
  int block_ld, n, T, k;
  float* rewards;                        // the partner's rollout-buffer rewards (T, n)
  int* pos;
  unsigned char *boundary, *term, *open, *prev_mask, *can;
  float* es;
  const int* actions;                    // (n) this step's sampled actions
  int* act_dst;                          // rank 0's slot row 1 + k
};
hipError_t launch_rr_send_block(const RRSend& a, int n_partners, hipStream_t s);
hipError_t launch_rr_env_step(const RREnvStep& a, hipStream_t s);
hipError_t launch_rr_partner_pre(const RRPartnerStep& a, hipStream_t s);
hipError_t launch_rr_partner_post(const RRPartnerStep& a, hipStream_t s);
// single-chunk / small-Discrete-head kernel (ph_ppo_fast.hip); eligible() says whether the spec fits it.  Its slabs are in
// the MFMA accumulators' register order -- [net][RS_NET] floats per workgroup (16-byte stores, 1 KB contiguous per wave
// instruction) -- and the reduce kernel maps slab positions to parameter indices through the table grad_slab_map fills.
constexpr int RS_NET = 8960;
constexpr int RS_W2 = 0, RS_W1 = 4096, RS_B1 = 8192, RS_B2 = 8256, RS_HW = 8320, RS_HB = 8832;
bool grad_fast_eligible(const NetDims& nd);
bool grad_uses_reg_slabs(const NetDims& nd);
bool grad_fast_fold(const NetDims& nd);   // the bias of layer 1 rides in row 63 of the staged W1 (Box observations, F < 64)
void grad_slab_map(const ph_layout& lay, int* map /* host, 2 * RS_NET */, bool fold);
// tiles (GradArgs.ntiles) and workgroups per net that launch_ppo_grad will use for a minibatch of nb rows
void grad_plan(const NetDims& nd, int nb, int num_cu, int* ntiles, int* nwg);
hipError_t launch_ppo_grad_fast(const GradArgs& a, int nwg, int gemm_mode, hipStream_t s);
// the same gradient with every product as six bf16 MFMA terms over three-plane operands (ph_ppo_split.hip): gemm_mode 2.
// Its slabs are in ITS accumulators' order (grad_slab_map_split); NetDims.split says the spec runs on it.
bool grad_split_eligible(const NetDims& nd);
// weight fragment image of the split kernel (ph_split.h): elements (bf16), distance between the planes of one fragment
constexpr int WIMG_PLANE = 64 * 8;
constexpr int WIMG_ELEMS = 2 * 4 * 3 * 2 * 3 * WIMG_PLANE;
void grad_weight_image_map(const ph_layout& lay, bool fold, int* map /* host, P x 2 */);
// image <- split(params) through the map (the image must have been zeroed by the caller)
hipError_t launch_weight_image(const float* params, unsigned short* image, const int* map, int P, hipStream_t s);
hipError_t launch_weight_image_check(const float* params, const unsigned short* image, const int* map, int P, int* mismatches,
                                     hipStream_t s);
void grad_slab_map_split(const ph_layout& lay, int* map /* host, 2 * RS_NET */, bool fold);
hipError_t launch_ppo_grad_split(const GradArgs& a, int nwg, hipStream_t s);
hipError_t launch_adv_stats(const AdvStatArgs& a, int n_total, hipStream_t s);
// observation rows (n, D) f32 -> the split kernel's plane image [n + 1][3][64] bf16 (features >= F zero; with `fold` feature 63 is
// 1: the first layer's bias rides as a feature; row n all zero)
constexpr int XIMG_ROW_U4 = 24;   // uint4 granules per image row: 3 planes x 8
hipError_t launch_obs_planes(const float* obs, int n, int D, int F, int fold, uint4* image, const float* adv, const float* logp,
                             const float* act, const float* ret, const float* val, uint4* rowrec, hipStream_t s);
hipError_t launch_ppo_reduce(const ReduceArgs& a, hipStream_t s);
int reduce_blocks(int slab_len);
// reduce + clip + Adam as one launch (ppo_step_kernel): `words` [reduce_blocks + 1] and `gen` are workspace that persists across
// launches (zeroed once); step_fused_fits says whether every block of the grid is resident at once on the current device
bool step_fused_fits(int nblk, int slab_len, int num_cu);
hipError_t launch_ppo_step(const ReduceArgs& r, const AdamArgs& ad, unsigned long long* words, unsigned int* gen,
                           unsigned int* sweep_error, unsigned long long timeout, hipStream_t st);
hipError_t launch_ppo_adam(const AdamArgs& a, hipStream_t s);
// the split-bf16 gradient kernel for one-hot (Discrete / MultiDiscrete) observations (ph_ppo_split_oh.hip): several feature
// chunks, heads of up to 32 logits in up to four components; slabs in its accumulator order, weight fragments from its own image
bool grad_split_oh_eligible(const NetDims& nd);
int grad_split_oh_slab_len(const NetDims& nd);
int grad_split_oh_wimage_elems(const NetDims& nd);
void grad_weight_image_map_oh(const ph_layout& lay, int nch, int* map /* host, P x 2 */);
void grad_slab_map_split_oh(const ph_layout& lay, int nch, int* map /* host, grad_split_oh_slab_len */);
hipError_t launch_ppo_grad_split_oh(const GradArgs& a, int nwg, hipStream_t s);
hipError_t launch_set_int(int* p, int v, hipStream_t s);
// behavioural cloning on the shared 32-32 policy (ph_bc.hip)
size_t bc_train_lds_bytes(int F, int L, int P, int A);
hipError_t launch_bc_train(const NetDims& nd, const ph_bc_layout& lay, float* params, float* adam_m, float* adam_v, int* step,
                           const float* obs, const float* acts, const int* order, int N, int batch, int n_epochs,
                           int max_batches, const ph_bc_hyper& hp, float* stats, hipStream_t s);
hipError_t launch_bc_forward(const NetDims& nd, const ph_bc_layout& lay, const float* params, const float* obs, int n,
                             const unsigned char* mask, const float* uniforms, const float* given, uint64_t seed,
                             uint64_t counter, int deterministic, int* act_i32, float* values, float* logp, float* entropy,
                             float* logits, hipStream_t s);
hipError_t launch_epoch_advance(unsigned long long* p, hipStream_t s);

// ---- ModularAlgorithm / ModularPolicy (ph_modular.hip) ---------------------------------------------------------------------
// one 64-64 tanh tower with its head: forward (latent + head output to HBM) or backward (head gradient and up to two external
// dL/dH2 terms from HBM -> register-order gradient slabs, optionally dL/dX)
struct TowerArgs {
  int nb, ntiles;
  const float* x;        // idx != null: rollout observations (T, E, x_ld) gathered through the minibatch order; else dense [nb][x_ld]
  int x_ld;              // floats per input row (D, or 64 for a tower reading a latent)
  int F;                 // features (<= 64); W1 has F rows
  const int* obs_off;    // device prefix sums of the observation nvec: one-hot features (x holds x_ld components), or null = Box
  const int* idx;        // (nb) env-major indices, or null
  int T, E;
  const float *W1, *b1, *W2, *b2, *hW, *hb;
  int head;              // 1: policy head, L logits (hW [64][L], hb [L]); 2: value head (hW [64], hb [1])
  int L;
  float* h2_out;         // forward: [nb][64] latent, or null
  float* head_out;       // forward: policy [nb][8] logits incl. bias (columns >= L zero) | value [nb]
  const float* dhead;    // backward: policy [nb][8] dL/dlogits | value [nb] dL/dv
  const float* ext0;     // backward: [nb][64] added to dL/dH2, or null
  const float* ext1;
  float* dx_out;         // backward: [nb][64] dL/dX (F == 64 towers), or null
  int dx_accumulate;     // 1: += into dx_out
  float* slab;           // backward: this tower's slab of workgroup w at slab + w * slab_stride (RS_NET floats)
  int slab_stride;
};
struct TowerLaunch {
  TowerArgs t[2];        // blockIdx.y
  int mode;              // 0 forward, 1 backward
  const int* stop_flag;
};
size_t tower_lds_bytes();
hipError_t launch_tower(const TowerLaunch& L, int nwg, int n_towers, int gemm_mode, hipStream_t s);
void tower_slab_map(int F, int L, int head, int oW1, int oB1, int oW2, int oB2, int oHW, int oHB, int* map /* host, RS_NET */);

struct ModLossArgs {
  int nb;
  const int* idx;
  int T, E;
  const float *rb_act, *rb_logp, *rb_adv, *rb_ret, *rb_val;
  const float* advstats;   // {mean, std} of this minibatch's advantages
  int L, n_mod, k_mod, nomain;
  const float* zm;         // [nb][8] main logits
  const float* zmod;       // [n_mod][nb][8] every module's logits
  float weight[PH_MOD_MAX];   // multiplicity of module m among the partners / num_partners
  const float *vm, *vk;    // [nb] main value, trained partner's value
  float clip, clip_vf, ent_coef, vf_coef, reg_coef;
  float* dzm;              // [nb][8] out
  float* dzmod;            // [n_mod][nb][8] out
  float* dv;               // [nb] out
  float* statpart;         // [gridDim.x][NSTATP]
  const int* stop_flag;
};
hipError_t launch_modular_loss(const ModLossArgs& a, hipStream_t s);
struct ModFinalizeArgs {
  const float* statpart;
  int nstatpart, nb;
  int* step;
  int* mod_first;          // [n_mod] optimizer step count before module m's value side first received a gradient, -1 = never
  int k_mod;
  float* kl_sum;           // running sum of this epoch's per-minibatch KLs
  float* stats_out;        // [PH_NSTAT] or null
  float ent_coef, vf_coef, reg_coef;
  const int* stop_flag;
};
hipError_t launch_modular_finalize(const ModFinalizeArgs& a, hipStream_t s);
hipError_t launch_modular_epoch_end(float* kl_sum, int n_mb, float target_kl, int* stop_flag, hipStream_t s);
struct ModAdamArgs {
  float *params, *m, *v;
  const float* grad;
  const float* blocksq;
  int nblk, P;
  const int* step;
  float lr, beta1, beta2, eps, max_norm;
  float* stats_out;
  int n_mod, k_mod, n_seg;
  int seg_lo[2 * PH_MOD_MAX], seg_hi[2 * PH_MOD_MAX], seg_mod[2 * PH_MOD_MAX];   // value-side parameter ranges of the modules
  const int* mod_first;
  const int* stop_flag;
};
hipError_t launch_modular_adam(const ModAdamArgs& a, hipStream_t s);
hipError_t launch_modular_act(const FwdArgs& a, const float* zm, const float* zk, const float* vm, const float* vk, int nomain,
                              float* logits_main, float* logits_partner, hipStream_t s);

}  // namespace ph

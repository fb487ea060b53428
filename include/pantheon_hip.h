/*
 * pantheon_hip.h -- C ABI of libpantheon_hip.so: the MI355X (gfx950) rollout-buffer / GAE / PPO-update
 * engine behind PantheonRL's OnPolicyAgent surface.
 *
 * The reference (Stanford-ILIAD/PantheonRL) is pure Python and delegates this path to
 * stable-baselines3==1.7.0 (reference setup.py:17).  Every entry point below replaces one reference call
 * site; the citation names it (paths relative to the reference root).  The reference-side binding a
 * maintainer would add is a ctypes stub -- see INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  Unless a parameter says "host", every pointer is a DEVICE
 *     pointer (HBM) owned by the caller; the library never frees caller memory and only allocates its own
 *     scratch inside ph_ctx (released by ph_ctx_destroy).
 *   - every call is asynchronous on the context's HIP stream (ph_ctx_set_stream); ph_ctx_sync() waits.
 *   - return value: 0 = ok, non-zero = error; ph_last_error() gives the thread-local message.  Nothing
 *     here calls abort().  No re-entrancy on one ph_ctx.
 *   - all floating point is IEEE float32 ("f32"), actions are stored float32 in the rollout buffer like
 *     SB3 does, and handed to environments as int32.
 *
 * Rollout buffer layout in HBM (same as SB3 RolloutBuffer): time-major (T, E, ...) row-major
 * float32 arrays; the PPO minibatch index n in [0, T*E) is SB3's env-major "swap_and_flatten" index
 * n = e*T + t and is translated to the physical row t*E + e inside the kernels (no flattened copy is made).
 *
 * Parameter vector layout (float32, P = ph_layout.P entries; H = 64), weights input-major [in][out]
 * (the transpose of torch.nn.Linear.weight):
 *   pi_W1[F][H] pi_b1[H] pi_W2[H][H] pi_b2[H]  vf_W1[F][H] vf_b1[H] vf_W2[H][H] vf_b2[H]
 *   act_W[H][L] act_b[L]  val_W[H] val_b[1]  (Box action spaces: + log_std[A], and L = A)
 */
#ifndef PANTHEON_HIP_H
#define PANTHEON_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PH_ABI_VERSION 7   /* 2: + ph_agent_*, ph_bc_*, ph_adap_*, ph_scripted_rollout, ph_liar_selfplay_rollout, ph_roundrobin_env_step,
                                  ph_buffer_compact_columns (additions only: every v1 signature is unchanged)
                              3: + ph_selfplay_rollout_persistent, PH_STEP_FIX_ILLEGAL / _MASK_ENV_ONLY, ph_modular_*, ph_roundrobin_*_iteration
                                  (additions only)
                              4: + ph_selfplay_rollout_persistent_capacity, ph_ppo_train's gradient pack (no signature changed)
                              5: + ph_policy_act_host, ph_buffer_add_reward_const, ph_adapmult_*, ph_ctx_set_joint_reward_rule (additions only)
                              6: + ph_bench_train_kernels, ph_debug_split_oh_tables, ph_ctx_step_errors (additions only)
                              7: ph_spec.act may be PH_SPACE_BOX on ph_layout_of / ph_policy_forward / ph_ppo_minibatch_grad / ph_ppo_train
                                  (an error before; no signature or struct changed) */
#define PH_HIDDEN 64     /* SB3 MlpPolicy default net_arch pi=[64,64], vf=[64,64] (modular/policies.py:112-114) */
#define PH_MAX_COMP 256  /* max MultiDiscrete components per space */
#define PH_MAX_LOGITS 64 /* max total policy logits L */
#define PH_MAX_BOX_ACT 16 /* max dimensions of a Box (continuous) action space: DiagGaussian head, general kernels */
#define PH_NSTAT 8       /* per-minibatch stats record, see ph_ppo_train */
#define PH_MOD_MAX 8     /* max partner modules of a ModularPolicy (ph_modular_*) */
#define PH_MAX_RANKS 16   /* ranks of one node in the peer-to-peer exchange layouts */

#define PH_SPACE_BOX 0      /* gym.spaces.Box, flattened length n                        */
#define PH_SPACE_DISCRETE 1 /* Discrete(k) (n=1, nvec={k}) or MultiDiscrete(nvec) (n=len) */

typedef struct ph_ctx ph_ctx;

typedef struct ph_space {
  int kind;              /* PH_SPACE_*                                                      */
  int n;                 /* Box: length.  Discrete family: number of components             */
  int nvec[PH_MAX_COMP]; /* Discrete family: categories per component                       */
} ph_space;

typedef struct ph_spec { /* observation_space / action_space of one agent (multiagentenv.py:72-79) */
  ph_space obs;
  ph_space act; /* PH_SPACE_DISCRETE: the categorical family PPO uses in every BASELINE config (every entry point).
                   PH_SPACE_BOX (n <= PH_MAX_BOX_ACT): SB3's DiagGaussianDistribution -- the head's n outputs are the means, log_std[n]
                   follows val_b in the parameter vector (ph_layout.P counts it); actions are float32 rows, clipped to the Box by the caller
                   (pantheonrl/common/util.py:84-99); ph_policy_forward, ph_ppo_minibatch_grad and ph_ppo_train only -- every other entry
                   point refuses the spec */
} ph_spec;

typedef struct ph_layout {
  int D; /* stored obs length per transition   */
  int F; /* feature length after one-hot       */
  int A; /* stored action length               */
  int L; /* total logits                       */
  int P; /* parameter count                    */
  int pi_W1, pi_b1, pi_W2, pi_b2, vf_W1, vf_b1, vf_W2, vf_b2, act_W, act_b, val_W, val_b; /* offsets */
} ph_layout;

typedef struct ph_rollout { /* SB3 RolloutBuffer storage; call sites agents.py:123-130,157,172-179,196-198 */
  int T, E;
  float *observations;   /* (T,E,D) */
  float *actions;        /* (T,E,A) */
  float *rewards;        /* (T,E)   */
  float *episode_starts; /* (T,E)   */
  float *values;         /* (T,E)   */
  float *log_probs;      /* (T,E)   */
  float *advantages;     /* (T,E)   */
  float *returns;        /* (T,E)   */
} ph_rollout;

typedef struct ph_ppo_hyper { /* SB3 PPO defaults mirrored at adap_learn.py:90-103; Adam eps modular/policies.py:84-88 */
  float learning_rate;      /* 3e-4 */
  float clip_range;         /* 0.2  */
  float clip_range_vf;      /* < 0 : None */
  float ent_coef;           /* 0.0  */
  float vf_coef;            /* 0.5  */
  float max_grad_norm;      /* 0.5  */
  float target_kl;          /* < 0 : None */
  int normalize_advantage;  /* 1    */
  float adam_beta1, adam_beta2, adam_eps; /* 0.9, 0.999, 1e-5 */
} ph_ppo_hyper;

typedef struct ph_opt_state {
  float *params; /* (P) */
  float *adam_m; /* (P) exp_avg    */
  float *adam_v; /* (P) exp_avg_sq */
  int *step;     /* (1) device counter: optimizer steps applied so far */
} ph_opt_state;

/* ---- library / context ------------------------------------------------------------------------------ */
int ph_abi_version(void);
const char *ph_last_error(void);
int ph_device_count(int *n_out /* host */);
/* create on HIP device `device`; fails (non-zero) when no gfx950 device is visible. <- PPO.__init__(device=) trainer.py:110,197 */
int ph_ctx_create(int device, ph_ctx **out /* host */);
int ph_ctx_destroy(ph_ctx *ctx);
int ph_ctx_set_stream(ph_ctx *ctx, void *hip_stream /* hipStream_t, may be NULL = default */);
int ph_ctx_sync(ph_ctx *ctx);
/* hipGraph capture of everything enqueued between begin/end on the ctx stream (launch-bound rollout loops) */
int ph_graph_begin(ph_ctx *ctx);
int ph_graph_end(ph_ctx *ctx, int *graph_id_out /* host */);
int ph_graph_launch(ph_ctx *ctx, int graph_id);
/* Device-resident RNG epoch.  Philox counters and permutation seeds are launch arguments, so a captured graph would
 * replay the same random numbers; when an epoch word (one device uint64, caller-owned) is attached, every random
 * stream is additionally keyed by *epoch, and ph_rng_epoch_advance enqueues *epoch += 1 (capturable). */
int ph_ctx_set_rng_epoch(ph_ctx *ctx, unsigned long long *epoch_dev /* device, or NULL to detach */);
int ph_rng_epoch_advance(ph_ctx *ctx);
/* Debug: while a buffer is attached, workgroup (bx,by) of policy_fwd / ppo_grad writes the shader clock at up to 16
 * phase boundaries to stamps[((by*gridDim.x)+bx)*16 + phase] (caller sizes it: 16 * workgroups int64).  NULL detaches. */
int ph_debug_set_profile_buffer(ph_ctx *ctx, long long *stamps_dev);
/* debug: how many elements of the context's pre-split weight image (the fragments gemm_mode 2's gradient kernel loads; kept in
 * step with the parameters by the optimizer kernel) differ from what `params` (device, P floats) split to right now.
 * *mismatches (host) = -1 when the context holds no image.  Synchronises the context's stream. */
int ph_debug_weight_image_mismatches(ph_ctx *ctx, const ph_spec *spec, const float *params, int *mismatches);
/* Scheduling hint: nothing else runs on this context's device while its training launches do (one learner per GPU --
 * north_star's "one agent per GPU", BASELINE config 5).  The slab reduction between two gradient launches then uses 1024-lane
 * blocks (it sits on the critical path) instead of the 256-lane blocks sized to run BESIDE another learner's gradient launch;
 * both walk the same summation tree, so results do not depend on the hint.  Default 0. */
int ph_set_exclusive_device(ph_ctx *ctx, int exclusive);
/* The one-launch reduce + clip + Adam step an exclusive learner uses (ppo_step_kernel) makes its blocks wait for each other's
 * partial results; a wait that runs into its bound (~2 s: never on a healthy, truly exclusive device) leaves that block's
 * parameters unchanged, marks the minibatch's statistics record (stats[7] = -1) and counts here.  *count_out = waits that expired
 * since the context was created; a non-zero count means an optimizer step was applied only in part: treat the run as invalid.
 * Synchronises the context's stream. */
int ph_ctx_step_errors(ph_ctx *ctx, unsigned int *count_out /* host */);
/* HIP-event timing on the ctx stream (bench.py roofline: events must sit on the stream the kernels run on) */
int ph_timer_start(ph_ctx *ctx);
int ph_timer_stop(ph_ctx *ctx, float *ms_out /* host */); /* synchronises */

/* ---- shapes --------------------------------------------------------------------------------------------- */
int ph_layout_of(const ph_spec *spec /* host */, ph_layout *out /* host */);
/* debug, host only (no device touched): the tables of gemm_mode 2's gradient kernel for `spec` -- slab_map[net * 8960 + position]
 * = parameter index held at that position of a workgroup's gradient slab (-1 = padding), image_map[2 * p + {0, 1}] = element
 * index (plane 0) of the weight-fragment image backed by parameter p (-1 = none).  *eligible = 0 (tables untouched) for specs
 * the kernel does not take. */
int ph_debug_split_tables(const ph_spec *spec, int *slab_map, int *image_map, int *eligible);
/* the same for the one-hot-observation kernel of gemm_mode 2 (ppo_grad_split_oh_kernel): *slab_len_out = ints slab_map needs
 * (a slab's floats, both nets), *image_elems_out = bf16 elements of its weight image; slab_map / image_map may be NULL (sizes only) */
int ph_debug_split_oh_tables(const ph_spec *spec, int *slab_map, int *image_map, int *slab_len_out, int *image_elems_out,
                             int *eligible);

/* ---- K1: rollout buffer writes --------------------------------------------------------------------------- */
/* RolloutBuffer.add(obs, action, reward=0, episode_start, value, log_prob) at row `pos` <- agents.py:172-179.
 * Inputs are copied (SB3 copies too), so callers may reuse them. */
int ph_buffer_add(ph_ctx *ctx, const ph_spec *spec, const ph_rollout *rb, int pos, const float *obs /* (E,D) */,
                  const float *actions /* (E,A) f32 */, const float *episode_start /* (E) */,
                  const float *values /* (E) */, const float *log_probs /* (E) */);
/* buf.rewards[pos][e] += reward[e] for e with env_mask[e] != 0 (NULL = all) <- Agent.update, agents.py:198 */
int ph_buffer_add_reward(ph_ctx *ctx, const ph_rollout *rb, int pos, const float *reward /* (E) */,
                         const unsigned char *env_mask /* (E) or NULL */);
/* Agent.update(reward, done) with a SCALAR reward (agents.py:198: buf.rewards[pos - 1][0] += reward; every environment of the
 * row receives it): the value travels as a kernel argument -- no host array, no copy, nothing to wait for. */
int ph_buffer_add_reward_const(ph_ctx *ctx, const ph_rollout *rb, int pos, float reward);
/* How the joint action of a SimultaneousEnv step enters every agent's reward in ph_buffer_add_reward_joint, ph_policy_step_multi,
 * ph_selfplay_rollout_* (multiagentenv.py:395-409 hands each agent ITS reward of the step).  A setting of the context:
 *   PH_JOINT_MATCH_BONUS (default): + bonus * [own action == partner's]   -- the synthetic driver's shared coordination term
 *   PH_JOINT_RPS:                   + bonus * rock-paper-scissors payoff of (own, partner's): (own - partner's + 3) % 3 == 1 wins,
 *                                     == 2 loses, 0 draws (rps.py:41-45; zero-sum: the partner's reward is the negative) -- a
 *                                     device-resident RPS self-play is the exchange rollout under this rule (bonus = 1, base reward 0,
 *                                     every step ends its episode) */
#define PH_JOINT_MATCH_BONUS 0
#define PH_JOINT_RPS 1
int ph_ctx_set_joint_reward_rule(ph_ctx *ctx, int rule);

/* Agent-per-GPU SimultaneousEnv step (multiagentenv.py:149-170 with the actions all-gathered over RCCL): seat `seat`
 * receives rewards[pos][e] += base_reward[e] + bonus * (joint[seat][e] == joint[*partner_seat][e]) -- the shared
 * coordination term of the synthetic transition, consuming the JOINT action.  joint_actions is (n_seats, E) int32;
 * partner_seat is a DEVICE int so the round-robin pairing (kept in Python) can change between graph replays. */
int ph_buffer_add_reward_joint(ph_ctx *ctx, const ph_rollout *rb, int pos, const float *base_reward /* (E) */,
                               const int *joint_actions, int n_seats, int seat, const int *partner_seat /* device */,
                               float bonus);
/* One-agent-per-GPU round-robin layout (BASELINE config 4: ego vs K OnPolicy partners, exactly one partner active per episode):
 * the synthetic 2-player SimultaneousEnv transition of n environments on the ego's rank <- MultiAgentEnv.step / reset,
 * multiagentenv.py:149-243, with the per-environment partner id the reference keeps in `partnerids` (:105-125).
 *   joint_actions (1 + n_partners, n) int32: row 0 the ego's actions, row 1 + k partner k's (all-gathered over RCCL)
 *   partnerid (n) int32 in/out: the partner of each environment; where done[e] != 0 it advances to (id + 1) % n_partners --
 *       resample_round_robin at that environment's own reset (:118-125,224)
 *   reward_out (n): base_reward[e] + bonus * [ego action == its partner's action] (Overcooked's reward is shared, so this is
 *       both players' reward); alt_action_out (n) or NULL: the partner action each environment consumed
 *   next_block (n, block_ld) f32 or NULL: header columns 0..2 of the next step's routing block are written --
 *       [partner id of the next step | reward_out | done] -- the block the ego's rank sends to the partner ranks together
 *       with the partner-seat observations in columns 3.. */
int ph_roundrobin_env_step(ph_ctx *ctx, const int *joint_actions, int *partnerid, const float *base_reward,
                           const float *done, float *reward_out, int *alt_action_out, float *next_block, int block_ld,
                           int n_partners, float bonus, int n);
/* The same layout with ONE native call per iteration and rank (no host tensor op, collective call or synchronisation per
 * environment step).  Every rank owns a fine-grained receive area (ph_p2p_alloc) that the others map through HIP IPC
 * (ph_p2p_open): [64 stamp words | slot 0 | slot 1], ph_rr_area_bytes() long; a partner's slot holds the block's header rows
 * (n, 4) = [partner id | reward | done | -] followed by the observations (n, D), rank 0's slot the actions (1 + K, n).  Per step t of iteration i (stamp i * T + t + 1):
 *   rank 0 : ego forward + rollout-buffer row (its actions land in row 0 of its own slot t & 1) -> the routing block of step t
 *            stored into every partner's slot t & 1, then the partner's block stamp -> wait for the K action stamps -> the
 *            transition of ph_roundrobin_env_step (writes the header of block t + 1)            multiagentenv.py:149-243
 *   rank 1+k: wait for the block stamp -> Agent.update of the previous step where it acted, this step's record mask, room and
 *            episode_start (agents.py:186-203,176) -> ph_policy_forward_ragged -> advance the write rows, store its actions into
 *            row 1 + k of rank 0's slot, then its action stamp.
 * A step is ping-pong between rank 0 and each partner, so two slots suffice.  Waits are bounded; a timeout bumps *error.
 * The caller (roundrobin.py) credits the ego's last reward, runs the learners' updates and decides ONCE per iteration whether a
 * partner whose columns are full trains (the reference's partner checks before every action, agents.py:126). */
typedef struct ph_rr_link {
  int n_partners, rank;                 /* rank 0 = ego, 1 + k = partner k */
  int n, block_ld;                      /* environments; floats per routing-block row (3 + D) */
  void *area[PH_MAX_RANKS];             /* [r] rank r's receive area as mapped in this process (own one included) */
  unsigned long long *error;            /* local device word */
  unsigned long long timeout_cycles;    /* bound of one wait, wall_clock64() ticks (100 MHz) */
} ph_rr_link;
typedef struct ph_rr_ego {
  const ph_spec *spec;                  /* host */
  const float *params;
  const float *obs_seq;                 /* (T, n, D) the ego's observations */
  const float *base_reward_seq;         /* (T, n) */
  const float *done_seq;                /* (T, n) */
  float *blocks;                        /* (T, n, block_ld): observation columns prefilled, headers written step by step */
  int *partnerid;                       /* (n) in/out */
  float *rewards;                       /* (T, n) out */
  int *alt_actions;                     /* (T, n) out */
  int *partner_trace;                   /* (T, n) out */
  const float *episode_start0;          /* (n) */
  unsigned long long seed, counter0;
  float *values, *log_probs;            /* (n) */
  const ph_rollout *rb;                 /* host; rows 0 .. T-1 */
  float bonus;
} ph_rr_ego;
typedef struct ph_rr_partner {
  const ph_spec *spec;                  /* host */
  const float *params;
  float *obs_scratch;                   /* unused (the forward reads the observations from the receive slot); may be NULL */
  float *es_scratch;                    /* (n) */
  unsigned char *can_scratch;           /* (n) */
  int *pos;                             /* (n) per-environment write rows, in/out */
  unsigned char *boundary, *term, *open, *prev_mask;   /* (n) RaggedVecOnPolicyAgent's book-keeping, in/out */
  unsigned long long seed, counter0;
  int *actions;                         /* (n) */
  float *values, *log_probs;            /* (n) */
  const ph_rollout *rb;                 /* host */
} ph_rr_partner;
int ph_rr_area_bytes(int n_partners, int n, int block_ld, size_t *bytes_out /* host */);
int ph_roundrobin_ego_iteration(ph_ctx *ctx, const ph_rr_link *link, const ph_rr_ego *ego, int T, unsigned long long iteration);
int ph_roundrobin_partner_iteration(ph_ctx *ctx, const ph_rr_link *link, const ph_rr_partner *partner, int T,
                                    unsigned long long iteration);
/* RolloutBuffer.reset() <- agents.py:157 : zero-fills every array */
int ph_buffer_reset(ph_ctx *ctx, const ph_spec *spec, const ph_rollout *rb);

/* ---- K2: GAE ------------------------------------------------------------------------------------------------ */
/* RolloutBuffer.compute_returns_and_advantage(last_values, dones) <- agents.py:127-130.
 * gamma / gae_lambda are doubles because the reference multiplies them as Python floats before the
 * float32 array arithmetic (gamma*gae_lambda is rounded to f32 once).
 * mode 0 = auto, 1 = serial-in-T / one lane per env (summation order and rounding identical to the numpy
 * loop: bit-exact), 2 = chunked wavefront suffix scan over T, any T (fp32 tolerance). */
int ph_gae(ph_ctx *ctx, const ph_rollout *rb, const float *last_values /* (E) */, const float *dones /* (E) */,
           double gamma, double gae_lambda, int mode);

/* ---- K4: policy forward --------------------------------------------------------------------------------------- */
/* ActorCriticPolicy.forward(obs) -> (actions, values, log_probs) <- util.py:63-81 (action_from_policy), agents.py:162.
 *  action_mask (n,L) u8 or NULL : logits -= 30*(1-mask)  (modular/policies.py:330-333)
 *  uniforms (n,A) or NULL       : teacher-forced inverse-CDF sampling; NULL = Philox4x32-10 keyed (seed, counter, row, comp)
 *  given_actions (n,A) f32 or NULL : evaluate these instead of sampling (evaluate_actions, modular/policies.py:364-383)
 *  deterministic != 0           : argmax
 * Box action spaces (spec->act.kind == PH_SPACE_BOX; SB3 DiagGaussianDistribution, L = A): the head's outputs are the means,
 *  action = mean + exp(log_std) * eps with eps ~ N(0, 1) -- `uniforms` (n,A) then carries the STANDARD-NORMAL draws eps (NULL:
 *  Box-Muller over two Philox uniforms per dimension), deterministic = the mean, log_prob / entropy = Normal's summed over the
 *  dimensions; actions leave through actions_f32 only (UNclipped: the caller clips for the environment, util.py:84-99),
 *  `logits` returns the means, action_mask must be NULL.
 * Outputs (any may be NULL): actions_i32 (n,A), actions_f32 (n,A), values (n), log_probs (n), entropy (n), logits (n,L).
 * When rb != NULL the transition is also written at row `pos` of the rollout buffer (fused RolloutBuffer.add with
 * reward 0 and episode_start = episode_start_in) -- n must equal rb->E; if pending_reward (E) is also given, it is
 * added to row pos-1's rewards in the same launch (the previous step's Agent.update, agents.py:198, folded in).
 * params must be 16-byte aligned. */
int ph_policy_forward(ph_ctx *ctx, const ph_spec *spec, const float *params, const float *obs, int n,
                      const unsigned char *action_mask, const float *uniforms, const float *given_actions,
                      unsigned long long seed, unsigned long long counter, int deterministic, int *actions_i32,
                      float *actions_f32, float *values, float *log_probs, float *entropy, float *logits,
                      const ph_rollout *rb, int pos, const float *episode_start_in, const float *pending_reward,
                      int gemm_mode);

/* The same call for an environment that lives on the HOST (the reference's own situation: one get_action per environment step,
 * the action needed back before the environment can move -- agents.py:111-184, util.py:63-81): observations, episode starts and
 * the three results are HOST arrays; the call stages them through pinned, device-visible memory of the context that the forward
 * kernel reads and writes directly (with the fused rollout-buffer write when rb != NULL, as above) and returns when the kernel
 * has finished.  One call, one launch and one wait per environment step instead of several tensor operations; the sampled
 * actions are bitwise those of ph_policy_forward with the same (seed, counter).  A launch of one 16-row tile (n <= 16, the
 * 16-row forward's shape class) signals its end through two words in the same pinned memory, stored by the policy and the value
 * workgroup after their last output, and the call polls those (bounded: 50 ms, then the stream wait) instead of synchronising
 * the stream; PH_ACT_HOST_WAIT=stream in the environment selects the stream wait for every launch.  Either way every write of the
 * launch, the rollout-buffer row included, is complete when the call returns.
 *   obs_host (n, D) f32; episode_start_host (n) f32 or NULL (required with rb); actions_host (n, A) i32; values_host (n);
 *   log_probs_host (n) -- any output may be NULL. */
int ph_policy_act_host(ph_ctx *ctx, const ph_spec *spec, const float *params, const float *obs_host, int n,
                       const float *episode_start_host, unsigned long long seed, unsigned long long counter,
                       int deterministic, int *actions_host, float *values_host, float *log_probs_host,
                       const ph_rollout *rb, int pos, int gemm_mode);

/* One environment step of SEVERAL local agents in one launch (agent-per-GPU self-play hosts two learners per GPU).
 * Each record is one agent's ph_policy_forward call with the fused rollout-buffer write; optionally the PREVIOUS step's
 * late reward (Agent.update, agents.py:198) is applied to row pos-1 in the same launch:
 *   rewards[pos-1][e] += pending_reward[e] + (joint_actions ? bonus * [joint[seat][e] == joint[*partner_seat][e]] : 0)
 * where joint_actions (n_seats, n) int32 is the all-gathered action matrix of that previous step.
 * All records must have the same n and the same padded logit count; n_calls <= 4. */
typedef struct ph_step_call {
  const ph_spec *spec;               /* host */
  const float *params;
  const float *obs;                  /* (n, D) */
  int n;
  const unsigned char *action_mask;  /* (n, L) or NULL */
  unsigned long long seed, counter;
  int deterministic;                 /* bit 0: argmax instead of sampling; PH_STEP_FIX_ILLEGAL / PH_STEP_MASK_ENV_ONLY: see below */
  int *actions_i32;                  /* (n, A) */
  float *values;                     /* (n) */
  float *log_probs;                  /* (n) */
  const ph_rollout *rb;              /* host; required */
  int pos;
  const float *episode_start_in;     /* (n) */
  const float *pending_reward;       /* (n) or NULL */
  const int *joint_actions;          /* (n_seats, n) or NULL */
  int n_seats, seat;
  const int *partner_seat;           /* device int; required when joint_actions != NULL */
  float bonus;
} ph_step_call;
/* Flags in `deterministic` besides bit 0, for a record with an action mask (single Discrete heads of <= 8 logits):
 *  PH_STEP_FIX_ILLEGAL: actions_i32 -- what the environment and the action exchange consume -- is the ENV-SIDE fix-up of an
 *      illegal sample (first legal index: pettingzoo.py:81-82 does this inside the environment, before base_env.step), while
 *      the rollout-buffer row and the log-prob keep the sampled action, as the reference's agent does (it is never told).
 *  PH_STEP_MASK_ENV_ONLY: the mask is NOT applied to the logits.  That is the reference's plain PPO agent: OnPolicyAgent
 *      hands its policy obs.obs only (agents.py:162; action_from_policy util.py:63-81), so only ModularPolicy
 *      (modular/policies.py:330-333) ever offsets logits; everyone else samples unmasked and the environment repairs. */
#define PH_STEP_FIX_ILLEGAL 2
#define PH_STEP_MASK_ENV_ONLY 4
int ph_policy_step_multi(ph_ctx *ctx, int n_calls, const ph_step_call *calls /* host */);

/* ---- agent-per-GPU action exchange over RCCL ------------------------------------------------------
 * What MultiAgentEnv._get_actions does in-process (multiagentenv.py:149-170) across ranks: every rank contributes the actions
 * of its local agents for the current step, every rank receives the joint action.  librccl.so is loaded on first use
 * (dlopen) -- a build or a node without it still loads this library and gets an error from these calls only.
 *  ph_comm_unique_id: ncclGetUniqueId on the calling rank (rank 0); the 128 bytes travel to the other ranks through the
 *      caller's rendezvous (torch.distributed's store).
 *  ph_comm_init:      ncclCommInitRank for this context (one process per GPU).
 *  ph_all_gather_i32: joint[r*count .. (r+1)*count) <- rank r's local[0..count), on the context's stream, no host sync.
 *      Without a communicator (single process) it degenerates to a device copy.
 *  ph_selfplay_rollout: T environment steps back to back -- step t = ph_policy_step_multi(calls + t*n_calls) followed by
 *      the all-gather of the step's actions -- in ONE host call: the per-step host work is two native enqueues. */
#define PH_COMM_ID_BYTES 128
int ph_comm_unique_id(unsigned char *id_out /* host, PH_COMM_ID_BYTES */);
int ph_comm_init(ph_ctx *ctx, const unsigned char *id /* host */, int world, int rank);
int ph_comm_destroy(ph_ctx *ctx);
int ph_all_gather_i32(ph_ctx *ctx, const int *local /* (count) */, int *joint /* (world*count) */, int count);
int ph_selfplay_rollout(ph_ctx *ctx, int n_calls, const ph_step_call *calls /* host, [T][n_calls] */, int T,
                        const int *local, int *joint, int count);

/* ---- the same exchange as direct peer-to-peer stores over xGMI -------------------------------------------------------
 * The per-step message is KB-sized, so the collective's cost is its latency.  Here every rank owns a fine-grained
 * (uncached, cross-device coherent) receive area that all ranks map through HIP IPC; after the step's forward launch a
 * one-workgroup kernel stores the local actions straight into every peer's receive area (slot = step parity), fences at
 * system scope and publishes a monotonic step stamp in the peer's flag array; the consumer is a one-wave kernel that waits
 * for all stamps (bounded: a timeout raises the error word instead of hanging the device).
 *  ph_p2p_alloc / ph_p2p_open / ph_p2p_close / ph_p2p_free: the shareable buffer and its 64-byte IPC handle.
 *  ph_p2p: the exchange as seen from one rank -- peers' receive areas and flag arrays as mapped here (own ones included).
 *  ph_p2p_push(t): joint[t&1][peer][rank*count ..] <- local[0..count) for every peer; flags[peer][rank] <- stamp(t).
 *  ph_p2p_wait(t): until flags_local[src] >= stamp(t) for every src.   stamp(t) = (*epoch) * T + t + 1.
 *  ph_selfplay_rollout_p2p: T x { ph_policy_step_multi, push, wait } in one host call.  When every call fits the 16-row
 *      step kernel the exchange is folded INTO the step launch with the stamp in-band: a policy workgroup stores each row's
 *      action as ONE 8-byte word (stamp << 32 | action) into every rank's `ll` area (slot = t mod ll_slots; 8-byte stores are
 *      single-copy atomic, so no fence, flag or arrival counter is needed), and the value workgroups of the next step poll
 *      exactly the two words they consume (own seat, partner seat) until the stamp matches.  A rank therefore waits for
 *      its partner's rank only and can run up to world-1 steps ahead of some other rank: ll_slots must be >= T so that no
 *      slot is reused inside an iteration (the unpack of step T-1 waits for every rank, which bounds the skew per iteration).  After the last step the
 *      words of step T-1 are unpacked into this rank's plain receive slot for ordinary consumers. */
#define PH_IPC_HANDLE_BYTES 64
typedef struct ph_p2p {
  int world, rank, count, T;
  int *joint[2][PH_MAX_RANKS];                 /* [parity][peer] -> that peer's (world*count) int32 receive slot */
  unsigned long long *flags[PH_MAX_RANKS];     /* [peer] -> that peer's (world) stamp array */
  unsigned long long *ll[PH_MAX_RANKS];        /* [peer] -> that peer's ll_slots x (world*count) stamp-in-band words */
  int ll_slots;                                /* slot of step t = t mod ll_slots; >= T (see above) */
  const unsigned long long *epoch;             /* device word advanced once per iteration (ph_rng_epoch_advance) */
  unsigned long long *error;                   /* local device word: number of timed-out waits */
  unsigned long long timeout_cycles;           /* bound of one wait in wall_clock64() ticks (100 MHz) */
} ph_p2p;
int ph_p2p_alloc(ph_ctx *ctx, size_t bytes, void **ptr_out, unsigned char *handle_out /* host, 64 */);
int ph_p2p_open(ph_ctx *ctx, const unsigned char *handle /* host, 64 */, void **ptr_out);
int ph_p2p_close(ph_ctx *ctx, void *ptr);
int ph_p2p_free(ph_ctx *ctx, void *ptr);
int ph_p2p_push(ph_ctx *ctx, const ph_p2p *x, const int *local, int t);
int ph_p2p_wait(ph_ctx *ctx, const ph_p2p *x, int t);
/* stamp-in-band variant as separate launches (what the fused step launch does per row; used by the route self-test):
 * ll_push stores (stamp(t) << 32 | local[i]) into every rank's word area, ll_unpack waits for step t's words of every rank
 * and writes their low halves to this rank's plain receive slot of parity t & 1 */
int ph_p2p_ll_push(ph_ctx *ctx, const ph_p2p *x, const int *local, int t);
int ph_p2p_ll_unpack(ph_ctx *ctx, const ph_p2p *x, int t);
int ph_selfplay_rollout_p2p(ph_ctx *ctx, int n_calls, const ph_step_call *calls /* host, [T][n_calls] */, int T,
                            const int *local, const ph_p2p *x);

/* The same T steps as ONE launch (the N > 1 counterpart of ph_scripted_rollout): when every local agent's observations,
 * base rewards and done flags of the rollout already sit in HBM (the synthetic rollout driver of SURVEY.md 8d), the only thing
 * that has to cross a step boundary is what multiagentenv.py:149-170 hands between agents -- each seat's action.  Here that
 * hand-off happens INSIDE the launch: the policy workgroup of 16 environments stores each sampled action as one stamped
 * 8-byte word into every rank's `ll` area (as ph_selfplay_rollout_p2p's fused step does), and the value workgroup of step
 * t + 1 polls the two words it consumes (own seat, partner seat) before it credits step t's reward
 *   rewards[t][e] += rew_seq[t][e] + bonus * [joint[seat][e] == joint[*partner_seat][e]]
 * (the last step's reward is credited at the end of the launch).  Bitwise the launch-per-step walk of
 * ph_selfplay_rollout_p2p followed by ph_buffer_add_reward_joint for step T - 1 (tests/test_gpu_parity.py).
 *   Word slots: step t of an iteration whose epoch word reads e uses slot (e & 1) * x->T + t (x->T >= T), so x->ll_slots must be
 *   >= 2 x->T (a
 *   launch's policy workgroups run ahead of its value workgroups; alternating halves keep a faster peer's next iteration
 *   from overwriting words this rank has not consumed).  After the launch the words of step T - 1 are unpacked into the plain
 *   receive slot of parity (T - 1) & 1, which also makes this rank wait for every peer once per iteration.
 *   Every workgroup of the launch must be resident at once (value workgroups poll; policy workgroups never wait):
 *   n_calls * 2 * ceil(n / 16) workgroups, times `ranks_on_device` when several ranks share one GPU, must not exceed
 *   what ph_selfplay_rollout_persistent_capacity reports -- the runtime's occupancy answer for the rollout kernel on this
 *   device (workgroups per CU x CUs, one workgroup per CU held back where the answer is not bounded by LDS: the occupancy API
 *   can be one high there) -- otherwise the call refuses (use ph_selfplay_rollout_p2p).  Shapes of the 16-row forward only. */
typedef struct ph_rollout_call {
  const ph_spec *spec;               /* host */
  const float *params;
  const float *obs_seq;              /* (T, n, D) */
  const float *rew_seq;              /* (T, n) base reward of every step */
  const float *done_seq;             /* (T, n) */
  const unsigned char *mask_seq;     /* (T, n, L) action masks or NULL */
  int n;
  const float *episode_start0;       /* (n) */
  unsigned long long seed, counter0; /* step t samples with counter0 + t */
  int mask_mode;                     /* with mask_seq: 0 = policy-side logit offset only, 1 = offset + env-side fix-up
                                        (PH_STEP_FIX_ILLEGAL), 2 = env-side fix-up only (+ PH_STEP_MASK_ENV_ONLY) */
  int *actions_i32;                  /* (n) the last step's environment-side actions */
  float *values;                     /* (n) */
  float *log_probs;                  /* (n) */
  const ph_rollout *rb;              /* host; rows 0 .. T-1 are written */
  int n_seats, seat;
  const int *partner_seat;           /* device int */
  float bonus;
} ph_rollout_call;
int ph_selfplay_rollout_persistent(ph_ctx *ctx, int n_calls, const ph_rollout_call *calls /* host */, int T,
                                   const ph_p2p *x, int ranks_on_device);
/* workgroups of the exchange rollout kernel the device keeps resident at once (hipOccupancyMaxActiveBlocksPerMultiprocessor,
 * not a formula): the bound ph_selfplay_rollout_persistent checks its grid against.  Callers that must agree across ranks
 * (vec.FusedSelfPlayRollout) read it, decide, and all-reduce the decision. */
int ph_selfplay_rollout_persistent_capacity(ph_ctx *ctx, int *workgroups_out);

/* Ragged rollout buffers for vectorised TURN-BASED games ("per-env pos"): a partner does not act in
 * every env at every step, so each env e has its own write row pos_env[e] (device int32, caller-owned).
 *  ph_policy_forward_ragged: forward for all n = rb->E envs; the transition of env e is recorded at row pos_env[e] iff
 *      record_mask[e] != 0 and pos_env[e] < rb->T; `values` (n) is only updated for recorded envs (it caches V of the
 *      last recorded action, the bootstrap OnPolicyAgent hands to GAE -- agents.py:127-130,183).
 *  ph_buffer_add_reward_ragged: rewards[pos_env[e]-1][e] += reward[e] for env_mask[e] != 0 with pos_env[e] >= 1.
 *  ph_ragged_advance: pos_env[e] += 1 for record_mask[e] != 0 while pos_env[e] < rb->T (call after the forward). */
int ph_policy_forward_ragged(ph_ctx *ctx, const ph_spec *spec, const float *params, const float *obs,
                             const unsigned char *action_mask, unsigned long long seed, unsigned long long counter,
                             int deterministic, int *actions_i32, float *values, float *log_probs,
                             const ph_rollout *rb, const int *pos_env, const unsigned char *record_mask,
                             const float *episode_start_in);
int ph_buffer_add_reward_ragged(ph_ctx *ctx, const ph_rollout *rb, const int *pos_env, const float *reward,
                                const unsigned char *env_mask);
int ph_ragged_advance(ph_ctx *ctx, const ph_rollout *rb, int *pos_env, const unsigned char *record_mask);
/* ph_buffer_compact_columns: dst (T, n, .) <- columns cols[0..n) (device int32, environment indices, any order) of src
 * (T, E, .), all eight arrays.  A partner whose environments reach it at different rates (turn-based games, round-robin
 * partner selection) trains on the columns that are full -- the E-environment reading of "train once n_steps of its own
 * transitions are in the buffer" (agents.py:126) -- instead of waiting for every environment. */
int ph_buffer_compact_columns(ph_ctx *ctx, const ph_spec *spec, const ph_rollout *src, const ph_rollout *dst,
                              const int *cols /* device (n) */, int n);

/* env-side illegal-action fix-up: action not legal -> first legal index <- pettingzoo.py:81-82.  Integer, bit-exact. */
int ph_fix_illegal_actions(ph_ctx *ctx, int *actions /* (n) in/out */, const unsigned char *action_mask /* (n,L) */,
                           int n, int L);

/* ---- vectorised integer game rules ------------------------------------------------------ */
/* RPSEnv.multi_step <- pantheonrl/envs/rpsgym/rps.py:41-45 for n environments: rewards (ego, partner) as f32 */
int ph_rps_step(ph_ctx *ctx, const int *ego_actions, const int *alt_actions, float *ego_reward, float *alt_reward, int n);
/* LiarEnv.player_step <- pantheonrl/envs/liargym/liar.py:58-83 (sanitize_action, eval_bluff, getObs) for every env with
 * active[e] != 0 (NULL = all).  hands (n,12) int32 = ego histogram then partner histogram; history (n,24) int32 moves
 * newest first, nmoves (n) int32 -- both updated in place; actions (n,2) int32 raw (side, count-1) of the mover,
 * is_ego (n) u8 = who moves.  Outputs: obs_next (n,30) f32 observation of the OTHER player, rewards (n,2) f32
 * (ego, partner), done (n) u8.  hands / history must be 16-byte aligned, actions / obs / rewards 8-byte aligned (the
 * kernels move a table's state with 16-byte accesses); the same holds for ph_liar_reset / ph_liar_obs and the arrays
 * of ph_liar_selfplay. */
int ph_liar_step(ph_ctx *ctx, const int *hands, int *history, int *nmoves, const int *actions,
                 const unsigned char *is_ego, const unsigned char *active, float *obs_next, float *rewards,
                 unsigned char *done, int n);

/* LiarEnv.multi_reset <- liar.py:96-102 for every env with reset_mask[e] != 0 (NULL = all): dice from Philox4x32-10
 * keyed (seed, counter [+ the context's RNG epoch word << 32, when one is attached], env, die/4; die%4 selects the word),
 * empty history; ego_first (n) u8 <- first mover ~ Bernoulli(probegostart)
 * (TurnBasedEnv.n_reset, multiagentenv.py:323-326). */
int ph_liar_reset(ph_ctx *ctx, int *hands, int *history, int *nmoves, const unsigned char *reset_mask,
                  unsigned char *ego_first, unsigned long long seed, unsigned long long counter, float probegostart,
                  int n);
/* LiarEnv.getObs(isego) <- liar.py:53-56: observation (n,30) f32 of the player is_ego[e] selects, active envs only */
int ph_liar_obs(ph_ctx *ctx, const int *hands, const int *history, const int *nmoves, const unsigned char *is_ego,
                const unsigned char *active, float *obs_out, int n);

/* One vectorised MultiAgentEnv.step of n Liar's Dice tables with a PPO ego and a PPO partner, entirely on the device
 * (multiagentenv.py:149-215 + TurnBasedEnv.n_step/n_reset :307-327 + LiarEnv, liar.py:53-102), as ONE host call that
 * enqueues ~15 launches and keeps every mask on the device: ego forward (recorded at row ego_pos of its rectangular
 * buffer) -> player_step -> partner credited / flagged (Agent.update) -> partner forward where the game goes on (ragged
 * rows) -> player_step -> both credited, ego observation, episode flags -> finished tables re-dealt -> where the partner
 * opens the new game it moves once -> every table is back at the ego's turn.  With deal_only != 0 only the re-deal half
 * runs, for the tables flagged in `done` (the initial deal).  RNG counters: ego forward `counter`, partner forwards
 * 2*counter and 2*counter+1, dice `counter`.  All pointers are device pointers owned by the caller. */
typedef struct ph_liar_selfplay {
  int n;
  const ph_spec *spec;                       /* LiarsDice spaces (both seats) */
  int *hands, *history, *nmoves;             /* (n,12) (n,24) (n) game state */
  unsigned char *ego_first;                  /* (n) */
  unsigned long long dice_seed;
  float probegostart;
  /* ego: rectangular rollout buffer */
  const float *ego_params;
  const ph_rollout *ego_rb;
  int *ego_actions;                          /* (n,2) */
  float *ego_values, *ego_log_probs;         /* (n) */
  float *ego_episode_start;                  /* (n) in: flags of the previous step; out: this step's done */
  unsigned long long ego_seed;
  /* partner: ragged rollout buffer */
  const float *alt_params;
  const ph_rollout *alt_rb;
  int *alt_actions;                          /* (n,2) */
  float *alt_values, *alt_log_probs;         /* (n) */
  int *alt_pos;                              /* (n) per-table write row */
  unsigned char *alt_boundary, *alt_term, *alt_open, *alt_acted;   /* (n) OnPolicyAgent book-keeping per table */
  unsigned long long alt_seed;
  /* observations of whoever moves next */
  float *obs_ego, *obs_alt;                  /* (n,30) */
  unsigned long long *episodes;              /* finished games */
  /* scratch */
  float *obs_next, *rew1, *rew2, *es_alt;    /* (n,30) (n,2) (n,2) (n) */
  unsigned char *done1, *done2, *running, *can, *alt_opens, *ego_opens, *done;   /* (n) */
  const unsigned char *zeros8, *ones8;       /* (n) constants */
} ph_liar_selfplay;
int ph_liar_selfplay_step(ph_ctx *ctx, const ph_liar_selfplay *s, int ego_pos, unsigned long long counter, int deal_only);
/* n_steps of those vectorised steps -- ego rows ego_pos .. ego_pos + n_steps - 1, step t with counter + t -- in ONE persistent
 * launch: tables are independent, so one workgroup owns up to 16 tables (as few as spreads n over the CUs) for the whole rollout
 * and runs the three forwards and the book-keeping of every step with workgroup barriers in place of kernel boundaries.  The
 * game state, observations, both rollout buffers, the partner's book-keeping, the cached values and the episode count are
 * bitwise the result of n_steps calls of ph_liar_selfplay_step(deal_only = 0).  A partner forward that none of a workgroup's
 * tables asks for (no game running after the ego's move; no fresh game opened by the partner) is not executed: the scratch
 * outputs such a forward would have left for tables that do not move (partner action / log-prob cache entries nothing reads)
 * are then stale.  Needs the 16-row one-hot forward's shape class (<= 64 observation components, <= 32 logits). */
int ph_liar_selfplay_rollout(ph_ctx *ctx, const ph_liar_selfplay *s, int ego_pos, int n_steps, unsigned long long counter);

/* n_steps vectorised agent steps against a SCRIPTED environment -- observations, rewards and dones of every step already
 * resident on the device (the synthetic rollout driver of SURVEY.md 8d; a recorded trajectory being replayed) -- in ONE launch:
 * a workgroup stages the network once and walks the steps of its 16 environments.  Exactly, bit for bit,
 *     for t in 0 .. n_steps-1:  ph_policy_forward(obs_seq[t], counter0 + t, rb, pos0 + t,
 *                                                 episode_start_in = t ? done_seq[t-1] : episode_start0,
 *                                                 pending_reward   = t ? rew_seq[t-1]  : NULL)
 *     ph_buffer_add_reward(rb, pos0 + n_steps - 1, rew_seq[n_steps-1], NULL)
 * i.e. OnPolicyAgent.get_action / update per step (agents.py:111-203) with the launch boundaries removed; actions / values /
 * log_probs (n, .) end up holding the last step's outputs, as after the per-step calls.  obs_seq (n_steps, n, D), rew_seq and
 * done_seq (n_steps, n) f32.  Shapes of the 16-row forward only (one feature chunk, one Discrete head of <= 8 logits,
 * n < 16384); anything else is an error, not a slower path. */
int ph_scripted_rollout(ph_ctx *ctx, const ph_spec *spec, const float *params, const float *obs_seq, const float *rew_seq,
                        const float *done_seq, int n, int n_steps, const float *episode_start0, unsigned long long seed,
                        unsigned long long counter0, int *actions_i32, float *values, float *log_probs, const ph_rollout *rb,
                        int pos0, int gemm_mode);

/* Frame stack as a device ring buffer <- HistoryQueue.add / reset, wrappers.py:37-71, applied to
 * n environments: stack (n, numframes*D) f32 holds the last numframes observations NEWEST FIRST; for envs with
 * reset_mask[e] != 0 the history is first refilled with default_obs (D, NULL = zeros), then obs (n, D) is pushed. */
int ph_framestack_push(ph_ctx *ctx, float *stack, const float *obs, const unsigned char *reset_mask,
                       const float *default_obs, int n, int D, int numframes);

/* ---- K3+K5+K6: PPO.train() ------------------------------------------------------------------------------------ */
/* PPO.train() <- agents.py:155 (SB3 semantics SURVEY.md A.3; in-tree witness adap_learn.py:229-371).
 * For each epoch, for each consecutive slice of `batch_size` indices (last may be short): gather by index,
 * advantage normalisation, clipped surrogate + value + entropy loss, backward, global-norm clip, Adam.
 *   perms      (n_epochs, T*E) int32 env-major indices (teacher-forced np.random.permutation) or NULL = a keyed
 *              Feistel permutation of [0,T*E) per epoch generated in-kernel from perm_seed.
 *   stats      (n_epochs * ceil(T*E/batch_size), PH_NSTAT) f32 or NULL: per minibatch
 *              {policy_loss, value_loss, entropy_loss, clip_fraction, approx_kl, loss, grad_norm, applied}.
 *   gemm_mode  how the 64x64x64 products are computed:
 *              0 = v_mfma_f32_32x32x2_f32, exact float32 (bit-for-bit the k-ordered fmaf chain);
 *              1 = VALU fmaf chain with the same tile order (debug cross-check of 0, bitwise equal to it);
 *              2 = float32 operands carried as three bf16 planes (x = h + m + l, 24 significand bits), each product six
 *                  v_mfma_f32_16x16x32_bf16 terms accumulated in float32 (ppo_grad_split_kernel): float32 accuracy -- against a
 *                  float64 gradient its error is not larger than mode 0's (tests/test_gpu_parity.py) -- at 6/16 of the matrix
 *                  cycles, beside the vector ALU instead of on it.  Applies to the gradient launches of specs that kernel
 *                  takes (Box observations, <= 64 features, one Discrete head of <= 8 logits); everywhere else 2 means 0.
 * target_kl early stop is evaluated on the device; later minibatches become no-ops (applied = 0). */
int ph_ppo_train(ph_ctx *ctx, const ph_spec *spec, const ph_opt_state *opt, const ph_rollout *rb,
                 const ph_ppo_hyper *hyper /* host */, int n_epochs, int batch_size, const int *perms,
                 unsigned long long perm_seed, float *stats, int gemm_mode);
/* The same call for several independent learners at once (trainer.py: `PPO PPO` self-play keeps two PPO objects, each
 * with its own buffer, policy and optimizer <- trainer.py:126,203; README.md:6).  Every learner's launches go to its own
 * context's stream exactly as ph_ppo_train would issue them -- results are bit-identical -- but the gradient launches of
 * the learners are chained round-robin with events, so that one learner's small reduce / Adam launches overlap the next
 * learner's device-filling gradient launch instead of all learners alternating between "everyone computes gradients" and
 * "everyone reduces".  Streams may be captured into one hipGraph (fork / join is the caller's). */
#define PH_MAX_TRAIN_CALLS 8
typedef struct ph_train_call {
  ph_ctx *ctx;
  const ph_spec *spec;
  const ph_opt_state *opt;
  const ph_rollout *rb;
  const ph_ppo_hyper *hyper;
  int n_epochs, batch_size;
  const int *perms;
  unsigned long long perm_seed;
  float *stats;
  int gemm_mode;
} ph_train_call;
int ph_ppo_train_multi(const ph_train_call *calls, int n_calls);

/* gradient of ONE minibatch (indices given, (nb) int32 env-major) without touching the optimizer state:
 * grad_out (P) = d loss / d params before clipping; stats_out (PH_NSTAT) as above.  For parity tests. */
int ph_ppo_minibatch_grad(ph_ctx *ctx, const ph_spec *spec, const float *params, const ph_rollout *rb,
                          const ph_ppo_hyper *hyper /* host */, const int *indices, int nb, float *grad_out,
                          float *stats_out, int gemm_mode);

/* ---- ADAP: PPO.train() with the context term ("ADAP loss terms as fused-update variants") ------------
 * ADAP.train() <- pantheonrl/algos/adap/adap_learn.py:229-371 is PPO.train() whose minibatch loss gains
 *     context_loss_coeff * get_context_kl_loss(...)                         adap_learn.py:313-320, adap/util.py:97-131
 * The rollout rows carry the context in their LAST context_size components (adap_learn.py:448-452, agent.py:117-121) and
 * AdapPolicy is the MlpPolicy over features ++ context (adap/policies.py:104-119, 136-146), so `spec->obs` is the Box of
 * length (environment observation + context_size).  Per minibatch: min(num_state_samples, nb) states of the minibatch
 * (th.randperm(B)[:num_state_samples]) are re-evaluated under num_context_samples sampled contexts; the term is the mean
 * over context pairs (a before b, itertools.combinations) of mean_s exp(-KL(pi(.|s,a) || pi(.|s,b))).
 *   state_idx   device (n_minibatches, num_state_samples) int32 positions WITHIN each minibatch (teacher-forced
 *               randperm; only the first min(num_state_samples, nb) of a row are read) or NULL = the head of a keyed
 *               Feistel permutation of [0, nb) drawn in the kernel from (seed, rng epoch, minibatch number)
 *   contexts    device (n_minibatches, num_context_samples, context_size) f32 (teacher-forced sampler) or NULL = drawn in
 *               the kernel from a Philox stream keyed the same way, through `sampler` (adap/util.py:42-77)
 *   context_loss, used_state_idx, used_contexts   device outputs with the shapes above ((n_minibatches) for the loss) or
 *               NULL: the raw term and the samples each minibatch actually used
 * n_minibatches = n_epochs * ceil(T*E / batch_size) for ph_adap_train, 1 for ph_adap_minibatch_grad.
 * stats[.][5] (loss) includes the term; the gradient norm, the clip and the Adam step see the gradient of the whole loss.
 * Box observations and categorical action families only; 2 <= num_context_samples <= 16. */
#define PH_CTX_L2 0              /* "l2": uniform in [-1,1)^n scaled to unit length      util.py:42-51 */
#define PH_CTX_UNIT_SQUARE 1     /* "unit_square": uniform in [-1,1)^n                   util.py:54-59 */
#define PH_CTX_POSITIVE_SQUARE 2 /* "positive_square": uniform in [0,1)^n                util.py:62-67 */
#define PH_CTX_CATEGORICAL 3     /* "categorical": one-hot                               util.py:70-77 */
#define PH_CTX_NATURAL_NUMBERS 4 /* "natural_numbers": ONE integer in [0, ctx_size); ctx_size must be 1   util.py:80-89 */
typedef struct ph_adap_loss {     /* defaults: ADAP.__init__ adap_learn.py:111-116 */
  int context_size;               /* 3    */
  int num_context_samples;        /* 5    */
  int num_state_samples;          /* 32   */
  int sampler;                    /* PH_CTX_L2 */
  float context_loss_coeff;       /* 0.1  */
  const int *state_idx;
  const float *contexts;
  unsigned long long seed;
  float *context_loss;
  int *used_state_idx;
  float *used_contexts;
} ph_adap_loss;
int ph_adap_train(ph_ctx *ctx, const ph_spec *spec, const ph_opt_state *opt, const ph_rollout *rb,
                  const ph_ppo_hyper *hyper /* host */, int n_epochs, int batch_size, const int *perms,
                  unsigned long long perm_seed, float *stats, int gemm_mode, const ph_adap_loss *adap /* host */);
/* ph_ppo_minibatch_grad with the context term: grad_out = d (ppo loss + coeff * context loss) / d params */
int ph_adap_minibatch_grad(ph_ctx *ctx, const ph_spec *spec, const float *params, const ph_rollout *rb,
                           const ph_ppo_hyper *hyper /* host */, const int *indices, int nb, float *grad_out,
                           float *stats_out, int gemm_mode, const ph_adap_loss *adap /* host */);

/* ---- AdapPolicyMult (pantheonrl/algos/adap/policies.py:136-283): AdapPolicy with MultModel as its extractor -----------------
 * Rows are features ++ context like AdapPolicy's (adap_learn.py:448-452); per net (policies.py:239-264)
 *   x = tanh(W1 o + b1)  (o = the row without its context),  x_a = tanh(Ws x + bs) viewed as (64, C),
 *   latent = tanh(W2 (x + x_a @ ctx) + b2),  then action_net / value_net.
 * spec: obs = Box(features + context_size) with features <= 64, context_size <= 4; act = Discrete(<= 8).
 * Parameter vector (input-major, like ph_layout): pi {W1 [Fo][64], b1, Ws [64][64 C] (column j C + c), bs, W2 [64][64], b2},
 * vf {the same}, act_W [64][L], act_b, val_W [64], val_b.  The entry points mirror ph_policy_forward / ph_ppo_minibatch_grad /
 * ph_adap_train (same argument meaning; `adap` may be NULL in the gradient call = PPO loss only); the network runs as a chain of
 * small launches over dense intermediates (csrc/ph_adapmult.hip), exact float32. */
typedef struct ph_adapmult_layout {
  int Fo, C, L, P;
  int pi_W1, pi_b1, pi_Ws, pi_bs, pi_W2, pi_b2;
  int vf_W1, vf_b1, vf_Ws, vf_bs, vf_W2, vf_b2;
  int act_W, act_b, val_W, val_b;
} ph_adapmult_layout;
int ph_adapmult_layout_of(const ph_spec *spec, int context_size, ph_adapmult_layout *out /* host */);
int ph_adapmult_forward(ph_ctx *ctx, const ph_spec *spec, int context_size, const float *params, const float *obs, int n,
                        const unsigned char *action_mask, const float *uniforms, const float *given_actions,
                        unsigned long long seed, unsigned long long counter, int deterministic, int *actions_i32,
                        float *actions_f32, float *values, float *log_probs, float *entropy, float *logits,
                        const ph_rollout *rb, int pos, const float *episode_start_in);
int ph_adapmult_minibatch_grad(ph_ctx *ctx, const ph_spec *spec, int context_size, const float *params, const ph_rollout *rb,
                               const ph_ppo_hyper *hyper /* host */, const int *indices, int nb, float *grad_out,
                               float *stats_out, const ph_adap_loss *adap /* host, or NULL */);
int ph_adapmult_train(ph_ctx *ctx, const ph_spec *spec, int context_size, const ph_opt_state *opt, const ph_rollout *rb,
                      const ph_ppo_hyper *hyper /* host */, int n_epochs, int batch_size, const int *perms,
                      unsigned long long perm_seed, float *stats, const ph_adap_loss *adap /* host */);

/* Measurement hook for bench.py's roofline: enqueue ONLY the ppo_grad kernel (the dominant kernel of PPO.train) `reps`
 * times between two HIP events on the ctx stream; *avg_ms_out = mean launch duration.  Launch i takes minibatch
 * i mod ceil(T*E / batch) of one in-kernel permutation of the buffer (size min(batch_size, T*E)), as the launches of an epoch do:
 * consecutive launches read DIFFERENT rows, so the figure does not flatter the kernel with rows the previous launch left in
 * the L2 (with PH_BENCH_GRAD_SAME_ROWS=1 every launch repeats the first minibatch -- the round-1..3 behaviour).
 * Optimizer state is not touched.  Synchronises. */
int ph_bench_ppo_grad(ph_ctx *ctx, const ph_spec *spec, const float *params, const ph_rollout *rb,
                      const ph_ppo_hyper *hyper /* host */, int batch_size, int reps, int gemm_mode,
                      float *avg_ms_out /* host */);
/* same for the GAE kernel (mode as in ph_gae); advantages/returns are overwritten with the same values each rep */
int ph_bench_gae(ph_ctx *ctx, const ph_rollout *rb, const float *last_values, const float *dones, double gamma,
                 double gae_lambda, int mode, int reps, float *avg_ms_out /* host */);

/* Measurement hook for bench.py's HBM-side roofline lines (K1 buffer write, K3 minibatch gather, K6 reduction):
 * the kernels of one PPO.train() call OTHER than the gradient kernel, each enqueued `reps` times between two HIP events on the ctx
 * stream; us_out[slot] = mean launch duration in microseconds (0 = the kernel does not run for this spec).  Runs the preparation of
 * a train() call with the in-kernel permutation, one gradient launch (so that the slabs hold a real minibatch), then the timed
 * launches; `opt` must be a SCRATCH copy of the optimizer state -- the reduction / Adam launches advance it.  Synchronises. */
enum {
  PH_BENCH_WEIGHT_IMAGE = 0, /* weight_image_kernel (split gradient kernel only) */
  PH_BENCH_OBS_PLANES = 1,   /* obs_planes_kernel: the call's observation planes + per-row scalar table (split only) */
  PH_BENCH_ADV_STATS = 2,    /* adv_stats_kernel + adv_finalize_kernel: all n_epochs x n_minibatches of the call */
  PH_BENCH_REDUCE = 3,       /* ppo_reduce_kernel, the shape used beside another learner's gradient launch */
  PH_BENCH_REDUCE_WIDE = 4,  /* ppo_reduce_kernel, the 16-byte-load shape of a learner alone on its device */
  PH_BENCH_ADAM = 5,         /* ppo_adam_kernel */
  PH_BENCH_STEP_FUSED = 6,   /* ppo_step_kernel (reduce + clip + Adam as one launch), 0 if its grid is not resident at once */
  PH_BENCH_BUFFER_ADD = 7,   /* buffer_add_kernel: RolloutBuffer.add of one step (row 1 <- row 0) */
  PH_BENCH_SLAB_FLOATS = 8,  /* not a time: floats of gradient slabs one minibatch writes (and the reduction reads) */
  PH_BENCH_NKERN = 9
};
int ph_bench_train_kernels(ph_ctx *ctx, const ph_spec *spec, const ph_opt_state *scratch_opt, const ph_rollout *rb,
                           const ph_ppo_hyper *hyper /* host */, int n_epochs, int batch_size, int reps, int gemm_mode,
                           float *us_out /* host, PH_BENCH_NKERN */);

/* Host-side evaluation of the keyed Feistel permutation ph_ppo_train uses when perms == NULL: writes
 * out[i] = perm_epoch(start + i) for i < count (env-major indices in [0, n)).  Pure CPU, needs no device;
 * the kernels run the identical integer code, so this is the bit-exact statement of the minibatch order. */
int ph_feistel_indices(int n, unsigned long long perm_seed, int epoch, int start, int count, int *out /* host */);

/* ---- ModularAlgorithm / ModularPolicy (pantheonrl/algos/modular) -----------------------------------------
 * ModularPolicy (modular/policies.py:243-395): the ordinary MlpPolicy ("main") plus one module per partner -- a 64-64 policy
 * tower and a 64-64 value tower that BOTH read the main policy latent (policies.py:254,281), an action head and a value head.
 * Logits = main + partner logits (the partner's alone with `nomain`, policies.py:325-328), value = main + partner value
 * (policies.py:286).  `baseline` shares one module between all partners (policies.py:255-257).
 * Parameter vector: the main network in ph_layout order (P_main floats), then module m at P_main + m * P_module in the ph_layout
 * order of a (Box(64), same action space) network.  Single Discrete head of <= 8 logits, <= 64 features.
 *   ph_modular_layout  : both layouts and the total length.
 *   ph_modular_forward : ModularPolicy.forward / evaluate_actions / get_action_logits_from_obs for partner `partner_idx`
 *       (policies.py:271-288,364-395): arguments as ph_policy_forward, plus the two logit vectors the marginal regulariser reads.
 *   ph_modular_train   : ModularAlgorithm.train (modular/learn.py:221-351): partner by partner over that partner's rollout buffer
 *       (rbs[k], learn.py:134-144), n_epochs passes each; every minibatch is the PPO loss on the composed heads with ALWAYS
 *       normalised advantages (learn.py:260-261) + marginal_reg_coef * marginal regularisation loss (learn.py:298-318), one
 *       clip_grad_norm_ over the main network and every module, one Adam step (torch's per-parameter step counts: mod_first);
 *       the target-KL test runs after a whole epoch on the mean of its KLs (learn.py:320-334) and ends that partner's epochs.
 *       stats: (num_partners * n_epochs * n_minibatches, PH_NSTAT) = policy_loss, value_loss, entropy_loss, clip_fraction,
 *       approx_kl (mean(old_log_prob - log_prob), learn.py:327), loss, grad_norm, marginal_reg; rows of skipped minibatches 0.
 *   ph_modular_minibatch_grad : the gradient of one minibatch's loss w.r.t. every parameter (tests). */
typedef struct ph_modular {
  int num_partners;
  int n_modules;               /* distinct modules: num_partners, or 1 with `baseline` */
  int module_of[PH_MOD_MAX];   /* partner -> module */
  int nomain;
} ph_modular;
int ph_modular_layout(const ph_spec *spec /* host */, const ph_modular *mod /* host */, ph_layout *main_out /* host */,
                      ph_layout *module_out /* host */, int *p_total_out /* host */);
int ph_modular_forward(ph_ctx *ctx, const ph_spec *spec, const ph_modular *mod, const float *params, int partner_idx,
                       const float *obs, int n, const unsigned char *action_mask, const float *uniforms,
                       const float *given_actions, unsigned long long seed, unsigned long long counter, int deterministic,
                       int *actions_i32, float *actions_f32, float *values, float *log_probs, float *entropy,
                       float *logits_main /* (n, L) or NULL */, float *logits_partner /* (n, L) or NULL */,
                       const ph_rollout *rb, int pos, const float *episode_start_in, const float *pending_reward,
                       int gemm_mode);
int ph_modular_train(ph_ctx *ctx, const ph_spec *spec, const ph_modular *mod, const ph_opt_state *opt,
                     int *mod_first /* device (n_modules): -1 until the module's value side first received a gradient */,
                     const ph_rollout *rbs /* host, [num_partners] */, const ph_ppo_hyper *hyper, int n_epochs, int batch_size,
                     const int *perms /* device (num_partners, n_epochs, T*E) or NULL */, unsigned long long perm_seed,
                     float *stats, float marginal_reg_coef, int gemm_mode);
int ph_modular_minibatch_grad(ph_ctx *ctx, const ph_spec *spec, const ph_modular *mod, const float *params, int partner_idx,
                              const ph_rollout *rb, const ph_ppo_hyper *hyper, const int *indices /* device (nb) */, int nb,
                              float marginal_reg_coef, float *grad_out /* device (P_total) */,
                              float *stats_out /* device (PH_NSTAT) or NULL */, int gemm_mode);

/* ---- behavioural cloning --------------------------------------------------------------------------
 * BC <- pantheonrl/algos/bc.py:180-366 on FeedForward32Policy <- pantheonrl/common/util.py:114-123: SB3 ActorCriticPolicy with
 * net_arch = [32, 32], ONE shared tanh trunk feeding action_net and value_net.  Parameter vector (float32, input-major):
 *   W1[F][32] b1[32] W2[32][32] b2[32] act_W[32][L] act_b[L] val_W[32] val_b[1]
 *  ph_bc_forward: policy.forward / evaluate_actions of that architecture (argument meaning as ph_policy_forward).
 *  ph_bc_train:   BC.train (bc.py:305-353): for every epoch, for every consecutive slice of `batch_size` entries of that epoch's
 *      visiting order (the last may be short; DataLoader(shuffle=True) draws one permutation per epoch -- `order` teacher-forces
 *      it): loss = -mean(log_prob) - ent_weight * mean(entropy) + l2_weight * sum(w^2) / 2 (bc.py:291-303), backward, one
 *      torch.optim.Adam step (no gradient clipping).  max_batches > 0 stops after that many minibatches (`n_batches` mode).
 *      The whole chain runs inside ONE launch of one persistent workgroup (parameters resident in LDS).
 *      stats (minibatches, PH_BC_NSTAT) or NULL: {neglogp, entropy, ent_loss, prob_true_act, l2_norm, l2_loss, loss, rows}
 *      (bc.py:305-313).  Limit: parameters, their gradient and one 32-row tile must fit the CU's 160 KiB of LDS (F <= ~500). */
#define PH_BC_HIDDEN 32
#define PH_BC_NSTAT 8
typedef struct ph_bc_layout {
  int D, F, A, L, P;
  int W1, b1, W2, b2, act_W, act_b, val_W, val_b; /* offsets */
} ph_bc_layout;
typedef struct ph_bc_hyper { /* BC defaults: bc.py:189-191; torch.optim.Adam defaults */
  float learning_rate; /* 1e-3  */
  float adam_beta1, adam_beta2, adam_eps; /* 0.9, 0.999, 1e-8 */
  float ent_weight;    /* 1e-3  */
  float l2_weight;     /* 0.0   */
} ph_bc_hyper;
int ph_bc_layout_of(const ph_spec *spec /* host */, ph_bc_layout *out /* host */);
int ph_bc_forward(ph_ctx *ctx, const ph_spec *spec, const float *params, const float *obs, int n,
                  const unsigned char *action_mask, const float *uniforms, const float *given_actions,
                  unsigned long long seed, unsigned long long counter, int deterministic, int *actions_i32, float *values,
                  float *log_probs, float *entropy, float *logits);
int ph_bc_train(ph_ctx *ctx, const ph_spec *spec, const ph_opt_state *opt, const float *obs /* (N,D) */,
                const float *acts /* (N,A) f32 */, const int *order /* (n_epochs, N) int32 */, int N, int batch_size,
                int n_epochs, int max_batches, const ph_bc_hyper *hyper /* host */, float *stats);

/* ---- owning handle: one agent = its rollout buffer, weights and Adam state on the device ----------------------------------
 *   The pointer-level entry points above take device memory owned by the caller (the Python host uses torch
 * allocations).  This layer is the same path for a binder that has NO device runtime of its own: the handle owns every device
 * allocation, every array crossing the boundary is a HOST array (copied in / out), every call is complete on return.
 * Mapping to the reference's call sites:
 *   ph_agent_create / destroy        <- PPO.__init__(env, n_steps, gamma, gae_lambda, seed, device)    trainer.py:108-126,196-203
 *   ph_agent_set/get_params          <- policy.load_state_dict / state_dict (flat layout: ph_layout)    trainer.py:140-157,419-432
 *   ph_agent_set/get_optimizer       <- policy.optimizer.state_dict()
 *   ph_agent_buffer_reset            <- rollout_buffer.reset()                                          agents.py:157
 *   ph_agent_act(record=1)           <- policy.forward(obs) + rollout_buffer.add(..., reward = 0, ...)  util.py:63-81, agents.py:162-179
 *   ph_agent_act(record=0)           <- policy.forward(obs) alone (StaticPolicyAgent)                   agents.py:54-79
 *   ph_agent_add_reward              <- rollout_buffer.rewards[pos - 1] += reward                       agents.py:198
 *   ph_agent_gae                     <- rollout_buffer.compute_returns_and_advantage(last_values, dones) agents.py:127-130
 *   ph_agent_train                   <- model.train()                                                   agents.py:155
 *   ph_agent_export / import_buffer  <- the buffer's arrays as SB3 lays them out, (T, E, ...) float32
 * Errors: non-zero return, message from ph_agent_last_error() (thread-local). */
typedef struct ph_agent ph_agent;
const char *ph_agent_last_error(void);
int ph_agent_create(int device, const ph_spec *spec /* host */, int n_envs, int n_steps, double gamma, double gae_lambda,
                    unsigned long long seed, ph_agent **out /* host */);
int ph_agent_destroy(ph_agent *agent);
int ph_agent_layout(const ph_agent *agent, ph_layout *out /* host */);
int ph_agent_set_params(ph_agent *agent, const float *params /* host (P) */);
int ph_agent_get_params(ph_agent *agent, float *params_out /* host (P) */);
int ph_agent_set_optimizer(ph_agent *agent, const float *adam_m /* host (P) */, const float *adam_v /* host (P) */, int step);
int ph_agent_get_optimizer(ph_agent *agent, float *adam_m_out, float *adam_v_out, int *step_out /* host, any may be NULL */);
int ph_agent_buffer_reset(ph_agent *agent);
int ph_agent_pos(const ph_agent *agent, int *pos_out /* host */);
/* obs (E,D) host; action_mask (E,L) u8 or NULL; uniforms (E,A) or NULL (NULL = Philox keyed by the handle's seed and call
 * counter); record != 0 writes the transition at the buffer's write row with episode_start (E) and advances it.
 * Outputs (host, any may be NULL): actions (E,A) int32, values (E), log_probs (E). */
int ph_agent_act(ph_agent *agent, const float *obs, const unsigned char *action_mask, const float *uniforms,
                 int deterministic, int record, const float *episode_start, int *actions_out, float *values_out,
                 float *log_probs_out);
int ph_agent_add_reward(ph_agent *agent, const float *reward /* host (E) */, const unsigned char *env_mask /* host (E) or NULL */);
int ph_agent_gae(ph_agent *agent, const float *last_values /* host (E) */, const float *dones /* host (E) */, int mode);
/* perms (n_epochs, T*E) host int32 or NULL (keyed Feistel order from perm_seed); stats_out host
 * (n_epochs * ceil(T*E / batch_size), PH_NSTAT) or NULL */
int ph_agent_train(ph_agent *agent, const ph_ppo_hyper *hyper /* host */, int n_epochs, int batch_size, const int *perms,
                   unsigned long long perm_seed, float *stats_out);
int ph_agent_export_buffer(ph_agent *agent, float *observations, float *actions, float *rewards, float *episode_starts,
                           float *values, float *log_probs, float *advantages, float *returns /* host, any may be NULL */);
int ph_agent_import_buffer(ph_agent *agent, const float *observations, const float *actions, const float *rewards,
                           const float *episode_starts, const float *values, const float *log_probs, const float *advantages,
                           const float *returns /* host, any may be NULL */, int pos);

#ifdef __cplusplus
}
#endif
#endif /* PANTHEON_HIP_H */

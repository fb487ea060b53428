// C ABI of libpantheon_hip.so (see include/pantheon_hip.h for the contract and the reference call sites).
// Host side only: argument checking, workspace management, kernel launches on the context stream.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "ph_launch.h"

namespace {

thread_local std::string g_err;

int fail(const std::string& m) {
  g_err = m;
  return 1;
}
int fail_hip(const char* what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return 2;
}
#define PH_HIP(call)                                   \
  do {                                                 \
    hipError_t _e = (call);                            \
    if (_e != hipSuccess) return fail_hip(#call, _e);  \
  } while (0)

struct SpecCache {
  ph_spec spec;
  int* obs_off = nullptr;  // device prefix sums
  int* act_off = nullptr;
  int* slab_map = nullptr; // device: slab position -> parameter index for register-order gradient slabs, or null
  int* slab_map_split = nullptr;  // device: the same table for the split-bf16 gradient kernel, or null
  int* wimage_map = nullptr;      // device: parameter -> weight fragment image elements of the split kernel (ph_split.h), or null
  int split_kind = 0;             // 1 ppo_grad_split_kernel, 2 ppo_grad_split_oh_kernel, 0 neither takes the spec
  int slab_len_split = 0;         // floats per slab in that kernel's order
  int wimage_elems = 0;           // bf16 elements of its weight image
  int id = 0;                     // 1, 2, ..: position in the context's cache, never reused
};

}  // namespace

struct ph_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // workspace
  float* slabs = nullptr;
  size_t slabs_cap = 0;
  float* statpart = nullptr;
  size_t statpart_cap = 0;
  float* grad = nullptr;
  size_t grad_cap = 0;
  float* blocksq = nullptr;
  size_t blocksq_cap = 0;
  float* advstats = nullptr;
  size_t advstats_cap = 0;
  // ph_policy_act_host: pinned, coherent host staging the kernel accesses directly (host view, device view of the same memory)
  float* act_stage_host = nullptr;
  float* act_stage_dev = nullptr;
  size_t act_stage_cap = 0;   // floats
  // the two completion words of a one-tile ph_policy_act_host launch (coherent host memory, both views), the sequence number
  // the next launch stores, and -- while ph_policy_act_host is inside ph_policy_forward -- the words that launch should signal
  unsigned int* act_done_host = nullptr;
  unsigned int* act_done_dev = nullptr;
  unsigned int act_seq = 0;
  unsigned int* fwd_host_done = nullptr;
  int joint_reward_rule = 0;     // ph_ctx_set_joint_reward_rule: how a step's joint action enters the agents' rewards
  int wimage_zeroed_for = 0;                // spec-cache id the weight image's unbacked elements were last zeroed for (ids are never reused)
  size_t wimage_cap = 0;                    // elements allocated
  bool exclusive = false;             // ph_set_exclusive_device: nothing else runs on the device beside this context's launches
  unsigned short* wimage = nullptr;   // split gradient kernel: pre-split weight fragments of the policy being trained (ph_split.h)
  double* advpart = nullptr;     // per-segment partial sums of the advantage statistics
  size_t advpart_cap = 0;
  int* perm_idx = nullptr;   // (n_epochs, N) minibatch order of the current train() call, written by adv_stats
  size_t perm_idx_cap = 0;
  int* perm_phys = nullptr;  // the same order as physical buffer rows (t * E + e)
  size_t perm_phys_cap = 0;
  // the split gradient kernel's pack of the current train() call (ph_split.h): observation rows as bf16 planes, per-row scalars
  // of either net in minibatch order
  uint4* ximg = nullptr;     // [rows + 1][24]
  size_t ximg_cap = 0;       // in uint4
  uint4* rec_pi = nullptr;   // (n_epochs, N)
  uint4* rec_vf = nullptr;
  size_t rec_cap = 0;        // elements of each
  uint4* rowrec = nullptr;   // [rows][2] the per-row scalars packed by physical row (input of the record pass)
  size_t rowrec_cap = 0;
  // the fused reduce + clip + Adam launch of an exclusive learner (ppo_step_kernel): one stamped word per block (+ 1), the
  // launch generation, the count of sweeps that timed out
  unsigned long long* step_words = nullptr;
  size_t step_words_cap = 0;
  unsigned int* step_gen = nullptr;   // [2]: generation, sweep errors
  float* am_buf = nullptr;       // AdapPolicyMult: the dense intermediates of one net (ph_adapmult.hip), one block
  size_t am_buf_cap = 0;         // floats
  float* adap_extra = nullptr;   // [workgroups][policy-side parameters] gradient slabs of ADAP's context term
  size_t adap_extra_cap = 0;
  float* adap_loss = nullptr;    // [workgroups] partial sums of the raw term
  size_t adap_loss_cap = 0;
  float* scalars = nullptr;  // [4]
  int* stop_flag = nullptr;  // [1]
  std::vector<SpecCache> specs;
  std::vector<hipGraphExec_t> graphs;
  bool capturing = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  ph_p2p* p2p_dev = nullptr;      // device copy of the peer-to-peer descriptor used by the fused step launches
  ph_p2p p2p_host;                // what p2p_dev currently holds
  bool p2p_valid = false;
  void* comm = nullptr;           // ncclComm_t of the agent-per-GPU exchange (ph_comm_init), or null
  int comm_world = 1, comm_rank = 0;
  hipEvent_t ev_grad = nullptr;   // ph_ppo_train_multi: "this learner's latest gradient launch"
  int num_cu = 256;
  int exchange_blocks_per_cu = 0;           // occupancy answer for the exchange rollout kernel (0 = not asked yet)
  unsigned long long* rng_epoch = nullptr;  // caller-owned device word
  long long* prof = nullptr;                // caller-owned debug stamp buffer
  // ModularAlgorithm workspace (ph_modular_*): activations and head gradients in minibatch order, the towers' slab maps
  struct ModWs {
    float* act = nullptr;        // one allocation carved into the arrays below
    size_t act_cap = 0;
    int* maps = nullptr;         // [n_modules][n_slots][RS_NET] slab position -> parameter index, one set per trained module
    ph_spec spec;
    ph_modular mod;
    bool maps_valid = false;
    float* kl_sum = nullptr;     // [1]
    int* scratch = nullptr;      // [2 + PH_MOD_MAX] throw-away optimizer counters of ph_modular_minibatch_grad's statistics
  } mw;
};

namespace {

// Every entry point that takes a context runs with the context's device current and restores the caller's device on
// return: a process that holds models on several GPUs (or whose torch current device differs) would otherwise allocate
// workspaces and launch on whichever device happened to be current (ph_ctx_create used to leave its device selected).
struct DevGuard {
  int prev = -1;
  bool switched = false;
  explicit DevGuard(const ph_ctx* ctx) {
    if (!ctx) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != ctx->device) switched = hipSetDevice(ctx->device) == hipSuccess;
  }
  ~DevGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DevGuard(const DevGuard&) = delete;
  DevGuard& operator=(const DevGuard&) = delete;
};

template <typename T>
int ensure(T*& p, size_t& cap, size_t n) {
  if (n <= cap) return 0;
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
  hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
  if (e != hipSuccess) return fail_hip("hipMalloc(workspace)", e);
  cap = n;
  return 0;
}

int check_space(const ph_space& s, const char* name) {
  if (s.kind != PH_SPACE_BOX && s.kind != PH_SPACE_DISCRETE) return fail(std::string(name) + ": unknown space kind");
  if (s.n <= 0) return fail(std::string(name) + ": empty space");
  if (s.kind == PH_SPACE_DISCRETE) {
    if (s.n > PH_MAX_COMP) return fail(std::string(name) + ": too many components");
    for (int i = 0; i < s.n; ++i)
      if (s.nvec[i] <= 0) return fail(std::string(name) + ": nvec entries must be positive");
  }
  return 0;
}

int layout_of(const ph_spec* spec, ph_layout* o) {
  if (!spec || !o) return fail("null spec/layout");
  if (check_space(spec->obs, "observation space")) return 1;
  if (check_space(spec->act, "action space")) return 1;
  const bool gauss = spec->act.kind == PH_SPACE_BOX;   // Box actions: DiagGaussian head, A means + log_std[A] (general kernels only)
  if (gauss && spec->act.n > PH_MAX_BOX_ACT) return fail("Box action spaces: at most PH_MAX_BOX_ACT dimensions");
  o->D = spec->obs.n;
  o->F = 0;
  if (spec->obs.kind == PH_SPACE_BOX) o->F = spec->obs.n;
  else
    for (int i = 0; i < spec->obs.n; ++i) o->F += spec->obs.nvec[i];
  o->A = spec->act.n;
  o->L = 0;
  if (gauss) o->L = spec->act.n;   // the head's outputs are the A means
  else
    for (int i = 0; i < spec->act.n; ++i) o->L += spec->act.nvec[i];
  if (o->L > PH_MAX_LOGITS) return fail("too many logits (PH_MAX_LOGITS)");
  const int H = PH_HIDDEN;
  int off = 0;
  o->pi_W1 = off; off += o->F * H;
  o->pi_b1 = off; off += H;
  o->pi_W2 = off; off += H * H;
  o->pi_b2 = off; off += H;
  o->vf_W1 = off; off += o->F * H;
  o->vf_b1 = off; off += H;
  o->vf_W2 = off; off += H * H;
  o->vf_b2 = off; off += H;
  o->act_W = off; off += H * o->L;
  o->act_b = off; off += o->L;
  o->val_W = off; off += H;
  o->val_b = off; off += 1;
  if (gauss) off += o->A;   // log_std[A] (SB3 DiagGaussianDistribution.proba_distribution_net's nn.Parameter), behind val_b
  o->P = off;
  return 0;
}

// device-resident prefix sums of the nvec arrays, cached per distinct spec
// box_act_ok: the caller's kernels know the DiagGaussian head (the general forward / gradient kernels: ph_policy_forward,
// ph_ppo_minibatch_grad, ph_ppo_train); every other entry point refuses a Box action space here
int resolve(ph_ctx* ctx, const ph_spec* spec, ph::NetDims* nd, bool box_act_ok = false) {
  if (layout_of(spec, &nd->lay)) return 1;
  const bool gauss = spec->act.kind == PH_SPACE_BOX;
  if (gauss && !box_act_ok)
    return fail("Box (continuous) action spaces run on ph_policy_forward / ph_ppo_minibatch_grad / ph_ppo_train only "
                "(the DiagGaussian head lives in the general kernels)");
  SpecCache* hit = nullptr;
  for (auto& c : ctx->specs)
    if (std::memcmp(&c.spec, spec, sizeof(ph_spec)) == 0) { hit = &c; break; }
  if (!hit) {
    if (ctx->capturing) return fail("first use of a spec inside graph capture: call it once outside capture first");
    SpecCache c;
    std::memcpy(&c.spec, spec, sizeof(ph_spec));
    std::vector<int> ao(spec->act.n + 1, 0);
    for (int i = 0; i < spec->act.n; ++i) ao[i + 1] = ao[i] + (gauss ? 1 : spec->act.nvec[i]);
    PH_HIP(hipMalloc((void**)&c.act_off, ao.size() * sizeof(int)));
    PH_HIP(hipMemcpy(c.act_off, ao.data(), ao.size() * sizeof(int), hipMemcpyHostToDevice));
    if (spec->obs.kind == PH_SPACE_DISCRETE) {
      std::vector<int> oo(spec->obs.n + 1, 0);
      for (int i = 0; i < spec->obs.n; ++i) oo[i + 1] = oo[i] + spec->obs.nvec[i];
      PH_HIP(hipMalloc((void**)&c.obs_off, oo.size() * sizeof(int)));
      PH_HIP(hipMemcpy(c.obs_off, oo.data(), oo.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    {   // does this spec run on the gradient kernel with register-order slabs?  then its slab table goes to the device once
      ph::NetDims probe;
      probe.lay = nd->lay;
      probe.nchunk = (nd->lay.F + PH_HIDDEN - 1) / PH_HIDDEN;
      probe.A = nd->lay.A;
      probe.L = nd->lay.L;
      probe.F = nd->lay.F;
      probe.obs_kind = spec->obs.kind;
      probe.gauss = gauss ? 1 : 0;
      if (ph::grad_uses_reg_slabs(probe)) {
        std::vector<int> m(2 * ph::RS_NET);
        ph::grad_slab_map(nd->lay, m.data(), ph::grad_fast_fold(probe));
        PH_HIP(hipMalloc((void**)&c.slab_map, m.size() * sizeof(int)));
        PH_HIP(hipMemcpy(c.slab_map, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice));
        if (ph::grad_split_eligible(probe)) {
          ph::grad_slab_map_split(nd->lay, m.data(), ph::grad_fast_fold(probe));
          PH_HIP(hipMalloc((void**)&c.slab_map_split, m.size() * sizeof(int)));
          PH_HIP(hipMemcpy(c.slab_map_split, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice));
          std::vector<int> wm(2 * (size_t)nd->lay.P);
          ph::grad_weight_image_map(nd->lay, ph::grad_fast_fold(probe), wm.data());
          PH_HIP(hipMalloc((void**)&c.wimage_map, wm.size() * sizeof(int)));
          PH_HIP(hipMemcpy(c.wimage_map, wm.data(), wm.size() * sizeof(int), hipMemcpyHostToDevice));
          c.split_kind = 1;
          c.slab_len_split = 2 * ph::RS_NET;
          c.wimage_elems = ph::WIMG_ELEMS;
        }
      } else {
        probe.D = nd->lay.D;
        if (ph::grad_split_oh_eligible(probe)) {   // one-hot observations: the wide split kernel's tables
          std::vector<int> m((size_t)ph::grad_split_oh_slab_len(probe));
          ph::grad_slab_map_split_oh(nd->lay, probe.nchunk, m.data());
          PH_HIP(hipMalloc((void**)&c.slab_map_split, m.size() * sizeof(int)));
          PH_HIP(hipMemcpy(c.slab_map_split, m.data(), m.size() * sizeof(int), hipMemcpyHostToDevice));
          std::vector<int> wm(2 * (size_t)nd->lay.P);
          ph::grad_weight_image_map_oh(nd->lay, probe.nchunk, wm.data());
          PH_HIP(hipMalloc((void**)&c.wimage_map, wm.size() * sizeof(int)));
          PH_HIP(hipMemcpy(c.wimage_map, wm.data(), wm.size() * sizeof(int), hipMemcpyHostToDevice));
          c.split_kind = 2;
          c.slab_len_split = ph::grad_split_oh_slab_len(probe);
          c.wimage_elems = ph::grad_split_oh_wimage_elems(probe);
        }
      }
    }
    c.id = (int)ctx->specs.size() + 1;
    ctx->specs.push_back(c);
    hit = &ctx->specs.back();
  }
  nd->slab_map = hit->slab_map;
  nd->slab_map_split = hit->slab_map_split;
  nd->wimage_map = hit->wimage_map;
  nd->split = 0;
  nd->split_kind = hit->split_kind;
  nd->slab_len_split = hit->slab_len_split;
  nd->wimage_elems = hit->wimage_elems;
  nd->spec_id = hit->id;
  nd->obs_kind = spec->obs.kind;
  nd->gauss = gauss ? 1 : 0;
  nd->D = nd->lay.D;
  nd->F = nd->lay.F;
  nd->A = nd->lay.A;
  nd->L = nd->lay.L;
  nd->Lp = ((nd->L + 31) / 32) * 32;
  nd->nchunk = (nd->F + PH_HIDDEN - 1) / PH_HIDDEN;
  nd->head16 = !gauss && spec->act.n <= 4;
  for (int i = 0; i < spec->act.n && i < 4 && !gauss; ++i) nd->head16 = nd->head16 && spec->act.nvec[i] <= 16;
  nd->obs_off = hit->obs_off;
  nd->act_off = hit->act_off;
  return 0;
}

int check_rb(const ph_rollout* rb) {
  if (!rb) return fail("null rollout buffer");
  if (rb->T <= 0 || rb->E <= 0) return fail("rollout buffer: T and E must be positive");
  if (!rb->observations || !rb->actions || !rb->rewards || !rb->episode_starts || !rb->values || !rb->log_probs ||
      !rb->advantages || !rb->returns)
    return fail("rollout buffer: null array");
  if ((long long)rb->T * rb->E >= (1ll << 31)) return fail("rollout buffer: T*E must fit in int32");
  return 0;
}

}  // namespace

extern "C" {

int ph_abi_version(void) { return PH_ABI_VERSION; }
const char* ph_last_error(void) { return g_err.c_str(); }

int ph_device_count(int* n_out) {
  if (!n_out) return fail("null n_out");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *n_out = 0;
    return fail_hip("hipGetDeviceCount", e);
  }
  *n_out = n;
  return 0;
}

int ph_ctx_create(int device, ph_ctx** out) {
  if (!out) return fail("null out");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail("no HIP device visible: the PantheonRL MI355X engine has no CPU fallback");
  if (device < 0 || device >= n) return fail("device index out of range");
  ph_ctx probe;                    // the caller's current device is restored on every return path
  probe.device = device;
  DevGuard dev_guard(&probe);
  hipDeviceProp_t prop;
  PH_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 (MI355X) only");
  ph_ctx* c = new ph_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount;
  if (hipMalloc((void**)&c->scalars, 4 * sizeof(float)) != hipSuccess ||
      hipMalloc((void**)&c->stop_flag, sizeof(int)) != hipSuccess) {
    delete c;
    return fail("hipMalloc(ctx scalars) failed");
  }
  (void)hipMemset(c->scalars, 0, 4 * sizeof(float));
  (void)hipMemset(c->stop_flag, 0, sizeof(int));
  (void)hipEventCreate(&c->ev0);
  (void)hipEventCreate(&c->ev1);
  *out = c;
  return 0;
}

int ph_ctx_destroy(ph_ctx* ctx) {
  DevGuard dev_guard(ctx);
  if (!ctx) return 0;
  (void)hipStreamSynchronize(ctx->stream);
  for (auto g : ctx->graphs)
    if (g) (void)hipGraphExecDestroy(g);
  for (auto& s : ctx->specs) {
    if (s.obs_off) (void)hipFree(s.obs_off);
    if (s.slab_map) (void)hipFree(s.slab_map);
    if (s.slab_map_split) (void)hipFree(s.slab_map_split);
    if (s.wimage_map) (void)hipFree(s.wimage_map);
    if (s.act_off) (void)hipFree(s.act_off);
  }
  void* ptrs[] = {ctx->wimage, ctx->mw.act, ctx->mw.maps, ctx->mw.kl_sum, ctx->mw.scratch, ctx->advpart, ctx->p2p_dev, ctx->slabs, ctx->statpart, ctx->grad, ctx->blocksq, ctx->advstats, ctx->perm_idx, ctx->perm_phys, ctx->ximg, ctx->rec_pi, ctx->rec_vf, ctx->rowrec, ctx->step_words, ctx->step_gen, ctx->scalars, ctx->stop_flag,
                  ctx->adap_extra, ctx->adap_loss, ctx->am_buf};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (ctx->act_stage_host) (void)hipHostFree(ctx->act_stage_host);
  if (ctx->act_done_host) (void)hipHostFree(ctx->act_done_host);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev_grad) (void)hipEventDestroy(ctx->ev_grad);
  (void)ph_comm_destroy(ctx);
  delete ctx;
  return 0;
}

int ph_ctx_set_joint_reward_rule(ph_ctx* ctx, int rule) {
  if (!ctx) return fail("null ctx");
  if (rule != PH_JOINT_MATCH_BONUS && rule != PH_JOINT_RPS) return fail("ph_ctx_set_joint_reward_rule: unknown rule");
  ctx->joint_reward_rule = rule;
  return 0;
}

int ph_ctx_set_stream(ph_ctx* ctx, void* hip_stream) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  ctx->stream = (hipStream_t)hip_stream;
  return 0;
}

int ph_ctx_sync(ph_ctx* ctx) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int ph_graph_begin(ph_ctx* ctx) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (ctx->capturing) return fail("already capturing");
  if (ctx->stream == nullptr) return fail("graph capture needs a non-default stream (ph_ctx_set_stream)");
  PH_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capturing = true;
  return 0;
}

int ph_graph_end(ph_ctx* ctx, int* graph_id_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !graph_id_out) return fail("null ctx/graph_id_out");
  if (!ctx->capturing) return fail("not capturing");
  ctx->capturing = false;
  hipGraph_t g = nullptr;
  PH_HIP(hipStreamEndCapture(ctx->stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return fail_hip("hipGraphInstantiate", e);
  ctx->graphs.push_back(ge);
  *graph_id_out = (int)ctx->graphs.size() - 1;
  return 0;
}

int ph_graph_launch(ph_ctx* ctx, int graph_id) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (graph_id < 0 || graph_id >= (int)ctx->graphs.size()) return fail("bad graph id");
  PH_HIP(hipGraphLaunch(ctx->graphs[graph_id], ctx->stream));
  return 0;
}

int ph_ctx_set_rng_epoch(ph_ctx* ctx, unsigned long long* epoch_dev) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  ctx->rng_epoch = epoch_dev;
  return 0;
}
int ph_rng_epoch_advance(ph_ctx* ctx) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!ctx->rng_epoch) return fail("ph_rng_epoch_advance: no epoch word attached");
  PH_HIP(ph::launch_epoch_advance(ctx->rng_epoch, ctx->stream));
  return 0;
}

int ph_debug_set_profile_buffer(ph_ctx* ctx, long long* stamps_dev) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  ctx->prof = stamps_dev;
  return 0;
}

int ph_debug_weight_image_mismatches(ph_ctx* ctx, const ph_spec* spec, const float* params, int* mismatches_host) {
  DevGuard dev_guard(ctx);
  if (!ctx || !params || !mismatches_host) return fail("ph_debug_weight_image_mismatches: null argument");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  *mismatches_host = -1;                       // -1: this context holds no image (the split kernel never ran for it)
  if (!ctx->wimage || !nd.wimage_map) return 0;
  int* d = nullptr;
  PH_HIP(hipMalloc((void**)&d, sizeof(int)));
  PH_HIP(hipMemsetAsync(d, 0, sizeof(int), ctx->stream));
  PH_HIP(ph::launch_weight_image_check(params, ctx->wimage, nd.wimage_map, nd.lay.P, d, ctx->stream));
  PH_HIP(hipMemcpyAsync(mismatches_host, d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  (void)hipFree(d);
  return 0;
}

int ph_set_exclusive_device(ph_ctx* ctx, int exclusive) {
  if (!ctx) return fail("null ctx");
  ctx->exclusive = exclusive != 0;
  return 0;
}

int ph_ctx_step_errors(ph_ctx* ctx, unsigned int* count_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !count_out) return fail("ph_ctx_step_errors: null argument");
  *count_out = 0u;
  if (!ctx->step_gen) return 0;   // the fused step never ran on this context
  PH_HIP(hipMemcpyAsync(count_out, ctx->step_gen + 1, sizeof(unsigned int), hipMemcpyDeviceToHost, ctx->stream));
  PH_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

int ph_timer_start(ph_ctx* ctx) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  PH_HIP(hipEventRecord(ctx->ev0, ctx->stream));
  return 0;
}
int ph_timer_stop(ph_ctx* ctx, float* ms_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !ms_out) return fail("null ctx/ms_out");
  PH_HIP(hipEventRecord(ctx->ev1, ctx->stream));
  PH_HIP(hipEventSynchronize(ctx->ev1));
  PH_HIP(hipEventElapsedTime(ms_out, ctx->ev0, ctx->ev1));
  return 0;
}

int ph_layout_of(const ph_spec* spec, ph_layout* out) { return layout_of(spec, out); }

// Host-only: the two tables that tie ppo_grad_split_kernel's register and fragment orders to parameter indices, for inspection
// (tests/test_host_logic.py checks them against the layout without a GPU).  Returns 0 and *eligible = 0 for specs the kernel
// does not take (the tables are then untouched).
int ph_debug_split_tables(const ph_spec* spec, int* slab_map /* 2 * 8960 */, int* image_map /* 2 * P */, int* eligible) {
  if (!spec || !slab_map || !image_map || !eligible) return fail("ph_debug_split_tables: null argument");
  ph_layout lay;
  if (layout_of(spec, &lay)) return 1;
  ph::NetDims probe;
  std::memset(&probe, 0, sizeof(probe));
  probe.lay = lay;
  probe.nchunk = (lay.F + PH_HIDDEN - 1) / PH_HIDDEN;
  probe.A = lay.A;
  probe.L = lay.L;
  probe.F = lay.F;
  probe.obs_kind = spec->obs.kind;
  *eligible = ph::grad_split_eligible(probe) ? 1 : 0;
  if (!*eligible) return 0;
  ph::grad_slab_map_split(lay, slab_map, ph::grad_fast_fold(probe));
  ph::grad_weight_image_map(lay, ph::grad_fast_fold(probe), image_map);
  return 0;
}

// The same for ppo_grad_split_oh_kernel (one-hot observations): *slab_len_out = floats per slab (both nets), *image_elems_out = bf16
// elements of its weight image; slab_map needs *slab_len_out ints (call once with slab_map = image_map = null to learn the sizes).
int ph_debug_split_oh_tables(const ph_spec* spec, int* slab_map, int* image_map /* 2 * P */, int* slab_len_out, int* image_elems_out,
                             int* eligible) {
  if (!spec || !slab_len_out || !image_elems_out || !eligible) return fail("ph_debug_split_oh_tables: null argument");
  ph_layout lay;
  if (layout_of(spec, &lay)) return 1;
  ph::NetDims probe;
  std::memset(&probe, 0, sizeof(probe));
  probe.lay = lay;
  probe.nchunk = (lay.F + PH_HIDDEN - 1) / PH_HIDDEN;
  probe.A = lay.A;
  probe.L = lay.L;
  probe.F = lay.F;
  probe.D = lay.D;
  probe.obs_kind = spec->obs.kind;
  *eligible = ph::grad_split_oh_eligible(probe) ? 1 : 0;
  if (!*eligible) return 0;
  *slab_len_out = ph::grad_split_oh_slab_len(probe);
  *image_elems_out = ph::grad_split_oh_wimage_elems(probe);
  if (slab_map) ph::grad_slab_map_split_oh(lay, probe.nchunk, slab_map);
  if (image_map) ph::grad_weight_image_map_oh(lay, probe.nchunk, image_map);
  return 0;
}

// ---- K1 ----
int ph_buffer_add(ph_ctx* ctx, const ph_spec* spec, const ph_rollout* rb, int pos, const float* obs,
                  const float* actions, const float* episode_start, const float* values, const float* log_probs) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  ph_layout lay;
  if (layout_of(spec, &lay) || check_rb(rb)) return 1;
  if (pos < 0 || pos >= rb->T) return fail("ph_buffer_add: pos out of range (buffer full?)");
  if (!obs || !actions || !episode_start || !values || !log_probs) return fail("ph_buffer_add: null input");
  const size_t E = rb->E, row = (size_t)pos * E;
  PH_HIP(ph::launch_buffer_add(rb->observations + row * lay.D, rb->actions + row * lay.A, rb->rewards + row,
                               rb->episode_starts + row, rb->values + row, rb->log_probs + row, obs, actions,
                               episode_start, values, log_probs, rb->E, lay.D, lay.A, ctx->stream));
  return 0;
}

int ph_buffer_compact_columns(ph_ctx* ctx, const ph_spec* spec, const ph_rollout* src, const ph_rollout* dst,
                              const int* cols, int n) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  ph_layout lay;
  if (layout_of(spec, &lay) || check_rb(src) || check_rb(dst)) return 1;
  if (!cols || n <= 0 || n > src->E) return fail("ph_buffer_compact_columns: bad column list");
  if (dst->T != src->T || dst->E != n) return fail("ph_buffer_compact_columns: destination must be (T, n)");
  PH_HIP(ph::launch_buffer_compact(*src, *dst, cols, n, lay.D, lay.A, ctx->stream));
  return 0;
}

int ph_buffer_add_reward(ph_ctx* ctx, const ph_rollout* rb, int pos, const float* reward,
                         const unsigned char* env_mask) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_rb(rb)) return 1;
  if (pos < 0 || pos >= rb->T) return fail("ph_buffer_add_reward: pos out of range");
  if (!reward) return fail("ph_buffer_add_reward: null reward");
  PH_HIP(ph::launch_reward_add(rb->rewards + (size_t)pos * rb->E, reward, env_mask, rb->E, ctx->stream));
  return 0;
}

int ph_buffer_add_reward_const(ph_ctx* ctx, const ph_rollout* rb, int pos, float reward) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_rb(rb)) return 1;
  if (pos < 0 || pos >= rb->T) return fail("ph_buffer_add_reward_const: pos out of range");
  PH_HIP(ph::launch_reward_add_const(rb->rewards + (size_t)pos * rb->E, reward, rb->E, ctx->stream));
  return 0;
}

int ph_buffer_add_reward_joint(ph_ctx* ctx, const ph_rollout* rb, int pos, const float* base_reward,
                               const int* joint_actions, int n_seats, int seat, const int* partner_seat, float bonus) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_rb(rb)) return 1;
  if (pos < 0 || pos >= rb->T) return fail("ph_buffer_add_reward_joint: pos out of range");
  if (!base_reward || !joint_actions || !partner_seat) return fail("ph_buffer_add_reward_joint: null argument");
  if (n_seats <= 0 || seat < 0 || seat >= n_seats) return fail("ph_buffer_add_reward_joint: bad seat");
  PH_HIP(ph::launch_reward_add_joint(rb->rewards + (size_t)pos * rb->E, base_reward, joint_actions, rb->E, n_seats, seat,
                                     partner_seat, bonus, ctx->joint_reward_rule, ctx->stream));
  return 0;
}

int ph_roundrobin_env_step(ph_ctx* ctx, const int* joint_actions, int* partnerid, const float* base_reward, const float* done,
                           float* reward_out, int* alt_action_out, float* next_block, int block_ld, int n_partners,
                           float bonus, int n) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!joint_actions || !partnerid || !base_reward || !done || !reward_out)
    return fail("ph_roundrobin_env_step: null argument");
  if (n <= 0 || n_partners <= 0) return fail("ph_roundrobin_env_step: bad sizes");
  if (next_block && block_ld < 3) return fail("ph_roundrobin_env_step: routing block rows need >= 3 header columns");
  PH_HIP(ph::launch_roundrobin_env_step(joint_actions, partnerid, base_reward, done, reward_out, alt_action_out, next_block,
                                        block_ld, n_partners, bonus, n, ctx->stream));
  return 0;
}

// ---- engine-side round-robin layout (config 4) -------------------------------------------------------------------------------
namespace {
constexpr size_t RR_STAMP_BYTES = 64 * sizeof(unsigned long long);
size_t rr_slot_bytes(int n_partners, int n, int block_ld) {
  const size_t blk = (size_t)n * (block_ld + 1) * sizeof(float), act = (size_t)(n_partners + 1) * n * sizeof(int);
  return ((blk > act ? blk : act) + 255) / 256 * 256;
}
int check_rr_link(const ph_rr_link* l) {
  if (!l) return fail("null ph_rr_link");
  if (l->n_partners < 1 || l->n_partners + 1 > PH_MAX_RANKS) return fail("ph_rr_link: n_partners out of range");
  if (l->rank < 0 || l->rank > l->n_partners || l->n <= 0 || l->block_ld <= 3) return fail("ph_rr_link: bad rank / n / block_ld");
  if (!l->error) return fail("ph_rr_link: error word required");
  for (int r = 0; r <= l->n_partners; ++r)
    if (!l->area[r]) return fail("ph_rr_link: unmapped peer area");
  return 0;
}
unsigned long long* rr_stamps(const ph_rr_link* l, int rank) { return (unsigned long long*)l->area[rank]; }
// rank 0's arrival counters of the block send live behind its stamp words (words 32 .. 63 of the stamp area)
unsigned int* rr_arrive(const ph_rr_link* l) { return (unsigned int*)((unsigned long long*)l->area[0] + 32); }
char* rr_slot(const ph_rr_link* l, int rank, int parity) {
  return (char*)l->area[rank] + RR_STAMP_BYTES + (size_t)parity * rr_slot_bytes(l->n_partners, l->n, l->block_ld);
}
}  // namespace

int ph_rr_area_bytes(int n_partners, int n, int block_ld, size_t* bytes_out) {
  if (!bytes_out || n_partners < 1 || n <= 0 || block_ld <= 3) return fail("ph_rr_area_bytes: bad argument");
  *bytes_out = RR_STAMP_BYTES + 2 * rr_slot_bytes(n_partners, n, block_ld);
  return 0;
}

int ph_roundrobin_ego_iteration(ph_ctx* ctx, const ph_rr_link* link, const ph_rr_ego* ego, int T, unsigned long long iteration) {
  DevGuard dev_guard(ctx);
  if (!ctx || !ego || T <= 0) return fail("ph_roundrobin_ego_iteration: bad argument");
  if (check_rr_link(link)) return 1;
  if (link->rank != 0) return fail("ph_roundrobin_ego_iteration: the ego lives on rank 0");
  if (!ego->params || !ego->obs_seq || !ego->base_reward_seq || !ego->done_seq || !ego->blocks || !ego->partnerid ||
      !ego->rewards || !ego->episode_start0 || !ego->rb)
    return fail("ph_roundrobin_ego_iteration: null argument");
  if (check_rb(ego->rb)) return 1;
  if (ego->rb->E != link->n || ego->rb->T < T) return fail("ph_roundrobin_ego_iteration: rollout buffer shape");
  ph::NetDims nd;
  if (resolve(ctx, ego->spec, &nd)) return 1;
  const int n = link->n, K = link->n_partners, ld = link->block_ld;
  for (int t = 0; t < T; ++t) {
    const unsigned long long want = iteration * (unsigned long long)T + (unsigned long long)t + 1ull;
    int* slot = (int*)rr_slot(link, 0, t & 1);   // (1 + K, n): row 0 = the ego's actions of this step
    // the routing block leaves first: the partners work on step t while the ego's own forward runs
    ph::RRSend sd;
    std::memset(&sd, 0, sizeof(sd));
    sd.src = ego->blocks + (size_t)t * n * ld;
    sd.n = n;
    sd.block_ld = ld;
    for (int k = 0; k < K; ++k) {
      sd.dst[k] = (float*)rr_slot(link, 1 + k, t & 1);
      sd.stamp[k] = rr_stamps(link, 1 + k);
    }
    sd.arrive = rr_arrive(link);
    sd.want = want;
    PH_HIP(ph::launch_rr_send_block(sd, K, ctx->stream));
    if (ph_policy_forward(ctx, ego->spec, ego->params, ego->obs_seq + (size_t)t * n * nd.D, n, nullptr, nullptr, nullptr, ego->seed,
                          ego->counter0 + (unsigned long long)t, 0, slot, nullptr, ego->values, ego->log_probs, nullptr, nullptr,
                          ego->rb, t, t == 0 ? ego->episode_start0 : ego->done_seq + (size_t)(t - 1) * n,
                          t == 0 ? nullptr : ego->rewards + (size_t)(t - 1) * n, 0))
      return 1;
    ph::RREnvStep es;
    std::memset(&es, 0, sizeof(es));
    es.stamps = rr_stamps(link, 0);
    es.want = want;
    es.timeout = link->timeout_cycles;
    es.error = link->error;
    es.joint = slot;
    es.partnerid = ego->partnerid;
    es.partner_trace = ego->partner_trace ? ego->partner_trace + (size_t)t * n : nullptr;
    es.base = ego->base_reward_seq + (size_t)t * n;
    es.done = ego->done_seq + (size_t)t * n;
    es.reward_out = ego->rewards + (size_t)t * n;
    es.alt_action_out = ego->alt_actions ? ego->alt_actions + (size_t)t * n : nullptr;
    es.next_block = ego->blocks + (size_t)((t + 1) % T) * n * ld;
    es.block_ld = ld;
    es.n_partners = K;
    es.n = n;
    es.bonus = ego->bonus;
    PH_HIP(ph::launch_rr_env_step(es, ctx->stream));
  }
  return 0;
}

int ph_roundrobin_partner_iteration(ph_ctx* ctx, const ph_rr_link* link, const ph_rr_partner* pa, int T,
                                    unsigned long long iteration) {
  DevGuard dev_guard(ctx);
  if (!ctx || !pa || T <= 0) return fail("ph_roundrobin_partner_iteration: bad argument");
  if (check_rr_link(link)) return 1;
  if (link->rank < 1) return fail("ph_roundrobin_partner_iteration: partners live on ranks 1 .. n_partners");
  if (!pa->params || !pa->es_scratch || !pa->can_scratch || !pa->pos || !pa->boundary || !pa->term ||
      !pa->open || !pa->prev_mask || !pa->actions || !pa->values || !pa->log_probs || !pa->rb)
    return fail("ph_roundrobin_partner_iteration: null argument");
  if (check_rb(pa->rb)) return 1;
  if (pa->rb->E != link->n) return fail("ph_roundrobin_partner_iteration: the rollout buffer must have one column per environment");
  const int n = link->n, k = link->rank - 1;
  for (int t = 0; t < T; ++t) {
    const unsigned long long want = iteration * (unsigned long long)T + (unsigned long long)t + 1ull;
    ph::RRPartnerStep st;
    std::memset(&st, 0, sizeof(st));
    st.block_stamp = rr_stamps(link, link->rank);
    st.act_stamp = rr_stamps(link, 0) + 1 + k;
    st.want = want;
    st.timeout = link->timeout_cycles;
    st.error = link->error;
    st.block = (const float*)rr_slot(link, link->rank, t & 1);
    st.block_ld = link->block_ld;
    st.n = n;
    st.T = pa->rb->T;
    st.k = k;
    st.rewards = pa->rb->rewards;
    st.pos = pa->pos;
    st.boundary = pa->boundary;
    st.term = pa->term;
    st.open = pa->open;
    st.prev_mask = pa->prev_mask;
    st.can = pa->can_scratch;
    st.es = pa->es_scratch;
    st.actions = pa->actions;
    st.act_dst = (int*)rr_slot(link, 0, t & 1) + (size_t)(1 + k) * n;
    PH_HIP(ph::launch_rr_partner_pre(st, ctx->stream));
    const float* obs_t = st.block + (size_t)n * 4;    // the observations follow the header rows in the slot
    if (ph_policy_forward_ragged(ctx, pa->spec, pa->params, obs_t, nullptr, pa->seed, pa->counter0 + (unsigned long long)t,
                                 0, pa->actions, pa->values, pa->log_probs, pa->rb, pa->pos, pa->can_scratch, pa->es_scratch))
      return 1;
    PH_HIP(ph::launch_rr_partner_post(st, ctx->stream));
  }
  return 0;
}

int ph_buffer_reset(ph_ctx* ctx, const ph_spec* spec, const ph_rollout* rb) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  ph_layout lay;
  if (layout_of(spec, &lay) || check_rb(rb)) return 1;
  const size_t n = (size_t)rb->T * rb->E;
  PH_HIP(hipMemsetAsync(rb->observations, 0, n * lay.D * sizeof(float), ctx->stream));
  PH_HIP(hipMemsetAsync(rb->actions, 0, n * lay.A * sizeof(float), ctx->stream));
  float* arrs[] = {rb->rewards, rb->episode_starts, rb->values, rb->log_probs, rb->advantages, rb->returns};
  for (float* a : arrs) PH_HIP(hipMemsetAsync(a, 0, n * sizeof(float), ctx->stream));
  return 0;
}

// ---- K2 ----
int ph_gae(ph_ctx* ctx, const ph_rollout* rb, const float* last_values, const float* dones, double gamma,
           double gae_lambda, int mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_rb(rb)) return 1;
  if (!last_values || !dones) return fail("ph_gae: null last_values/dones");
  if (mode < 0 || mode > 2) return fail("ph_gae: mode must be 0, 1 or 2");
  PH_HIP(ph::launch_gae(rb->rewards, rb->values, rb->episode_starts, last_values, dones, rb->advantages, rb->returns,
                        rb->T, rb->E, gamma, gae_lambda, mode, ctx->stream));
  return 0;
}

// ---- K4 ----
int ph_policy_forward(ph_ctx* ctx, const ph_spec* spec, const float* params, const float* obs, int n,
                      const unsigned char* action_mask, const float* uniforms, const float* given_actions,
                      unsigned long long seed, unsigned long long counter, int deterministic, int* actions_i32,
                      float* actions_f32, float* values, float* log_probs, float* entropy, float* logits,
                      const ph_rollout* rb, int pos, const float* episode_start_in, const float* pending_reward,
                      int gemm_mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs) return fail("ph_policy_forward: null params/obs");
  if ((uintptr_t)params % 16 != 0) return fail("ph_policy_forward: params must be 16-byte aligned");
  if (n <= 0) return fail("ph_policy_forward: n must be positive");
  ph::FwdArgs a;
  std::memset(&a, 0, sizeof(a));
  if (resolve(ctx, spec, &a.nd, true)) return 1;
  if (a.nd.gauss && action_mask) return fail("ph_policy_forward: action masks belong to the categorical heads");
  a.params = params;
  a.obs = obs;
  a.n = n;
  a.mask = action_mask;
  a.uniforms = uniforms;
  a.given_actions = given_actions;
  a.seed = seed;
  a.counter = counter;
  a.epoch = ctx->rng_epoch;
  a.prof = ctx->prof;
  a.deterministic = deterministic;
  a.act_i32 = actions_i32;
  a.act_f32 = actions_f32;
  a.values = values;
  a.logp = log_probs;
  a.entropy = entropy;
  a.logits = logits;
  if (rb) {
    if (check_rb(rb)) return 1;
    if (n != rb->E) return fail("ph_policy_forward: fused add needs n == rollout E");
    if (pos < 0 || pos >= rb->T) return fail("ph_policy_forward: pos out of range (buffer full?)");
    if (!episode_start_in) return fail("ph_policy_forward: fused add needs episode_start_in");
    const size_t row = (size_t)pos * rb->E;
    a.rb_obs = rb->observations + row * a.nd.D;
    a.rb_act = rb->actions + row * a.nd.A;
    a.rb_rew = rb->rewards + row;
    a.rb_es = rb->episode_starts + row;
    a.rb_val = rb->values + row;
    a.rb_logp = rb->log_probs + row;
    a.es_in = episode_start_in;
    if (pending_reward) {
      if (pos < 1) return fail("ph_policy_forward: pending_reward needs pos >= 1");
      a.prev_rew = rb->rewards + (row - rb->E);
      a.pending_reward = pending_reward;
    }
  } else if (pending_reward) {
    return fail("ph_policy_forward: pending_reward needs the fused rollout-buffer write");
  }
  a.host_done = ctx->fwd_host_done;
  a.host_seq = ctx->act_seq;
  PH_HIP(ph::launch_policy_fwd(a, gemm_mode, ctx->stream));
  return 0;
}

// How ph_policy_act_host waits for a one-tile launch: 1 = on the launch's two completion words in host memory, 0 = on the stream.
// PH_ACT_HOST_WAIT=stream selects the latter (the A/B switch of the measurement in DESIGN.md section 3.1).
static bool act_host_waits_on_words() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PH_ACT_HOST_WAIT");
    v = (e && std::strcmp(e, "stream") == 0) ? 0 : 1;
  }
  return v == 1;
}

int ph_policy_act_host(ph_ctx* ctx, const ph_spec* spec, const float* params, const float* obs_host, int n,
                       const float* episode_start_host, unsigned long long seed, unsigned long long counter, int deterministic,
                       int* actions_host, float* values_host, float* log_probs_host, const ph_rollout* rb, int pos, int gemm_mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs_host) return fail("ph_policy_act_host: null params/obs");
  if (n <= 0) return fail("ph_policy_act_host: n must be positive");
  if (rb && !episode_start_host) return fail("ph_policy_act_host: the fused rollout-buffer write needs episode_start_host");
  if (ctx->capturing) return fail("ph_policy_act_host synchronises: not inside graph capture");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  // staging layout (floats): obs n*D | episode_start n || actions n*A (as int32) | values n | log_probs n -- pinned, coherent host
  // memory the kernel reads and writes DIRECTLY (a few hundred bytes over the host link): no copy launches, one kernel, one wait
  const size_t n_in = (size_t)n * nd.D + (size_t)n, n_out = (size_t)n * nd.A + 2 * (size_t)n, need = n_in + n_out;
  if (need > ctx->act_stage_cap) {
    if (ctx->act_stage_host) (void)hipHostFree(ctx->act_stage_host);
    ctx->act_stage_host = nullptr;
    ctx->act_stage_dev = nullptr;
    ctx->act_stage_cap = 0;
    const size_t cap = need < 4096 ? 4096 : need;
    PH_HIP(hipHostMalloc((void**)&ctx->act_stage_host, cap * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
    PH_HIP(hipHostGetDevicePointer((void**)&ctx->act_stage_dev, ctx->act_stage_host, 0));
    ctx->act_stage_cap = cap;
  }
  float *h = ctx->act_stage_host, *d = ctx->act_stage_dev;   // the same memory, host and device view
  std::memcpy(h, obs_host, (size_t)n * nd.D * sizeof(float));
  if (episode_start_host) std::memcpy(h + (size_t)n * nd.D, episode_start_host, (size_t)n * sizeof(float));
  else std::memset(h + (size_t)n * nd.D, 0, (size_t)n * sizeof(float));
  float* d_out = d + n_in;
  int* d_act = reinterpret_cast<int*>(d_out);
  float *d_val = d_out + (size_t)n * nd.A, *d_lp = d_val + n;
  // One tile of the 16-row forward = one policy and one value workgroup: each stores the launch's sequence number into its word
  // after its last output, and the host polls the two words (the outputs are in the same coherent memory, ordered before them).
  const bool words = act_host_waits_on_words() && n <= 16 && ph::fwd16_eligible(nd, n);
  if (words && !ctx->act_done_host) {
    PH_HIP(hipHostMalloc((void**)&ctx->act_done_host, 64, hipHostMallocMapped | hipHostMallocCoherent));
    PH_HIP(hipHostGetDevicePointer((void**)&ctx->act_done_dev, ctx->act_done_host, 0));
    ctx->act_done_host[0] = ctx->act_done_host[1] = 0;
  }
  if (words) {
    if (++ctx->act_seq == 0) ctx->act_seq = 1;
    ctx->fwd_host_done = ctx->act_done_dev;
  }
  const int frc = ph_policy_forward(ctx, spec, params, d, n, nullptr, nullptr, nullptr, seed, counter, deterministic, d_act,
                                    nullptr, d_val, d_lp, nullptr, nullptr, rb, pos, rb ? d + (size_t)n * nd.D : nullptr, nullptr,
                                    gemm_mode);
  ctx->fwd_host_done = nullptr;
  if (frc) return 1;
  bool arrived = false;
  if (words) {
    // a launch takes ~10 us end to end; a launch that has not signalled within 50 ms is waited for on the stream, which also
    // surfaces whatever error kept it from running
    const unsigned int seq = ctx->act_seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
      if (__atomic_load_n(ctx->act_done_host, __ATOMIC_ACQUIRE) == seq &&
          __atomic_load_n(ctx->act_done_host + 1, __ATOMIC_ACQUIRE) == seq) {
        arrived = true;
        break;
      }
      if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
      __builtin_ia32_pause();
    }
  }
  if (!arrived) PH_HIP(hipStreamSynchronize(ctx->stream));
  const float* h_out = h + n_in;
  if (actions_host) std::memcpy(actions_host, h_out, (size_t)n * nd.A * sizeof(int));
  if (values_host) std::memcpy(values_host, h_out + (size_t)n * nd.A, (size_t)n * sizeof(float));
  if (log_probs_host) std::memcpy(log_probs_host, h_out + (size_t)n * nd.A + n, (size_t)n * sizeof(float));
  return 0;
}

int ph_scripted_rollout(ph_ctx* ctx, const ph_spec* spec, const float* params, const float* obs_seq, const float* rew_seq,
                        const float* done_seq, int n, int n_steps, const float* episode_start0, unsigned long long seed,
                        unsigned long long counter0, int* actions_i32, float* values, float* log_probs, const ph_rollout* rb,
                        int pos0, int gemm_mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs_seq || !rew_seq || !done_seq || !episode_start0) return fail("ph_scripted_rollout: null argument");
  if ((uintptr_t)params % 16 != 0) return fail("ph_scripted_rollout: params must be 16-byte aligned");
  if (n <= 0 || n_steps <= 0) return fail("ph_scripted_rollout: n and n_steps must be positive");
  if (check_rb(rb)) return 1;
  if (n != rb->E) return fail("ph_scripted_rollout: n must equal the rollout buffer's E");
  if (pos0 < 0 || pos0 + n_steps > rb->T) return fail("ph_scripted_rollout: rows pos0 .. pos0 + n_steps - 1 must lie in the buffer");
  ph::FwdArgs a;
  std::memset(&a, 0, sizeof(a));
  if (resolve(ctx, spec, &a.nd)) return 1;
  if (!ph::fwd16_eligible(a.nd, n))
    return fail("ph_scripted_rollout: needs the 16-row forward's shape class (one feature chunk, one Discrete head of <= 8 logits, "
                "n < 16384); use ph_policy_forward per step");
  a.params = params;
  a.obs = obs_seq;
  a.n = n;
  a.seed = seed;
  a.counter = counter0;
  a.epoch = ctx->rng_epoch;
  a.prof = ctx->prof;
  a.act_i32 = actions_i32;
  a.values = values;
  a.logp = log_probs;
  const size_t row = (size_t)pos0 * rb->E;
  a.rb_obs = rb->observations + row * a.nd.D;
  a.rb_act = rb->actions + row * a.nd.A;
  a.rb_rew = rb->rewards + row;
  a.rb_es = rb->episode_starts + row;
  a.rb_val = rb->values + row;
  a.rb_logp = rb->log_probs + row;
  a.es_in = episode_start0;
  ph::ScriptedSteps sc;
  sc.n_steps = n_steps;
  sc.obs_seq = obs_seq;
  sc.rew_seq = rew_seq;
  sc.done_seq = done_seq;
  sc.mask_seq = nullptr;
  PH_HIP(ph::launch_policy_fwd16_rollout(a, sc, gemm_mode, ctx->stream));
  return 0;
}

namespace {
int step_multi_impl(ph_ctx* ctx, int n_calls, const ph_step_call* calls, const ph_p2p* x, int t);
int ensure_p2p_dev(ph_ctx* ctx, const ph_p2p* x);
}
int ph_policy_step_multi(ph_ctx* ctx, int n_calls, const ph_step_call* calls) {
  DevGuard dev_guard(ctx);
  return step_multi_impl(ctx, n_calls, calls, nullptr, 0);
}
namespace {
int step_multi_impl(ph_ctx* ctx, int n_calls, const ph_step_call* calls, const ph_p2p* x, int t) {
  if (!ctx) return fail("null ctx");
  if (!calls) return fail("ph_policy_step_multi: null calls");
  if (n_calls <= 0 || n_calls > ph::MAX_LOCAL_AGENTS) return fail("ph_policy_step_multi: 1..4 calls per launch");
  ph::FwdMulti m;
  std::memset(&m, 0, sizeof(m));
  for (int i = 0; i < n_calls; ++i) {
    const ph_step_call& c = calls[i];
    ph::FwdArgs& a = m.a[i];
    if (!c.params || !c.obs || !c.rb || !c.episode_start_in) return fail("ph_policy_step_multi: null argument");
    if ((uintptr_t)c.params % 16 != 0) return fail("ph_policy_step_multi: params must be 16-byte aligned");
    if (resolve(ctx, c.spec, &a.nd)) return 1;
    if (check_rb(c.rb)) return 1;
    if (c.n != c.rb->E) return fail("ph_policy_step_multi: n must equal the rollout E");
    if (c.pos < 0 || c.pos >= c.rb->T) return fail("ph_policy_step_multi: pos out of range (buffer full?)");
    if (i > 0 && (c.n != calls[0].n || a.nd.Lp != m.a[0].nd.Lp))
      return fail("ph_policy_step_multi: all calls must share n and the padded logit count");
    a.params = c.params;
    a.obs = c.obs;
    a.n = c.n;
    a.mask = c.action_mask;
    a.seed = c.seed;
    a.counter = c.counter;
    a.epoch = ctx->rng_epoch;
    a.deterministic = c.deterministic & 1;
    if (c.deterministic & (PH_STEP_FIX_ILLEGAL | PH_STEP_MASK_ENV_ONLY)) {
      if (!c.action_mask) return fail("PH_STEP_FIX_ILLEGAL / PH_STEP_MASK_ENV_ONLY need an action mask");
      if (!(a.nd.A == 1 && a.nd.L <= 8))
        return fail("PH_STEP_FIX_ILLEGAL: single Discrete head of at most 8 logits only (use ph_fix_illegal_actions)");
      a.env_mask = c.action_mask;
      if (c.deterministic & PH_STEP_MASK_ENV_ONLY) a.mask = nullptr;   // the policy never sees the mask (plain PPO partner)
    }
    a.act_i32 = c.actions_i32;
    a.values = c.values;
    a.logp = c.log_probs;
    const size_t row = (size_t)c.pos * c.rb->E;
    a.rb_obs = c.rb->observations + row * a.nd.D;
    a.rb_act = c.rb->actions + row * a.nd.A;
    a.rb_rew = c.rb->rewards + row;
    a.rb_es = c.rb->episode_starts + row;
    a.rb_val = c.rb->values + row;
    a.rb_logp = c.rb->log_probs + row;
    a.es_in = c.episode_start_in;
    if (c.pending_reward) {
      if (c.pos < 1) return fail("ph_policy_step_multi: pending_reward needs pos >= 1");
      a.prev_rew = c.rb->rewards + (row - c.rb->E);
      a.pending_reward = c.pending_reward;
      if (c.joint_actions) {
        if (!c.partner_seat || c.n_seats <= 0 || c.seat < 0 || c.seat >= c.n_seats)
          return fail("ph_policy_step_multi: bad joint-action description");
        a.joint = c.joint_actions;
        a.n_seats = c.n_seats;
        a.seat = c.seat;
        a.partner_seat = c.partner_seat;
        a.bonus = c.bonus;
        a.reward_rule = ctx->joint_reward_rule;
      }
    }
  }
  if (x) {  // exchange fused into the launch (16-row kernel only; the caller checked eligibility)
    if (!ph::fwd16_eligible(m.a[0].nd, m.a[0].n)) return fail("fused peer-to-peer step needs the 16-row forward kernel");
    if (x->count != n_calls * m.a[0].n) return fail("ph_p2p.count must be local agents x n");
    if (ensure_p2p_dev(ctx, x)) return 1;
    m.px.x = ctx->p2p_dev;
    m.px.t = t;
    m.px.a_local = n_calls;
    for (int i = 0; i < n_calls; ++i) {
      if (!m.a[i].joint) continue;   // consumers of the previous step's joint action read the stamp-in-band words
      const int prev = t >= 1 ? t - 1 : x->T - 1;   // step whose joint action this launch consumes (t = 0: last step of the previous iteration)
      m.a[i].joint_ll = x->ll[x->rank] + (size_t)(prev % x->ll_slots) * x->world * x->count;
      m.a[i].ll_epoch = x->epoch;
      m.a[i].ll_T = x->T;
      m.a[i].ll_t = t - 1;
      m.a[i].ll_timeout = x->timeout_cycles;
      m.a[i].ll_error = x->error;
    }
  }
  PH_HIP(ph::launch_policy_fwd_multi(m, n_calls, ctx->stream));
  return 0;
}
}  // namespace

// ---- RCCL exchange (librccl.so resolved at first use) ------------------------------------------------------------------
namespace {
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, ncclUniqueId, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = (int (*)(void*))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (int (*)(void**, int, ncclUniqueId, int))dlsym(api.lib, "ncclCommInitRank");
      api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
      api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
      api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.AllGather || !api.CommDestroy) api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}
int fail_rccl(const char* what, int rc) {
  RcclApi* r = rccl_api();
  std::string msg = std::string(what) + ": " + ((r && r->GetErrorString) ? r->GetErrorString(rc) : "RCCL error");
  return fail(msg.c_str());
}
}  // namespace

int ph_comm_unique_id(unsigned char* id_out) {
  if (!id_out) return fail("ph_comm_unique_id: null output");
  RcclApi* r = rccl_api();
  if (!r) return fail("ph_comm_unique_id: librccl.so could not be loaded");
  ncclUniqueId id;
  const int rc = r->GetUniqueId(&id);
  if (rc != 0) return fail_rccl("ncclGetUniqueId", rc);
  std::memcpy(id_out, id.internal, PH_COMM_ID_BYTES);
  return 0;
}

int ph_comm_init(ph_ctx* ctx, const unsigned char* id, int world, int rank) {
  DevGuard dev_guard(ctx);
  if (!ctx || !id) return fail("ph_comm_init: null argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("ph_comm_init: bad world / rank");
  if (ctx->comm) return fail("ph_comm_init: this context already has a communicator");
  RcclApi* r = rccl_api();
  if (!r) return fail("ph_comm_init: librccl.so could not be loaded");
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, PH_COMM_ID_BYTES);
  void* comm = nullptr;
  const int rc = r->CommInitRank(&comm, world, uid, rank);
  if (rc != 0) return fail_rccl("ncclCommInitRank", rc);
  ctx->comm = comm;
  ctx->comm_world = world;
  ctx->comm_rank = rank;
  return 0;
}

int ph_comm_destroy(ph_ctx* ctx) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (ctx->comm) {
    RcclApi* r = rccl_api();
    if (r) (void)r->CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
  }
  return 0;
}

int ph_all_gather_i32(ph_ctx* ctx, const int* local, int* joint, int count) {
  DevGuard dev_guard(ctx);
  if (!ctx || !local || !joint || count <= 0) return fail("ph_all_gather_i32: bad argument");
  if (!ctx->comm) {  // single process: the joint action is the local one
    PH_HIP(hipMemcpyAsync(joint, local, (size_t)count * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
  }
  RcclApi* r = rccl_api();
  const int rc = r->AllGather(local, joint, (size_t)count, (int)ncclInt32, ctx->comm, ctx->stream);
  if (rc != 0) return fail_rccl("ncclAllGather", rc);
  return 0;
}

int ph_selfplay_rollout(ph_ctx* ctx, int n_calls, const ph_step_call* calls, int T, const int* local, int* joint,
                        int count) {
  DevGuard dev_guard(ctx);
  if (!ctx || !calls || T <= 0) return fail("ph_selfplay_rollout: bad argument");
  for (int t = 0; t < T; ++t) {
    if (ph_policy_step_multi(ctx, n_calls, calls + (size_t)t * n_calls)) return 1;
    if (ph_all_gather_i32(ctx, local, joint, count)) return 1;
  }
  return 0;
}

// ---- peer-to-peer exchange buffers ---------------------------------------------------------------------------------------
int ph_p2p_alloc(ph_ctx* ctx, size_t bytes, void** ptr_out, unsigned char* handle_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !ptr_out || !handle_out || bytes == 0) return fail("ph_p2p_alloc: bad argument");
  void* p = nullptr;
  PH_HIP(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained));
  hipError_t e = hipMemset(p, 0, bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  hipIpcMemHandle_t hd;
  if (e == hipSuccess) e = hipIpcGetMemHandle(&hd, p);
  if (e != hipSuccess) {
    (void)hipFree(p);
    return fail_hip("ph_p2p_alloc", e);
  }
  static_assert(sizeof(hd) == PH_IPC_HANDLE_BYTES, "IPC handle size");
  std::memcpy(handle_out, &hd, PH_IPC_HANDLE_BYTES);
  *ptr_out = p;
  return 0;
}
int ph_p2p_open(ph_ctx* ctx, const unsigned char* handle, void** ptr_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !handle || !ptr_out) return fail("ph_p2p_open: bad argument");
  hipIpcMemHandle_t hd;
  std::memcpy(&hd, handle, PH_IPC_HANDLE_BYTES);
  void* p = nullptr;
  PH_HIP(hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess));
  *ptr_out = p;
  return 0;
}
int ph_p2p_close(ph_ctx* ctx, void* ptr) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (ptr) PH_HIP(hipIpcCloseMemHandle(ptr));
  return 0;
}
int ph_p2p_free(ph_ctx* ctx, void* ptr) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (ptr) PH_HIP(hipFree(ptr));
  return 0;
}
namespace {
int check_p2p(const ph_p2p* x) {
  if (!x) return fail("null ph_p2p");
  if (x->world < 1 || x->world > PH_MAX_RANKS || x->rank < 0 || x->rank >= x->world || x->count <= 0 || x->T <= 0)
    return fail("ph_p2p: bad world / rank / count / T");
  if (!x->epoch || !x->error) return fail("ph_p2p: epoch and error words are required");
  for (int p = 0; p < x->world; ++p)
    if (!x->joint[0][p] || !x->joint[1][p] || !x->flags[p] || !x->ll[p]) return fail("ph_p2p: unmapped peer");
  if (x->ll_slots < x->T) return fail("ph_p2p: ll_slots must be >= T (a slot must not be reused inside an iteration)");
  return 0;
}
}  // namespace
int ph_p2p_push(ph_ctx* ctx, const ph_p2p* x, const int* local, int t) {
  DevGuard dev_guard(ctx);
  if (!ctx || !local) return fail("ph_p2p_push: null argument");
  if (check_p2p(x)) return 1;
  PH_HIP(ph::launch_p2p_push(*x, local, t, ctx->stream));
  return 0;
}
int ph_p2p_wait(ph_ctx* ctx, const ph_p2p* x, int t) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_p2p(x)) return 1;
  PH_HIP(ph::launch_p2p_wait(*x, t, ctx->stream));
  return 0;
}
int ph_p2p_ll_push(ph_ctx* ctx, const ph_p2p* x, const int* local, int t) {
  DevGuard dev_guard(ctx);
  if (!ctx || !local) return fail("ph_p2p_ll_push: null argument");
  if (check_p2p(x)) return 1;
  PH_HIP(ph::launch_p2p_ll_push(*x, local, t, ctx->stream));
  return 0;
}
int ph_p2p_ll_unpack(ph_ctx* ctx, const ph_p2p* x, int t) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_p2p(x)) return 1;
  PH_HIP(ph::launch_p2p_ll_unpack(*x, t, ctx->stream));
  return 0;
}
int ph_selfplay_rollout_p2p(ph_ctx* ctx, int n_calls, const ph_step_call* calls, int T, const int* local,
                            const ph_p2p* x) {
  DevGuard dev_guard(ctx);
  if (!ctx || !calls || !local || T <= 0) return fail("ph_selfplay_rollout_p2p: bad argument");
  if (check_p2p(x)) return 1;
  bool fused = x->count == n_calls * calls[0].n && getenv("PH_P2P_UNFUSED") == nullptr;
  for (int i = 0; i < n_calls && fused; ++i) {
    ph::NetDims nd;
    if (resolve(ctx, calls[i].spec, &nd)) return 1;
    fused = ph::fwd16_eligible(nd, calls[i].n);
  }
  if (fused) {
    // push and wait live inside the step launch; the stamp of the last step is awaited once, for whoever reads it next
    for (int t = 0; t < T; ++t)
      if (step_multi_impl(ctx, n_calls, calls + (size_t)t * n_calls, x, t)) return 1;
    PH_HIP(ph::launch_p2p_ll_unpack(*x, T - 1, ctx->stream));
    return 0;
  }
  for (int t = 0; t < T; ++t) {
    if (ph_policy_step_multi(ctx, n_calls, calls + (size_t)t * n_calls)) return 1;
    PH_HIP(ph::launch_p2p_push(*x, local, t, ctx->stream));
    PH_HIP(ph::launch_p2p_wait(*x, t, ctx->stream));
  }
  return 0;
}

namespace {
int ensure_p2p_dev(ph_ctx* ctx, const ph_p2p* x) {
  if (!ctx->p2p_dev) PH_HIP(hipMalloc((void**)&ctx->p2p_dev, sizeof(ph_p2p)));
  if (!ctx->p2p_valid || std::memcmp(&ctx->p2p_host, x, sizeof(ph_p2p)) != 0) {
    if (ctx->capturing) return fail("the peer-to-peer descriptor changed inside graph capture");
    ctx->p2p_host = *x;
    PH_HIP(hipMemcpyAsync(ctx->p2p_dev, &ctx->p2p_host, sizeof(ph_p2p), hipMemcpyHostToDevice, ctx->stream));
    PH_HIP(hipStreamSynchronize(ctx->stream));
    ctx->p2p_valid = true;
  }
  return 0;
}
}  // namespace

int ph_selfplay_rollout_persistent_capacity(ph_ctx* ctx, int* workgroups_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !workgroups_out) return fail("ph_selfplay_rollout_persistent_capacity: null argument");
  if (ctx->exchange_blocks_per_cu <= 0) PH_HIP(ph::exchange_rollout_blocks_per_cu(&ctx->exchange_blocks_per_cu));
  *workgroups_out = ctx->exchange_blocks_per_cu * ctx->num_cu;
  return 0;
}

int ph_selfplay_rollout_persistent(ph_ctx* ctx, int n_calls, const ph_rollout_call* calls, int T, const ph_p2p* x,
                                   int ranks_on_device) {
  DevGuard dev_guard(ctx);
  if (!ctx || !calls || T <= 0) return fail("ph_selfplay_rollout_persistent: bad argument");
  if (n_calls <= 0 || n_calls > ph::MAX_LOCAL_AGENTS) return fail("ph_selfplay_rollout_persistent: 1..4 local agents");
  if (check_p2p(x)) return 1;
  if (x->T < T) return fail("ph_selfplay_rollout_persistent: ph_p2p.T must be >= the rollout length (stamps are epoch * x.T + t + 1)");
  if (x->ll_slots < 2 * x->T) return fail("ph_selfplay_rollout_persistent: ph_p2p.ll_slots must be >= 2 * ph_p2p.T (two alternating halves)");
  if (x->count != n_calls * calls[0].n) return fail("ph_p2p.count must be local agents x n");
  if (ranks_on_device < 1) ranks_on_device = 1;
  const long long wgs = (long long)n_calls * 2 * ((calls[0].n + 15) / 16) * ranks_on_device;
  int capacity = 0;
  if (ph_selfplay_rollout_persistent_capacity(ctx, &capacity)) return 1;
  if (wgs > (long long)capacity)
    return fail("ph_selfplay_rollout_persistent: the launch's workgroups would not all be resident (value workgroups poll): use "
                "ph_selfplay_rollout_p2p");
  ph::FwdMulti m;
  ph::ScriptedMulti sm;
  std::memset(&m, 0, sizeof(m));
  std::memset(&sm, 0, sizeof(sm));
  for (int i = 0; i < n_calls; ++i) {
    const ph_rollout_call& c = calls[i];
    ph::FwdArgs& a = m.a[i];
    if (!c.params || !c.obs_seq || !c.rew_seq || !c.done_seq || !c.rb || !c.episode_start0 || !c.partner_seat)
      return fail("ph_selfplay_rollout_persistent: null argument");
    if ((uintptr_t)c.params % 16 != 0) return fail("ph_selfplay_rollout_persistent: params must be 16-byte aligned");
    if (resolve(ctx, c.spec, &a.nd)) return 1;
    if (!ph::fwd16_eligible(a.nd, c.n))
      return fail("ph_selfplay_rollout_persistent: needs the 16-row forward's shape class (one feature chunk, one Discrete head "
                  "of <= 8 logits, n < 16384)");
    if (check_rb(c.rb)) return 1;
    if (c.n != c.rb->E || c.n != calls[0].n) return fail("ph_selfplay_rollout_persistent: every call's n must equal its rollout E and agree");
    if (c.rb->T < T) return fail("ph_selfplay_rollout_persistent: the rollout buffer has fewer than T rows");
    if (c.n_seats <= 0 || c.seat < 0 || c.seat >= c.n_seats) return fail("ph_selfplay_rollout_persistent: bad seat description");
    a.params = c.params;
    a.obs = c.obs_seq;
    a.n = c.n;
    if (c.mask_mode < 0 || c.mask_mode > 3) return fail("ph_selfplay_rollout_persistent: mask_mode is 0..3");
    sm.sc[i].mask_policy = (c.mask_seq && (c.mask_mode == 0 || c.mask_mode == 1)) ? 1 : 0;
    sm.sc[i].mask_env = (c.mask_seq && (c.mask_mode == 1 || c.mask_mode == 2)) ? 1 : 0;
    a.mask = sm.sc[i].mask_policy ? c.mask_seq : nullptr;
    a.env_mask = sm.sc[i].mask_env ? c.mask_seq : nullptr;
    a.seed = c.seed;
    a.counter = c.counter0;
    a.epoch = ctx->rng_epoch;
    a.act_i32 = c.actions_i32;
    a.values = c.values;
    a.logp = c.log_probs;
    a.rb_obs = c.rb->observations;
    a.rb_act = c.rb->actions;
    a.rb_rew = c.rb->rewards;
    a.rb_es = c.rb->episode_starts;
    a.rb_val = c.rb->values;
    a.rb_logp = c.rb->log_probs;
    a.es_in = c.episode_start0;
    a.joint = x->joint[0][x->rank];      // non-null marks "the reward has a joint-action term"; the kernel reads the words
    a.n_seats = c.n_seats;
    a.seat = c.seat;
    a.partner_seat = c.partner_seat;
    a.bonus = c.bonus;
    a.reward_rule = ctx->joint_reward_rule;
    a.ll_epoch = x->epoch;
    a.ll_T = x->T;
    a.ll_timeout = x->timeout_cycles;
    a.ll_error = x->error;
    sm.sc[i].n_steps = T;
    sm.sc[i].obs_seq = c.obs_seq;
    sm.sc[i].rew_seq = c.rew_seq;
    sm.sc[i].done_seq = c.done_seq;
    sm.sc[i].mask_seq = c.mask_seq;
  }
  if (ensure_p2p_dev(ctx, x)) return 1;
  m.px.x = ctx->p2p_dev;
  m.px.t = 0;
  m.px.a_local = n_calls;
  m.px.persistent = 1;
  PH_HIP(ph::launch_policy_fwd16_exchange_rollout(m, sm, n_calls, ctx->stream));
  // the last step's words -> the plain receive slot (ordinary consumers, route verification); waits for every peer once
  PH_HIP(ph::launch_p2p_ll_unpack(*x, T - 1, ctx->stream, -2));
  return 0;
}

int ph_policy_forward_ragged(ph_ctx* ctx, const ph_spec* spec, const float* params, const float* obs,
                             const unsigned char* action_mask, unsigned long long seed, unsigned long long counter,
                             int deterministic, int* actions_i32, float* values, float* log_probs, const ph_rollout* rb,
                             const int* pos_env, const unsigned char* record_mask, const float* episode_start_in) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs || !pos_env || !record_mask || !episode_start_in)
    return fail("ph_policy_forward_ragged: null argument");
  if ((uintptr_t)params % 16 != 0) return fail("ph_policy_forward_ragged: params must be 16-byte aligned");
  if (check_rb(rb)) return 1;
  ph::FwdArgs a;
  std::memset(&a, 0, sizeof(a));
  if (resolve(ctx, spec, &a.nd)) return 1;
  a.params = params;
  a.obs = obs;
  a.n = rb->E;
  a.mask = action_mask;
  a.seed = seed;
  a.counter = counter;
  a.epoch = ctx->rng_epoch;
  a.prof = ctx->prof;
  a.deterministic = deterministic;
  a.act_i32 = actions_i32;
  a.values = values;
  a.logp = log_probs;
  a.rb_obs = rb->observations;  // array bases: rows are selected per env
  a.rb_act = rb->actions;
  a.rb_rew = rb->rewards;
  a.rb_es = rb->episode_starts;
  a.rb_val = rb->values;
  a.rb_logp = rb->log_probs;
  a.es_in = episode_start_in;
  a.pos_env = pos_env;
  a.rec_mask = record_mask;
  a.rb_T = rb->T;
  PH_HIP(ph::launch_policy_fwd(a, 0, ctx->stream));
  return 0;
}

int ph_buffer_add_reward_ragged(ph_ctx* ctx, const ph_rollout* rb, const int* pos_env, const float* reward,
                                const unsigned char* env_mask) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_rb(rb)) return 1;
  if (!pos_env || !reward) return fail("ph_buffer_add_reward_ragged: null argument");
  PH_HIP(ph::launch_reward_add_ragged(rb->rewards, pos_env, reward, env_mask, rb->T, rb->E, ctx->stream));
  return 0;
}

int ph_ragged_advance(ph_ctx* ctx, const ph_rollout* rb, int* pos_env, const unsigned char* record_mask) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (check_rb(rb)) return 1;
  if (!pos_env || !record_mask) return fail("ph_ragged_advance: null argument");
  PH_HIP(ph::launch_ragged_advance(pos_env, record_mask, rb->T, rb->E, ctx->stream));
  return 0;
}

int ph_fix_illegal_actions(ph_ctx* ctx, int* actions, const unsigned char* action_mask, int n, int L) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!actions || !action_mask) return fail("ph_fix_illegal_actions: null argument");
  if (n <= 0 || L <= 0) return fail("ph_fix_illegal_actions: bad sizes");
  PH_HIP(ph::launch_fix_illegal(actions, action_mask, n, L, ctx->stream));
  return 0;
}

// ---- vectorised game rules ----
int ph_rps_step(ph_ctx* ctx, const int* ego_actions, const int* alt_actions, float* ego_reward, float* alt_reward, int n) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!ego_actions || !alt_actions || !ego_reward || !alt_reward) return fail("ph_rps_step: null argument");
  if (n <= 0) return fail("ph_rps_step: n must be positive");
  PH_HIP(ph::launch_rps_step(ego_actions, alt_actions, ego_reward, alt_reward, n, ctx->stream));
  return 0;
}

int ph_liar_step(ph_ctx* ctx, const int* hands, int* history, int* nmoves, const int* actions,
                 const unsigned char* is_ego, const unsigned char* active, float* obs_next, float* rewards,
                 unsigned char* done, int n) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!hands || !history || !nmoves || !actions || !is_ego || !obs_next || !rewards || !done)
    return fail("ph_liar_step: null argument");
  if (n <= 0) return fail("ph_liar_step: n must be positive");
  if (((uintptr_t)hands | (uintptr_t)history) % 16 || ((uintptr_t)actions | (uintptr_t)obs_next | (uintptr_t)rewards) % 8)
    return fail("ph_liar_step: hands/history must be 16-byte aligned, actions/obs_next/rewards 8-byte aligned");
  PH_HIP(ph::launch_liar_step(hands, history, nmoves, actions, is_ego, active, obs_next, rewards, done, n, ctx->stream));
  return 0;
}

int ph_liar_reset(ph_ctx* ctx, int* hands, int* history, int* nmoves, const unsigned char* reset_mask,
                  unsigned char* ego_first, unsigned long long seed, unsigned long long counter, float probegostart,
                  int n) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!hands || !history || !nmoves || !ego_first) return fail("ph_liar_reset: null argument");
  if (n <= 0) return fail("ph_liar_reset: n must be positive");
  if (((uintptr_t)hands | (uintptr_t)history) % 16) return fail("ph_liar_reset: hands/history must be 16-byte aligned");
  PH_HIP(ph::launch_liar_reset(hands, history, nmoves, reset_mask, ego_first, seed, counter, ctx->rng_epoch, probegostart, n,
                               ctx->stream));
  return 0;
}

int ph_liar_obs(ph_ctx* ctx, const int* hands, const int* history, const int* nmoves, const unsigned char* is_ego,
                const unsigned char* active, float* obs_out, int n) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!hands || !history || !nmoves || !is_ego || !obs_out) return fail("ph_liar_obs: null argument");
  if (n <= 0) return fail("ph_liar_obs: n must be positive");
  if (((uintptr_t)hands | (uintptr_t)history) % 16 || (uintptr_t)obs_out % 8)
    return fail("ph_liar_obs: hands/history must be 16-byte aligned, obs_out 8-byte aligned");
  PH_HIP(ph::launch_liar_obs(hands, history, nmoves, is_ego, active, obs_out, n, ctx->stream));
  return 0;
}

int ph_liar_selfplay_step(ph_ctx* ctx, const ph_liar_selfplay* sp, int ego_pos, unsigned long long counter, int deal_only) {
  DevGuard dev_guard(ctx);
  if (!ctx || !sp) return fail("ph_liar_selfplay_step: null argument");
  const ph_liar_selfplay& s = *sp;
  if (s.n <= 0 || !s.spec || !s.ego_rb || !s.alt_rb) return fail("ph_liar_selfplay_step: incomplete description");
  if (check_rb(s.ego_rb) || check_rb(s.alt_rb)) return 1;
  if (s.ego_rb->E != s.n || s.alt_rb->E != s.n) return fail("ph_liar_selfplay_step: buffers must have E = n");
  if (((uintptr_t)s.hands | (uintptr_t)s.history) % 16 ||
      ((uintptr_t)s.ego_actions | (uintptr_t)s.alt_actions | (uintptr_t)s.obs_ego | (uintptr_t)s.obs_alt | (uintptr_t)s.obs_next |
       (uintptr_t)s.rew1 | (uintptr_t)s.rew2) % 8)
    return fail("ph_liar_selfplay_step: hands/history must be 16-byte aligned, actions/observations/rewards 8-byte aligned");
  hipStream_t st = ctx->stream;
  if (!deal_only) {
    if (ego_pos < 0 || ego_pos >= s.ego_rb->T) return fail("ph_liar_selfplay_step: ego_pos out of range (buffer full?)");
    // ego moves in every table
    if (ph_policy_forward(ctx, s.spec, s.ego_params, s.obs_ego, s.n, nullptr, nullptr, nullptr, s.ego_seed, counter, 0,
                          s.ego_actions, nullptr, s.ego_values, s.ego_log_probs, nullptr, nullptr, s.ego_rb, ego_pos,
                          s.ego_episode_start, nullptr, 0))
      return 1;
    PH_HIP(ph::launch_liar_sp_after_ego(s, st));
    // partner replies where the game goes on (obs_next = its observation there)
    if (ph_policy_forward_ragged(ctx, s.spec, s.alt_params, s.obs_next, nullptr, s.alt_seed, 2 * counter, 0, s.alt_actions,
                                 s.alt_values, s.alt_log_probs, s.alt_rb, s.alt_pos, s.can, s.es_alt))
      return 1;
  }
  // reply played and credited; finished tables (flagged in s.done) are re-dealt; where the partner opens the new game it
  // moves once
  PH_HIP(ph::launch_liar_sp_after_reply(s, deal_only ? nullptr : s.ego_rb->rewards + (size_t)ego_pos * s.n, counter,
                                        ctx->rng_epoch, deal_only, st));
  if (ph_policy_forward_ragged(ctx, s.spec, s.alt_params, s.obs_alt, nullptr, s.alt_seed, 2 * counter + 1, 0, s.alt_actions,
                               s.alt_values, s.alt_log_probs, s.alt_rb, s.alt_pos, s.can, s.es_alt))
    return 1;
  PH_HIP(ph::launch_liar_sp_after_opening(s, st));
  return 0;
}

int ph_liar_selfplay_rollout(ph_ctx* ctx, const ph_liar_selfplay* sp, int ego_pos, int n_steps, unsigned long long counter) {
  DevGuard dev_guard(ctx);
  if (!ctx || !sp) return fail("ph_liar_selfplay_rollout: null argument");
  const ph_liar_selfplay& s = *sp;
  if (s.n <= 0 || !s.spec || !s.ego_rb || !s.alt_rb) return fail("ph_liar_selfplay_rollout: incomplete description");
  if (check_rb(s.ego_rb) || check_rb(s.alt_rb)) return 1;
  if (s.ego_rb->E != s.n || s.alt_rb->E != s.n) return fail("ph_liar_selfplay_rollout: buffers must have E = n");
  if (n_steps <= 0 || ego_pos < 0 || ego_pos + n_steps > s.ego_rb->T)
    return fail("ph_liar_selfplay_rollout: ego_pos + n_steps exceeds the ego's buffer");
  if (((uintptr_t)s.hands | (uintptr_t)s.history) % 16 ||
      ((uintptr_t)s.ego_actions | (uintptr_t)s.alt_actions | (uintptr_t)s.obs_ego | (uintptr_t)s.obs_alt | (uintptr_t)s.obs_next |
       (uintptr_t)s.rew1 | (uintptr_t)s.rew2) % 8)
    return fail("ph_liar_selfplay_rollout: hands/history must be 16-byte aligned, actions/observations/rewards 8-byte aligned");
  if (((uintptr_t)s.ego_params | (uintptr_t)s.alt_params) % 16) return fail("ph_liar_selfplay_rollout: params must be 16-byte aligned");
  ph::FwdArgs ego, reply, opening;
  std::memset(&ego, 0, sizeof(ego));
  if (resolve(ctx, s.spec, &ego.nd)) return 1;
  if (!ph::liar_rollout_eligible(ego.nd, s.n))
    return fail("ph_liar_selfplay_rollout: the spec does not fit the 16-row one-hot forward (use ph_liar_selfplay_step)");
  // exactly the argument records ph_liar_selfplay_step's three forwards build (ph_policy_forward / _ragged)
  ego.params = s.ego_params;
  ego.obs = s.obs_ego;
  ego.n = s.n;
  ego.seed = s.ego_seed;
  ego.epoch = ctx->rng_epoch;
  ego.act_i32 = s.ego_actions;
  ego.values = s.ego_values;
  ego.logp = s.ego_log_probs;
  const size_t row = (size_t)ego_pos * s.n;
  ego.rb_obs = s.ego_rb->observations + row * ego.nd.D;
  ego.rb_act = s.ego_rb->actions + row * ego.nd.A;
  ego.rb_rew = s.ego_rb->rewards + row;
  ego.rb_es = s.ego_rb->episode_starts + row;
  ego.rb_val = s.ego_rb->values + row;
  ego.rb_logp = s.ego_rb->log_probs + row;
  ego.es_in = s.ego_episode_start;
  ego.prof = ctx->prof;   // debug stamps (scripts/liar_rollout_profile.py)
  std::memset(&reply, 0, sizeof(reply));
  reply.nd = ego.nd;
  reply.params = s.alt_params;
  reply.obs = s.obs_next;
  reply.n = s.n;
  reply.seed = s.alt_seed;
  reply.epoch = ctx->rng_epoch;
  reply.act_i32 = s.alt_actions;
  reply.values = s.alt_values;
  reply.logp = s.alt_log_probs;
  reply.rb_obs = s.alt_rb->observations;
  reply.rb_act = s.alt_rb->actions;
  reply.rb_rew = s.alt_rb->rewards;
  reply.rb_es = s.alt_rb->episode_starts;
  reply.rb_val = s.alt_rb->values;
  reply.rb_logp = s.alt_rb->log_probs;
  reply.es_in = s.es_alt;
  reply.pos_env = s.alt_pos;
  reply.rec_mask = s.can;
  reply.rb_T = s.alt_rb->T;
  opening = reply;
  opening.obs = s.obs_alt;
  PH_HIP(ph::launch_liar_rollout(s, ego, reply, opening, n_steps, counter, ctx->rng_epoch, s.ego_rb->rewards + row, ctx->stream));
  return 0;
}

int ph_framestack_push(ph_ctx* ctx, float* stack, const float* obs, const unsigned char* reset_mask,
                       const float* default_obs, int n, int D, int numframes) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!stack || !obs) return fail("ph_framestack_push: null argument");
  if (n <= 0 || D <= 0 || numframes <= 0) return fail("ph_framestack_push: bad sizes");
  PH_HIP(ph::launch_framestack_push(stack, obs, reset_mask, default_obs, n, D, numframes, ctx->stream));
  return 0;
}

// ---- K3 + K5 + K6 ----
namespace {

struct MbPlan {
  int nb, ntiles, nwg;
};

MbPlan plan_minibatch(const ph_ctx* ctx, const ph::NetDims& nd, int nb) {
  MbPlan p;
  p.nb = nb;
  ph::grad_plan(nd, nb, ctx->num_cu, &p.ntiles, &p.nwg);
  return p;
}

void fill_grad_args(ph::GradArgs& g, const ph::NetDims& nd, const float* params, const ph_rollout* rb,
                    const ph_ppo_hyper* hp, ph_ctx* ctx) {
  g.nd = nd;
  g.params = params;
  g.rb_obs = rb->observations;
  g.rb_act = rb->actions;
  g.rb_val = rb->values;
  g.rb_logp = rb->log_probs;
  g.rb_adv = rb->advantages;
  g.rb_ret = rb->returns;
  g.T = rb->T;
  g.E = rb->E;
  g.clip = hp->clip_range;
  g.clip_vf = hp->clip_range_vf;
  g.ent_coef = hp->ent_coef;
  g.vf_coef = hp->vf_coef;
  g.norm_adv = hp->normalize_advantage;
  g.slabs = ctx->slabs;
  g.statpart = ctx->statpart;
  g.stop_flag = ctx->stop_flag;
  g.prof = ctx->prof;
  g.wimage = ctx->wimage;
  g.ximg = ctx->ximg;
  g.ximg_zero_row = rb->T * rb->E;
  g.rec_pi = ctx->rec_pi;   // callers offset these to the minibatch
  g.rec_vf = ctx->rec_vf;
}

// the split kernel's weight fragment image of `params`, rebuilt from scratch (entries no parameter backs are zero): at the start
// of every train() / gradient call, whatever happened to the parameters in between; ppo_adam_kernel keeps it current afterwards
int rebuild_weight_image(ph_ctx* ctx, const ph::NetDims& nd, const float* params) {
  if (!nd.split) return 0;
  if ((size_t)nd.wimage_elems > ctx->wimage_cap) {
    if (ctx->capturing) return fail("first split-kernel use inside graph capture: call it once outside capture first");
    if (ensure(ctx->wimage, ctx->wimage_cap, (size_t)nd.wimage_elems)) return 1;
    ctx->wimage_zeroed_for = 0;
  }
  // Elements no parameter backs are written by nobody (weight_image_kernel / ppo_adam_kernel go through the map), so they are
  // zeroed when the image starts serving a spec, not before every call (a 4.5 us fill node in every iteration graph).  The fill
  // and every later user of the image are enqueued on the context's stream of that moment; a context whose stream is switched
  // (ph_ctx_set_stream) is synchronised by its owner at the switch, as for every other piece of workspace.
  if (ctx->wimage_zeroed_for != nd.spec_id) {
    if (ctx->capturing) return fail("the weight image changes its spec inside graph capture: run the same call once outside capture first");
    PH_HIP(hipMemsetAsync(ctx->wimage, 0, (size_t)nd.wimage_elems * sizeof(unsigned short), ctx->stream));
    ctx->wimage_zeroed_for = nd.spec_id;
  }
  PH_HIP(ph::launch_weight_image(params, ctx->wimage, nd.wimage_map, nd.lay.P, ctx->stream));
  return 0;
}

// The split kernel's gradient pack: workspace for `n_rec` row records per net and the plane image of the buffer's rows, and the
// image itself (obs_planes_kernel) -- built at the start of every train() / gradient call from the observations as they are
// then.  The row records are written by the advantage-statistics launch that follows (fill_adv_records).
int build_grad_pack(ph_ctx* ctx, const ph::NetDims& nd, const ph_rollout* rb, size_t n_rec) {
  if (nd.split != 1) return 0;   // the one-hot kernel gathers its rows itself (D integers per row: nothing to split ahead)
  const size_t rows = (size_t)rb->T * rb->E;
  const size_t need_img = (rows + 1) * ph::XIMG_ROW_U4;
  if (ctx->capturing) {
    if (need_img > ctx->ximg_cap || n_rec > ctx->rec_cap || 2 * rows > ctx->rowrec_cap)
      return fail("workspace would grow inside graph capture: run the same call once outside capture first");
  } else {
    if (ensure(ctx->ximg, ctx->ximg_cap, need_img)) return 1;
    if (ensure(ctx->rowrec, ctx->rowrec_cap, 2 * rows)) return 1;
    if (n_rec > ctx->rec_cap) {
      size_t c1 = ctx->rec_cap, c2 = ctx->rec_cap;
      if (ensure(ctx->rec_pi, c1, n_rec)) return 1;
      if (ensure(ctx->rec_vf, c2, n_rec)) return 1;
      ctx->rec_cap = n_rec;
    }
  }
  PH_HIP(ph::launch_obs_planes(rb->observations, (int)rows, nd.D, nd.F, ph::grad_fast_fold(nd) ? 1 : 0, ctx->ximg, rb->advantages,
                               rb->log_probs, rb->actions, rb->returns, rb->values, ctx->rowrec, ctx->stream));
  return 0;
}
void fill_adv_records(ph::AdvStatArgs& aa, const ph_ctx* ctx, const ph::NetDims& nd, const ph_rollout* rb) {
  aa.rec_pi_out = nd.split == 1 ? ctx->rec_pi : nullptr;
  aa.rec_vf_out = nd.split == 1 ? ctx->rec_vf : nullptr;
  aa.rowrec = (nd.split == 1 && rb->advantages && rb->log_probs && rb->actions && rb->returns && rb->values) ? ctx->rowrec : nullptr;
  aa.rb_logp = rb->log_probs;
  aa.rb_act = rb->actions;
  aa.rb_ret = rb->returns;
  aa.rb_val = rb->values;
}

// gemm_mode 2 (products as six bf16 MFMA terms over three-plane operands, float32 accuracy) applies to the gradient launches
// of specs ppo_grad_split_kernel takes; everywhere else it means 0.  The split kernel's slabs have their own order.
void select_gemm(ph::NetDims& nd, int gemm_mode) {
  nd.split = (gemm_mode == 2 && nd.slab_map_split != nullptr) ? nd.split_kind : 0;
  if (nd.split) nd.slab_map = nd.slab_map_split;
}

// floats per gradient slab: the canonical parameter layout, the register-order layout of ppo_grad_fast_kernel, or a split kernel's
int slab_len_of(const ph::NetDims& nd) { return nd.split ? nd.slab_len_split : (nd.slab_map ? 2 * ph::RS_NET : nd.lay.P); }

int ensure_train_ws(ph_ctx* ctx, int P, int slab_len, int nwg_max, int n_mb_total, size_t n_idx = 0, size_t n_phys = 0) {
  if (ctx->capturing) {
    if ((size_t)nwg_max * slab_len > ctx->slabs_cap || (size_t)n_mb_total * 2 > ctx->advstats_cap || n_idx > ctx->perm_idx_cap ||
        n_phys > ctx->perm_phys_cap || (size_t)ph::reduce_blocks(slab_len) + 1 > ctx->step_words_cap || !ctx->step_gen)
      return fail("workspace would grow inside graph capture: run the same call once outside capture first");
    return 0;
  }
  if (n_phys && ensure(ctx->perm_phys, ctx->perm_phys_cap, n_phys)) return 1;
  if (ensure(ctx->slabs, ctx->slabs_cap, (size_t)nwg_max * slab_len)) return 1;
  if (ensure(ctx->statpart, ctx->statpart_cap, (size_t)2 * nwg_max * ph::NSTATP)) return 1;
  if (ensure(ctx->grad, ctx->grad_cap, (size_t)P)) return 1;
  if (ensure(ctx->blocksq, ctx->blocksq_cap, (size_t)ph::reduce_blocks(slab_len))) return 1;
  if (ensure(ctx->advstats, ctx->advstats_cap, (size_t)n_mb_total * 2)) return 1;
  if (ensure(ctx->advpart, ctx->advpart_cap, (size_t)n_mb_total * 2 * ph::ADV_SPLIT)) return 1;
  if (n_idx && ensure(ctx->perm_idx, ctx->perm_idx_cap, n_idx)) return 1;
  const size_t n_words = (size_t)ph::reduce_blocks(slab_len) + 1;
  if (n_words > ctx->step_words_cap) {
    if (ensure(ctx->step_words, ctx->step_words_cap, n_words)) return 1;
    PH_HIP(hipMemsetAsync(ctx->step_words, 0, n_words * sizeof(unsigned long long), ctx->stream));
  }
  if (!ctx->step_gen) {
    PH_HIP(hipMalloc((void**)&ctx->step_gen, 4 * sizeof(unsigned int)));   // generation, timed-out waits, -, -
    PH_HIP(hipMemsetAsync(ctx->step_gen, 0, 4 * sizeof(unsigned int), ctx->stream));
  }
  return 0;
}

// reduce + clip + Adam of an exclusive learner's minibatch as ONE launch?  (PH_STEP_FUSED=0 keeps the two launches)
bool step_fused_wanted(const ph_ctx* ctx, int slab_len, bool alone) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_STEP_FUSED");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled && alone && ctx->exclusive && ctx->step_words && ctx->step_gen &&
         ph::step_fused_fits(ph::reduce_blocks(slab_len), slab_len, ctx->num_cu);
}

}  // namespace

namespace {

// one PPO.train() call resolved into launch parameters
struct TrainPlan {
  int alone = 1;          // not part of a joint call (ph_ppo_train_multi chains several learners' launches)
  ph_ctx* ctx;
  ph::NetDims nd;
  const ph_opt_state* opt;
  const ph_rollout* rb;
  const ph_ppo_hyper* hp;
  const int* perms;
  unsigned long long perm_seed;
  float* stats;
  int n_epochs, batch_size, gemm_mode, N, n_mb, P;
  uint32_t hb;
  const ph_adap_loss* adap = nullptr;   // ADAP's context term, or null = plain PPO
};

// ---- ADAP's context term: validation, workspace, the launch next to a minibatch's gradient launch ----
int adap_check(ph_ctx* ctx, const ph::NetDims& nd, const ph_adap_loss* ad, const char* who) {
  const std::string w(who);
  if (nd.obs_kind != PH_SPACE_BOX) return fail(w + ": the context term needs Box observations (observation ++ context)");
  if (ad->context_size <= 0 || ad->context_size >= nd.F) return fail(w + ": context_size must be in (0, observation length)");
  if (ad->num_context_samples < 2 || ad->num_context_samples > ph::ADAP_ROWS)
    return fail(w + ": num_context_samples must be in [2, 16]");
  if (ad->num_state_samples <= 0) return fail(w + ": num_state_samples must be positive");
  if (ad->sampler < PH_CTX_L2 || ad->sampler > PH_CTX_NATURAL_NUMBERS) return fail(w + ": unknown context sampler");
  if (ad->sampler == PH_CTX_NATURAL_NUMBERS && ad->context_size != 1)
    return fail(w + ": the natural_numbers sampler draws (num, 1) contexts: context_size must be 1");
  if (ph::adap_lds_bytes(nd, ad->num_context_samples, ad->context_size) > 160 * 1024)
    return fail(w + ": observation / action space too large for the context kernel's LDS tile");
  const size_t nwg = (size_t)ph::adap_workgroups(ad->num_context_samples, ad->num_state_samples);
  if (ctx->capturing) {
    if (nwg * ph::adap_slab_floats(nd.lay) > ctx->adap_extra_cap || nwg > ctx->adap_loss_cap)
      return fail("workspace would grow inside graph capture: run the same call once outside capture first");
    return 0;
  }
  if (ensure(ctx->adap_extra, ctx->adap_extra_cap, nwg * ph::adap_slab_floats(nd.lay))) return 1;
  if (ensure(ctx->adap_loss, ctx->adap_loss_cap, nwg)) return 1;
  return 0;
}

// the context launch of minibatch number `mbi` (rows idx[0..nb)); fills the reduce launch's additional-term fields
int adap_launch(ph_ctx* ctx, const ph::NetDims& nd, const float* params, const ph_rollout* rb, const ph_adap_loss* ad,
                const int* idx, int nb, int mbi, ph::ReduceArgs* r) {
  const int C = ad->num_context_samples, S = ad->num_state_samples, cs = ad->context_size;
  const int n_states = S < nb ? S : nb;   // util.py:106-108
  const int nwg = ph::adap_workgroups(C, n_states);
  ph::AdapArgs a;
  std::memset(&a, 0, sizeof(a));
  a.nd = nd;
  a.params = params;
  a.rb_obs = rb->observations;
  a.T = rb->T;
  a.E = rb->E;
  a.idx = idx;
  a.nb = nb;
  a.ctx_size = cs;
  a.n_ctx = C;
  a.n_states = n_states;
  a.sampler = ad->sampler;
  a.coef = ad->context_loss_coeff;
  a.state_idx = ad->state_idx ? ad->state_idx + (size_t)mbi * S : nullptr;
  a.contexts = ad->contexts ? ad->contexts + (size_t)mbi * C * cs : nullptr;
  a.seed = ad->seed;
  a.epoch = ctx->rng_epoch;
  a.mbi = (uint32_t)mbi;
  a.nb_hb = ph::feistel_half_bits((uint32_t)nb);
  a.extra = ctx->adap_extra;
  a.loss_part = ctx->adap_loss;
  a.used_state_idx = ad->used_state_idx ? ad->used_state_idx + (size_t)mbi * S : nullptr;
  a.used_contexts = ad->used_contexts ? ad->used_contexts + (size_t)mbi * C * cs : nullptr;
  a.stop_flag = ctx->stop_flag;
  a.prof = ctx->prof;
  PH_HIP(ph::launch_adap_context(a, nwg, ctx->stream));
  r->extra = ctx->adap_extra;
  r->n_extra = nwg;
  r->extra_len = ph::adap_slab_floats(nd.lay);
  r->extra_cut = nd.lay.vf_W1;
  r->extra_lo = nd.lay.act_W;
  r->extra_hi = nd.lay.val_W;
  r->extra_loss = ctx->adap_loss;
  r->extra_norm = 1.0f / (float)((C * (C - 1) / 2) * n_states);
  r->extra_coef = ad->context_loss_coeff;
  r->extra_loss_out = ad->context_loss ? ad->context_loss + mbi : nullptr;
  return 0;
}

// the advantage-statistics launch of a train() call: statistics of every minibatch, the minibatch order, the row records
void train_adv_args(const TrainPlan& t, bool need_idx, ph::AdvStatArgs& aa) {
  ph_ctx* ctx = t.ctx;
  aa.clear_flag = ctx->stop_flag;    // the call's KL stop flag starts at 0 (no launch of its own: nothing reads it before the first gradient launch)
  aa.rb_adv = t.rb->advantages;
  aa.T = t.rb->T;
  aa.E = t.rb->E;
  aa.perms = t.perms;
  aa.perm_n = (uint32_t)t.N;
  aa.perm_hb = ph::feistel_half_bits((uint32_t)t.N);
  aa.perm_seed = t.perm_seed;
  aa.epoch = ctx->rng_epoch;
  aa.N = t.N;
  aa.batch = t.batch_size;
  aa.n_mb = t.n_mb;
  aa.out = ctx->advstats;
  aa.partial = ctx->advpart;
  aa.idx_out = t.perms ? nullptr : ctx->perm_idx;
  aa.phys_out = ctx->perm_phys;      // the order once more as physical rows: the tile walk then has no index arithmetic
  // the split kernel reads the row records only: the materialised order (8 of the launch's 40 bytes per element) is written
  // for whoever else walks it -- the other gradient kernels, ADAP's context launch (need_idx)
  if (t.nd.split == 1 && !need_idx) aa.idx_out = aa.phys_out = nullptr;
  fill_adv_records(aa, ctx, t.nd, t.rb);
}

// validation, workspace, stop-flag reset and the advantage statistics / minibatch order of every minibatch
int train_prepare(TrainPlan& t, ph_ctx* ctx, const ph_spec* spec, const ph_opt_state* opt, const ph_rollout* rb,
                  const ph_ppo_hyper* hp, int n_epochs, int batch_size, const int* perms, unsigned long long perm_seed,
                  float* stats, int gemm_mode, bool need_idx = false) {
  if (!ctx) return fail("null ctx");
  if (!opt || !opt->params || !opt->adam_m || !opt->adam_v || !opt->step) return fail("ph_ppo_train: null optimizer state");
  if ((uintptr_t)opt->params % 16 != 0) return fail("ph_ppo_train: params must be 16-byte aligned");
  if (!hp) return fail("ph_ppo_train: null hyper-parameters");
  if (check_rb(rb)) return 1;
  if (n_epochs <= 0 || batch_size <= 0) return fail("ph_ppo_train: n_epochs and batch_size must be positive");
  if (resolve(ctx, spec, &t.nd, true)) return 1;
  select_gemm(t.nd, gemm_mode);
  t.ctx = ctx;
  t.opt = opt;
  t.rb = rb;
  t.hp = hp;
  t.perms = perms;
  t.perm_seed = perm_seed;
  t.stats = stats;
  t.n_epochs = n_epochs;
  t.batch_size = batch_size;
  t.gemm_mode = gemm_mode;
  t.N = rb->T * rb->E;
  t.n_mb = (t.N + batch_size - 1) / batch_size;
  t.P = t.nd.lay.P;
  const MbPlan big = plan_minibatch(ctx, t.nd, batch_size < t.N ? batch_size : t.N);
  if (ensure_train_ws(ctx, t.P, slab_len_of(t.nd), big.nwg, n_epochs * t.n_mb, perms ? 0 : (size_t)n_epochs * t.N,
                      (size_t)n_epochs * t.N))
    return 1;
  hipStream_t s = ctx->stream;
  if (rebuild_weight_image(ctx, t.nd, opt->params)) return 1;
  if (build_grad_pack(ctx, t.nd, rb, (size_t)n_epochs * t.N)) return 1;
  t.hb = ph::feistel_half_bits((uint32_t)t.N);
  ph::AdvStatArgs aa;
  train_adv_args(t, need_idx, aa);
  PH_HIP(ph::launch_adv_stats(aa, n_epochs * t.n_mb, s));
  return 0;
}

void fill_step_args(const TrainPlan& t, int mbi, const MbPlan& pl, ph::ReduceArgs& r, ph::AdamArgs& ad);
// ~2 s of wall_clock64 ticks (100 MHz): the bound of a wait for a workgroup that cannot be scheduled, far above any healthy one
constexpr unsigned long long STEP_WAIT_TICKS = 200000000ull;

// gradient launch of minibatch mbi = ep * n_mb + k
int train_launch_grad(const TrainPlan& t, int mbi, MbPlan* pl_out) {
  ph_ctx* ctx = t.ctx;
  const int ep = mbi / t.n_mb, k = mbi - ep * t.n_mb;
  const int start = k * t.batch_size;
  const int nb = (t.N - start < t.batch_size) ? t.N - start : t.batch_size;
  const MbPlan pl = plan_minibatch(ctx, t.nd, nb);
  ph::GradArgs g;
  std::memset(&g, 0, sizeof(g));
  fill_grad_args(g, t.nd, t.opt->params, t.rb, t.hp, ctx);
  g.idx = (t.perms ? t.perms : ctx->perm_idx) + (size_t)ep * t.N + start;
  g.idx_phys = ctx->perm_phys + (size_t)ep * t.N + start;
  if (t.nd.split == 1) {
    g.rec_pi = ctx->rec_pi + (size_t)ep * t.N + start;
    g.rec_vf = ctx->rec_vf + (size_t)ep * t.N + start;
  }
  g.perm_n = (uint32_t)t.N;
  g.perm_hb = t.hb;
  g.perm_seed = t.perm_seed;
  g.perm_epoch = ep;
  g.epoch = ctx->rng_epoch;
  g.mb_start = start;
  g.nb = nb;
  g.advstats = ctx->advstats + 2 * (size_t)mbi;
  g.ntiles = pl.ntiles;
  *pl_out = pl;
  PH_HIP(ph::launch_ppo_grad(g, pl.nwg, t.gemm_mode, ctx->stream));
  return 0;
}

// the argument records of minibatch mbi's reduce (+ statistics, KL decision) and clip + Adam
void fill_step_args(const TrainPlan& t, int mbi, const MbPlan& pl, ph::ReduceArgs& r, ph::AdamArgs& ad) {
  ph_ctx* ctx = t.ctx;
  r.slabs = ctx->slabs;
  r.nslab = pl.nwg;
  r.nstatpart = 2 * pl.nwg;
  r.P = t.P;
  r.slab_len = slab_len_of(t.nd);
  r.map = t.nd.slab_map;
  r.grad = ctx->grad;
  r.blocksq = ctx->blocksq;
  r.statpart = ctx->statpart;
  r.stats_out = t.stats ? t.stats + (size_t)mbi * PH_NSTAT : nullptr;
  r.nb = pl.nb;
  r.ent_coef = t.hp->ent_coef;
  r.vf_coef = t.hp->vf_coef;
  r.target_kl = t.hp->target_kl;
  r.stop_flag = ctx->stop_flag;
  r.step = t.opt->step;
  r.scalars = ctx->scalars;
  r.wide = t.alone && ctx->exclusive;
  ad.params = t.opt->params;
  ad.m = t.opt->adam_m;
  ad.v = t.opt->adam_v;
  ad.grad = ctx->grad;
  ad.blocksq = ctx->blocksq;
  ad.nblk = ph::reduce_blocks(slab_len_of(t.nd));
  ad.P = t.P;
  ad.step = t.opt->step;
  ad.scalars = ctx->scalars;
  ad.stop_flag = ctx->stop_flag;
  ad.lr = t.hp->learning_rate;
  ad.beta1 = t.hp->adam_beta1;
  ad.beta2 = t.hp->adam_beta2;
  ad.eps = t.hp->adam_eps;
  ad.max_norm = t.hp->max_grad_norm;
  ad.stats_out = r.stats_out;
  ad.wimage = t.nd.split ? ctx->wimage : nullptr;
  ad.wimage_map = t.nd.wimage_map;
}

// slab reduction + statistics + KL decision, then clip + Adam, of the minibatch whose gradient launch returned `pl`
int train_launch_step(const TrainPlan& t, int mbi, const MbPlan& pl) {
  ph_ctx* ctx = t.ctx;
  hipStream_t s = ctx->stream;
  ph::ReduceArgs r;
  ph::AdamArgs ad;
  fill_step_args(t, mbi, pl, r, ad);
  if (t.adap) {
    const int ep = mbi / t.n_mb, start = (mbi - ep * t.n_mb) * t.batch_size;
    const int* idx = (t.perms ? t.perms : ctx->perm_idx) + (size_t)ep * t.N + start;
    if (adap_launch(ctx, t.nd, t.opt->params, t.rb, t.adap, idx, pl.nb, mbi, &r)) return 1;
  }
  const bool fused = step_fused_wanted(ctx, slab_len_of(t.nd), t.alone != 0);
#ifdef PH_EXPERIMENT_SKIP_STEP   // TIMING EXPERIMENTS ONLY (never in the default build; scripts/build_variants.sh): what the gradient launches
  (void)fused;                   // of two learners cost each other WITHOUT the other learner's reduce / Adam kernels beside them
  if (PH_EXPERIMENT_SKIP_STEP == 1) return 0;                                        // 1: neither kernel
  if (PH_EXPERIMENT_SKIP_STEP == 2) { PH_HIP(ph::launch_ppo_reduce(r, s)); return 0; }   // 2: the reduction only
  if (PH_EXPERIMENT_SKIP_STEP == 3) { PH_HIP(ph::launch_ppo_adam(ad, s)); return 0; }    // 3: clip + Adam only (on a stale gradient)
#endif                           // (profiles/r06_bn_*)
  if (fused) {
    PH_HIP(ph::launch_ppo_step(r, ad, ctx->step_words, ctx->step_gen, ctx->step_gen + 1, STEP_WAIT_TICKS, s));
  } else {
    PH_HIP(ph::launch_ppo_reduce(r, s));
    PH_HIP(ph::launch_ppo_adam(ad, s));
  }
  return 0;
}

}  // namespace

namespace {
int train_run(ph_ctx* ctx, const ph_spec* spec, const ph_opt_state* opt, const ph_rollout* rb, const ph_ppo_hyper* hp,
              int n_epochs, int batch_size, const int* perms, unsigned long long perm_seed, float* stats, int gemm_mode,
              const ph_adap_loss* adap) {
  TrainPlan t;
  if (train_prepare(t, ctx, spec, opt, rb, hp, n_epochs, batch_size, perms, perm_seed, stats, gemm_mode, adap != nullptr)) return 1;
  if (adap && t.nd.gauss) return fail("ph_adap_train: the context term is written for the categorical heads");
  if (adap && adap_check(ctx, t.nd, adap, "ph_adap_train")) return 1;
  t.adap = adap;
  for (int mbi = 0; mbi < n_epochs * t.n_mb; ++mbi) {
    MbPlan pl;
    if (train_launch_grad(t, mbi, &pl)) return 1;
    if (train_launch_step(t, mbi, pl)) return 1;
  }
  return 0;
}
}  // namespace

int ph_ppo_train(ph_ctx* ctx, const ph_spec* spec, const ph_opt_state* opt, const ph_rollout* rb,
                 const ph_ppo_hyper* hp, int n_epochs, int batch_size, const int* perms,
                 unsigned long long perm_seed, float* stats, int gemm_mode) {
  DevGuard dev_guard(ctx);
  return train_run(ctx, spec, opt, rb, hp, n_epochs, batch_size, perms, perm_seed, stats, gemm_mode, nullptr);
}

int ph_adap_train(ph_ctx* ctx, const ph_spec* spec, const ph_opt_state* opt, const ph_rollout* rb,
                  const ph_ppo_hyper* hp, int n_epochs, int batch_size, const int* perms,
                  unsigned long long perm_seed, float* stats, int gemm_mode, const ph_adap_loss* adap) {
  DevGuard dev_guard(ctx);
  if (!adap) return fail("ph_adap_train: null context-term description");
  return train_run(ctx, spec, opt, rb, hp, n_epochs, batch_size, perms, perm_seed, stats, gemm_mode, adap);
}

int ph_ppo_train_multi(const ph_train_call* calls, int n_calls) {
  if (!calls || n_calls <= 0) return fail("ph_ppo_train_multi: no calls");
  if (n_calls > PH_MAX_TRAIN_CALLS) return fail("ph_ppo_train_multi: too many calls");
  TrainPlan t[PH_MAX_TRAIN_CALLS];
  int total[PH_MAX_TRAIN_CALLS], longest = 0;
  for (int k = 0; k < n_calls; ++k) {
    const ph_train_call& c = calls[k];
    if (train_prepare(t[k], c.ctx, c.spec, c.opt, c.rb, c.hyper, c.n_epochs, c.batch_size, c.perms, c.perm_seed, c.stats,
                      c.gemm_mode))
      return 1;
    t[k].alone = n_calls == 1;
    if (!t[k].ctx->ev_grad) PH_HIP(hipEventCreateWithFlags(&t[k].ctx->ev_grad, hipEventDisableTiming));
    total[k] = c.n_epochs * t[k].n_mb;
    longest = total[k] > longest ? total[k] : longest;
  }
  // Round-robin over the learners, minibatch by minibatch.  Gradient launches fill the whole device, so two of them side by
  // side only slow each other down; chaining them (each waits for the previous learner's gradient launch) lets one
  // learner's small reduce / Adam launches run in the shadow of the next learner's gradient launch.
  // (Round 4 measured the alternatives on MI355X, 2 learners, whole iterations as hipGraphs: this chaining 3.62 ms; every gradient
  // launch on ONE queue with each learner's reduce / Adam on a side stream tied in by two events per minibatch 3.87 ms; NO
  // chaining at all -- one linear graph per learner, free-running, bench.py's default -- 2.74 ms.  Cross-stream edges inside a
  // graph cost far more than the overlap they arrange; the hardware's own interleaving of two independent queues wins.)
  hipEvent_t prev = nullptr;
  hipStream_t prev_stream = nullptr;
  for (int mbi = 0; mbi < longest; ++mbi) {
    for (int k = 0; k < n_calls; ++k) {
      if (mbi >= total[k]) continue;
      hipStream_t s = t[k].ctx->stream;
      if (prev && prev_stream != s) PH_HIP(hipStreamWaitEvent(s, prev, 0));
      MbPlan pl;
      if (train_launch_grad(t[k], mbi, &pl)) return 1;
      PH_HIP(hipEventRecord(t[k].ctx->ev_grad, s));
      prev = t[k].ctx->ev_grad;
      prev_stream = s;
      if (train_launch_step(t[k], mbi, pl)) return 1;
    }
  }
  return 0;
}

namespace {
int minibatch_grad_run(ph_ctx* ctx, const ph_spec* spec, const float* params, const ph_rollout* rb, const ph_ppo_hyper* hp,
                       const int* indices, int nb, float* grad_out, float* stats_out, int gemm_mode,
                       const ph_adap_loss* adap);
}  // namespace

int ph_ppo_minibatch_grad(ph_ctx* ctx, const ph_spec* spec, const float* params, const ph_rollout* rb,
                          const ph_ppo_hyper* hp, const int* indices, int nb, float* grad_out, float* stats_out,
                          int gemm_mode) {
  DevGuard dev_guard(ctx);
  return minibatch_grad_run(ctx, spec, params, rb, hp, indices, nb, grad_out, stats_out, gemm_mode, nullptr);
}

int ph_adap_minibatch_grad(ph_ctx* ctx, const ph_spec* spec, const float* params, const ph_rollout* rb,
                           const ph_ppo_hyper* hp, const int* indices, int nb, float* grad_out, float* stats_out,
                           int gemm_mode, const ph_adap_loss* adap) {
  DevGuard dev_guard(ctx);
  if (!adap) return fail("ph_adap_minibatch_grad: null context-term description");
  return minibatch_grad_run(ctx, spec, params, rb, hp, indices, nb, grad_out, stats_out, gemm_mode, adap);
}

namespace {
int minibatch_grad_run(ph_ctx* ctx, const ph_spec* spec, const float* params, const ph_rollout* rb, const ph_ppo_hyper* hp,
                       const int* indices, int nb, float* grad_out, float* stats_out, int gemm_mode,
                       const ph_adap_loss* adap) {
  if (!ctx) return fail("null ctx");
  if (!params || !hp || !indices || !grad_out) return fail("ph_ppo_minibatch_grad: null argument");
  if ((uintptr_t)params % 16 != 0) return fail("ph_ppo_minibatch_grad: params must be 16-byte aligned");
  if (check_rb(rb)) return 1;
  if (nb <= 0) return fail("ph_ppo_minibatch_grad: nb must be positive");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd, adap == nullptr)) return 1;
  select_gemm(nd, gemm_mode);
  const int P = nd.lay.P;
  const MbPlan pl = plan_minibatch(ctx, nd, nb);
  if (ensure_train_ws(ctx, P, slab_len_of(nd), pl.nwg, 1, 0, (size_t)nb)) return 1;
  hipStream_t s = ctx->stream;
  PH_HIP(ph::launch_set_int(ctx->stop_flag, 0, s));
  if (rebuild_weight_image(ctx, nd, params)) return 1;
  if (build_grad_pack(ctx, nd, rb, (size_t)nb)) return 1;
  ph::AdvStatArgs aa;
  aa.rb_adv = rb->advantages;
  aa.T = rb->T;
  aa.E = rb->E;
  aa.perms = indices;  // one "epoch" whose first nb entries are the minibatch
  aa.perm_n = 0;
  aa.perm_hb = 1;
  aa.perm_seed = 0;
  aa.epoch = nullptr;
  aa.N = nb;
  aa.batch = nb;
  aa.n_mb = 1;
  aa.out = ctx->advstats;
  aa.partial = ctx->advpart;
  aa.idx_out = nullptr;
  aa.phys_out = ctx->perm_phys;
  fill_adv_records(aa, ctx, nd, rb);
  PH_HIP(ph::launch_adv_stats(aa, 1, s));
  ph::GradArgs g;
  std::memset(&g, 0, sizeof(g));
  fill_grad_args(g, nd, params, rb, hp, ctx);
  g.idx = indices;
  g.idx_phys = ctx->perm_phys;
  g.nb = nb;
  g.advstats = ctx->advstats;
  g.ntiles = pl.ntiles;
  PH_HIP(ph::launch_ppo_grad(g, pl.nwg, gemm_mode, s));
  ph::ReduceArgs r;
  r.slabs = ctx->slabs;
  r.nslab = pl.nwg;
  r.nstatpart = 2 * pl.nwg;
  r.P = P;
  r.slab_len = slab_len_of(nd);
  r.map = nd.slab_map;
  r.grad = grad_out;
  r.blocksq = ctx->blocksq;
  r.statpart = ctx->statpart;
  r.stats_out = stats_out;
  r.nb = nb;
  r.ent_coef = hp->ent_coef;
  r.vf_coef = hp->vf_coef;
  r.target_kl = -1.f;
  r.stop_flag = ctx->stop_flag;
  r.step = nullptr;
  r.scalars = ctx->scalars;
  if (adap) {
    if (adap_check(ctx, nd, adap, "ph_adap_minibatch_grad")) return 1;
    if (adap_launch(ctx, nd, params, rb, adap, indices, nb, 0, &r)) return 1;
  }
  PH_HIP(ph::launch_ppo_reduce(r, s));
  return 0;
}
}  // namespace

// ---- AdapPolicyMult (adap/policies.py:136-283; kernels: ph_adapmult.hip) ------------------------------------------------------
int ph_adapmult_layout_of(const ph_spec* spec, int C, ph_adapmult_layout* o) {
  if (!o) return fail("ph_adapmult_layout_of: null layout");
  ph_layout big;
  if (layout_of(spec, &big)) return 1;
  if (spec->obs.kind != PH_SPACE_BOX) return fail("AdapPolicyMult: rows are Box rows (features ++ context)");
  if (big.A != 1 || big.L > 8) return fail("AdapPolicyMult: one Discrete head of at most 8 logits");
  if (C < 1 || C > 4) return fail("AdapPolicyMult: context_size must be in [1, 4]");
  if (big.F - C < 1 || big.F - C > 64) return fail("AdapPolicyMult: 1 .. 64 features besides the context");
  const int H = PH_HIDDEN;
  o->Fo = big.F - C;
  o->C = C;
  o->L = big.L;
  int off = 0;
  o->pi_W1 = off; off += o->Fo * H;
  o->pi_b1 = off; off += H;
  o->pi_Ws = off; off += H * H * C;
  o->pi_bs = off; off += H * C;
  o->pi_W2 = off; off += H * H;
  o->pi_b2 = off; off += H;
  o->vf_W1 = off; off += o->Fo * H;
  o->vf_b1 = off; off += H;
  o->vf_Ws = off; off += H * H * C;
  o->vf_bs = off; off += H * C;
  o->vf_W2 = off; off += H * H;
  o->vf_b2 = off; off += H;
  o->act_W = off; off += H * o->L;
  o->act_b = off; off += o->L;
  o->val_W = off; off += H;
  o->val_b = off; off += 1;
  o->P = off;
  return 0;
}

namespace {
// the dense intermediates for `rows` rows of width D, carved out of one block (every array starts on a 64-float boundary)
int am_work(ph_ctx* ctx, const ph_adapmult_layout& L, int D, int rows, ph::AmWork* w) {
  const size_t H = PH_HIDDEN, C = (size_t)L.C, R = (size_t)rows;
  auto up = [](size_t n) { return (n + 63) & ~(size_t)63; };
  const size_t sizes[] = {R * H, R * H * C, R * H, R * H, R * 8, R, R * 8, R, R * H, R * H, R * H * C, R * (size_t)D,
                          R, R, R, R, R};
  size_t total = 0;
  for (size_t n : sizes) total += up(n);
  if (total > ctx->am_buf_cap) {
    if (ctx->capturing) return fail("workspace would grow inside graph capture: run the same call once outside capture first");
    if (ensure(ctx->am_buf, ctx->am_buf_cap, total)) return 1;
  }
  float* p = ctx->am_buf;
  float** fields[] = {&w->x, &w->xa, &w->y, &w->h, &w->z, &w->v, &w->dz, &w->dv, &w->dzh, &w->dy, &w->dza, &w->xg,
                      &w->act, &w->oldlp, &w->adv, &w->ret, &w->oldv};
  for (size_t i = 0; i < sizeof(sizes) / sizeof(sizes[0]); ++i) {
    *fields[i] = p;
    p += up(sizes[i]);
  }
  return 0;
}
int am_nslab(int nb) {
  const int n = (nb + 15) / 16;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}
// PPO loss gradient of rows indices[0..nb) into ctx->slabs (nslab slabs of P floats, canonical order) + statistics partials; with
// `adap` the context term's slabs and loss shares as well (fields of *r).  advstats: {mean, std} of the minibatch's advantages.
int am_minibatch(ph_ctx* ctx, const ph_adapmult_layout& L, int D, const float* params, const ph_rollout* rb, const ph_ppo_hyper* hp,
                 const int* indices, int nb, const float* advstats, const ph_adap_loss* adap, int mbi, ph::ReduceArgs* r) {
  hipStream_t s = ctx->stream;
  const int S = adap ? adap->num_state_samples : 0, Cs = adap ? adap->num_context_samples : 0;
  const int n_states = adap ? (S < nb ? S : nb) : 0;
  const int rows_max = nb > n_states * Cs ? nb : n_states * Cs;
  ph::AmWork w;
  if (am_work(ctx, L, D, rows_max, &w)) return 1;
  const int nslab = am_nslab(nb);
  ph::AmGather g;
  g.rb_obs = rb->observations;
  g.rb_act = rb->actions;
  g.rb_logp = rb->log_probs;
  g.rb_adv = rb->advantages;
  g.rb_ret = rb->returns;
  g.rb_val = rb->values;
  g.idx = indices;
  g.nb = nb;
  g.T = rb->T;
  g.E = rb->E;
  g.D = D;
  g.norm_adv = hp->normalize_advantage;
  g.advstats = advstats;
  g.xg = w.xg;
  g.act = w.act;
  g.oldlp = w.oldlp;
  g.adv = w.adv;
  g.ret = w.ret;
  g.oldv = w.oldv;
  PH_HIP(ph::launch_am_gather(g, s));
  for (int net = 0; net < 2; ++net) {
    PH_HIP(ph::am_forward_net(L, params, net, w.xg, D, nb, w, s));
    PH_HIP(ph::launch_am_loss(w, L.L, nb, *hp, ctx->statpart, nslab, net, s));
    PH_HIP(ph::am_backward_net(L, params, net, w.xg, D, nb, w, ctx->slabs, nslab, L.P, net == 0 ? L.act_W : L.val_W,
                               net == 0 ? L.act_b : L.val_b, s));
  }
  r->slabs = ctx->slabs;
  r->nslab = nslab;
  r->nstatpart = 2 * nslab;
  r->P = L.P;
  r->slab_len = L.P;
  r->map = nullptr;
  if (adap) {   // adap/util.py:97-131: n_states sampled states under Cs sampled contexts, policy net only
    const int cs = adap->context_size, extra_len = L.vf_W1 + (L.val_W - L.act_W);
    if (ensure(ctx->adap_extra, ctx->adap_extra_cap, (size_t)n_states * extra_len)) return 1;
    if (ensure(ctx->adap_loss, ctx->adap_loss_cap, (size_t)n_states)) return 1;
    ph::AmCtx a;
    std::memset(&a, 0, sizeof(a));
    a.rb_obs = rb->observations;
    a.idx = indices;
    a.nb = nb;
    a.T = rb->T;
    a.E = rb->E;
    a.D = D;
    a.ctx_size = cs;
    a.n_ctx = Cs;
    a.n_states = n_states;
    a.sampler = adap->sampler;
    a.state_idx = adap->state_idx ? adap->state_idx + (size_t)mbi * S : nullptr;
    a.contexts = adap->contexts ? adap->contexts + (size_t)mbi * Cs * cs : nullptr;
    a.seed = adap->seed;
    a.epoch = ctx->rng_epoch;
    a.mbi = (uint32_t)mbi;
    a.nb_hb = ph::feistel_half_bits((uint32_t)nb);
    a.rows = w.xg;
    a.used_state_idx = adap->used_state_idx ? adap->used_state_idx + (size_t)mbi * S : nullptr;
    a.used_contexts = adap->used_contexts ? adap->used_contexts + (size_t)mbi * Cs * cs : nullptr;
    PH_HIP(ph::launch_am_ctx_rows(a, s));
    const int R = n_states * Cs, npairs = Cs * (Cs - 1) / 2;
    PH_HIP(ph::am_forward_net(L, params, 0, w.xg, D, R, w, s));
    PH_HIP(ph::launch_am_ctx_loss(w.z, L.L, n_states, Cs, adap->context_loss_coeff / (float)(npairs * n_states), w.dz,
                                  ctx->adap_loss, s));
    // one slab per state (its Cs rows): policy-side parameters at their own offsets, the action head behind them
    PH_HIP(ph::am_backward_net(L, params, 0, w.xg, D, R, w, ctx->adap_extra, n_states, extra_len, L.vf_W1,
                               L.vf_W1 + (L.act_b - L.act_W), s));
    r->extra = ctx->adap_extra;
    r->n_extra = n_states;
    r->extra_len = extra_len;
    r->extra_cut = L.vf_W1;
    r->extra_lo = L.act_W;
    r->extra_hi = L.val_W;
    r->extra_loss = ctx->adap_loss;
    r->extra_norm = 1.0f / (float)(npairs * n_states);
    r->extra_coef = adap->context_loss_coeff;
    r->extra_loss_out = adap->context_loss ? adap->context_loss + mbi : nullptr;
  }
  return 0;
}
int am_adap_ok(const ph_adapmult_layout& L, const ph_adap_loss* ad, const char* who) {
  const std::string w(who);
  if (ad->context_size != L.C) return fail(w + ": the context term's context_size differs from the policy's");
  if (ad->num_context_samples < 2 || ad->num_context_samples > ph::ADAP_ROWS) return fail(w + ": num_context_samples must be in [2, 16]");
  if (ad->num_state_samples <= 0 || ad->num_state_samples > 256) return fail(w + ": num_state_samples must be in [1, 256]");
  if (ad->sampler < PH_CTX_L2 || ad->sampler > PH_CTX_NATURAL_NUMBERS) return fail(w + ": unknown context sampler");
  if (ad->sampler == PH_CTX_NATURAL_NUMBERS && ad->context_size != 1)
    return fail(w + ": the natural_numbers sampler draws (num, 1) contexts: context_size must be 1");
  return 0;
}
}  // namespace

int ph_adapmult_forward(ph_ctx* ctx, const ph_spec* spec, int context_size, const float* params, const float* obs, int n,
                        const unsigned char* action_mask, const float* uniforms, const float* given_actions,
                        unsigned long long seed, unsigned long long counter, int deterministic, int* actions_i32,
                        float* actions_f32, float* values, float* log_probs, float* entropy, float* logits, const ph_rollout* rb,
                        int pos, const float* episode_start_in) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs) return fail("ph_adapmult_forward: null params/obs");
  if (n <= 0) return fail("ph_adapmult_forward: n must be positive");
  ph_adapmult_layout L;
  if (ph_adapmult_layout_of(spec, context_size, &L)) return 1;
  ph::FwdArgs a;
  std::memset(&a, 0, sizeof(a));
  if (resolve(ctx, spec, &a.nd)) return 1;
  a.params = params;
  a.obs = obs;
  a.n = n;
  a.mask = action_mask;
  a.uniforms = uniforms;
  a.given_actions = given_actions;
  a.seed = seed;
  a.counter = counter;
  a.epoch = ctx->rng_epoch;
  a.deterministic = deterministic;
  a.act_i32 = actions_i32;
  a.act_f32 = actions_f32;
  a.values = values;
  a.logp = log_probs;
  a.entropy = entropy;
  a.logits = logits;
  if (rb) {
    if (check_rb(rb)) return 1;
    if (n != rb->E) return fail("ph_adapmult_forward: fused add needs n == rollout E");
    if (pos < 0 || pos >= rb->T) return fail("ph_adapmult_forward: pos out of range (buffer full?)");
    if (!episode_start_in) return fail("ph_adapmult_forward: fused add needs episode_start_in");
    const size_t row = (size_t)pos * rb->E;
    a.rb_obs = rb->observations + row * a.nd.D;
    a.rb_act = rb->actions + row * a.nd.A;
    a.rb_rew = rb->rewards + row;
    a.rb_es = rb->episode_starts + row;
    a.rb_val = rb->values + row;
    a.rb_logp = rb->log_probs + row;
    a.es_in = episode_start_in;
  }
  ph::AmWork w;
  if (am_work(ctx, L, a.nd.D, n, &w)) return 1;
  PH_HIP(ph::am_forward_net(L, params, 0, obs, a.nd.D, n, w, ctx->stream));
  PH_HIP(ph::am_forward_net(L, params, 1, obs, a.nd.D, n, w, ctx->stream));
  PH_HIP(ph::launch_am_act(a, w.z, w.v, ctx->stream));
  return 0;
}

int ph_adapmult_minibatch_grad(ph_ctx* ctx, const ph_spec* spec, int context_size, const float* params, const ph_rollout* rb,
                               const ph_ppo_hyper* hp, const int* indices, int nb, float* grad_out, float* stats_out,
                               const ph_adap_loss* adap) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !hp || !indices || !grad_out) return fail("ph_adapmult_minibatch_grad: null argument");
  if (check_rb(rb)) return 1;
  if (nb <= 0) return fail("ph_adapmult_minibatch_grad: nb must be positive");
  ph_adapmult_layout L;
  if (ph_adapmult_layout_of(spec, context_size, &L)) return 1;
  if (adap && am_adap_ok(L, adap, "ph_adapmult_minibatch_grad")) return 1;
  const int nslab = am_nslab(nb);
  if (ensure_train_ws(ctx, L.P, L.P, nslab, 1, 0, (size_t)nb)) return 1;
  hipStream_t s = ctx->stream;
  PH_HIP(ph::launch_set_int(ctx->stop_flag, 0, s));
  ph::AdvStatArgs aa;
  aa.rb_adv = rb->advantages;
  aa.T = rb->T;
  aa.E = rb->E;
  aa.perms = indices;  // one "epoch" whose first nb entries are the minibatch
  aa.perm_n = 0;
  aa.perm_hb = 1;
  aa.perm_seed = 0;
  aa.epoch = nullptr;
  aa.N = nb;
  aa.batch = nb;
  aa.n_mb = 1;
  aa.out = ctx->advstats;
  aa.partial = ctx->advpart;
  aa.idx_out = nullptr;
  aa.phys_out = ctx->perm_phys;
  PH_HIP(ph::launch_adv_stats(aa, 1, s));
  ph::ReduceArgs r;
  if (am_minibatch(ctx, L, spec->obs.n, params, rb, hp, indices, nb, ctx->advstats, adap, 0, &r)) return 1;
  r.grad = grad_out;
  r.blocksq = ctx->blocksq;
  r.statpart = ctx->statpart;
  r.stats_out = stats_out;
  r.nb = nb;
  r.ent_coef = hp->ent_coef;
  r.vf_coef = hp->vf_coef;
  r.target_kl = -1.f;
  r.stop_flag = ctx->stop_flag;
  r.step = nullptr;
  r.scalars = ctx->scalars;
  PH_HIP(ph::launch_ppo_reduce(r, s));
  return 0;
}

int ph_adapmult_train(ph_ctx* ctx, const ph_spec* spec, int context_size, const ph_opt_state* opt, const ph_rollout* rb,
                      const ph_ppo_hyper* hp, int n_epochs, int batch_size, const int* perms, unsigned long long perm_seed,
                      float* stats, const ph_adap_loss* adap) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!opt || !opt->params || !opt->adam_m || !opt->adam_v || !opt->step) return fail("ph_adapmult_train: null optimizer state");
  if (!hp || !adap) return fail("ph_adapmult_train: null hyper-parameters / context-term description");
  if (check_rb(rb)) return 1;
  if (n_epochs <= 0 || batch_size <= 0) return fail("ph_adapmult_train: n_epochs and batch_size must be positive");
  ph_adapmult_layout L;
  if (ph_adapmult_layout_of(spec, context_size, &L)) return 1;
  if (am_adap_ok(L, adap, "ph_adapmult_train")) return 1;
  const int N = rb->T * rb->E, n_mb = (N + batch_size - 1) / batch_size;
  const int nb_max = batch_size < N ? batch_size : N;
  if (ensure_train_ws(ctx, L.P, L.P, am_nslab(nb_max), n_epochs * n_mb, perms ? 0 : (size_t)n_epochs * N, (size_t)n_epochs * N))
    return 1;
  hipStream_t s = ctx->stream;
  const uint32_t hb = ph::feistel_half_bits((uint32_t)N);
  ph::AdvStatArgs aa;
  aa.rb_adv = rb->advantages;
  aa.T = rb->T;
  aa.E = rb->E;
  aa.perms = perms;
  aa.perm_n = (uint32_t)N;
  aa.perm_hb = hb;
  aa.perm_seed = perm_seed;
  aa.epoch = ctx->rng_epoch;
  aa.N = N;
  aa.batch = batch_size;
  aa.n_mb = n_mb;
  aa.out = ctx->advstats;
  aa.partial = ctx->advpart;
  aa.idx_out = perms ? nullptr : ctx->perm_idx;
  aa.phys_out = ctx->perm_phys;
  aa.clear_flag = ctx->stop_flag;
  PH_HIP(ph::launch_adv_stats(aa, n_epochs * n_mb, s));
  for (int mbi = 0; mbi < n_epochs * n_mb; ++mbi) {
    const int ep = mbi / n_mb, k = mbi - ep * n_mb, start = k * batch_size;
    const int nb = (N - start < batch_size) ? N - start : batch_size;
    const int* idx = (perms ? perms : ctx->perm_idx) + (size_t)ep * N + start;
    ph::ReduceArgs r;
    if (am_minibatch(ctx, L, spec->obs.n, opt->params, rb, hp, idx, nb, ctx->advstats + 2 * (size_t)mbi, adap, mbi, &r)) return 1;
    r.grad = ctx->grad;
    r.blocksq = ctx->blocksq;
    r.statpart = ctx->statpart;
    r.stats_out = stats ? stats + (size_t)mbi * PH_NSTAT : nullptr;
    r.nb = nb;
    r.ent_coef = hp->ent_coef;
    r.vf_coef = hp->vf_coef;
    r.target_kl = hp->target_kl;
    r.stop_flag = ctx->stop_flag;
    r.step = opt->step;
    r.scalars = ctx->scalars;
    PH_HIP(ph::launch_ppo_reduce(r, s));
    ph::AdamArgs ad;
    ad.params = opt->params;
    ad.m = opt->adam_m;
    ad.v = opt->adam_v;
    ad.grad = ctx->grad;
    ad.blocksq = ctx->blocksq;
    ad.nblk = ph::reduce_blocks(L.P);
    ad.P = L.P;
    ad.step = opt->step;
    ad.scalars = ctx->scalars;
    ad.stop_flag = ctx->stop_flag;
    ad.lr = hp->learning_rate;
    ad.beta1 = hp->adam_beta1;
    ad.beta2 = hp->adam_beta2;
    ad.eps = hp->adam_eps;
    ad.max_norm = hp->max_grad_norm;
    ad.stats_out = r.stats_out;
    ad.wimage = nullptr;
    ad.wimage_map = nullptr;
    PH_HIP(ph::launch_ppo_adam(ad, s));
  }
  return 0;
}

int ph_bench_ppo_grad(ph_ctx* ctx, const ph_spec* spec, const float* params, const ph_rollout* rb,
                      const ph_ppo_hyper* hp, int batch_size, int reps, int gemm_mode, float* avg_ms_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !params || !hp || !avg_ms_out) return fail("ph_bench_ppo_grad: null argument");
  if (check_rb(rb)) return 1;
  if (batch_size <= 0 || reps <= 0) return fail("ph_bench_ppo_grad: bad sizes");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  select_gemm(nd, gemm_mode);
  const int N = rb->T * rb->E;
  const int nb = batch_size < N ? batch_size : N;
  const MbPlan pl = plan_minibatch(ctx, nd, nb);
  const char* same_env = getenv("PH_BENCH_GRAD_SAME_ROWS");
  const int n_mb = (same_env && same_env[0] == '1') ? 1 : N / nb;   // whole minibatches of one epoch's order
  if (ensure_train_ws(ctx, nd.lay.P, slab_len_of(nd), pl.nwg, n_mb, (size_t)N, (size_t)N)) return 1;
  hipStream_t s = ctx->stream;
  PH_HIP(ph::launch_set_int(ctx->stop_flag, 0, s));
  if (rebuild_weight_image(ctx, nd, params)) return 1;
  if (build_grad_pack(ctx, nd, rb, (size_t)N)) return 1;
  ph::AdvStatArgs aa;
  aa.rb_adv = rb->advantages;
  aa.T = rb->T;
  aa.E = rb->E;
  aa.perms = nullptr;
  aa.perm_n = (uint32_t)N;
  aa.perm_hb = ph::feistel_half_bits((uint32_t)N);
  aa.perm_seed = 12345;
  aa.epoch = nullptr;
  aa.N = N;
  aa.batch = nb;
  aa.n_mb = n_mb;
  aa.out = ctx->advstats;
  aa.partial = ctx->advpart;
  aa.idx_out = nd.split == 1 ? nullptr : ctx->perm_idx;   // as in ph_ppo_train: the grad launches read the materialised order,
  aa.phys_out = nd.split == 1 ? nullptr : ctx->perm_phys; // the split kernel the row records
  fill_adv_records(aa, ctx, nd, rb);
  PH_HIP(ph::launch_adv_stats(aa, n_mb, s));
  ph::GradArgs g;
  std::memset(&g, 0, sizeof(g));
  fill_grad_args(g, nd, params, rb, hp, ctx);
  g.perm_n = aa.perm_n;
  g.perm_hb = aa.perm_hb;
  g.perm_seed = aa.perm_seed;
  g.nb = nb;
  g.ntiles = pl.ntiles;
  auto launch = [&](int i) -> hipError_t {
    const size_t start = (size_t)(i % n_mb) * nb;
    g.idx = ctx->perm_idx + start;
    g.idx_phys = ctx->perm_phys + start;
    g.mb_start = (int)start;
    g.rec_pi = ctx->rec_pi ? ctx->rec_pi + start : nullptr;
    g.rec_vf = ctx->rec_vf ? ctx->rec_vf + start : nullptr;
    g.advstats = ctx->advstats + 2 * (size_t)(i % n_mb);
    return ph::launch_ppo_grad(g, pl.nwg, gemm_mode, s);
  };
  for (int i = 0; i < n_mb; ++i) PH_HIP(launch(i));  // warm
  PH_HIP(hipEventRecord(ctx->ev0, s));
  for (int i = 0; i < reps; ++i) PH_HIP(launch(i));
  PH_HIP(hipEventRecord(ctx->ev1, s));
  PH_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_ms_out = ms / (float)reps;
  return 0;
}

int ph_bench_gae(ph_ctx* ctx, const ph_rollout* rb, const float* last_values, const float* dones, double gamma,
                 double gae_lambda, int mode, int reps, float* avg_ms_out) {
  DevGuard dev_guard(ctx);
  if (!ctx || !last_values || !dones || !avg_ms_out) return fail("ph_bench_gae: null argument");
  if (check_rb(rb)) return 1;
  if (reps <= 0 || mode < 0 || mode > 2) return fail("ph_bench_gae: bad arguments");
  hipStream_t s = ctx->stream;
  auto once = [&]() {
    return ph::launch_gae(rb->rewards, rb->values, rb->episode_starts, last_values, dones, rb->advantages, rb->returns,
                          rb->T, rb->E, gamma, gae_lambda, mode, s);
  };
  PH_HIP(once());
  PH_HIP(hipEventRecord(ctx->ev0, s));
  for (int i = 0; i < reps; ++i) PH_HIP(once());
  PH_HIP(hipEventRecord(ctx->ev1, s));
  PH_HIP(hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  PH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_ms_out = ms / (float)reps;
  return 0;
}

int ph_bench_train_kernels(ph_ctx* ctx, const ph_spec* spec, const ph_opt_state* opt, const ph_rollout* rb,
                           const ph_ppo_hyper* hp, int n_epochs, int batch_size, int reps, int gemm_mode, float* us_out) {
  DevGuard dev_guard(ctx);
  if (!us_out) return fail("ph_bench_train_kernels: null argument");
  if (reps <= 0) return fail("ph_bench_train_kernels: reps must be positive");
  for (int i = 0; i < PH_BENCH_NKERN; ++i) us_out[i] = 0.f;
  TrainPlan t;
  if (train_prepare(t, ctx, spec, opt, rb, hp, n_epochs, batch_size, nullptr, 12345ull, nullptr, gemm_mode)) return 1;
  hipStream_t s = ctx->stream;
  int rc = 0;
  auto timed = [&](int slot, auto&& fn) -> int {
    if (fn()) return 1;   // warm
    PH_HIP(hipEventRecord(ctx->ev0, s));
    for (int i = 0; i < reps; ++i)
      if (fn()) return 1;
    PH_HIP(hipEventRecord(ctx->ev1, s));
    PH_HIP(hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    PH_HIP(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    us_out[slot] = 1e3f * ms / (float)reps;
    return 0;
  };
  auto hip = [&](hipError_t e) -> int {
    if (e != hipSuccess) { rc = fail(std::string("ph_bench_train_kernels: ") + hipGetErrorString(e)); return 1; }
    return 0;
  };
  if (t.nd.split) {
    if (timed(PH_BENCH_WEIGHT_IMAGE, [&] { return rebuild_weight_image(ctx, t.nd, opt->params); })) return 1;
    if (timed(PH_BENCH_OBS_PLANES, [&] { return build_grad_pack(ctx, t.nd, rb, (size_t)n_epochs * t.N); })) return 1;
  }
  ph::AdvStatArgs aa;
  train_adv_args(t, false, aa);
  if (timed(PH_BENCH_ADV_STATS, [&] { return hip(ph::launch_adv_stats(aa, n_epochs * t.n_mb, s)); })) return rc ? rc : 1;
  MbPlan pl;
  if (train_launch_grad(t, 0, &pl)) return 1;   // slabs and partial statistics of minibatch 0 for the reduction to read
  ph::ReduceArgs r;
  ph::AdamArgs ad;
  fill_step_args(t, 0, pl, r, ad);
  r.wide = 0;
  if (timed(PH_BENCH_REDUCE, [&] { return hip(ph::launch_ppo_reduce(r, s)); })) return rc ? rc : 1;
  r.wide = 1;
  if (timed(PH_BENCH_REDUCE_WIDE, [&] { return hip(ph::launch_ppo_reduce(r, s)); })) return rc ? rc : 1;
  if (timed(PH_BENCH_ADAM, [&] { return hip(ph::launch_ppo_adam(ad, s)); })) return rc ? rc : 1;
  if (ctx->step_words && ctx->step_gen && ph::step_fused_fits(ph::reduce_blocks(slab_len_of(t.nd)), slab_len_of(t.nd), ctx->num_cu)) {
    if (timed(PH_BENCH_STEP_FUSED, [&] {
          return hip(ph::launch_ppo_step(r, ad, ctx->step_words, ctx->step_gen, ctx->step_gen + 1, STEP_WAIT_TICKS, s));
        }))
      return rc ? rc : 1;
  }
  if (rb->T >= 2) {   // RolloutBuffer.add of one step (row 1 <- row 0): buffer_add_kernel
    const size_t E = (size_t)rb->E;
    if (timed(PH_BENCH_BUFFER_ADD, [&] {
          return hip(ph::launch_buffer_add(rb->observations + E * t.nd.D, rb->actions + E * t.nd.A, rb->rewards + E,
                                           rb->episode_starts + E, rb->values + E, rb->log_probs + E, rb->observations,
                                           rb->actions, rb->episode_starts, rb->values, rb->log_probs, rb->E, t.nd.D, t.nd.A, s));
        }))
      return rc ? rc : 1;
  }
  us_out[PH_BENCH_SLAB_FLOATS] = (float)((double)pl.nwg * slab_len_of(t.nd));
  return 0;
}


// ---- ModularAlgorithm / ModularPolicy (ph_modular.hip) ---------------------------------------------------------------------
namespace {

int module_layout(int L, ph_layout* o) {   // a (Box(64), same action space) network: the partner module's parameter block
  const int H = PH_HIDDEN;
  o->D = H;
  o->F = H;
  o->A = 1;
  o->L = L;
  int off = 0;
  o->pi_W1 = off; off += H * H;
  o->pi_b1 = off; off += H;
  o->pi_W2 = off; off += H * H;
  o->pi_b2 = off; off += H;
  o->vf_W1 = off; off += H * H;
  o->vf_b1 = off; off += H;
  o->vf_W2 = off; off += H * H;
  o->vf_b2 = off; off += H;
  o->act_W = off; off += H * L;
  o->act_b = off; off += L;
  o->val_W = off; off += H;
  o->val_b = off; off += 1;
  o->P = off;
  return 0;
}

int check_modular(const ph_modular* m, const ph::NetDims& nd, const char* who) {
  const std::string w(who);
  if (!m) return fail(w + ": null ph_modular");
  if (m->num_partners < 1 || m->num_partners > PH_MOD_MAX) return fail(w + ": num_partners must be in [1, PH_MOD_MAX]");
  if (m->n_modules < 1 || m->n_modules > m->num_partners) return fail(w + ": n_modules must be in [1, num_partners]");
  for (int k = 0; k < m->num_partners; ++k)
    if (m->module_of[k] < 0 || m->module_of[k] >= m->n_modules) return fail(w + ": module_of out of range");
  if (nd.nchunk != 1 || nd.A != 1 || nd.L > 8)
    return fail(w + ": the ModularPolicy path takes observations of at most 64 features and one Discrete head of at most 8 logits");
  return 0;
}

// carve the minibatch-order activations out of one allocation
struct ModBufs {
  float *Lp, *zm, *zmod, *vm, *vk, *dzm, *dzmod, *dv, *dLa, *dLb;
};
int mod_buffers(ph_ctx* ctx, int nb, int n_mod, ModBufs* b) {
  const size_t rows = ((size_t)nb + 63) / 64 * 64;
  const size_t need = rows * (64 * 3 + 8 * 2 + 8 * 2 * (size_t)n_mod + 3);
  if (need > ctx->mw.act_cap) {
    if (ctx->capturing) return fail("workspace would grow inside graph capture: run the same call once outside capture first");
    if (ctx->mw.act) (void)hipFree(ctx->mw.act);
    ctx->mw.act = nullptr;
    ctx->mw.act_cap = 0;
    PH_HIP(hipMalloc((void**)&ctx->mw.act, need * sizeof(float)));
    ctx->mw.act_cap = need;
  }
  if (!ctx->mw.kl_sum) {
    if (ctx->capturing) return fail("first ModularPolicy call inside graph capture: call it once outside capture first");
    PH_HIP(hipMalloc((void**)&ctx->mw.kl_sum, sizeof(float)));
    PH_HIP(hipMemset(ctx->mw.kl_sum, 0, sizeof(float)));
    PH_HIP(hipMalloc((void**)&ctx->mw.scratch, (2 + PH_MOD_MAX) * sizeof(int)));
  }
  float* p = ctx->mw.act;
  b->Lp = p; p += rows * 64;
  b->dLa = p; p += rows * 64;
  b->dLb = p; p += rows * 64;
  b->zm = p; p += rows * 8;
  b->dzm = p; p += rows * 8;
  b->zmod = p; p += rows * 8 * n_mod;
  b->dzmod = p; p += rows * 8 * n_mod;
  b->vm = p; p += rows;
  b->vk = p; p += rows;
  b->dv = p; p += rows;
  return 0;
}

// slab slots of one minibatch: [main pi, main vf, module 0 pi .. module M-1 pi, trained module's vf]
int mod_slots(const ph_modular* m) { return m->n_modules + 3; }

int mod_maps(ph_ctx* ctx, const ph_spec* spec, const ph_modular* mod, const ph_layout& lay, const ph_layout& ml) {
  if (ctx->mw.maps_valid && std::memcmp(&ctx->mw.spec, spec, sizeof(ph_spec)) == 0 &&
      std::memcmp(&ctx->mw.mod, mod, sizeof(ph_modular)) == 0)
    return 0;
  if (ctx->capturing) return fail("first use of a ModularPolicy inside graph capture: call it once outside capture first");
  const int M = mod->n_modules, NS = mod_slots(mod);
  std::vector<int> h((size_t)M * NS * ph::RS_NET, -1);
  for (int k = 0; k < M; ++k) {
    int* base = h.data() + (size_t)k * NS * ph::RS_NET;
    ph::tower_slab_map(lay.F, lay.L, 1, lay.pi_W1, lay.pi_b1, lay.pi_W2, lay.pi_b2, lay.act_W, lay.act_b, base);
    ph::tower_slab_map(lay.F, lay.L, 2, lay.vf_W1, lay.vf_b1, lay.vf_W2, lay.vf_b2, lay.val_W, lay.val_b, base + ph::RS_NET);
    for (int m = 0; m < M; ++m) {
      const int o = lay.P + m * ml.P;
      ph::tower_slab_map(64, lay.L, 1, o + ml.pi_W1, o + ml.pi_b1, o + ml.pi_W2, o + ml.pi_b2, o + ml.act_W, o + ml.act_b,
                         base + (size_t)(2 + m) * ph::RS_NET);
    }
    const int o = lay.P + k * ml.P;
    ph::tower_slab_map(64, lay.L, 2, o + ml.vf_W1, o + ml.vf_b1, o + ml.vf_W2, o + ml.vf_b2, o + ml.val_W, o + ml.val_b,
                       base + (size_t)(2 + M) * ph::RS_NET);
  }
  if (ctx->mw.maps) (void)hipFree(ctx->mw.maps);
  ctx->mw.maps = nullptr;
  PH_HIP(hipMalloc((void**)&ctx->mw.maps, h.size() * sizeof(int)));
  PH_HIP(hipMemcpy(ctx->mw.maps, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  std::memcpy(&ctx->mw.spec, spec, sizeof(ph_spec));
  std::memcpy(&ctx->mw.mod, mod, sizeof(ph_modular));
  ctx->mw.maps_valid = true;
  return 0;
}

// the towers' argument records
void tower_main(ph::TowerArgs& t, const ph::NetDims& nd, const float* params, bool policy, const float* x, int x_ld,
                const int* idx, int T, int E, int nb) {
  const ph_layout& lay = nd.lay;
  std::memset(&t, 0, sizeof(t));
  t.nb = nb;
  t.ntiles = (nb + 63) / 64;
  t.x = x;
  t.x_ld = x_ld;
  t.F = nd.F;
  t.obs_off = nd.obs_kind == PH_SPACE_BOX ? nullptr : nd.obs_off;
  t.idx = idx;
  t.T = T;
  t.E = E;
  t.W1 = params + (policy ? lay.pi_W1 : lay.vf_W1);
  t.b1 = params + (policy ? lay.pi_b1 : lay.vf_b1);
  t.W2 = params + (policy ? lay.pi_W2 : lay.vf_W2);
  t.b2 = params + (policy ? lay.pi_b2 : lay.vf_b2);
  t.hW = params + (policy ? lay.act_W : lay.val_W);
  t.hb = params + (policy ? lay.act_b : lay.val_b);
  t.head = policy ? 1 : 2;
  t.L = nd.L;
}
void tower_module(ph::TowerArgs& t, const ph_layout& lay, const ph_layout& ml, const float* params, int m, bool policy,
                  const float* latent, int nb) {
  const float* base = params + lay.P + (size_t)m * ml.P;
  std::memset(&t, 0, sizeof(t));
  t.nb = nb;
  t.ntiles = (nb + 63) / 64;
  t.x = latent;
  t.x_ld = 64;
  t.F = 64;
  t.W1 = base + (policy ? ml.pi_W1 : ml.vf_W1);
  t.b1 = base + (policy ? ml.pi_b1 : ml.vf_b1);
  t.W2 = base + (policy ? ml.pi_W2 : ml.vf_W2);
  t.b2 = base + (policy ? ml.pi_b2 : ml.vf_b2);
  t.hW = base + (policy ? ml.act_W : ml.val_W);
  t.hb = base + (policy ? ml.act_b : ml.val_b);
  t.head = policy ? 1 : 2;
  t.L = lay.L;
}

// forward of the whole DAG for `nb` rows: main towers, then every module in `mods` (policy tower; value tower too for k_mod)
int mod_forward_towers(ph_ctx* ctx, const ph::NetDims& nd, const ph_layout& ml, const float* params, const float* x, int x_ld,
                       const int* idx, int T, int E, int nb, const ModBufs& b, int n_mod, int k_mod, bool all_modules,
                       int gemm_mode, const int* stop_flag) {
  const int ntiles = (nb + 63) / 64, nwg = ntiles < ctx->num_cu ? ntiles : ctx->num_cu;
  ph::TowerLaunch L;
  std::memset(&L, 0, sizeof(L));
  L.mode = 0;
  L.stop_flag = stop_flag;
  tower_main(L.t[0], nd, params, true, x, x_ld, idx, T, E, nb);
  L.t[0].h2_out = b.Lp;
  L.t[0].head_out = b.zm;
  tower_main(L.t[1], nd, params, false, x, x_ld, idx, T, E, nb);
  L.t[1].head_out = b.vm;
  PH_HIP(ph::launch_tower(L, nwg, 2, gemm_mode, ctx->stream));
  for (int m = 0; m < n_mod; ++m) {
    if (!all_modules && m != k_mod) continue;
    tower_module(L.t[0], nd.lay, ml, params, m, true, b.Lp, nb);
    L.t[0].head_out = b.zmod + (size_t)m * nb * 8;
    int ny = 1;
    if (m == k_mod) {
      tower_module(L.t[1], nd.lay, ml, params, m, false, b.Lp, nb);
      L.t[1].head_out = b.vk;
      ny = 2;
    }
    PH_HIP(ph::launch_tower(L, nwg, ny, gemm_mode, ctx->stream));
  }
  return 0;
}

struct ModMinibatch {
  const ph_rollout* rb;
  const int* idx;
  int nb;
  const float* advstats;
  int k_mod;
  float reg_coef;
  float* stats_out;
};

// forward, loss, backward and the slab reduction of one minibatch; the gradient lands in `grad`
int mod_minibatch(ph_ctx* ctx, const ph_spec* spec, const ph_modular* mod, const ph::NetDims& nd, const ph_layout& ml,
                  const float* params, const ph_ppo_hyper* hp, const ModMinibatch& mb, float* grad, int gemm_mode) {
  const int nb = mb.nb, M = mod->n_modules, NS = mod_slots(mod);
  const int ntiles = (nb + 63) / 64, nwg = ntiles < ctx->num_cu ? ntiles : ctx->num_cu;
  ModBufs b;
  if (mod_buffers(ctx, nb, M, &b)) return 1;
  hipStream_t s = ctx->stream;
  const int T = mb.rb->T, E = mb.rb->E;
  if (mod_forward_towers(ctx, nd, ml, params, mb.rb->observations, nd.D, mb.idx, T, E, nb, b, M, mb.k_mod, true, gemm_mode,
                         ctx->stop_flag))
    return 1;
  ph::ModLossArgs la;
  std::memset(&la, 0, sizeof(la));
  la.nb = nb;
  la.idx = mb.idx;
  la.T = T;
  la.E = E;
  la.rb_act = mb.rb->actions;
  la.rb_logp = mb.rb->log_probs;
  la.rb_adv = mb.rb->advantages;
  la.rb_ret = mb.rb->returns;
  la.rb_val = mb.rb->values;
  la.advstats = mb.advstats;
  la.L = nd.L;
  la.n_mod = M;
  la.k_mod = mb.k_mod;
  la.nomain = mod->nomain;
  la.zm = b.zm;
  la.zmod = b.zmod;
  for (int m = 0; m < M; ++m) {
    int cnt = 0;
    for (int k = 0; k < mod->num_partners; ++k) cnt += mod->module_of[k] == m;
    la.weight[m] = (float)cnt / (float)mod->num_partners;
  }
  la.vm = b.vm;
  la.vk = b.vk;
  la.clip = hp->clip_range;
  la.clip_vf = hp->clip_range_vf;
  la.ent_coef = hp->ent_coef;
  la.vf_coef = hp->vf_coef;
  la.reg_coef = mb.reg_coef;
  la.dzm = b.dzm;
  la.dzmod = b.dzmod;
  la.dv = b.dv;
  la.statpart = ctx->statpart;
  la.stop_flag = ctx->stop_flag;
  PH_HIP(ph::launch_modular_loss(la, s));

  // backward: the modules first (their dL/dX is the main policy latent's gradient), the main towers last
  float* slabs = ctx->slabs;   // [nwg][NS][RS_NET]: tower slot t of workgroup w at (w * NS + t) * RS_NET
  auto slot_base = [&](int slot) { return slabs + (size_t)slot * ph::RS_NET; };
  ph::TowerLaunch L;
  std::memset(&L, 0, sizeof(L));
  L.mode = 1;
  L.stop_flag = ctx->stop_flag;
  bool first = true;
  for (int m = 0; m < M; ++m) {
    tower_module(L.t[0], nd.lay, ml, params, m, true, b.Lp, nb);
    L.t[0].dhead = b.dzmod + (size_t)m * nb * 8;
    L.t[0].dx_out = b.dLa;
    L.t[0].dx_accumulate = first ? 0 : 1;
    L.t[0].slab = slot_base(2 + m);
    L.t[0].slab_stride = NS * ph::RS_NET;
    first = false;
    int ny = 1;
    if (m == mb.k_mod) {
      tower_module(L.t[1], nd.lay, ml, params, m, false, b.Lp, nb);
      L.t[1].dhead = b.dv;
      L.t[1].dx_out = b.dLb;
      L.t[1].dx_accumulate = 0;
      L.t[1].slab = slot_base(2 + M);
      L.t[1].slab_stride = NS * ph::RS_NET;
      ny = 2;
    }
    PH_HIP(ph::launch_tower(L, nwg, ny, gemm_mode, s));
  }
  tower_main(L.t[0], nd, params, true, mb.rb->observations, nd.D, mb.idx, T, E, nb);
  L.t[0].dhead = b.dzm;
  L.t[0].ext0 = b.dLa;
  L.t[0].ext1 = b.dLb;
  L.t[0].slab = slot_base(0);
  L.t[0].slab_stride = NS * ph::RS_NET;
  tower_main(L.t[1], nd, params, false, mb.rb->observations, nd.D, mb.idx, T, E, nb);
  L.t[1].dhead = b.dv;
  L.t[1].slab = slot_base(1);
  L.t[1].slab_stride = NS * ph::RS_NET;
  PH_HIP(ph::launch_tower(L, nwg, 2, gemm_mode, s));

  ph::ReduceArgs r;
  r.slabs = slabs;
  r.nslab = nwg;
  r.nstatpart = 0;
  r.P = nd.lay.P + M * ml.P;
  r.slab_len = NS * ph::RS_NET;
  r.map = ctx->mw.maps + (size_t)mb.k_mod * NS * ph::RS_NET;
  r.grad = grad;
  r.blocksq = ctx->blocksq;
  r.statpart = ctx->statpart;
  r.stats_out = nullptr;
  r.nb = nb;
  r.ent_coef = 0.f;
  r.vf_coef = 0.f;
  r.target_kl = -1.f;
  r.stop_flag = ctx->stop_flag;
  r.step = nullptr;
  r.scalars = ctx->scalars;
  PH_HIP(ph::launch_ppo_reduce(r, s));
  return 0;
}

int mod_workspace(ph_ctx* ctx, const ph_modular* mod, int P_total, int nb_max, int n_mb_total, size_t n_idx) {
  const int ntiles = (nb_max + 63) / 64, nwg = ntiles < ctx->num_cu ? ntiles : ctx->num_cu;
  const int loss_blocks = (nb_max + 255) / 256;
  if (ensure_train_ws(ctx, P_total, mod_slots(mod) * ph::RS_NET, nwg, n_mb_total, n_idx)) return 1;
  if (!ctx->capturing && ensure(ctx->statpart, ctx->statpart_cap, (size_t)(loss_blocks > 2 * nwg ? loss_blocks : 2 * nwg) * ph::NSTATP))
    return 1;
  return 0;
}
}  // namespace

int ph_modular_layout(const ph_spec* spec, const ph_modular* mod, ph_layout* main_out, ph_layout* module_out, int* p_total_out) {
  if (!spec || !mod) return fail("ph_modular_layout: null argument");
  ph_layout lay, ml;
  if (layout_of(spec, &lay)) return 1;
  module_layout(lay.L, &ml);
  if (mod->n_modules < 1 || mod->n_modules > PH_MOD_MAX) return fail("ph_modular_layout: n_modules must be in [1, PH_MOD_MAX]");
  if (main_out) *main_out = lay;
  if (module_out) *module_out = ml;
  if (p_total_out) *p_total_out = lay.P + mod->n_modules * ml.P;
  return 0;
}

int ph_modular_forward(ph_ctx* ctx, const ph_spec* spec, const ph_modular* mod, const float* params, int partner_idx,
                       const float* obs, int n, const unsigned char* action_mask, const float* uniforms,
                       const float* given_actions, unsigned long long seed, unsigned long long counter, int deterministic,
                       int* actions_i32, float* actions_f32, float* values, float* log_probs, float* entropy,
                       float* logits_main, float* logits_partner, const ph_rollout* rb, int pos,
                       const float* episode_start_in, const float* pending_reward, int gemm_mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs) return fail("ph_modular_forward: null params/obs");
  if ((uintptr_t)params % 16 != 0) return fail("ph_modular_forward: params must be 16-byte aligned");
  if (n <= 0) return fail("ph_modular_forward: n must be positive");
  ph::FwdArgs a;
  std::memset(&a, 0, sizeof(a));
  if (resolve(ctx, spec, &a.nd)) return 1;
  if (check_modular(mod, a.nd, "ph_modular_forward")) return 1;
  if (partner_idx < 0 || partner_idx >= mod->num_partners) return fail("ph_modular_forward: partner_idx out of range");
  ph_layout ml;
  module_layout(a.nd.L, &ml);
  const int k_mod = mod->module_of[partner_idx];
  ModBufs b;
  if (mod_buffers(ctx, n, mod->n_modules, &b)) return 1;
  if (mod_forward_towers(ctx, a.nd, ml, params, obs, a.nd.D, nullptr, 0, 0, n, b, mod->n_modules, k_mod, false, gemm_mode,
                         nullptr))
    return 1;
  a.params = params;
  a.obs = obs;
  a.n = n;
  a.mask = action_mask;
  a.uniforms = uniforms;
  a.given_actions = given_actions;
  a.seed = seed;
  a.counter = counter;
  a.epoch = ctx->rng_epoch;
  a.deterministic = deterministic;
  a.act_i32 = actions_i32;
  a.act_f32 = actions_f32;
  a.values = values;
  a.logp = log_probs;
  a.entropy = entropy;
  if (rb) {
    if (check_rb(rb)) return 1;
    if (n != rb->E) return fail("ph_modular_forward: fused add needs n == rollout E");
    if (pos < 0 || pos >= rb->T) return fail("ph_modular_forward: pos out of range (buffer full?)");
    if (!episode_start_in) return fail("ph_modular_forward: fused add needs episode_start_in");
    const size_t row = (size_t)pos * rb->E;
    a.rb_obs = rb->observations + row * a.nd.D;
    a.rb_act = rb->actions + row * a.nd.A;
    a.rb_rew = rb->rewards + row;
    a.rb_es = rb->episode_starts + row;
    a.rb_val = rb->values + row;
    a.rb_logp = rb->log_probs + row;
    a.es_in = episode_start_in;
    if (pending_reward) {
      if (pos < 1) return fail("ph_modular_forward: pending_reward needs pos >= 1");
      a.prev_rew = rb->rewards + (row - rb->E);
      a.pending_reward = pending_reward;
    }
  } else if (pending_reward) {
    return fail("ph_modular_forward: pending_reward needs the fused rollout-buffer write");
  }
  PH_HIP(ph::launch_modular_act(a, b.zm, b.zmod + (size_t)k_mod * n * 8, b.vm, b.vk, mod->nomain, logits_main, logits_partner,
                                ctx->stream));
  return 0;
}

int ph_modular_minibatch_grad(ph_ctx* ctx, const ph_spec* spec, const ph_modular* mod, const float* params, int partner_idx,
                              const ph_rollout* rb, const ph_ppo_hyper* hp, const int* indices, int nb, float marginal_reg_coef,
                              float* grad_out, float* stats_out, int gemm_mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !hp || !indices || !grad_out) return fail("ph_modular_minibatch_grad: null argument");
  if ((uintptr_t)params % 16 != 0) return fail("ph_modular_minibatch_grad: params must be 16-byte aligned");
  if (check_rb(rb)) return 1;
  if (nb <= 0) return fail("ph_modular_minibatch_grad: nb must be positive");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  if (check_modular(mod, nd, "ph_modular_minibatch_grad")) return 1;
  if (partner_idx < 0 || partner_idx >= mod->num_partners) return fail("ph_modular_minibatch_grad: partner_idx out of range");
  ph_layout ml;
  module_layout(nd.L, &ml);
  const int P_total = nd.lay.P + mod->n_modules * ml.P;
  if (mod_workspace(ctx, mod, P_total, nb, 1, 0)) return 1;
  if (mod_maps(ctx, spec, mod, nd.lay, ml)) return 1;
  hipStream_t s = ctx->stream;
  PH_HIP(ph::launch_set_int(ctx->stop_flag, 0, s));
  ph::AdvStatArgs aa;
  aa.rb_adv = rb->advantages;
  aa.T = rb->T;
  aa.E = rb->E;
  aa.perms = indices;
  aa.perm_n = 0;
  aa.perm_hb = 1;
  aa.perm_seed = 0;
  aa.epoch = nullptr;
  aa.N = nb;
  aa.batch = nb;
  aa.n_mb = 1;
  aa.out = ctx->advstats;
  aa.partial = ctx->advpart;
  aa.idx_out = nullptr;
  aa.phys_out = nullptr;
  PH_HIP(ph::launch_adv_stats(aa, 1, s));
  PH_HIP(hipMemsetAsync(grad_out, 0, (size_t)P_total * sizeof(float), s));
  ModMinibatch mb;
  mb.rb = rb;
  mb.idx = indices;
  mb.nb = nb;
  mb.advstats = ctx->advstats;
  mb.k_mod = mod->module_of[partner_idx];
  mb.reg_coef = marginal_reg_coef;
  mb.stats_out = stats_out;
  if (mod_minibatch(ctx, spec, mod, nd, ml, params, hp, mb, grad_out, gemm_mode)) return 1;
  if (stats_out) {   // statistics without touching any optimizer state: a scratch step counter / first-use table
    PH_HIP(hipMemsetAsync(ctx->mw.scratch, 0, (2 + PH_MOD_MAX) * sizeof(int), s));
    ph::ModFinalizeArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.statpart = ctx->statpart;
    fa.nstatpart = (nb + 255) / 256;
    fa.nb = nb;
    fa.step = ctx->mw.scratch;
    fa.mod_first = ctx->mw.scratch + 1;
    fa.k_mod = 0;
    fa.kl_sum = ctx->mw.kl_sum;
    fa.stats_out = stats_out;
    fa.ent_coef = hp->ent_coef;
    fa.vf_coef = hp->vf_coef;
    fa.reg_coef = marginal_reg_coef;
    fa.stop_flag = ctx->stop_flag;
    PH_HIP(ph::launch_modular_finalize(fa, s));
  }
  return 0;
}

int ph_modular_train(ph_ctx* ctx, const ph_spec* spec, const ph_modular* mod, const ph_opt_state* opt, int* mod_first,
                     const ph_rollout* rbs, const ph_ppo_hyper* hp, int n_epochs, int batch_size, const int* perms,
                     unsigned long long perm_seed, float* stats, float marginal_reg_coef, int gemm_mode) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!opt || !opt->params || !opt->adam_m || !opt->adam_v || !opt->step || !mod_first)
    return fail("ph_modular_train: null optimizer state");
  if ((uintptr_t)opt->params % 16 != 0) return fail("ph_modular_train: params must be 16-byte aligned");
  if (!hp || !rbs) return fail("ph_modular_train: null argument");
  if (n_epochs <= 0 || batch_size <= 0) return fail("ph_modular_train: n_epochs and batch_size must be positive");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  if (check_modular(mod, nd, "ph_modular_train")) return 1;
  ph_layout ml;
  module_layout(nd.L, &ml);
  const int M = mod->n_modules, P_total = nd.lay.P + M * ml.P;
  for (int k = 0; k < mod->num_partners; ++k) {
    if (check_rb(&rbs[k])) return 1;
    if (rbs[k].T != rbs[0].T || rbs[k].E != rbs[0].E) return fail("ph_modular_train: the partners' rollout buffers must have one shape");
  }
  const int N = rbs[0].T * rbs[0].E;
  const int n_mb = (N + batch_size - 1) / batch_size;
  const int nb_max = batch_size < N ? batch_size : N;
  if (mod_workspace(ctx, mod, P_total, nb_max, n_epochs * n_mb, perms ? 0 : (size_t)n_epochs * N)) return 1;
  if (mod_maps(ctx, spec, mod, nd.lay, ml)) return 1;
  {
    ModBufs probe;
    if (mod_buffers(ctx, nb_max, M, &probe)) return 1;
  }
  hipStream_t s = ctx->stream;
  const uint32_t hb = ph::feistel_half_bits((uint32_t)N);
  for (int k = 0; k < mod->num_partners; ++k) {
    const ph_rollout* rb = &rbs[k];
    const int k_mod = mod->module_of[k];
    PH_HIP(ph::launch_set_int(ctx->stop_flag, 0, s));
    PH_HIP(hipMemsetAsync(ctx->mw.kl_sum, 0, sizeof(float), s));
    const int* perms_k = perms ? perms + (size_t)k * n_epochs * N : nullptr;
    ph::AdvStatArgs aa;
    aa.rb_adv = rb->advantages;
    aa.T = rb->T;
    aa.E = rb->E;
    aa.perms = perms_k;
    aa.perm_n = (uint32_t)N;
    aa.perm_hb = hb;
    aa.perm_seed = perm_seed + (unsigned long long)k * 0x9E3779B97F4A7C15ull;
    aa.epoch = ctx->rng_epoch;
    aa.N = N;
    aa.batch = batch_size;
    aa.n_mb = n_mb;
    aa.out = ctx->advstats;
    aa.partial = ctx->advpart;
    aa.idx_out = perms_k ? nullptr : ctx->perm_idx;
    aa.phys_out = nullptr;
    PH_HIP(ph::launch_adv_stats(aa, n_epochs * n_mb, s));
    for (int ep = 0; ep < n_epochs; ++ep) {
      for (int j = 0; j < n_mb; ++j) {
        const int mbi = ep * n_mb + j, start = j * batch_size;
        const int nb = (N - start < batch_size) ? N - start : batch_size;
        float* st = stats ? stats + ((size_t)k * n_epochs * n_mb + mbi) * PH_NSTAT : nullptr;
        ModMinibatch mb;
        mb.rb = rb;
        mb.idx = (perms_k ? perms_k : ctx->perm_idx) + (size_t)ep * N + start;
        mb.nb = nb;
        mb.advstats = ctx->advstats + 2 * (size_t)mbi;
        mb.k_mod = k_mod;
        mb.reg_coef = marginal_reg_coef;
        mb.stats_out = st;
        if (mod_minibatch(ctx, spec, mod, nd, ml, opt->params, hp, mb, ctx->grad, gemm_mode)) return 1;
        ph::ModFinalizeArgs fa;
        std::memset(&fa, 0, sizeof(fa));
        fa.statpart = ctx->statpart;
        fa.nstatpart = (nb + 255) / 256;
        fa.nb = nb;
        fa.step = opt->step;
        fa.mod_first = mod_first;
        fa.k_mod = k_mod;
        fa.kl_sum = ctx->mw.kl_sum;
        fa.stats_out = st;
        fa.ent_coef = hp->ent_coef;
        fa.vf_coef = hp->vf_coef;
        fa.reg_coef = marginal_reg_coef;
        fa.stop_flag = ctx->stop_flag;
        PH_HIP(ph::launch_modular_finalize(fa, s));
        ph::ModAdamArgs ad;
        std::memset(&ad, 0, sizeof(ad));
        ad.params = opt->params;
        ad.m = opt->adam_m;
        ad.v = opt->adam_v;
        ad.grad = ctx->grad;
        ad.blocksq = ctx->blocksq;
        ad.nblk = ph::reduce_blocks(mod_slots(mod) * ph::RS_NET);
        ad.P = P_total;
        ad.step = opt->step;
        ad.lr = hp->learning_rate;
        ad.beta1 = hp->adam_beta1;
        ad.beta2 = hp->adam_beta2;
        ad.eps = hp->adam_eps;
        ad.max_norm = hp->max_grad_norm;
        ad.stats_out = st;
        ad.n_mod = M;
        ad.k_mod = k_mod;
        ad.n_seg = 2 * M;
        for (int m = 0; m < M; ++m) {
          const int o = nd.lay.P + m * ml.P;
          ad.seg_lo[2 * m] = o + ml.vf_W1;
          ad.seg_hi[2 * m] = o + ml.act_W;
          ad.seg_mod[2 * m] = m;
          ad.seg_lo[2 * m + 1] = o + ml.val_W;
          ad.seg_hi[2 * m + 1] = o + ml.P;
          ad.seg_mod[2 * m + 1] = m;
        }
        ad.mod_first = mod_first;
        ad.stop_flag = ctx->stop_flag;
        PH_HIP(ph::launch_modular_adam(ad, s));
      }
      PH_HIP(ph::launch_modular_epoch_end(ctx->mw.kl_sum, n_mb, hp->target_kl, ctx->stop_flag, s));
    }
  }
  PH_HIP(ph::launch_set_int(ctx->stop_flag, 0, s));
  return 0;
}

// ---- behavioural cloning on the shared 32-32 policy ----
int ph_bc_layout_of(const ph_spec* spec, ph_bc_layout* o) {
  ph_layout big;
  if (!o) return fail("null layout");
  if (layout_of(spec, &big)) return 1;
  const int H = PH_BC_HIDDEN;
  o->D = big.D;
  o->F = big.F;
  o->A = big.A;
  o->L = big.L;
  int off = 0;
  o->W1 = off; off += big.F * H;
  o->b1 = off; off += H;
  o->W2 = off; off += H * H;
  o->b2 = off; off += H;
  o->act_W = off; off += H * big.L;
  o->act_b = off; off += big.L;
  o->val_W = off; off += H;
  o->val_b = off; off += 1;
  o->P = off;
  return 0;
}

int ph_bc_forward(ph_ctx* ctx, const ph_spec* spec, const float* params, const float* obs, int n,
                  const unsigned char* action_mask, const float* uniforms, const float* given_actions,
                  unsigned long long seed, unsigned long long counter, int deterministic, int* actions_i32, float* values,
                  float* log_probs, float* entropy, float* logits) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!params || !obs || n <= 0) return fail("ph_bc_forward: bad argument");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  ph_bc_layout lay;
  if (ph_bc_layout_of(spec, &lay)) return 1;
  if ((size_t)lay.P * sizeof(float) > 150 * 1024) return fail("ph_bc_forward: policy too large for the LDS-resident forward");
  PH_HIP(ph::launch_bc_forward(nd, lay, params, obs, n, action_mask, uniforms, given_actions, seed, counter, deterministic,
                               actions_i32, values, log_probs, entropy, logits, ctx->stream));
  return 0;
}

int ph_bc_train(ph_ctx* ctx, const ph_spec* spec, const ph_opt_state* opt, const float* obs, const float* acts,
                const int* order, int N, int batch_size, int n_epochs, int max_batches, const ph_bc_hyper* hyper,
                float* stats) {
  DevGuard dev_guard(ctx);
  if (!ctx) return fail("null ctx");
  if (!opt || !opt->params || !opt->adam_m || !opt->adam_v || !opt->step) return fail("ph_bc_train: null optimizer state");
  if (!obs || !acts || !order || !hyper) return fail("ph_bc_train: null argument");
  if (N <= 0 || batch_size <= 0 || n_epochs <= 0) return fail("ph_bc_train: N, batch_size and n_epochs must be positive");
  ph::NetDims nd;
  if (resolve(ctx, spec, &nd)) return 1;
  ph_bc_layout lay;
  if (ph_bc_layout_of(spec, &lay)) return 1;
  if (ph::bc_train_lds_bytes(nd.F, nd.L, lay.P, nd.A) > 160 * 1024) return fail("ph_bc_train: working set exceeds the CU's 160 KiB of LDS");
  PH_HIP(ph::launch_bc_train(nd, lay, opt->params, opt->adam_m, opt->adam_v, opt->step, obs, acts, order, N, batch_size,
                             n_epochs, max_batches, *hyper, stats, ctx->stream));
  return 0;
}

int ph_feistel_indices(int n, unsigned long long perm_seed, int epoch, int start, int count, int* out) {
  if (!out) return fail("ph_feistel_indices: null out");
  if (n <= 0 || start < 0 || count < 0 || (long long)start + count > n) return fail("ph_feistel_indices: bad range");
  const uint32_t hb = ph::feistel_half_bits((uint32_t)n);
  const uint64_t key = ph::epoch_key(perm_seed, epoch);
  for (int i = 0; i < count; ++i) out[i] = (int)ph::feistel_perm((uint32_t)(start + i), (uint32_t)n, hb, key);
  return 0;
}

}  // extern "C"

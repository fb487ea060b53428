// Owning-handle layer of the C ABI ("one opaque handle per agent holding device buffers + weights + Adam
// state ... caller-owned host arrays, callee-owned device memory freed only by destroy").  Everything here is a thin composition
// of the pointer-level entry points of ph_abi.hip: the handle allocates the rollout buffer, the parameter vector, the Adam
// moments and the staging areas on its device, every array crossing the boundary is a HOST array, and every call is complete
// (stream-synchronised) on return.  A binder needs ctypes/cgo/JNI and nothing else -- no torch, no HIP runtime of its own.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <vector>

#include "pantheon_hip.h"

namespace {

thread_local std::string g_agent_err;

struct DeviceScope {   // the handle's device is current inside a call, the caller's device afterwards
  int prev = -1;
  bool switched = false;
  explicit DeviceScope(int device) {
    if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceScope() {
    if (switched) (void)hipSetDevice(prev);
  }
};

}  // namespace

struct ph_agent {
  int device = 0;
  ph_ctx* ctx = nullptr;
  hipStream_t stream = nullptr;
  ph_spec spec;
  ph_layout lay;
  int E = 0, T = 0, pos = 0;
  double gamma = 0.99, gae_lambda = 0.95;
  unsigned long long seed = 0, counter = 0, train_calls = 0;
  ph_rollout rb;
  float *params = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  int* opt_step = nullptr;
  // staging (device)
  float *d_obs = nullptr, *d_uniforms = nullptr, *d_es = nullptr, *d_vec = nullptr, *d_vec2 = nullptr, *d_values = nullptr,
        *d_logp = nullptr, *d_stats = nullptr;
  unsigned char* d_mask = nullptr;
  int *d_actions = nullptr, *d_perms = nullptr;
  size_t perms_cap = 0, stats_cap = 0;
  std::vector<void*> owned;
};

namespace {

int afail(const std::string& m) {
  g_agent_err = m;
  return 1;
}
int afail_hip(const char* what, hipError_t e) { return afail(std::string(what) + ": " + hipGetErrorString(e)); }
int lower(int rc) {   // a failed pointer-level call: carry its message
  if (rc) g_agent_err = ph_last_error();
  return rc;
}
#define PA_HIP(call)                                      \
  do {                                                    \
    hipError_t _e = (call);                               \
    if (_e != hipSuccess) return afail_hip(#call, _e);    \
  } while (0)

template <typename T>
int dalloc(ph_agent* a, T*& p, size_t n) {
  p = nullptr;
  hipError_t e = hipMalloc((void**)&p, (n ? n : 1) * sizeof(T));
  if (e != hipSuccess) return afail_hip("hipMalloc(agent)", e);
  a->owned.push_back((void*)p);
  return 0;
}
int h2d(ph_agent* a, void* dst, const void* src, size_t bytes) {
  PA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, a->stream));
  return 0;
}
int d2h(ph_agent* a, void* dst, const void* src, size_t bytes) {
  PA_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, a->stream));
  return 0;
}

}  // namespace

extern "C" {

const char* ph_agent_last_error(void) { return g_agent_err.c_str(); }

int ph_agent_create(int device, const ph_spec* spec, int n_envs, int n_steps, double gamma, double gae_lambda,
                    unsigned long long seed, ph_agent** out) {
  if (!out) return afail("ph_agent_create: null out");
  *out = nullptr;
  if (!spec || n_envs <= 0 || n_steps <= 0) return afail("ph_agent_create: bad arguments");
  ph_layout lay;
  if (lower(ph_layout_of(spec, &lay))) return 1;
  ph_ctx* ctx = nullptr;
  if (lower(ph_ctx_create(device, &ctx))) return 1;
  DeviceScope scope(device);
  ph_agent* a = new ph_agent();
  a->device = device;
  a->ctx = ctx;
  a->spec = *spec;
  a->lay = lay;
  a->E = n_envs;
  a->T = n_steps;
  a->gamma = gamma;
  a->gae_lambda = gae_lambda;
  a->seed = seed;
  auto bail = [&](int rc) {
    ph_agent_destroy(a);
    return rc;
  };
  if (hipStreamCreate(&a->stream) != hipSuccess) return bail(afail("hipStreamCreate failed"));
  if (lower(ph_ctx_set_stream(ctx, a->stream))) return bail(1);
  const size_t TE = (size_t)n_steps * n_envs, E = (size_t)n_envs;
  a->rb.T = n_steps;
  a->rb.E = n_envs;
  if (dalloc(a, a->rb.observations, TE * lay.D) || dalloc(a, a->rb.actions, TE * lay.A) || dalloc(a, a->rb.rewards, TE) ||
      dalloc(a, a->rb.episode_starts, TE) || dalloc(a, a->rb.values, TE) || dalloc(a, a->rb.log_probs, TE) ||
      dalloc(a, a->rb.advantages, TE) || dalloc(a, a->rb.returns, TE) || dalloc(a, a->params, (size_t)lay.P) ||
      dalloc(a, a->adam_m, (size_t)lay.P) || dalloc(a, a->adam_v, (size_t)lay.P) || dalloc(a, a->opt_step, 1) ||
      dalloc(a, a->d_obs, E * lay.D) || dalloc(a, a->d_uniforms, E * lay.A) || dalloc(a, a->d_es, E) ||
      dalloc(a, a->d_vec, E) || dalloc(a, a->d_vec2, E) || dalloc(a, a->d_values, E) || dalloc(a, a->d_logp, E) ||
      dalloc(a, a->d_mask, E * (size_t)(lay.L > 0 ? lay.L : 1)) || dalloc(a, a->d_actions, E * lay.A))
    return bail(1);
  if (hipMemsetAsync(a->params, 0, lay.P * sizeof(float), a->stream) != hipSuccess ||
      hipMemsetAsync(a->adam_m, 0, lay.P * sizeof(float), a->stream) != hipSuccess ||
      hipMemsetAsync(a->adam_v, 0, lay.P * sizeof(float), a->stream) != hipSuccess ||
      hipMemsetAsync(a->opt_step, 0, sizeof(int), a->stream) != hipSuccess)
    return bail(afail("hipMemset(agent) failed"));
  if (lower(ph_buffer_reset(ctx, &a->spec, &a->rb))) return bail(1);
  if (lower(ph_ctx_sync(ctx))) return bail(1);
  *out = a;
  return 0;
}

int ph_agent_destroy(ph_agent* a) {
  if (!a) return 0;
  DeviceScope scope(a->device);
  if (a->stream) (void)hipStreamSynchronize(a->stream);
  if (a->ctx) (void)ph_ctx_destroy(a->ctx);
  for (void* p : a->owned) (void)hipFree(p);
  if (a->d_perms) (void)hipFree(a->d_perms);
  if (a->d_stats) (void)hipFree(a->d_stats);
  if (a->stream) (void)hipStreamDestroy(a->stream);
  delete a;
  return 0;
}

int ph_agent_layout(const ph_agent* a, ph_layout* out) {
  if (!a || !out) return afail("ph_agent_layout: null argument");
  *out = a->lay;
  return 0;
}

int ph_agent_set_params(ph_agent* a, const float* params) {
  if (!a || !params) return afail("ph_agent_set_params: null argument");
  DeviceScope scope(a->device);
  if (h2d(a, a->params, params, a->lay.P * sizeof(float))) return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_get_params(ph_agent* a, float* params_out) {
  if (!a || !params_out) return afail("ph_agent_get_params: null argument");
  DeviceScope scope(a->device);
  if (d2h(a, params_out, a->params, a->lay.P * sizeof(float))) return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_set_optimizer(ph_agent* a, const float* adam_m, const float* adam_v, int step) {
  if (!a || !adam_m || !adam_v || step < 0) return afail("ph_agent_set_optimizer: bad argument");
  DeviceScope scope(a->device);
  if (h2d(a, a->adam_m, adam_m, a->lay.P * sizeof(float)) || h2d(a, a->adam_v, adam_v, a->lay.P * sizeof(float)) ||
      h2d(a, a->opt_step, &step, sizeof(int)))
    return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_get_optimizer(ph_agent* a, float* adam_m_out, float* adam_v_out, int* step_out) {
  if (!a) return afail("ph_agent_get_optimizer: null agent");
  DeviceScope scope(a->device);
  if (adam_m_out && d2h(a, adam_m_out, a->adam_m, a->lay.P * sizeof(float))) return 1;
  if (adam_v_out && d2h(a, adam_v_out, a->adam_v, a->lay.P * sizeof(float))) return 1;
  if (step_out && d2h(a, step_out, a->opt_step, sizeof(int))) return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_buffer_reset(ph_agent* a) {
  if (!a) return afail("ph_agent_buffer_reset: null agent");
  DeviceScope scope(a->device);
  if (lower(ph_buffer_reset(a->ctx, &a->spec, &a->rb))) return 1;
  a->pos = 0;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_pos(const ph_agent* a, int* pos_out) {
  if (!a || !pos_out) return afail("ph_agent_pos: null argument");
  *pos_out = a->pos;
  return 0;
}

int ph_agent_act(ph_agent* a, const float* obs, const unsigned char* action_mask, const float* uniforms, int deterministic,
                 int record, const float* episode_start, int* actions_out, float* values_out, float* log_probs_out) {
  if (!a || !obs) return afail("ph_agent_act: null argument");
  if (record && !episode_start) return afail("ph_agent_act: record needs episode_start");
  if (record && a->pos >= a->T) return afail("ph_agent_act: RolloutBuffer.add on a full buffer");
  DeviceScope scope(a->device);
  const size_t E = (size_t)a->E;
  if (h2d(a, a->d_obs, obs, E * a->lay.D * sizeof(float))) return 1;
  if (action_mask && h2d(a, a->d_mask, action_mask, E * a->lay.L)) return 1;
  if (uniforms && h2d(a, a->d_uniforms, uniforms, E * a->lay.A * sizeof(float))) return 1;
  if (record && h2d(a, a->d_es, episode_start, E * sizeof(float))) return 1;
  a->counter += 1;
  if (lower(ph_policy_forward(a->ctx, &a->spec, a->params, a->d_obs, a->E, action_mask ? a->d_mask : nullptr,
                              uniforms ? a->d_uniforms : nullptr, nullptr, a->seed, a->counter, deterministic, a->d_actions,
                              nullptr, a->d_values, a->d_logp, nullptr, nullptr, record ? &a->rb : nullptr,
                              record ? a->pos : 0, record ? a->d_es : nullptr, nullptr, 0)))
    return 1;
  if (actions_out && d2h(a, actions_out, a->d_actions, E * a->lay.A * sizeof(int))) return 1;
  if (values_out && d2h(a, values_out, a->d_values, E * sizeof(float))) return 1;
  if (log_probs_out && d2h(a, log_probs_out, a->d_logp, E * sizeof(float))) return 1;
  if (lower(ph_ctx_sync(a->ctx))) return 1;
  if (record) a->pos += 1;
  return 0;
}

int ph_agent_add_reward(ph_agent* a, const float* reward, const unsigned char* env_mask) {
  if (!a || !reward) return afail("ph_agent_add_reward: null argument");
  if (a->pos < 1) return afail("ph_agent_add_reward: no recorded action to credit");
  DeviceScope scope(a->device);
  if (h2d(a, a->d_vec, reward, (size_t)a->E * sizeof(float))) return 1;
  if (env_mask && h2d(a, a->d_mask, env_mask, (size_t)a->E)) return 1;
  if (lower(ph_buffer_add_reward(a->ctx, &a->rb, a->pos - 1, a->d_vec, env_mask ? a->d_mask : nullptr))) return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_gae(ph_agent* a, const float* last_values, const float* dones, int mode) {
  if (!a || !last_values || !dones) return afail("ph_agent_gae: null argument");
  DeviceScope scope(a->device);
  if (h2d(a, a->d_vec, last_values, (size_t)a->E * sizeof(float)) || h2d(a, a->d_vec2, dones, (size_t)a->E * sizeof(float)))
    return 1;
  if (lower(ph_gae(a->ctx, &a->rb, a->d_vec, a->d_vec2, a->gamma, a->gae_lambda, mode))) return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_train(ph_agent* a, const ph_ppo_hyper* hyper, int n_epochs, int batch_size, const int* perms,
                   unsigned long long perm_seed, float* stats_out) {
  if (!a || !hyper) return afail("ph_agent_train: null argument");
  if (n_epochs <= 0 || batch_size <= 0) return afail("ph_agent_train: n_epochs and batch_size must be positive");
  DeviceScope scope(a->device);
  const size_t N = (size_t)a->T * a->E, n_mb = (N + batch_size - 1) / batch_size;
  if (perms) {
    if ((size_t)n_epochs * N > a->perms_cap) {
      if (a->d_perms) (void)hipFree(a->d_perms);
      a->d_perms = nullptr;
      PA_HIP(hipMalloc((void**)&a->d_perms, (size_t)n_epochs * N * sizeof(int)));
      a->perms_cap = (size_t)n_epochs * N;
    }
    if (h2d(a, a->d_perms, perms, (size_t)n_epochs * N * sizeof(int))) return 1;
  }
  const size_t n_stats = (size_t)n_epochs * n_mb * PH_NSTAT;
  if (n_stats > a->stats_cap) {
    if (a->d_stats) (void)hipFree(a->d_stats);
    a->d_stats = nullptr;
    PA_HIP(hipMalloc((void**)&a->d_stats, n_stats * sizeof(float)));
    a->stats_cap = n_stats;
  }
  ph_opt_state opt;
  opt.params = a->params;
  opt.adam_m = a->adam_m;
  opt.adam_v = a->adam_v;
  opt.step = a->opt_step;
  if (lower(ph_ppo_train(a->ctx, &a->spec, &opt, &a->rb, hyper, n_epochs, batch_size, perms ? a->d_perms : nullptr, perm_seed,
                         a->d_stats, 0)))
    return 1;
  if (stats_out && d2h(a, stats_out, a->d_stats, n_stats * sizeof(float))) return 1;
  a->train_calls += 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_export_buffer(ph_agent* a, float* observations, float* actions, float* rewards, float* episode_starts,
                           float* values, float* log_probs, float* advantages, float* returns) {
  if (!a) return afail("ph_agent_export_buffer: null agent");
  DeviceScope scope(a->device);
  const size_t TE = (size_t)a->T * a->E;
  if (observations && d2h(a, observations, a->rb.observations, TE * a->lay.D * sizeof(float))) return 1;
  if (actions && d2h(a, actions, a->rb.actions, TE * a->lay.A * sizeof(float))) return 1;
  if (rewards && d2h(a, rewards, a->rb.rewards, TE * sizeof(float))) return 1;
  if (episode_starts && d2h(a, episode_starts, a->rb.episode_starts, TE * sizeof(float))) return 1;
  if (values && d2h(a, values, a->rb.values, TE * sizeof(float))) return 1;
  if (log_probs && d2h(a, log_probs, a->rb.log_probs, TE * sizeof(float))) return 1;
  if (advantages && d2h(a, advantages, a->rb.advantages, TE * sizeof(float))) return 1;
  if (returns && d2h(a, returns, a->rb.returns, TE * sizeof(float))) return 1;
  return lower(ph_ctx_sync(a->ctx));
}

int ph_agent_import_buffer(ph_agent* a, const float* observations, const float* actions, const float* rewards,
                           const float* episode_starts, const float* values, const float* log_probs, const float* advantages,
                           const float* returns, int pos) {
  if (!a) return afail("ph_agent_import_buffer: null agent");
  if (pos < 0 || pos > a->T) return afail("ph_agent_import_buffer: pos out of range");
  DeviceScope scope(a->device);
  const size_t TE = (size_t)a->T * a->E;
  if (observations && h2d(a, a->rb.observations, observations, TE * a->lay.D * sizeof(float))) return 1;
  if (actions && h2d(a, a->rb.actions, actions, TE * a->lay.A * sizeof(float))) return 1;
  if (rewards && h2d(a, a->rb.rewards, rewards, TE * sizeof(float))) return 1;
  if (episode_starts && h2d(a, a->rb.episode_starts, episode_starts, TE * sizeof(float))) return 1;
  if (values && h2d(a, a->rb.values, values, TE * sizeof(float))) return 1;
  if (log_probs && h2d(a, a->rb.log_probs, log_probs, TE * sizeof(float))) return 1;
  if (advantages && h2d(a, a->rb.advantages, advantages, TE * sizeof(float))) return 1;
  if (returns && h2d(a, a->rb.returns, returns, TE * sizeof(float))) return 1;
  a->pos = pos;
  return lower(ph_ctx_sync(a->ctx));
}

}  // extern "C"

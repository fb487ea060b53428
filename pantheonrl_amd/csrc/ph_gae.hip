// K1 + K2: rollout-buffer row writes and the GAE(lambda) return/advantage pass.
// Reference: RolloutBuffer.add / rewards[pos-1] += r / compute_returns_and_advantage, called from
// pantheonrl/common/agents.py:127-130,172-179,198 (SB3 1.7.0 semantics: SURVEY.md A.1, A.2).
//
// All arrays are time-major (T,E) float32, so every access below is coalesced along the env axis.
// HBM-bound: 20 algorithmic bytes per transition (read r, V, episode_start; write A, R).
#include "ph_launch.h"

namespace ph {

// ---- serial in T, one lane per env: bit-faithful to the numpy loop ---------------------------------------------
//   delta = r[t] + gamma*V[t+1]*nnt - V[t];  A = delta + (gamma*lambda)*nnt*A;  numpy evaluates left to right in f32
//   with one rounding per operation, so contraction to FMA must stay off here.
__global__ __launch_bounds__(64) void gae_serial_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                        const float* __restrict__ es, const float* __restrict__ last_values,
                                                        const float* __restrict__ dones, float* __restrict__ adv,
                                                        float* __restrict__ ret, int T, int E, float g, float gl) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  float nv = last_values[e];
  float nnt = 1.0f - dones[e];
  float last = 0.f;
  int t = T - 1;
  // 8 independent row loads in flight per lane before the dependent chain consumes them
  for (; t >= 7; t -= 8) {
    float r[8], v[8], s[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = (size_t)(t - u) * E + e;
      r[u] = rew[o];
      v[u] = val[o];
      s[u] = es[o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = (size_t)(t - u) * E + e;
      const float delta = (r[u] + (g * nv) * nnt) - v[u];
      last = delta + ((gl * nnt) * last);
      adv[o] = last;
      ret[o] = last + v[u];
      nv = v[u];
      nnt = 1.0f - s[u];
    }
  }
  for (; t >= 0; --t) {
    const size_t o = (size_t)t * E + e;
    const float r = rew[o], v = val[o], s = es[o];
    const float delta = (r + (g * nv) * nnt) - v;
    last = delta + ((gl * nnt) * last);
    adv[o] = last;
    ret[o] = last + v;
    nv = v;
    nnt = 1.0f - s;
  }
}

// ---- chunked wavefront suffix scan over T --------------------------------------------------------------------------
// A_t = delta_t + c_t * A_{t+1} is a first-order linear recurrence: the map A_{t+1} -> A_t composes associatively,
// (S1,P1) o (S2,P2) = (S1 + P1*S2, P1*P2).  A workgroup owns EB = 32 consecutive envs (one full 128-byte line per
// row: 16-env / 64-byte segments measured 2x FETCH_SIZE) and walks T from the end in super-chunks of NCH*LC steps;
// inside a super-chunk lane (chunk, env) runs its LC steps serially keeping the local advantages and running
// coefficient products in registers, the NCH chunk composites are suffix-scanned through LDS (Hillis-Steele,
// log2 NCH rounds), each lane fixes up its LC outputs with the carried-in advantage, and the advantage at the first
// step of the super-chunk is carried to the next one.  One HBM read + one write per element, any T.
#ifndef PH_GAE_NT
#define PH_GAE_NT 1   // streaming (nontemporal) loads and stores of the five arrays -- every element is read once and written once:
                      // 4.40 -> 4.64 TB/s at E = 16384, T = 2048 (profiles/r06_w_gae_nontemporal_and_chunk_sweep.txt); 0 = plain accesses
#endif
__device__ __forceinline__ float gae_ld(const float* p) { return PH_GAE_NT ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void gae_st(float* p, float v) {
  if (PH_GAE_NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

template <int LC, int EB>
__global__ __launch_bounds__(1024) void gae_scan_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                        const float* __restrict__ es,
                                                        const float* __restrict__ last_values,
                                                        const float* __restrict__ dones, float* __restrict__ adv,
                                                        float* __restrict__ ret, int T, int E, int NCH, float g,
                                                        float gl) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sS = sm;                 // [NCH][EB]
  float* sP = sm + NCH * EB;      // [NCH][EB]
  float* carry = sP + NCH * EB;   // [EB] advantage at the first step of the super-chunk processed before this one
  const int el = threadIdx.x % EB, ch = threadIdx.x / EB;
  const int e = blockIdx.x * EB + el;
  const bool env_ok = e < E;
  const int span = NCH * LC;
  if (ch == 0) carry[el] = 0.f;
  // super-chunks cover [base, base+span) with base = T - span, T - 2*span, ... (the first one may start below 0)
  for (int top = T; top > 0; top -= span) {
    const int base = top - span;
    const int t0 = base + ch * LC;              // first step of this lane's chunk (may be negative: masked)
    const int tend = t0 + LC;                   // exclusive; tend <= top <= T
    const bool live = env_ok && tend > 0;
    float a[LC], cp[LC], vv[LC];
    float S = 0.f, P = 1.f;
    if (live) {
      float nv, nnt;
      if (tend == T) {
        nv = last_values[e];
        nnt = 1.0f - dones[e];
      } else {
        nv = val[(size_t)tend * E + e];
        nnt = 1.0f - es[(size_t)tend * E + e];
      }
      float r[LC], s[LC];
#pragma unroll
      for (int k = 0; k < LC; ++k) {
        const int t = t0 + k;
        const size_t o = (size_t)(t >= 0 ? t : 0) * E + e;
        r[k] = gae_ld(rew + o);
        vv[k] = gae_ld(val + o);
        s[k] = gae_ld(es + o);
      }
      float A = 0.f, Pacc = 1.f;
#pragma unroll
      for (int k = LC - 1; k >= 0; --k) {
        if (t0 + k >= 0) {
          const float delta = r[k] + g * nv * nnt - vv[k];
          const float c = gl * nnt;
          A = delta + c * A;
          Pacc = c * Pacc;
          nv = vv[k];
          nnt = 1.0f - s[k];
        }
        a[k] = A;
        cp[k] = Pacc;
      }
      S = A;
      P = Pacc;
    }
    __syncthreads();  // carry of the previous super-chunk visible / LDS of the previous round consumed
    sS[ch * EB + el] = S;
    sP[ch * EB + el] = P;
    // suffix scan: afterwards (S,P) of chunk ch is the composite of chunks [ch, NCH) of this super-chunk
    for (int d = 1; d < NCH; d <<= 1) {
      __syncthreads();
      float S2 = 0.f, P2 = 1.f;
      const bool has = (ch + d < NCH);
      if (has) {
        S2 = sS[(ch + d) * EB + el];
        P2 = sP[(ch + d) * EB + el];
      }
      __syncthreads();
      if (has) {
        S = S + P * S2;
        P = P * P2;
        sS[ch * EB + el] = S;
        sP[ch * EB + el] = P;
      }
    }
    __syncthreads();
    const float cin = carry[el];
    // true advantage entering this lane's chunk from above: composite of the later chunks applied to the carry
    const float Ain = (ch + 1 < NCH) ? sS[(ch + 1) * EB + el] + sP[(ch + 1) * EB + el] * cin : cin;
    const float Afirst = sS[el] + sP[el] * cin;  // advantage at the first (valid) step of the super-chunk
    __syncthreads();
    if (ch == 0) carry[el] = Afirst;
    if (live) {
#pragma unroll
      for (int k = 0; k < LC; ++k) {
        const int t = t0 + k;
        if (t >= 0) {
          const size_t o = (size_t)t * E + e;
          const float A = a[k] + cp[k] * Ain;
          gae_st(adv + o, A);
          gae_st(ret + o, A + vv[k]);
        }
      }
    }
  }
}

template <int LC, int EB>
static hipError_t launch_scan(const float* rew, const float* val, const float* es, const float* lv, const float* dn,
                              float* adv, float* ret, int T, int E, float g, float gl, hipStream_t s, int max_chunks = 1024 / EB) {
  int NCH = (T + LC - 1) / LC;
  int maxch = max_chunks < 1024 / EB ? max_chunks : 1024 / EB;
  static int nch_env = -1;   // PH_GAE_NCH: measurement override of the chunk count cap (workgroup size = NCH * EB lanes)
  if (nch_env < 0) {
    const char* e = getenv("PH_GAE_NCH");
    nch_env = e ? atoi(e) : 0;
  }
  if (nch_env > 0 && nch_env < maxch) maxch = nch_env;
  if (NCH > maxch) NCH = maxch;
  dim3 grid((E + EB - 1) / EB), block(NCH * EB);
  const size_t lds = sizeof(float) * (2 * NCH * EB + EB);
  hipLaunchKernelGGL((gae_scan_kernel<LC, EB>), grid, block, lds, s, rew, val, es, lv, dn, adv, ret, T, E, NCH, g, gl);
  return hipGetLastError();
}

// mode: 1 serial, 2 scan, 0 auto.
hipError_t launch_gae(const float* rew, const float* val, const float* es, const float* lv, const float* dn, float* adv,
                      float* ret, int T, int E, double gamma, double lam, int mode, hipStream_t s) {
  const float g = (float)gamma;
  const float gl = (float)(gamma * lam);  // Python multiplies the two floats in double first
  if (mode == 0) mode = 2;  // the scan exposes T-parallelism as well as E-parallelism: faster at every measured shape
  if (mode == 1) {
    hipLaunchKernelGGL(gae_serial_kernel, dim3((E + 63) / 64), dim3(64), 0, s, rew, val, es, lv, dn, adv, ret, T, E, g,
                       gl);
    return hipGetLastError();
  }
  // lanes per workgroup = min(ceil(T/LC), 1024/32) * 32
  static int lc_env = -1;   // PH_GAE_LC = 8 | 16 | 32: measurement override of the steps per lane
  if (lc_env < 0) {
    const char* e = getenv("PH_GAE_LC");
    lc_env = e ? atoi(e) : 0;
  }
  if (lc_env == 8) return launch_scan<8, 32>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s);
  if (lc_env == 16) return launch_scan<16, 32>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s);
  if (lc_env == 32) return launch_scan<32, 32>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s);
  // Longer T: 16 steps per lane as well.  32 steps per lane (120 VGPRs, one 1024-lane workgroup per CU whose load, scan and store
  // phases do not overlap with anything) measured 4.19 TB/s at E = 16384, T = 2048; 16 steps per lane (72 VGPRs) 4.57 TB/s; 8 steps
  // (51 VGPRs, two workgroups per CU, twice the LDS scan rounds per element) 4.37 TB/s -- profiles/r05_b_gae_lc_sweep.txt
  // Round 6: a rollout longer than one super-chunk walks it with EIGHT chunks per workgroup (256 lanes, 128 steps per super-chunk):
  // several such workgroups share a CU (72 VGPRs) and their load / chain / scan / store phases overlap, which one 1024-lane
  // workgroup per CU cannot do with itself -- 4.64 -> 5.0 TB/s with the streaming accesses (32 chunks: 4.64, 16: 4.73, 4: 4.53,
  // profiles/r06_w_gae_nontemporal_and_chunk_sweep.txt).  T <= 256 is ONE super-chunk of T / 8 chunks: a 3.9 us launch at
  // BASELINE's sizes, where splitting it in two costs a second pass (5.1 us).
  if (T <= 256) return launch_scan<8, 32>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s);
  // Round 6, third session: at E >= 16384 a workgroup takes 64 environments (one 256-byte segment per row and array, a wave's load
  // one contiguous piece) and sixteen chunks: 5.07-5.16 -> 5.33-5.36 TB/s at E = 16384, T = 2048 (profiles/r06_ae_gae_*).  Below that
  // the grid of 64-environment workgroups no longer covers the CUs.  128 environments: 4.4 TB/s; requesting the next super-chunk's
  // elements ahead of the scan (a second register set): SLOWER, 4.6 against 4.9-5.4 (profiles/r06_ad_gae_explicit_prefetch_ab.txt).
  static int eb_env = -1;   // PH_GAE_EB = 32 | 64 | 128: measurement override of the environments per workgroup
  if (eb_env < 0) {
    const char* e = getenv("PH_GAE_EB");
    eb_env = e ? atoi(e) : 0;
  }
  const int eb = eb_env ? eb_env : (E >= 16384 ? 64 : 32);
  if (eb == 64) return launch_scan<16, 64>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s, 16);
  if (eb == 128) return launch_scan<16, 128>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s, 8);
  return launch_scan<16, 32>(rew, val, es, lv, dn, adv, ret, T, E, g, gl, s, 8);
}

// ---- K1: RolloutBuffer.add / reward += / reset -----------------------------------------------------------------------
__global__ void buffer_add_kernel(float* __restrict__ d_obs, float* __restrict__ d_act, float* __restrict__ d_rew,
                                  float* __restrict__ d_es, float* __restrict__ d_val, float* __restrict__ d_lp,
                                  const float* __restrict__ obs, const float* __restrict__ act,
                                  const float* __restrict__ es, const float* __restrict__ val,
                                  const float* __restrict__ lp, int E, int D, int A) {
  const int stride = gridDim.x * blockDim.x;
  const int g0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = g0; i < E * D; i += stride) d_obs[i] = obs[i];
  for (int i = g0; i < E * A; i += stride) d_act[i] = act[i];
  for (int i = g0; i < E; i += stride) {
    d_rew[i] = 0.f;
    d_es[i] = es[i];
    d_val[i] = val[i];
    d_lp[i] = lp[i];
  }
}
hipError_t launch_buffer_add(float* d_obs, float* d_act, float* d_rew, float* d_es, float* d_val, float* d_lp,
                             const float* obs, const float* act, const float* es, const float* val, const float* lp,
                             int E, int D, int A, hipStream_t s) {
  const int n = E * D;
  int blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(buffer_add_kernel, dim3(blocks), dim3(256), 0, s, d_obs, d_act, d_rew, d_es, d_val, d_lp, obs, act,
                     es, val, lp, E, D, A);
  return hipGetLastError();
}

// columns cols[0..n) of a (T, E, .) rollout buffer -> a compact (T, n, .) buffer (all eight arrays).  A partner whose
// environments reach it at different rates trains on the columns that are full (envs/vec.py::RaggedVecOnPolicyAgent).
struct CompactArgs {
  ph_rollout src, dst;
  const int* cols;
  int n, D, A;
};
__global__ void buffer_compact_kernel(CompactArgs a) {
  const int T = a.src.T, E = a.src.E, n = a.n;
  const size_t stride = (size_t)gridDim.x * blockDim.x, g0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = g0; i < (size_t)T * n * a.D; i += stride) {
    const size_t row = i / a.D, f = i - row * a.D, t = row / n, c = row - t * n;
    a.dst.observations[i] = a.src.observations[(t * E + a.cols[c]) * a.D + f];
  }
  for (size_t i = g0; i < (size_t)T * n * a.A; i += stride) {
    const size_t row = i / a.A, f = i - row * a.A, t = row / n, c = row - t * n;
    a.dst.actions[i] = a.src.actions[(t * E + a.cols[c]) * a.A + f];
  }
  for (size_t i = g0; i < (size_t)T * n; i += stride) {
    const size_t t = i / n, c = i - t * n, j = t * E + a.cols[c];
    a.dst.rewards[i] = a.src.rewards[j];
    a.dst.episode_starts[i] = a.src.episode_starts[j];
    a.dst.values[i] = a.src.values[j];
    a.dst.log_probs[i] = a.src.log_probs[j];
    a.dst.advantages[i] = a.src.advantages[j];
    a.dst.returns[i] = a.src.returns[j];
  }
}
hipError_t launch_buffer_compact(const ph_rollout& src, const ph_rollout& dst, const int* cols, int n, int D, int A,
                                 hipStream_t s) {
  CompactArgs a;
  a.src = src;
  a.dst = dst;
  a.cols = cols;
  a.n = n;
  a.D = D;
  a.A = A;
  size_t work = (size_t)src.T * n * (D > 0 ? D : 1);
  int blocks = (int)((work + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(buffer_compact_kernel, dim3(blocks), dim3(256), 0, s, a);
  return hipGetLastError();
}

__global__ void reward_add_kernel(float* __restrict__ rew_row, const float* __restrict__ reward,
                                  const unsigned char* __restrict__ env_mask, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (env_mask && !env_mask[e]) return;
  rew_row[e] += reward[e];
}
hipError_t launch_reward_add(float* rew_row, const float* reward, const unsigned char* env_mask, int E, hipStream_t s) {
  hipLaunchKernelGGL(reward_add_kernel, dim3((E + 255) / 256), dim3(256), 0, s, rew_row, reward, env_mask, E);
  return hipGetLastError();
}

__global__ void reward_add_const_kernel(float* __restrict__ rew_row, float reward, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) rew_row[e] += reward;
}
hipError_t launch_reward_add_const(float* rew_row, float reward, int E, hipStream_t s) {
  hipLaunchKernelGGL(reward_add_const_kernel, dim3((E + 255) / 256), dim3(256), 0, s, rew_row, reward, E);
  return hipGetLastError();
}

// ragged (per-env write position) forms of Agent.update and of the position advance after a recorded action
__global__ void reward_add_ragged_kernel(float* __restrict__ rewards, const int* __restrict__ pos_env,
                                         const float* __restrict__ reward, const unsigned char* __restrict__ mask, int T,
                                         int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (mask && !mask[e]) return;
  const int p = pos_env[e];
  if (p < 1 || p > T) return;   // nothing recorded yet in this column
  rewards[(size_t)(p - 1) * E + e] += reward[e];
}
hipError_t launch_reward_add_ragged(float* rewards, const int* pos_env, const float* reward, const unsigned char* mask,
                                    int T, int E, hipStream_t s) {
  hipLaunchKernelGGL(reward_add_ragged_kernel, dim3((E + 255) / 256), dim3(256), 0, s, rewards, pos_env, reward, mask, T, E);
  return hipGetLastError();
}
__global__ void ragged_advance_kernel(int* __restrict__ pos_env, const unsigned char* __restrict__ mask, int T, int E) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  if (mask[e] && pos_env[e] < T) pos_env[e] += 1;
}
hipError_t launch_ragged_advance(int* pos_env, const unsigned char* mask, int T, int E, hipStream_t s) {
  hipLaunchKernelGGL(ragged_advance_kernel, dim3((E + 255) / 256), dim3(256), 0, s, pos_env, mask, T, E);
  return hipGetLastError();
}

__global__ void reward_add_joint_kernel(float* __restrict__ rew_row, const float* __restrict__ base,
                                        const int* __restrict__ joint, int E, int n_seats, int seat,
                                        const int* __restrict__ partner_seat, float bonus, int rule) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int p = *partner_seat;
  p = p < 0 ? 0 : (p >= n_seats ? n_seats - 1 : p);
  const float b = joint_reward(joint[(size_t)seat * E + e], joint[(size_t)p * E + e], bonus, rule);
  rew_row[e] += base[e] + b;
}
hipError_t launch_reward_add_joint(float* rew_row, const float* base, const int* joint, int E, int n_seats, int seat,
                                   const int* partner_seat, float bonus, int rule, hipStream_t s) {
  hipLaunchKernelGGL(reward_add_joint_kernel, dim3((E + 255) / 256), dim3(256), 0, s, rew_row, base, joint, E, n_seats,
                     seat, partner_seat, bonus, rule);
  return hipGetLastError();
}

}  // namespace ph

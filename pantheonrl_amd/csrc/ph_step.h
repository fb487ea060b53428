// Slab reduction, clip and Adam as device code shared by ppo_reduce_kernel / ppo_step_kernel / ppo_adam_kernel (ph_ppo.hip).
// `blk` is the 64-position slab block a workgroup works on.
#pragma once
#include "ph_launch.h"
#include "ph_split.h"

namespace ph {

// ---- slab reduction: grad[p] = sum_g slab[g][p] (fixed order), per-block sum of squares, minibatch statistics ----
// The slabs are summed in ONE FIXED tree per slab position, whatever lane layout walks it (bit-reproducible run to run, and
// ph_ppo_train / ph_ppo_train_multi / the fused step launch give identical parameters):
//   * RED_SUB = 16 consecutive ranges of per = ceil(nslab / 16) slabs;
//   * inside a range, slab j goes to accumulator j % 8, every accumulator adds its slabs in increasing j (a slab past the end of a
//     ragged range adds 0.0f); range sum = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
//   * four consecutive range sums chained into a quarter, A_g = ((s_4g + s_4g+1) + s_4g+2) + s_4g+3;  total (A_0 + A_1) + (A_2 + A_3).
// A block is 256 lanes for 64 slab positions; wave w always holds the four ranges of quarter w.  What differs is how many ADJACENT
// positions a lane carries (VEC), i.e. how wide its loads are and how many ranges of the quarter it walks in turn (4 / VEC):
//   VEC = 1: 4-byte loads, four ranges in turn                              (slab lengths that are odd: canonical slabs of some specs);
//   VEC = 2: 8-byte loads, two ranges in turn, 16 loads = 128 B in flight per lane -- 58 VGPRs: one wave of it still fits into the
//            registers two resident gradient waves leave on a SIMD, so one learner's reduce blocks run beside the OTHER learner's
//            gradient launch instead of waiting for a CU to drain (tests/test_kernel_resources.py);
//   VEC = 4: 16-byte loads, one range per lane, 16 loads = 256 B in flight -- a learner that has the device to itself
//            (ph_set_exclusive_device; the fused step launch), where the reduction sits on the critical path.
// Round 3's layout (one position per lane, 64 B in flight) read the 17 MB of a bench minibatch in eight dependent rounds of
// 256-byte wave accesses: latency, not bandwidth (9.5 us = 1.8 TB/s).
constexpr int RED_PARAMS = 64, RED_SUB = 16, RED_SHIFT = 6;

// clip_grad_norm_'s scaling and torch.optim.Adam's single-tensor update of ONE parameter (eps = 1e-5 default of SB3).  One
// definition for ppo_adam_kernel and the fused ppo_step_kernel, with floating-point contraction OFF: which multiply-add pairs
// the compiler fuses otherwise depends on the surrounding kernel, and the two paths must give bitwise the same parameters.
struct AdamScalars {
  float coef, ss, bc2s;   // clip coefficient, lr / (1 - beta1^t), sqrt(1 - beta2^t)
};
__device__ __forceinline__ AdamScalars adam_scalars(float total_norm, float max_norm, int step, float lr, float beta1, float beta2) {
  AdamScalars k;
  const float cc = max_norm / (total_norm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
  k.coef = cc < 1.0f ? cc : 1.0f;
  const double t = (double)step;
  const double bc1 = 1.0 - pow((double)beta1, t);
  const double bc2 = 1.0 - pow((double)beta2, t);
  k.ss = (float)((double)lr / bc1);
  k.bc2s = (float)sqrt(bc2);
  return k;
}
// value form: the caller fetched m0 / v0 / p0 (possibly long before the clip coefficient is known) and stores the results
__device__ __forceinline__ float adam_apply(float grad, const AdamScalars& k, float beta1, float beta2, float eps, float m0, float v0,
                                            float p0, float* m_out, float* v_out) {
#pragma clang fp contract(off)
  const float g = grad * k.coef;
  const float d = g - m0;
  const float m = m0 + d * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1-beta1)
  const float gg = g * g;
  const float v = v0 * beta2 + (1.0f - beta2) * gg;        // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
  const float denom = sqrtf(v) / k.bc2s + eps;
  *m_out = m;
  *v_out = v;
  const float step = k.ss * (m / denom);
  return p0 - step;                                        // param.addcdiv_(exp_avg, denom, -step_size)
}
__device__ __forceinline__ float adam_update(float grad, const AdamScalars& k, float beta1, float beta2, float eps, float* m_p,
                                             float* v_p, float* param_p) {
  float m, v;
  const float pn = adam_apply(grad, k, beta1, beta2, eps, *m_p, *v_p, *param_p, &m, &v);
  *m_p = m;
  *v_p = v;
  *param_p = pn;
  return pn;
}
// the slab sum of this block's 64 positions in the fixed tree: lane tid < 64 returns the gradient entry of position
// blockIdx.x * 64 + tid (0 for padding) and the parameter it belongs to (dst, -1 = padding); other lanes return 0 / -1
template <int VEC>
struct RedVec;
template <>
struct RedVec<1> { using T = float; };
template <>
struct RedVec<2> { using T = float2; };
template <>
struct RedVec<4> { using T = float4; };
// the parameter slab position blockIdx.x * 64 + tid belongs to (wave 0; -1 = padding / other waves): fetched by the callers BEFORE
// the slab walk -- the table lookup is a memory round trip of its own and nothing in it depends on the slabs
__device__ __forceinline__ int reduce_dst(const ReduceArgs& a, int blk) {
  const int tid = threadIdx.x;
  if (tid >= RED_PARAMS) return -1;
  const int p = blk * RED_PARAMS + tid;
  return p < a.slab_len ? (a.map ? a.map[p] : p) : -1;
}
// the sum of ONE range (slabs k0 .. k0 + per - 1, clipped to nslab) for the VEC positions p0 .. of a lane: eight interleaved
// accumulators, folded pairwise
template <int VEC, bool COH, typename RSRC>
__device__ __forceinline__ void range_walk(const ReduceArgs& a, const RSRC& rsrc, unsigned stride, int per, bool full, int k0, int p0,
                                           float (&range)[VEC]) {
  using VT = typename RedVec<VEC>::T;
  const int n = (k0 + per <= a.nslab ? per : a.nslab - k0);   // slabs of this range (<= 0: none)
  float acc[8][VEC];
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[u][c] = 0.f;
  if (p0 < a.slab_len) {
    // buffer loads: descriptor + one 32-bit lane offset (range start + position) + a scalar offset (slab j of the range) --
    // no 64-bit address per load in flight (flat loads cost the 8-byte shape 32 VGPRs of addresses)
    const unsigned lane_off = (unsigned)k0 * stride + (unsigned)p0 * (unsigned)sizeof(float);
#ifndef PH_REDUCE_NT
#define PH_REDUCE_NT 0   // experiment (scripts/build_variants.sh): 1 = the slab walk's loads carry the nontemporal bit (each slab byte is read once)
#endif
    constexpr int AUX = COH ? 16 : (PH_REDUCE_NT ? 2 : 0);   // 16 = sc1, 2 = nt
    auto ld = [&](int j) -> VT {
      if constexpr (VEC == 4) return __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, (unsigned)j * stride, AUX));
      else if constexpr (VEC == 2) return __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane_off, (unsigned)j * stride, AUX));
      else return __builtin_bit_cast(VT, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_off, (unsigned)j * stride, AUX));
    };
    auto add = [&](float (&dst)[VEC], const VT& x) {
      const float* xv = reinterpret_cast<const float*>(&x);
#pragma unroll
      for (int c = 0; c < VEC; ++c) dst[c] += xv[c];
    };
    int j = 0;
    if (full) {
      for (; j + 15 < per; j += 16) {   // two rounds of loads in flight; the adds keep the one-round-at-a-time order
        VT x0[8], x1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x0[u] = ld(j + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) x1[u] = ld(j + 8 + u);
        if constexpr (VEC == 4) __builtin_amdgcn_sched_barrier(0);   // all sixteen in flight (the scheduler otherwise rolls a window of eight)
#pragma unroll
        for (int u = 0; u < 8; ++u) add(acc[u], x0[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) add(acc[u], x1[u]);
      }
    }
    for (; j < per; j += 8) {           // ragged ranges, and the tail of a full one: a slab past the range's end adds 0
      VT x0[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        x0[u] = VT{};
        if (j + u < n) x0[u] = ld(j + u);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) add(acc[u], x0[u]);
    }
  }
#pragma unroll
  for (int c = 0; c < VEC; ++c)
    range[c] = ((acc[0][c] + acc[1][c]) + (acc[2][c] + acc[3][c])) + ((acc[4][c] + acc[5][c]) + (acc[6][c] + acc[7][c]));
}
// the additional loss term's slabs (ADAP's context term), added to the entry of parameter dst in a fixed order
__device__ __forceinline__ float extra_terms(const ReduceArgs& a, int dst, float g) {
  if (a.n_extra <= 0) return g;
  const int e = dst < a.extra_cut ? dst : ((dst >= a.extra_lo && dst < a.extra_hi) ? a.extra_cut + (dst - a.extra_lo) : -1);
  if (e < 0) return g;
  for (int k0 = 0; k0 < a.n_extra; k0 += 8) {   // loads batched, adds in slab order
    float x[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) x[u] = (k0 + u < a.n_extra) ? a.extra[(size_t)(k0 + u) * a.extra_len + e] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u < a.n_extra) g += x[u];
  }
  return g;
}
// COH: the slabs were written (write-through) by other workgroups of the SAME launch -- read them with sc1 loads, which do not hit
// lines an XCD's L2 kept from an earlier launch.  Used by the round-4 experiment that ran the minibatch step inside the gradient
// launch (CHANGELOG.md: bitwise the separate launches, and no faster -- the step is a chain of memory round trips, the kernel
// boundary is not its cost); the stand-alone kernels read plain.
template <int VEC, bool COH = false>
__device__ __forceinline__ float reduce_positions(const ReduceArgs& a, float (*gsum)[RED_PARAMS], int dst, int blk) {
  static_assert(VEC == 1 || VEC == 2 || VEC == 4, "positions per lane");
  constexpr int LP = RED_PARAMS / VEC;   // lanes that cover the block's 64 positions
  constexpr int TURNS = 4 / VEC;         // ranges of the wave's quarter a lane walks in turn
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> RED_SHIFT);
  const int lp = lane % LP, gw = lane / LP;
  const int p0 = blk * RED_PARAMS + lp * VEC;   // first of this lane's VEC positions (slab_len % VEC == 0: all in or all out)
  const int per = (a.nslab + RED_SUB - 1) / RED_SUB;
  const bool full = per * RED_SUB == a.nslab;          // no ragged range (every bench shape): no predicates
  const unsigned stride = (unsigned)a.slab_len * (unsigned)sizeof(float);   // nslab * stride < 4 GB (the launcher checks)
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.slabs), 0, (int)((unsigned)a.nslab * stride), 0x00020000);
  float rs[TURNS][VEC];
#pragma unroll
  for (int t = 0; t < TURNS; ++t)
#pragma unroll
    for (int c = 0; c < VEC; ++c) rs[t][c] = 0.f;
#pragma unroll 1   // one copy of the range walk: its registers are the kernel's (the 8-byte shape has 64 to stay within)
  for (int turn = 0; turn < TURNS; ++turn) {
    const int sub = wave * 4 + gw * TURNS + turn;
    float range[VEC];
    range_walk<VEC, COH>(a, rsrc, stride, per, full, sub * per, p0, range);
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
#pragma unroll
      for (int t = 0; t < TURNS; ++t) rs[t][c] = (turn == t) ? range[c] : rs[t][c];   // turn is uniform: a scalar select
    }
  }
  // quarter of this wave: the chain over its four ranges -- range 4 w + g * TURNS + turn sits in lane group g, slot turn
  {
    float q[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) q[c] = rs[0][c];
#pragma unroll
    for (int g = 0; g < VEC; ++g)
#pragma unroll
      for (int turn = 0; turn < TURNS; ++turn) {
        if (g == 0 && turn == 0) continue;
#pragma unroll
        for (int c = 0; c < VEC; ++c) q[c] += (g == 0) ? rs[turn][c] : __shfl(rs[turn][c], lp + g * LP, 64);
      }
    if (gw == 0) {
#pragma unroll
      for (int c = 0; c < VEC; ++c) gsum[wave][lp * VEC + c] = q[c];
    }
  }
  __syncthreads();
  float g = 0.f;
  if (tid < RED_PARAMS) {  // wave 0: fold the quarters
    g = (gsum[0][tid] + gsum[1][tid]) + (gsum[2][tid] + gsum[3][tid]);
    // (dst: canonical slabs: position = parameter index; register-order slabs (ppo_grad_fast_kernel): through the table)
    g = dst >= 0 ? extra_terms(a, dst, g) : 0.f;
  }
  return g;
}
// minibatch statistics (means over the nb rows) and the KL decision, by one block (>= 256 threads); thread 0 returns `stop`
template <bool COH = false>
__device__ __forceinline__ bool reduce_statistics(const ReduceArgs& a, float (*part)[NSTATP], float* means, bool bump_step) {
  const int tid = threadIdx.x;
  bool stop = false;
  if (tid < 256) {  // 32 lanes per statistic, strided over the workgroup partials, then a fixed-order fold
    const int kst = tid & (NSTATP - 1), j = tid >> 3;
    float v = 0.f;
    // sixteen loads in flight per lane (the bench shape's 512 partial records: ONE round trip; written as `v += load` the compiler
    // waits for every load before it issues the next -- 16 dependent round trips, which made this block the long pole of the launch)
    for (int w0 = j; w0 < a.nstatpart; w0 += 32 * 16) {
      float x[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int w = w0 + 32 * u;
        x[u] = 0.f;
        if (w < a.nstatpart) {
          if constexpr (COH) x[u] = __hip_atomic_load(a.statpart + (size_t)w * NSTATP + kst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else x[u] = a.statpart[(size_t)w * NSTATP + kst];
        }
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (w0 + 32 * u < a.nstatpart) v += x[u];
    }
    part[j][kst] = v;
  }
  __syncthreads();
  if (tid < NSTATP) {
    float v = 0.f;
    for (int j = 0; j < 32; ++j) v += part[j][tid];
    means[tid] = v / (float)a.nb;
  }
  __syncthreads();
  if (tid == 0) {
    const float pl_ = means[0], vl = means[1], el = means[2], cf = means[3], kl = means[4];
    stop = (a.target_kl >= 0.f) && (kl > 1.5f * a.target_kl);
    if (bump_step && !stop && a.step) *a.step += 1;
    a.scalars[0] = kl;
    a.scalars[1] = stop ? 0.f : 1.f;
    a.scalars[2] = stop ? 1.f : 0.f;  // ppo_adam_kernel raises stop_flag (next launch), never mid-kernel
    if (a.stats_out) {
      a.stats_out[0] = pl_;
      a.stats_out[1] = vl;
      a.stats_out[2] = el;
      a.stats_out[3] = cf;
      a.stats_out[4] = kl;
      a.stats_out[5] = pl_ + a.ent_coef * el + a.vf_coef * vl;
      a.stats_out[6] = 0.f;
      a.stats_out[7] = stop ? 0.f : 1.f;
    }
    if (a.n_extra > 0) {   // raw additional term of this minibatch (adap_learn.py:313-320: loss += coeff * context_loss)
      float raw = 0.f;
      for (int k0 = 0; k0 < a.n_extra; k0 += 16) {   // loads batched, adds in index order
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = (k0 + u < a.n_extra) ? a.extra_loss[k0 + u] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (k0 + u < a.n_extra) raw += x[u];
      }
      raw *= a.extra_norm;
      if (a.extra_loss_out) *a.extra_loss_out = raw;
      if (a.stats_out) a.stats_out[5] += a.extra_coef * raw;
    }
  }
  return stop;
}

// ---- reduce + clip + Adam as ONE launch (a learner that has the device to itself: the three launches of a minibatch step sit on
// its critical path, and two of the three kernel boundaries plus the separate pass over the gradient go away) -----------------
// Every block reduces its 64 slab positions (the tree above), publishes its sum of squares as ONE 8-byte word {tag, value}
// (tag = launch generation + 1: a single-copy-atomic store, nothing else crosses blocks), sweeps all blocks' words until they carry
// this launch's tag (all blocks are resident: the launcher checks the grid against the occupancy query), folds them in
// ppo_adam_kernel's order -- so the fused and the two-launch path give bitwise the same parameters -- and applies clip + Adam to its
// own 64 entries straight from registers.  Block 0 also does the minibatch statistics and the KL decision and carries `stop` in
// a word of its own.  A sweep that does not complete within `timeout` ticks (never, unless a block cannot be scheduled) counts in
// *sweep_error and skips the update instead of hanging the device.
// What ONE wave of a step does once its block's words are out: sweep all words (the nblk 64-position sums of squares and the stop
// word at index nblk) until they carry this launch's tag, fold the norm in ppo_adam_kernel's order, clip + Adam of the lane's
// entry g of parameter dst (-1: none) whose moments / value / image positions the caller fetched.  first: the one wave of the
// launch that advances the generation, the optimizer step and the stop flag.
__device__ __forceinline__ void step_finish(const StepArgs& s, int nblk, unsigned tag, int step_new, int lane, bool first, float g,
                                            int dst, float m0, float v0, float p0, int i0, int i1) {
  const ReduceArgs& a = s.r;
  const AdamArgs& ad = s.ad;
  // Adam's bias corrections need the step count only: two double-precision pow() under the wait for the other blocks' words
  AdamScalars k = adam_scalars(0.f, ad.max_norm, step_new, ad.lr, ad.beta1, ad.beta2);
  // ---- wave 0: sweep the nblk + 1 words; lane l takes words l, l + 64, ... ----
  constexpr int MAXW = 16;   // nblk + 1 <= 64 * MAXW (the launcher checks)
  float qv[MAXW];
  bool ok = false;
  const long long t0 = wall_clock64();
  while (true) {
    bool all = true;
#pragma unroll
    for (int i = 0; i < MAXW; ++i) {
      const int k = lane + 64 * i;
      qv[i] = 0.f;
      if (k <= nblk) {
        const unsigned long long w = __hip_atomic_load(s.words + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        all = all && ((unsigned)(w >> 32) == tag);
        qv[i] = __uint_as_float((unsigned)w);
      }
    }
    if (__all(all)) { ok = true; break; }
    if ((unsigned long long)(wall_clock64() - t0) > s.timeout) break;
    __builtin_amdgcn_s_sleep(4);
  }
  if (!ok) {   // never on a healthy device; this block's parameters stay as they are, the host is told (ph_ctx_step_errors)
    if (lane == 0) {
      atomicAdd(s.sweep_error, 1u);
      // (the generation is left alone: every later launch of this context bails out on *sweep_error before it sweeps --
      // ppo_step_kernel -- so words this launch's late blocks still publish are never taken for a later launch's)
      if (first && a.stats_out) a.stats_out[7] = -1.f;
    }
    return;
  }
  // the stop word sits at index nblk: lane nblk % 64, slot nblk / 64
  int stop_i = 0;
#pragma unroll
  for (int i = 0; i < MAXW; ++i)
    if (lane + 64 * i == nblk) stop_i = __float_as_int(qv[i]) & 1;
  const bool stop = __any(stop_i != 0);
  // total = sum of the nblk squares in ppo_adam_kernel's order: its thread t (256 of them) adds entries t, t + 256, ...; each of its
  // four waves folds by shuffles; (w0 + w1) + (w2 + w3).  Entry k lives in lane k % 64, slot k / 64: thread t = 64 v + lane of
  // "virtual wave" v owns slots v, v + 4, v + 8, ...
  float shv[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    float q = 0.f;
#pragma unroll
    for (int i = v; i < MAXW; i += 4)
      if (lane + 64 * i < nblk) q += qv[i];
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
    shv[v] = __shfl(q, 0, 64);
  }
  const float total = sqrtf((shv[0] + shv[1]) + (shv[2] + shv[3]));
  if (first && lane == 0) {   // every block has published, i.e. has read gen / step / stop_flag: now they may change
    if (!stop && ad.step) *const_cast<int*>(ad.step) = step_new;
    if (stop) *a.stop_flag = 1;
    *s.gen = tag;
    if (!stop && a.stats_out) a.stats_out[6] = total;
  }
  if (stop || dst < 0) return;
  {
    const float cc = ad.max_norm / (total + 1e-6f);  // torch.nn.utils.clip_grad_norm_ (as in adam_scalars)
    k.coef = cc < 1.0f ? cc : 1.0f;
  }
  const int p = dst;
  float m, v;
  const float pn = adam_apply(g, k, ad.beta1, ad.beta2, ad.eps, m0, v0, p0, &m, &v);
  ad.m[p] = m;
  ad.v[p] = v;
  ad.params[p] = pn;
  if (ad.wimage) wimage_put_at(ad.wimage, i0, i1, pn);
}


// what block 0 of a minibatch step leaves behind when an earlier minibatch of the train() call hit the KL early stop
__device__ __forceinline__ void step_stopped(const ReduceArgs& a) {
  const int tid = threadIdx.x;
  if (tid < PH_NSTAT && a.stats_out) a.stats_out[tid] = 0.f;
  if (tid == 0) {
    a.scalars[1] = 0.f;
    a.scalars[2] = 0.f;
  }
}
// The step of slab block `blk` of `nblk` (blk == nblk: the statistics block), 256 lanes.  tag = launch generation + 1 and
// step_new = optimizer step + 1 were read by the caller before anything of this launch was published.
// `stopped` (an earlier minibatch of this train() call hit the KL early stop) and `err` (an earlier wait of this context expired) are
// the launch's two reasons to do nothing.  They arrive as VALUES the caller loaded at the top of the kernel and are TESTED here behind
// the slab walk: tested at the top, every block waited for that round trip of scalar loads before its first slab load went out
// (PH_STEP_LATE_CHECKS=0 restores the test at the top; same-box A/B profiles/r06_bl_*).  A block that walked for nothing publishes and
// writes nothing; the statistics block, whose reduction has side effects, still tests first.
#ifndef PH_STEP_LATE_CHECKS
#define PH_STEP_LATE_CHECKS 1
#endif
__device__ __forceinline__ bool step_refused(const ReduceArgs& a, int blk, int stopped, unsigned err) {
  if (err != 0u) {
    // (ph_ctx_step_errors) blocks of the expired launch which could not be scheduled in time may still publish words -- tagged with a
    // generation a later launch would reuse -- so nothing after the first expiry sweeps at all: the update is skipped and marked
    // (stats[7] = -1), the host raises
    if (blk == 0 && threadIdx.x == 0 && a.stats_out) a.stats_out[7] = -1.f;
    return true;
  }
  if (stopped != 0) {  // (stable for the whole launch)
    if (blk == 0) step_stopped(a);
    return true;
  }
  return false;
}
template <int VEC, bool COH = false>
__device__ __forceinline__ void step_body(const StepArgs& s, int blk, int nblk, unsigned tag, int step_new, float (*gsum)[RED_PARAMS],
                                          float (*part)[NSTATP], float* means, const void* /*unused*/, int stopped = 0, unsigned err = 0u) {
  const ReduceArgs& a = s.r;
  const AdamArgs& ad = s.ad;
  const int tid = threadIdx.x;
  if (blk == nblk) {   // the extra block: statistics + KL decision while the slab blocks reduce; its word carries `stop`
    if (step_refused(a, blk, stopped, err)) return;
    const bool stop = reduce_statistics<COH>(a, part, means, false);
    if (tid == 0)
      __hip_atomic_store(s.words + nblk, ((unsigned long long)tag << 32) | (stop ? 1ull : 0ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // wave 0's lanes: the parameter behind their slab position, then its moments, value and image positions -- two dependent round
  // trips that complete under the slab walk
  const int dst = reduce_dst(a, blk);
  float m0 = 0.f, v0 = 0.f, p0 = 0.f;
  int i0 = -1, i1 = -1;
  if (dst >= 0) {
    m0 = ad.m[dst];
    v0 = ad.v[dst];
    p0 = ad.params[dst];
    if (ad.wimage) {
      i0 = ad.wimage_map[2 * dst];
      i1 = ad.wimage_map[2 * dst + 1];
    }
  }
  const float g = reduce_positions<VEC, COH>(a, gsum, dst, blk);
  if (step_refused(a, blk, stopped, err)) return;
  if (tid < 64) {
    float q = g * g;
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
    if (tid == 0)
      __hip_atomic_store(s.words + blk, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(q), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid >= 64) return;
  step_finish(s, nblk, tag, step_new, tid, blk == 0, g, dst, m0, v0, p0, i0, i1);
}


__host__ __device__ inline int reduce_blocks_of(int slab_len) { return (slab_len + RED_PARAMS - 1) / RED_PARAMS; }

}  // namespace ph

// K3 + K5 + K6: one PPO minibatch step = {advantage statistics, gather + forward + loss + backward, slab
// reduction + global-norm, clip + Adam}.   Reference: SB3 PPO.train() called from
// pantheonrl/common/agents.py:155; arithmetic restated from SURVEY.md A.3 and the in-tree copy
// pantheonrl/algos/adap/adap_learn.py:253-344.
//
// ppo_grad_kernel: grid (nWG, 2).  blockIdx.y selects the policy net (0) or the value net (1): the two SB3
// MLPs share nothing but the input, so each workgroup keeps ONE net's 64x64 blocks in LDS (~69 KB -> two
// workgroups per CU).  A workgroup walks row tiles of R=64 minibatch rows (gathered straight from the time-major
// rollout buffer through the env-major index n = e*T + t); every layer, its transpose-products for the weight
// gradients and the activation back-propagation run as 32x32 v_mfma_f32_32x32x2_f32 tiles on LDS operands, one
// tile per wave.  Weight-gradient tiles accumulate in the workgroup's private slab in HBM/L2 (the MFMA accumulator
// is initialised from the slab, so the add is free); a second kernel sums the slabs in a fixed order
// (deterministic), a third applies clip_grad_norm_ + Adam.
#include "ph_launch.h"
#include "ph_split.h"
#include "ph_step.h"

namespace ph {


// Weight-gradient tiles accumulate across the tiles of a workgroup in its private slab: the MFMA accumulator is
// initialised from the slab (the loads hide under the operand prefetch of the tile product), then stored back.
// (No-return L2 float atomics instead of the reload measured 8 % slower on the whole kernel.)
// OH: Discrete-family (one-hot) observations -- the chunk loops below differ; H16: the per-component head phase (nd.head16).
// Compile-time so that the Box / small-head instantiations keep their register budget (the one-hot loop holds the next
// chunk's W1 rows in registers across the products, the per-component head 32 logit slots).
// s + sum_{r < n} p[r * stride] in row order, 16 LDS reads in flight at a time (written as a plain loop the compiler emits one
// read - wait - add per element: 64 LDS latencies on the wave that also owns an MFMA tile of the phase)
template <int N>
__device__ __forceinline__ float lds_colsum(const float* p, int stride, float s) {
#pragma unroll
  for (int r0 = 0; r0 < N; r0 += 16) {
    float t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = p[(r0 + i) * stride];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s += t[i];
  }
  return s;
}
// s + sum_{r < n} p[r * stride] * q[r], same batching
template <int N>
__device__ __forceinline__ float lds_coldot(const float* p, int stride, const float* q, int qstride, float s) {
#pragma unroll
  for (int r0 = 0; r0 < N; r0 += 16) {
    float t[16], u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      t[i] = p[(r0 + i) * stride];
      u[i] = q[(r0 + i) * qstride];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; ++i) s = __builtin_fmaf(t[i], u[i], s);
  }
  return s;
}

// ---- one-hot first layer without a materialised X (OHR instantiations) ---------------------------------------------------------
// A one-hot observation row is a set of D hot features.  Per tile the workgroup builds two bit tables from the observations:
// rowmask[r][w] -- bit b set iff feature 32 w + b of row r is hot -- and colmask[f][w] -- bit b set iff row 32 w + b has feature
// f hot.  The MFMA A operands of X W1 (lane = row, k = feature) and of X^T dZ1 (lane = feature, k = row) are then one v_bfe +
// v_cvt on a register: no X chunk in LDS, no zero fill, and the dW1 chunk products need no barrier between them.  (Keeping
// every chunk of W1 resident as well -- 159 KB for Liar's Dice -- measured slower in situ: the two learners' update launches
// then cannot share a CU.)
// Same k order and the same 0.0 / 1.0 operand values as the dense products: bitwise the same accumulators.
template <bool VALU>
__device__ __forceinline__ f32x16 tile_mma_onehot_fwd(const unsigned* rowmask, int RW, int c, const float* B, int ldb, int m0, int n0,
                                                      f32x16 acc, int lane) {
  const int i = lane & 31, h = lane >> 5;
  if constexpr (!VALU) {
    const unsigned w0 = rowmask[(m0 + i) * RW + 2 * c], w1 = rowmask[(m0 + i) * RW + 2 * c + 1];
    const float* bp = B + h * ldb + n0 + i;
    float b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) b0[u] = bp[2 * u * ldb];
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      bp += 8 * ldb;
      if (s + 4 < 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) b1[u] = bp[2 * u * ldb];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned w = (s + u) < 16 ? w0 : w1;
        const float av = (float)((w >> (((2 * (s + u)) & 31) + h)) & 1u);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[u], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) b0[u] = b1[u];
    }
  } else {
    const int col = n0 + i;
    for (int k = 0; k < HID; ++k) {
      const float b = B[k * ldb + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + drow(r, h);
        const float av = (float)((rowmask[row * RW + 2 * c + (k >> 5)] >> (k & 31)) & 1u);
        acc[r] = __builtin_fmaf(av, b, acc[r]);
      }
    }
  }
  return acc;
}
// acc[f = m0 + .., n] += sum over the 64 rows of X[row][c*64 + f] * B[row][n]
template <bool VALU>
__device__ __forceinline__ f32x16 tile_mma_onehot_bwd(const unsigned* colmask, int c, const float* B, int ldb, int m0, int n0,
                                                      f32x16 acc, int lane) {
  const int i = lane & 31, h = lane >> 5;
  if constexpr (!VALU) {
    const unsigned w0 = colmask[(c * HID + m0 + i) * 2], w1 = colmask[(c * HID + m0 + i) * 2 + 1];
    const float* bp = B + h * ldb + n0 + i;
    float b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) b0[u] = bp[2 * u * ldb];
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      bp += 8 * ldb;
      if (s + 4 < 32) {
#pragma unroll
        for (int u = 0; u < 4; ++u) b1[u] = bp[2 * u * ldb];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned w = (s + u) < 16 ? w0 : w1;
        const float av = (float)((w >> (((2 * (s + u)) & 31) + h)) & 1u);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0[u], acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) b0[u] = b1[u];
    }
  } else {
    const int col = n0 + i;
    for (int k = 0; k < HID; ++k) {   // k = row
      const float b = B[k * ldb + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = c * HID + m0 + drow(r, h);
        const float av = (float)((colmask[f * 2 + (k >> 5)] >> (k & 31)) & 1u);
        acc[r] = __builtin_fmaf(av, b, acc[r]);
      }
    }
  }
  return acc;
}

template <int R, int LP, bool VALU, bool OH, bool H16, bool OHR = false>
__global__ __launch_bounds__(R * 4, 2) void ppo_grad_kernel(GradArgs a) {
  if (*a.stop_flag) return;
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = R * 4;
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  constexpr int Lp = LP, LDO = LP + 1;  // logits padded to the MFMA tile width: compile-time so LDS offsets fold
  float* bufA = smem;                  // [R][LDH]  X chunk -> H2 -> dZ2 -> X chunk
  float* bufB = bufA + R * LDH;        // [R][LDH]  H1 -> dZ1
  float* regW = bufB + R * LDH;        // W1 chunk [64][LDH]  |  Wo [64][LDO] + OUT [R][LDO]
  constexpr int regW_sz = (HID * LDH > (HID + R) * LDO) ? HID * LDH : (HID + R) * LDO;
  float* w2s = regW + regW_sz;         // [64][LDH]
  float* b1s = w2s + HID * LDH;        // [64]
  float* b2s = b1s + HID;              // [64]
  float* bos = b2s + HID;              // act_b [Lp]  (policy)  |  val_W [64] (value)
  float* radv = bos + 64;              // [R] normalised advantage (policy) | returns (value)
  float* rold = radv + R;              // [R] old log-prob (policy) | old values (value)
  float* rdv = rold + R;               // [R] dL/dv (value net)
  float* red = rdv + R;                // [NSTATP * 4] cross-wave stat reduction
  float* hlp = red + NSTATP * 4;       // [R][4] log-prob of the taken action, per action component (head16 shapes)
  float* hen = hlp + 4 * R;            // [R][4] entropy per action component
  int* rowphys = (int*)(hen + 4 * R);  // [R]
  int* feat = rowphys + R;             // [R][D] one-hot positions of the tile (Discrete-family observations only)
  constexpr bool onehot = OH;
  int* fcomp = feat + R * nd.D;        // [nchunk * 64] observation component of every feature (one-hot only)
  // OHR: no feat / fcomp; bit tables of the tile's hot features and every chunk of W1 resident
  const int RW = 2 * nd.nchunk;                        // mask words per row
  unsigned* rowmask = (unsigned*)(rowphys + R);        // [R][RW]
  unsigned* colmask = rowmask + R * RW;                // [nchunk * 64][2]
  float* wos = regW;
  float* outs = regW + HID * LDO;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int net = blockIdx.y;
  const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
  const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
  float* slab = a.slabs + (size_t)blockIdx.x * lay.P;
  const float inv_nb = 1.0f / (float)a.nb;

  // prologue: W2 and the bias vectors are issued now and committed after the first tile's row metadata, so their
  // latency overlaps the index gathers of S0
  WStage<NT> w2r;
  w2r.issue(a.params + oW2, 0, HID);
  float bias1 = 0.f, bias2 = 0.f, bias3 = 0.f;
  if (tid < HID) {
    bias1 = a.params[oB1 + tid];
    bias2 = a.params[oB2 + tid];
    bias3 = (net == 0) ? ((tid < nd.L) ? a.params[lay.act_b + tid] : 0.f) : a.params[lay.val_W + tid];
  }

  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;

  // S0: row metadata of one tile (gather indices -> physical rows, per-row scalars)
  auto stage_rows = [&](int tile) {
    if (tid < R) {
      const int gi = tile * R + tid;
      int phys = -1;
      float adv = 0.f, old = 0.f;
      if (gi < a.nb) {
        phys = minibatch_row(a, gi);
        if (net == 0) {
          adv = a.rb_adv[phys];
          if (a.norm_adv && a.nb > 1) adv = (adv - a.advstats[0]) / (a.advstats[1] + 1e-8f);
          old = a.rb_logp[phys];
        } else {
          adv = a.rb_ret[phys];
          old = a.rb_val[phys];
        }
      }
      rowphys[tid] = phys;
      radv[tid] = adv;
      rold[tid] = old;
    }
  };
  stage_rows(blockIdx.x);  // overlaps the W2 / bias loads issued above
  if constexpr (onehot && !OHR) XStage<R, NT>::build_fcomp(fcomp, nd, tid);
  w2r.commit(w2s);

  if (tid < HID) {
    b1s[tid] = bias1;
    b2s[tid] = bias2;
    bos[tid] = bias3;
  }

  bool first = true;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, first = false) {
    __syncthreads();  // row metadata (and, first time, W2 / biases) visible
    if (first) PH_STAMP(a.prof, 1);
    // every thread-id-derived coordinate is re-materialised per tile from an opaque copy of threadIdx.x: otherwise the
    // compiler hoists ~160 loop-invariant per-register LDS / slab addresses out of the tile loop and pins them in
    // VGPRs for the whole kernel (217 VGPRs, spills at 3 waves/SIMD); with this the kernel needs 170
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63, wave = tid >> 6;
    const int mt = wave >> 1, nt = wave & 1;  // this wave's 32x32 tile of every [R x 64] / [64 x 64] product
    const int li = lane & 31, lh = lane >> 5;

    // ---- S1: Z1 = X W1 over feature chunks; H1 = tanh(Z1 + b1) -> bufB ----
    f32x16 acc = {0};
    XStage<R, NT> xr;
    WStage<NT> w1r;
    WoStage<NT, LP> wor;   // sized by the head width of this instantiation
    if constexpr (OHR) {
      // bit tables of this tile: zero, then one LDS OR per (row, component) -- order-independent, so deterministic
      for (int e = tid; e < R * RW + nd.nchunk * HID * 2; e += NT) rowmask[e] = 0u;   // rowmask and colmask are contiguous
      w1r.issue(a.params + oW1, 0, nd.F, tid);
      if (net == 0) wor.issue(a.params + lay.act_W, nd.L, Lp, tid);   // consumed after the chunk loop
      __syncthreads();
      for (int e = tid; e < R * nd.D; e += NT) {
        const int r = e / nd.D, comp = e - r * nd.D;
        const int ph_row = rowphys[r];
        if (ph_row < 0) continue;
        const int lo = nd.obs_off[comp], n = nd.obs_off[comp + 1] - lo;
        int x = (int)a.rb_obs[(size_t)ph_row * nd.D + comp];
        x = x < 0 ? 0 : (x >= n ? n - 1 : x);
        const int f = lo + x;
        atomicOr(&rowmask[r * RW + (f >> 5)], 1u << (f & 31));
        atomicOr(&colmask[f * 2 + (r >> 5)], 1u << (r & 31));
      }
      // the W1 chunks stream through the one LDS buffer (two workgroups per CU: the two learners' updates run side by side);
      // chunk c + 1 is loaded into registers while chunk c's products run
      for (int c = 0; c < nd.nchunk; ++c) {
        if (c > 0) __syncthreads();   // previous chunk consumed
        w1r.commit(regW, tid);
        __syncthreads();              // (first chunk: the bit tables are complete as well)
        if (first) PH_STAMP(a.prof, 2);
        if (c + 1 < nd.nchunk) w1r.issue(a.params + oW1, (c + 1) * HID, nd.F, tid);
        acc = tile_mma_onehot_fwd<VALU>(rowmask, RW, c, regW, LDH, mt * 32, nt * 32, acc, lane);
      }
    } else if constexpr (onehot) {
      // One-hot observations: the hot feature row of every (row, component) once per tile; the chunks of S1 and S7 are then
      // built from LDS (commit_onehot: no zero fill, no scatter).  (Replacing S1 by a gather-sum of W1 rows, as the 16-row
      // forward kernel does, measured no faster at 64 rows per workgroup: 491 KB of gathered rows per workgroup against 69 KB
      // of W1 streamed once through LDS.)  A minibatch of this shape class is one tile per workgroup and one workgroup per CU
      // (8 192 rows = 128 tiles), so nothing else hides a chunk's W1 load: chunk c + 1's rows are loaded into registers while
      // chunk c's products run and committed after the barrier that frees the single LDS buffer -- the loop pays the load
      // latency once, not once per chunk.
      w1r.issue(a.params + oW1, 0, nd.F, tid);
      xr.build_feat(feat, rowphys, a.rb_obs, nd, tid);
      if (net == 0) wor.issue(a.params + lay.act_W, nd.L, Lp, tid);   // consumed after the loop
      for (int c = 0; c < nd.nchunk; ++c) {
        __syncthreads();  // previous chunk consumed; (first chunk) the tile's hot positions visible
        xr.commit_onehot(bufA, fcomp, nd, c, tid);
        w1r.commit(regW, tid);
        __syncthreads();
        if (first) PH_STAMP(a.prof, 2);
        if (c + 1 < nd.nchunk) w1r.issue(a.params + oW1, (c + 1) * HID, nd.F, tid);   // in flight during the products
        acc = tile_mma<false, false, VALU>(bufA, LDH, regW, LDH, mt * 32, nt * 32, 0, HID, acc, lane);
      }
    } else {
      for (int c = 0; c < nd.nchunk; ++c) {
        if (c > 0) __syncthreads();  // previous chunk consumed
        xr.issue(rowphys, a.rb_obs, nd, c, tid);
        w1r.issue(a.params + oW1, c * HID, nd.F, tid);
        xr.commit(bufA, rowphys, a.rb_obs, nd, c, tid);
        w1r.commit(regW, tid);
        __syncthreads();
        if (first) PH_STAMP(a.prof, 2);
        if (net == 0 && c == nd.nchunk - 1) wor.issue(a.params + lay.act_W, nd.L, Lp, tid);  // lands during the MFMAs
        acc = tile_mma<false, false, VALU>(bufA, LDH, regW, LDH, mt * 32, nt * 32, 0, HID, acc, lane);
      }
    }
    if (first) PH_STAMP(a.prof, 3);
    {
      const int col = nt * 32 + li;
      const float bb = b1s[col];   // once: inside the loop it is re-read (and waited for) after every store to bufB
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(mt * 32 + drow(r, lh)) * LDH + col] = fast_tanh(acc[r] + bb);
    }
    __syncthreads();  // every wave is done with the W1 chunk in regW and H1 is complete
    if (net == 0) wor.commit(wos, Lp, LDO, tid);
    if (first) PH_STAMP(a.prof, 4);

    // ---- S2: H2 = tanh(H1 W2 + b2) -> bufA ----
    {
      f32x16 acc2 = {0};
      acc2 = tile_mma<false, false, VALU>(bufB, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, acc2, lane);
      const int col = nt * 32 + li;
      const float bb = b2s[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) bufA[(mt * 32 + drow(r, lh)) * LDH + col] = fast_tanh(acc2[r] + bb);
    }
    __syncthreads();
    if (first) PH_STAMP(a.prof, 5);

    if (net == 0) {
      // ---- S3: logits = H2 Wo + bo -> OUT ----
      const int ntn = Lp >> 5;
      if (wave < (R >> 5) * ntn) {
        const int hm = wave / ntn, hn = wave - hm * ntn;
        f32x16 acc3 = {0};
        acc3 = tile_mma<false, false, VALU>(bufA, LDH, wos, LDO, hm * 32, hn * 32, 0, HID, acc3, lane);
        const int col = hn * 32 + li;
        const float bb = bos[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) outs[(hm * 32 + drow(r, lh)) * LDO + col] = acc3[r] + bb;
      }
      __syncthreads();
      if (first) PH_STAMP(a.prof, 6);

      // ---- S4: clipped-surrogate + entropy loss per row, dL/dlogits written over OUT ----
      if constexpr (H16) {
        // MultiDiscrete heads / up to 16 logits per component (Liar's Dice: 7 + 12): wave c takes action component c of every
        // row; a thread reads its component's logits in ONE batch of LDS reads and keeps them in registers -- one exp per
        // logit -- instead of one lane per row walking every component in five dependent passes over LDS (17 k cycles per
        // tile for 19 logits).  The per-component log-prob and entropy meet in LDS; the row's scalars are then recomputed by
        // each of its component threads.  log-prob = z[a] - (m + log sum exp(z - m)), accumulated over components in order.
        const int hrow = tid & (R - 1), hcomp = tid / R;
        const bool hon = hcomp < nd.A;
        const int physh = rowphys[hrow];
        int lo = 0, nk = 0, act = 0;
        float zc[16], pc[16], hc = 0.f;
        float* zrow = outs + hrow * LDO;
        if (hon) {
          lo = nd.act_off[hcomp];
          nk = nd.act_off[hcomp + 1] - lo;
          if (physh >= 0) {
            act = (int)a.rb_act[(size_t)physh * nd.A + hcomp];   // in flight during the LDS reads
#pragma unroll
            for (int k = 0; k < 16; ++k) zc[k] = (k < nk) ? zrow[lo + k] : -3.0e38f;
            float m = zc[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) m = fmaxf(m, zc[k]);
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              pc[k] = (k < nk) ? fast_exp(zc[k] - m) : 0.f;
              if (k < nk) se += pc[k];
            }
            const float lse = m + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
            act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
            float zact = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              zact = (k == act) ? zc[k] : zact;
              pc[k] *= inv;
              zc[k] -= lse;                                   // log-probability of the slot
              if (k < nk) hc -= pc[k] * zc[k];
            }
            hlp[hrow * 4 + hcomp] = zact - lse;
            hen[hrow * 4 + hcomp] = hc;
          }
        }
        __syncthreads();
        if (hon) {
          if (physh < 0) {
            for (int k = 0; k < nk; ++k) zrow[lo + k] = 0.f;
          } else {
            float logp = 0.f, ent = 0.f;
            for (int c = 0; c < nd.A; ++c) {
              logp += hlp[hrow * 4 + c];
              ent += hen[hrow * 4 + c];
            }
            const float adv = radv[hrow];
            const float lr = logp - rold[hrow];
            const float ratio = fast_exp(lr);
            const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
            const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
            const float pl1 = adv * ratio, pl2 = adv * rc;
            const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
            const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);   // torch.min / clamp backward
            const float g_lp = -inv_nb * adv * ratio * gate;
            const float g_en = -a.ent_coef * inv_nb;
            if (hcomp == 0) {
              st[0] += -fminf(pl1, pl2);
              st[2] += -ent;
              st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
              st[4] += (ratio - 1.0f) - lr;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k)
              if (k < nk) zrow[lo + k] = g_lp * (((k == act) ? 1.f : 0.f) - pc[k]) + g_en * (-pc[k] * (zc[k] + hc));
          }
        }
        if (tid < R)
          for (int k = nd.L; k < Lp; ++k) zrow[k] = 0.f;
      } else if (tid < R) {
        float* z = outs + tid * LDO;
        const int phys = rowphys[tid];
        if (phys < 0) {
          for (int k = 0; k < Lp; ++k) z[k] = 0.f;
        } else if (nd.gauss) {
          // Box action space (SB3 DiagGaussianDistribution; A <= 16 <= Lp / 2): z[0..A) are the means.  log-prob and entropy as in
          // general_row_tail; dL/dmean_c = g_lp (a - mu) / sigma^2 goes over z[c] (the head's dOut), dL/dlog_std_c =
          // g_lp (((a - mu) / sigma)^2 - 1) + g_en to column 16 + c of the row: S5a sums that column over the rows like a bias
          // gradient, the head products see it times Wo's zero padding columns
          const float* ls = a.params + lay.val_b + 1;
          float logp = 0.f, ent = 0.f;
          for (int c = 0; c < nd.A; ++c) {
            const float lsd = ls[c];
            const float d = (a.rb_act[(size_t)phys * nd.A + c] - z[c]) * fast_exp(-lsd);
            logp += (-0.5f * d * d - lsd) - 0.91893853320467274178f;
            ent += 1.41893853320467274178f + lsd;
          }
          const float adv = radv[tid];
          const float lr = logp - rold[tid];
          const float ratio = fast_exp(lr);
          const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
          const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
          const float pl1 = adv * ratio, pl2 = adv * rc;
          const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
          const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
          const float g_lp = -inv_nb * adv * ratio * gate;
          const float g_en = -a.ent_coef * inv_nb;
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
          for (int k = nd.A; k < Lp; ++k) z[k] = 0.f;
          for (int c = 0; c < nd.A; ++c) {
            const float is = fast_exp(-ls[c]);
            const float d = (a.rb_act[(size_t)phys * nd.A + c] - z[c]) * is;
            z[16 + c] = g_lp * (d * d - 1.0f) + g_en;
            z[c] = g_lp * d * is;
          }
        } else if (nd.A == 1 && nd.L <= 8) {
          // fast path (Discrete action space, <= 8 logits: every BASELINE config but Liar's Dice): the row lives in
          // registers, one LDS read and one LDS write per logit, one exp per logit
          const int nk = nd.L;
          float zr[8], pr[8];
          float m = -3.0e38f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            zr[k] = (k < nk) ? z[k] : -3.0e38f;
            m = fmaxf(m, zr[k]);
          }
          float se = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            pr[k] = (k < nk) ? fast_exp(zr[k] - m) : 0.f;
            se += pr[k];
          }
          const float lse = m + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
          int act = (int)a.rb_act[phys];
          act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
          float ent = 0.f, zact = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            pr[k] *= inv;
            const float lp = zr[k] - lse;
            ent -= (k < nk) ? pr[k] * lp : 0.f;
            zact = (k == act) ? zr[k] : zact;
          }
          const float logp = zact - lse;
          const float adv = radv[tid];
          const float lr = logp - rold[tid];
          const float ratio = fast_exp(lr);
          const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
          const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
          const float pl1 = adv * ratio, pl2 = adv * rc;
          const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
          const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
          const float g_lp = -inv_nb * adv * ratio * gate;
          const float g_en = -a.ent_coef * inv_nb;
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (k < nk) {
              const float dlogp = ((k == act) ? 1.f : 0.f) - pr[k];
              const float dent = -pr[k] * ((zr[k] - lse) + ent);
              z[k] = g_lp * dlogp + g_en * dent;
            }
          }
          for (int k = nk; k < Lp; ++k) z[k] = 0.f;
        } else {
          float logp = 0.f, ent = 0.f;
          // pass 1: log-prob and entropy (MultiDiscrete: sums over components)
          for (int c = 0; c < nd.A; ++c) {
            const int lo = nd.act_off[c], nk = nd.act_off[c + 1] - lo;
            float m = z[lo];
            for (int k = 1; k < nk; ++k) m = fmaxf(m, z[lo + k]);
            float se = 0.f;
            for (int k = 0; k < nk; ++k) se += fast_exp(z[lo + k] - m);
            const float lse = m + fast_log(se);
            int act = (int)a.rb_act[(size_t)phys * nd.A + c];
            act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
            float e = 0.f;
            for (int k = 0; k < nk; ++k) {
              const float lp = z[lo + k] - lse;
              e -= fast_exp(lp) * lp;
            }
            logp += z[lo + act] - lse;
            ent += e;
          }
          const float adv = radv[tid];
          const float lr = logp - rold[tid];
          const float ratio = fast_exp(lr);
          const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
          const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
          const float pl1 = adv * ratio, pl2 = adv * rc;
          // torch.min backward: the smaller branch gets the gradient, ties split 1/2 + 1/2; clamp passes the
          // gradient iff lo <= ratio <= hi.
          const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
          const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
          const float g_lp = -inv_nb * adv * ratio * gate;   // dL/dlogp
          const float g_en = -a.ent_coef * inv_nb;            // dL/dH
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
          // pass 2: dL/dz
          for (int c = 0; c < nd.A; ++c) {
            const int lo = nd.act_off[c], nk = nd.act_off[c + 1] - lo;
            float m = z[lo];
            for (int k = 1; k < nk; ++k) m = fmaxf(m, z[lo + k]);
            float se = 0.f;
            for (int k = 0; k < nk; ++k) se += fast_exp(z[lo + k] - m);
            const float lse = m + fast_log(se);
            int act = (int)a.rb_act[(size_t)phys * nd.A + c];
            act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
            float hc = 0.f;
            for (int k = 0; k < nk; ++k) {
              const float lp = z[lo + k] - lse;
              hc -= fast_exp(lp) * lp;
            }
            for (int k = 0; k < nk; ++k) {
              const float lp = z[lo + k] - lse;
              const float p = fast_exp(lp);
              const float dlogp = ((k == act) ? 1.f : 0.f) - p;
              const float dent = -p * (lp + hc);
              z[lo + k] = g_lp * dlogp + g_en * dent;
            }
          }
          for (int k = nd.L; k < Lp; ++k) z[k] = 0.f;
        }
      }
      __syncthreads();
      if (first) PH_STAMP(a.prof, 7);

      // ---- S5a: dWo = H2^T dOut (tiles 2 x Lp/32), d bo = column sums of dOut ----
      if (wave < 2 * ntn) {
        const int hm = wave / ntn, hn = wave - hm * ntn;
        f32x16 g = {0};
        if (!first) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int j = hm * 32 + drow(r, lh), col = hn * 32 + li;
            if (col < nd.L) g[r] = slab[lay.act_W + j * nd.L + col];
          }
        }
        g = tile_mma<true, false, VALU>(bufA, LDH, outs, LDO, hm * 32, hn * 32, 0, R, g, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = hm * 32 + drow(r, lh), col = hn * 32 + li;
          if (col < nd.L) slab[lay.act_W + j * nd.L + col] = g[r];
        }
      }
      if (tid >= NT - 64 && tid - (NT - 64) < nd.L) {  // last wave: bias gradient
        const int k = tid - (NT - 64);
        float s = first ? 0.f : slab[lay.act_b + k];
        s = lds_colsum<R>(outs + k, LDO, s);
        slab[lay.act_b + k] = s;
      } else if (nd.gauss && tid >= NT - 64 + 16 && tid - (NT - 64 + 16) < nd.A) {  // d log_std: column 16 + c of the rows (S4)
        const int k = tid - (NT - 64 + 16);
        float s = first ? 0.f : slab[lay.val_b + 1 + k];
        s = lds_colsum<R>(outs + 16 + k, LDO, s);
        slab[lay.val_b + 1 + k] = s;
      }
      __syncthreads();
      if (first) PH_STAMP(a.prof, 8);

      // ---- S5b: dH2 = dOut Wo^T ; dZ2 = dH2 * (1 - H2^2) in place over H2 ----
      {
        f32x16 d = {0};
        d = tile_mma<false, true, VALU>(outs, LDO, wos, LDO, mt * 32, nt * 32, 0, Lp, d, lane);
        float hv[16];   // all reads, then all writes: interleaved, every read waits behind the previous (possibly aliasing) store
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = bufA[(mt * 32 + drow(r, lh)) * LDH + nt * 32 + li];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 16; ++r) bufA[(mt * 32 + drow(r, lh)) * LDH + nt * 32 + li] = d[r] * (1.0f - hv[r] * hv[r]);
      }
      __syncthreads();
    } else {
      // ---- value net S3/S4: v = H2 . val_W + val_b ; value loss ; dv ----
      if (tid < R) {
        const int phys = rowphys[tid];
        float dv = 0.f;
        if (phys >= 0) {
          float v = 0.f;
          v = lds_coldot<HID>(bufA + tid * LDH, 1, bos, 1, v);
          v += a.params[lay.val_b];
          const float retn = radv[tid], oldv = rold[tid];
          float vp = v, pass = 1.f;
          if (a.clip_vf >= 0.f) {
            const float dlt = v - oldv;
            pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
            vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
          }
          const float err = vp - retn;
          st[1] += err * err;
          dv = a.vf_coef * 2.0f * err * inv_nb * pass;
        }
        rdv[tid] = dv;
      }
      __syncthreads();
      // ---- S5a: d val_W[j] = sum_r H2[r][j] dv[r] ; d val_b = sum_r dv[r] ----
      if (tid < HID) {
        float s = first ? 0.f : slab[lay.val_W + tid];
        s = lds_coldot<R>(bufA + tid, LDH, rdv, 1, s);
        slab[lay.val_W + tid] = s;
      } else if (tid == HID) {
        float s = first ? 0.f : slab[lay.val_b];
        s = lds_colsum<R>(rdv, 1, s);
        slab[lay.val_b] = s;
      }
      __syncthreads();
      // ---- S5b: dZ2[r][j] = dv[r] * val_W[j] * (1 - H2^2) in place ----
      for (int e = tid; e < R * HID; e += NT) {
        const int r = e >> 6, j = e & 63;
        const float h = bufA[r * LDH + j];
        bufA[r * LDH + j] = rdv[r] * bos[j] * (1.0f - h * h);
      }
      __syncthreads();
    }
    if (first) PH_STAMP(a.prof, 9);

    // ---- S6a: dW2 = H1^T dZ2 ; d b2 ; dH1 = dZ2 W2^T (kept in registers) ----
    // S7 needs X chunk 0 again: with one feature chunk the staged registers of S1 are still live (no second
    // gather); otherwise it is re-issued here and lands during the MFMAs.
    f32x16 dh1 = {0};
    if (!OHR && nd.nchunk > 1) xr.issue(rowphys, a.rb_obs, nd, 0, tid);
    {
      f32x16 g = {0};
      if (!first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) g[r] = slab[oW2 + (mt * 32 + drow(r, lh)) * HID + nt * 32 + li];
      }
      g = tile_mma<true, false, VALU>(bufB, LDH, bufA, LDH, mt * 32, nt * 32, 0, R, g, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[oW2 + (mt * 32 + drow(r, lh)) * HID + nt * 32 + li] = g[r];
      if (tid < HID) {
        float s = first ? 0.f : slab[oB2 + tid];
        s = lds_colsum<R>(bufA + tid, LDH, s);
        slab[oB2 + tid] = s;
      }
      dh1 = tile_mma<false, true, VALU>(bufA, LDH, w2s, LDH, mt * 32, nt * 32, 0, HID, dh1, lane);
    }
    __syncthreads();  // dZ2 (bufA) and H1 (bufB) fully consumed
    if (first) PH_STAMP(a.prof, 10);
    // ---- S6b: dZ1 = dH1 * (1 - H1^2) in place over H1; X chunk 0 lands in bufA ----
    {
      float hv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[r] = bufB[(mt * 32 + drow(r, lh)) * LDH + nt * 32 + li];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) bufB[(mt * 32 + drow(r, lh)) * LDH + nt * 32 + li] = dh1[r] * (1.0f - hv[r] * hv[r]);
    }
    if constexpr (OHR) {
    } else if constexpr (onehot) xr.commit_onehot(bufA, fcomp, nd, 0, tid);
    else xr.commit(bufA, rowphys, a.rb_obs, nd, 0, tid);
    __syncthreads();
    if (first) PH_STAMP(a.prof, 11);
    // ---- S7: dW1 = X^T dZ1 per feature chunk ; d b1 ----
    if (tid < HID) {
      float s = first ? 0.f : slab[oB1 + tid];
      s = lds_colsum<R>(bufB + tid, LDH, s);
      slab[oB1 + tid] = s;
    }
    for (int c = 0; c < nd.nchunk; ++c) {
      if constexpr (OHR) {     // dW1 chunk c straight from the bit table: no X in LDS, no barrier between chunks
        f32x16 g = {0};
        if (!first) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = c * HID + mt * 32 + drow(r, lh);
            if (k < nd.F) g[r] = slab[oW1 + (size_t)k * HID + nt * 32 + li];
          }
        }
        g = tile_mma_onehot_bwd<VALU>(colmask, c, bufB, LDH, mt * 32, nt * 32, g, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = c * HID + mt * 32 + drow(r, lh);
          if (k < nd.F) slab[oW1 + (size_t)k * HID + nt * 32 + li] = g[r];
        }
        continue;
      }
      if (c > 0) {
        __syncthreads();  // previous chunk consumed
        if constexpr (onehot) {
          xr.commit_onehot(bufA, fcomp, nd, c, tid);
        } else {
          xr.issue(rowphys, a.rb_obs, nd, c, tid);
          xr.commit(bufA, rowphys, a.rb_obs, nd, c, tid);
        }
        __syncthreads();
      }
      f32x16 g = {0};
      if (!first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = c * HID + mt * 32 + drow(r, lh);
          if (k < nd.F) g[r] = slab[oW1 + (size_t)k * HID + nt * 32 + li];
        }
      }
      g = tile_mma<true, false, VALU>(bufA, LDH, bufB, LDH, mt * 32, nt * 32, 0, R, g, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = c * HID + mt * 32 + drow(r, lh);
        if (k < nd.F) slab[oW1 + (size_t)k * HID + nt * 32 + li] = g[r];
      }
    }
    if (tile + (int)gridDim.x < a.ntiles) {
      __syncthreads();  // this tile's row metadata fully consumed
      stage_rows(tile + gridDim.x);
    }
  }

  PH_STAMP(a.prof, 12);
  // ---- per-workgroup partial statistics (fixed reduction tree -> deterministic) ----
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) {
    float v = st[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    st[k] = v;
  }
  __syncthreads();
  if (lane == 0 && wave < 4) {
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) red[wave * NSTATP + k] = st[k];
  }
  __syncthreads();
  if (tid < NSTATP) {
    float v = 0.f;
    for (int w = 0; w < NT / 64 && w < 4; ++w) v += red[w * NSTATP + tid];
    a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + tid] = v;
  }
  PH_STAMP(a.prof, 13);
}


size_t grad_lds_bytes(int R, int Lp, int onehot_D, int nchunk) {
  const int LDO = Lp + 1;
  const int regW_sz = (HID * LDH > (HID + R) * LDO) ? HID * LDH : (HID + R) * LDO;
  return sizeof(float) * (size_t)(2 * R * LDH + regW_sz + HID * LDH + 3 * 64 + 3 * R + NSTATP * 4 + 8 * R + R + R * onehot_D +
                                  (onehot_D ? nchunk * HID : 0));
}
// OHR instantiations: bit tables instead of the hot-position / feature-component tables
static size_t grad_lds_bytes_ohr(int R, int Lp, int nchunk) {
  const int LDO = Lp + 1;
  const int regW_sz = (HID * LDH > (HID + R) * LDO) ? HID * LDH : (HID + R) * LDO;
  return sizeof(float) * (size_t)(2 * R * LDH + regW_sz + HID * LDH + 3 * 64 + 3 * R + NSTATP * 4 + 8 * R + R + R * 2 * nchunk +
                                  nchunk * HID * 2);
}
static bool grad_ohr_enabled() {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_GRAD_OHR");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled != 0;
}

template <int LP, bool VALU, bool OH, bool H16, bool OHR>
static hipError_t launch_grad_inst(const GradArgs& a, int nwg, size_t lds, hipStream_t s) {
  constexpr int R = 64;
  dim3 grid(nwg, 2), block(R * 4);
  static size_t allowed[64] = {0};  // > 64 KiB of dynamic LDS is opt-in per kernel and device (kept out of graph capture)
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (lds > allowed[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_kernel<R, LP, VALU, OH, H16, OHR>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed[dev] = lds;
  }
  hipLaunchKernelGGL((ppo_grad_kernel<R, LP, VALU, OH, H16, OHR>), grid, block, lds, s, a);
  return hipGetLastError();
}
template <int LP, bool VALU, bool OH, bool H16>
static hipError_t launch_grad_variant(const GradArgs& a, int nwg, hipStream_t s) {
  constexpr int R = 64;
  if constexpr (OH) {
    // one-hot observations: the first layer's products take their X operands from bit tables, no materialised X
    const size_t ohr = grad_lds_bytes_ohr(R, LP, a.nd.nchunk);
    if (grad_ohr_enabled() && ohr <= 80 * 1024) return launch_grad_inst<LP, VALU, true, H16, true>(a, nwg, ohr, s);
  }
  return launch_grad_inst<LP, VALU, OH, H16, false>(a, nwg, grad_lds_bytes(R, LP, OH ? a.nd.D : 0, a.nd.nchunk), s);
}
template <int LP, bool VALU>
static hipError_t launch_grad_shape(const GradArgs& a, int nwg, hipStream_t s) {
  const bool oh = a.nd.obs_kind != PH_SPACE_BOX;
  const bool h16 = a.nd.head16 && !(a.nd.A == 1 && a.nd.L <= 8);   // a small Discrete head keeps its one-lane-per-row phase
  if (oh) return h16 ? launch_grad_variant<LP, VALU, true, true>(a, nwg, s) : launch_grad_variant<LP, VALU, true, false>(a, nwg, s);
  return h16 ? launch_grad_variant<LP, VALU, false, true>(a, nwg, s) : launch_grad_variant<LP, VALU, false, false>(a, nwg, s);
}

// The single-chunk / small-head kernel (ph_ppo_fast.hip) writes its slabs in the MFMA accumulators' register order; the
// general kernel accumulates in canonical parameter order.
bool grad_uses_reg_slabs(const NetDims& nd) { return grad_fast_eligible(nd); }

void grad_plan(const NetDims& nd, int nb, int num_cu, int* ntiles, int* nwg) {
  // 64-row tiles; 2 nets x nwg workgroups, two resident per CU: nwg = #CUs covers the chip, more tiles are walked
  *ntiles = (nb + 63) / 64;
  *nwg = *ntiles < num_cu ? *ntiles : num_cu;
  // the wide split kernel holds ONE workgroup per CU (its LDS): #CUs / 2 workgroups per net are all resident at once and walk more
  // tiles each -- one prologue and one set of slabs per CU instead of two
  if (nd.split == 2 && *nwg > num_cu / 2 && num_cu >= 2) *nwg = num_cu / 2;
}

hipError_t launch_ppo_grad(const GradArgs& a, int nwg, int gemm_mode, hipStream_t s) {
  if (a.nd.split == 2) return launch_ppo_grad_split_oh(a, nwg, s);   // gemm_mode 2 on one-hot observations
  if (a.nd.split) return launch_ppo_grad_split(a, nwg, s);   // gemm_mode 2 on a spec the split kernel takes (ph_abi.hip: select_gemm)
  if (grad_fast_eligible(a.nd)) return launch_ppo_grad_fast(a, nwg, gemm_mode, s);
  if (a.nd.Lp != 32 && a.nd.Lp != 64) return hipErrorInvalidValue;
  const bool lp64 = a.nd.Lp == 64;
  if (gemm_mode == 1) return lp64 ? launch_grad_shape<64, true>(a, nwg, s) : launch_grad_shape<32, true>(a, nwg, s);
  return lp64 ? launch_grad_shape<64, false>(a, nwg, s) : launch_grad_shape<32, false>(a, nwg, s);
}

// ---- advantage statistics of every minibatch of a train() call: mean and unbiased std (torch .mean()/.std()) ----
// a 16-byte row record is written once and read once, launches later, by ONE gradient workgroup: a streaming store (the 42 MB of a
// call's records otherwise displace the per-row table the kernel gathers from out of the L2)
#ifndef PH_ADV_STORE
#define PH_ADV_STORE 1
#endif
__device__ __forceinline__ void st_rec(uint4* p, const uint4& v) {
#if PH_ADV_STORE == 1
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store((u32x4_t){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4_t*>(p));
#else
  *p = v;
#endif
}
__global__ __launch_bounds__(ADV_THREADS) void adv_stats_kernel(AdvStatArgs a) {
  __shared__ double sh[2][ADV_THREADS / 64];
  const int mb = blockIdx.x / ADV_SPLIT, seg = blockIdx.x - mb * ADV_SPLIT;   // ADV_SPLIT workgroups share a minibatch
  const int ep = mb / a.n_mb, k = mb - ep * a.n_mb;
  const int start = k * a.batch;
  const int nb = (a.N - start < a.batch) ? a.N - start : a.batch;
  const uint64_t key = epoch_key(a.perm_seed + (a.epoch ? *a.epoch : 0ull), ep);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (a.clear_flag && blockIdx.x == 0 && tid == 0) *a.clear_flag = 0;
  const int chunk = (nb + ADV_SPLIT - 1) / ADV_SPLIT;
  const int lo = seg * chunk, hi = (lo + chunk < nb) ? lo + chunk : nb;
  auto value = [&](int i) -> double {
    if (i >= hi) return 0.0;
    int n;
    if (a.perms) {
      n = a.perms[(size_t)ep * a.N + start + i];
    } else {
      n = (int)feistel_perm((uint32_t)(start + i), a.perm_n, a.perm_hb, key);
      if (a.idx_out) a.idx_out[(size_t)ep * a.N + start + i] = n;   // each element is visited exactly once
    }
    const int phys = env_major_to_phys(n, a.T, a.E);
    if (a.phys_out) a.phys_out[(size_t)ep * a.N + start + i] = phys;
    float adv;
    if (a.rec_pi_out && a.rowrec) {   // the row's scalars from the packed table obs_planes_kernel wrote (same bits as the arrays)
      const uint4 rp = a.rowrec[2 * (size_t)phys], rv = a.rowrec[2 * (size_t)phys + 1];
      const size_t o = (size_t)ep * a.N + start + i;
      adv = __uint_as_float(rp.x);
      st_rec(a.rec_pi_out + o, make_uint4((unsigned)phys, rp.x, rp.y, rp.z));
      st_rec(a.rec_vf_out + o, make_uint4((unsigned)phys, rv.x, rv.y, 0u));
    } else {
      adv = a.rb_adv[phys];
      if (a.rec_pi_out) {   // the five per-row scalars of the gradient launches, gathered once per train() instead of once per launch
        const size_t o = (size_t)ep * a.N + start + i;
        st_rec(a.rec_pi_out + o, make_uint4((unsigned)phys, __float_as_uint(adv), __float_as_uint(a.rb_logp[phys]), __float_as_uint(a.rb_act[phys])));
        st_rec(a.rec_vf_out + o, make_uint4((unsigned)phys, __float_as_uint(a.rb_ret[phys]), __float_as_uint(a.rb_val[phys]), 0u));
      }
    }
    return (double)adv;
  };
  // one pass: sum and sum of squares in fp64 (exact products of f32 values), 4 independent gathers per round
  double s = 0.0, q = 0.0;
  for (int i = lo + tid; i < hi; i += 4 * ADV_THREADS) {
    const double v0 = value(i), v1 = value(i + ADV_THREADS), v2 = value(i + 2 * ADV_THREADS), v3 = value(i + 3 * ADV_THREADS);
    s += (v0 + v1) + (v2 + v3);
    q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off, 64);
    q += __shfl_down(q, off, 64);
  }
  if (lane == 0) {
    sh[0][wave] = s;
    sh[1][wave] = q;
  }
  __syncthreads();
  if (tid == 0) {
    double ts = 0.0, tq = 0.0;
    for (int w = 0; w < ADV_THREADS / 64; ++w) {
      ts += sh[0][w];
      tq += sh[1][w];
    }
    a.partial[((size_t)mb * ADV_SPLIT + seg) * 2 + 0] = ts;
    a.partial[((size_t)mb * ADV_SPLIT + seg) * 2 + 1] = tq;
  }
}
// mean and unbiased std of every minibatch from the segment partials, folded in a fixed order
__global__ void adv_finalize_kernel(AdvStatArgs a, int n_total) {
  const int mb = blockIdx.x * blockDim.x + threadIdx.x;
  if (mb >= n_total) return;
  const int k = mb % a.n_mb;
  const int start = k * a.batch;
  const int nb = (a.N - start < a.batch) ? a.N - start : a.batch;
  double ts = 0.0, tq = 0.0;
  for (int seg = 0; seg < ADV_SPLIT; ++seg) {
    ts += a.partial[((size_t)mb * ADV_SPLIT + seg) * 2 + 0];
    tq += a.partial[((size_t)mb * ADV_SPLIT + seg) * 2 + 1];
  }
  const double mean = ts / (double)nb;
  double var = (nb > 1) ? (tq - (double)nb * mean * mean) / (double)(nb - 1) : 0.0;
  if (var < 0.0) var = 0.0;
  a.out[2 * mb + 0] = (float)mean;
  a.out[2 * mb + 1] = (float)sqrt(var);
}
hipError_t launch_adv_stats(const AdvStatArgs& a, int n_total, hipStream_t s) {
  hipLaunchKernelGGL(adv_stats_kernel, dim3(n_total * ADV_SPLIT), dim3(ADV_THREADS), 0, s, a);
  hipLaunchKernelGGL(adv_finalize_kernel, dim3((n_total + 63) / 64), dim3(64), 0, s, a, n_total);
  return hipGetLastError();
}

// ---- observation rows -> bf16 planes (ph_split.h), once per train() ----------------------------------------------------------
// A workgroup takes 32 rows: their 32 D floats are contiguous in the buffer and come in as one coalesced stream into LDS
// ([row][65]: odd stride, the 8-float reads below are conflict-free), then one lane per (row, 8-feature granule) splits eight
// features and stores three 16-byte plane granules (8 lanes = one 128-byte plane row).  Reads 4 D and writes 384 bytes per row:
// HBM-bound.
constexpr int OP_ROWS = 32;
struct RowScalars {   // the rollout buffer's per-row scalars (action length 1), or all null
  const float *adv, *logp, *act, *ret, *val;
  uint4* rowrec;      // [n][2]: {advantage, old log-prob, action, 0} {return, old value, 0, 0} by physical row
};
__global__ __launch_bounds__(256) void obs_planes_kernel(const float* __restrict__ obs, int n, int D, int F, int fold,
                                                         uint4* __restrict__ image, RowScalars rs) {
  __shared__ float xs[OP_ROWS][65];
  const int tid = threadIdx.x;
  const size_t row0 = (size_t)blockIdx.x * OP_ROWS;
  const int nrow = (row0 + OP_ROWS <= (size_t)n) ? OP_ROWS : (int)((size_t)n > row0 ? (size_t)n - row0 : 0);
  const float* src = obs + row0 * D;
  for (int e = tid; e < nrow * D; e += 256) {
    const int r = e / D, f = e - r * D;
    if (f < F) xs[r][f] = __builtin_nontemporal_load(src + e);
  }
  __syncthreads();
  const int r = tid >> 3, g = tid & 7;
  const size_t row = row0 + r;
  if (row > (size_t)n) return;
  // the row's five scalars side by side (32 bytes): adv_stats_kernel then builds a minibatch-order record from ONE random 32-byte
  // read of this L2-sized table instead of five random 4-byte reads that each pull a line
  if (rs.rowrec && row < (size_t)n && g < 2) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (g == 0) v = make_uint4(__float_as_uint(rs.adv[row]), __float_as_uint(rs.logp[row]), __float_as_uint(rs.act[row]), 0u);
    else v = make_uint4(__float_as_uint(rs.ret[row]), __float_as_uint(rs.val[row]), 0u, 0u);
    rs.rowrec[2 * row + g] = v;
  }
  typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
  bf8 pl[3];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int f = 8 * g + e;
    float x = 0.f;
    if (row < (size_t)n) x = f < F ? xs[r][f] : ((fold && f == HID - 1) ? 1.f : 0.f);
    __bf16 h, m, l;
    split1(x, h, m, l);
    pl[0][e] = h;
    pl[1][e] = m;
    pl[2][e] = l;
  }
#pragma unroll
  for (int p = 0; p < 3; ++p) {
#if defined(PH_OBS_STORE_NT)
    st_rec(image + row * XIMG_ROW_U4 + p * 8 + g, __builtin_bit_cast(uint4, pl[p]));
#else
    image[row * XIMG_ROW_U4 + p * 8 + g] = __builtin_bit_cast(uint4, pl[p]);
#endif
  }
}
hipError_t launch_obs_planes(const float* obs, int n, int D, int F, int fold, uint4* image, const float* adv, const float* logp,
                             const float* act, const float* ret, const float* val, uint4* rowrec, hipStream_t s) {
  const size_t blocks = ((size_t)n + 1 + OP_ROWS - 1) / OP_ROWS;   // rows 0 .. n (row n = the zero row)
  RowScalars rs{adv, logp, act, ret, val, (adv && logp && act && ret && val) ? rowrec : nullptr};
  hipLaunchKernelGGL(obs_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, obs, n, D, F, fold, image, rs);
  return hipGetLastError();
}

// (slab reduction, clip and Adam device code: ph_step.h)
template <int VEC>
__global__ __launch_bounds__(RED_PARAMS * 4) void ppo_reduce_kernel(ReduceArgs a) {
  // ONE kilobyte of LDS, not two: a slab block uses gsum, the statistics block part / means -- never both.  Beside another learner's
  // gradient launch that is the difference between fitting into what two resident gradient workgroups leave of a CU's LDS (2 x 79 680
  // of ~160.7 KB usable: ~1.4 KB) and waiting for one of them to leave -- or, placed first, keeping the second one out for the
  // reduction's 5.7 us (PH_REDUCE_LDS_UNION=0: the two arrays side by side, 2 112 bytes; same-box A/B profiles/r06_bn_*)
#ifndef PH_REDUCE_LDS_UNION
#define PH_REDUCE_LDS_UNION 1
#endif
#if PH_REDUCE_LDS_UNION
  __shared__ float lds_union[4 * RED_PARAMS];
  static_assert(4 * RED_PARAMS >= 32 * NSTATP + NSTATP || 4 * RED_PARAMS >= 32 * NSTATP, "statistics scratch within the slab scratch");
  float (*gsum)[RED_PARAMS] = reinterpret_cast<float (*)[RED_PARAMS]>(lds_union);
  float (*part)[NSTATP] = reinterpret_cast<float (*)[NSTATP]>(lds_union);
  __shared__ float means[NSTATP];
#else
  __shared__ float gsum[4][RED_PARAMS];
  __shared__ float part[32][NSTATP];
  __shared__ float means[NSTATP];
#endif
  const int tid = threadIdx.x;
#ifdef PH_REDUCE_START_DELAY_TICKS   // experiment (scripts/build_variants.sh; 100 MHz ticks): the reduction's loads held back so that they do not
  {                                  // meet the OTHER learner's gradient prologue, which starts when this learner's gradient launch ends
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)(PH_REDUCE_START_DELAY_TICKS)) __builtin_amdgcn_s_sleep(8);
  }
#endif
  if (*a.stop_flag != 0) {  // a previous minibatch of this train() call hit the KL early stop
    if (blockIdx.x == 0 && tid < PH_NSTAT && a.stats_out) a.stats_out[tid] = 0.f;
    if (blockIdx.x == 0 && tid == 0) {
      a.scalars[1] = 0.f;
      a.scalars[2] = 0.f;
    }
    return;
  }
  if (blockIdx.x == gridDim.x - 1) {   // the extra block: minibatch statistics and the KL decision, beside the slab blocks
    (void)reduce_statistics(a, part, means, true);
    return;
  }
  const int dst = reduce_dst(a, blockIdx.x);
  const float g = reduce_positions<VEC>(a, gsum, dst, blockIdx.x);
  if (tid < 64) {  // wave 0: store, square, wave-reduce
    if (dst >= 0) a.grad[dst] = g;
    float q = g * g;
    for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
    if (tid == 0) a.blocksq[blockIdx.x] = q;
  }
}

// ---- reduce + clip + Adam as ONE launch (ph_step.h: step_body) for a learner that has the device to itself ----
template <int VEC>
__global__ __launch_bounds__(RED_PARAMS * 4) void ppo_step_kernel(StepArgs s) {
  __shared__ float gsum[4][RED_PARAMS];
  __shared__ float part[32][NSTATP];
  __shared__ float means[NSTATP];
  const int nblk = gridDim.x - 1;   // slab blocks; block nblk does the statistics
  // the three device words every block starts from, as one batch of scalar loads.  gen / step are read before anything of this
  // launch is published: block 0 advances both only after every block has published
  const int stopped = *s.r.stop_flag;
  const unsigned tag = *s.gen + 1u;
  const int step_new = *s.ad.step + 1;
  const unsigned err = *s.sweep_error;
#if !PH_STEP_LATE_CHECKS
  if (step_refused(s.r, blockIdx.x, stopped, err)) return;
#endif
  // (the two reasons to do nothing -- an expired wait of this context, a KL early stop earlier in this train() call -- are tested
  // inside, behind the slab walk: ph_step.h)
  step_body<VEC>(s, blockIdx.x, nblk, tag, step_new, gsum, part, means, nullptr, stopped, err);
}

int reduce_blocks(int slab_len) { return (slab_len + RED_PARAMS - 1) / RED_PARAMS; }
// positions per lane of the reduction (see reduce_positions): the widest load the slab length allows, 16 bytes only for a learner
// that has the device to itself (beside another learner's gradient launch the 8-byte shape's register count is what fits)
static int reduce_vec(int slab_len, int exclusive) { return (slab_len % 4 == 0 && exclusive) ? 4 : (slab_len % 2 == 0 ? 2 : 1); }
// Can the fused step kernel run this grid with every block resident (its blocks wait for each other)?  The runtime's occupancy
// answer for the kernel on the current device, one block per CU held back (the API can be one high: MI355X_MICROARCH.md).
bool step_fused_fits(int nblk, int slab_len, int num_cu) {
  if (nblk + 1 > 64 * 16) return false;
  static int per_cu[3] = {-1, -1, -1};
  const int vec = reduce_vec(slab_len, 1);
  int& v = per_cu[vec == 4 ? 2 : vec - 1];
  if (v < 0) {
    int api = 0;
    const void* fn = vec == 4 ? (const void*)ppo_step_kernel<4> : vec == 2 ? (const void*)ppo_step_kernel<2> : (const void*)ppo_step_kernel<1>;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, fn, RED_PARAMS * 4, 0);
    v = (e == hipSuccess && api > 1) ? api - 1 : 0;
  }
  return (long long)v * num_cu >= (long long)nblk + 1;   // + the statistics block
}
hipError_t launch_ppo_step(const ReduceArgs& r, const AdamArgs& ad, unsigned long long* words, unsigned int* gen,
                           unsigned int* sweep_error, unsigned long long timeout, hipStream_t st) {
  StepArgs s;
  s.r = r;
  s.ad = ad;
  s.words = words;
  s.gen = gen;
  s.timeout = timeout;
  s.sweep_error = sweep_error;
  const int nblk = reduce_blocks(r.slab_len);
  if ((unsigned long long)r.nslab * (unsigned long long)r.slab_len * sizeof(float) >= (1ull << 32)) return hipErrorInvalidValue;
  switch (reduce_vec(r.slab_len, 1)) {
    case 4: hipLaunchKernelGGL(ppo_step_kernel<4>, dim3(nblk + 1), dim3(RED_PARAMS * 4), 0, st, s); break;
    case 2: hipLaunchKernelGGL(ppo_step_kernel<2>, dim3(nblk + 1), dim3(RED_PARAMS * 4), 0, st, s); break;
    default: hipLaunchKernelGGL(ppo_step_kernel<1>, dim3(nblk + 1), dim3(RED_PARAMS * 4), 0, st, s); break;
  }
  return hipGetLastError();
}
hipError_t launch_ppo_reduce(const ReduceArgs& a, hipStream_t s) {
  const dim3 grid(reduce_blocks(a.slab_len) + 1), block(RED_PARAMS * 4);   // + the statistics block
  // 32-bit buffer offsets (the largest slab area of any shape in use: 512 x 177 KB = 90 MB)
  if ((unsigned long long)a.nslab * (unsigned long long)a.slab_len * sizeof(float) >= (1ull << 32)) return hipErrorInvalidValue;
  switch (reduce_vec(a.slab_len, a.wide)) {
    case 4: hipLaunchKernelGGL(ppo_reduce_kernel<4>, grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL(ppo_reduce_kernel<2>, grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL(ppo_reduce_kernel<1>, grid, block, 0, s, a); break;
  }
  return hipGetLastError();
}

// ---- clip_grad_norm_ + Adam (torch.optim.Adam single-tensor maths, eps = 1e-5) ----------------------------------------
__global__ __launch_bounds__(256) void ppo_adam_kernel(AdamArgs a) {
  __shared__ float sh[4];
  __shared__ AdamScalars ks;
  const int tid = threadIdx.x;
  // everything this thread's entry needs goes out first: the gradient, the moments, the parameter and its image positions do not
  // depend on the norm, so they travel under the sum of squares, the two pow() of the bias corrections and the barriers
  const int p = blockIdx.x * blockDim.x + tid;
  const bool live = p < a.P;
  float gr = 0.f, m0 = 0.f, v0 = 0.f, p0 = 0.f;
  int i0 = -1, i1 = -1;
  if (live) {
    gr = a.grad[p];
    m0 = a.m[p];
    v0 = a.v[p];
    p0 = a.params[p];
    if (a.wimage) {
      i0 = a.wimage_map[2 * p];
      i1 = a.wimage_map[2 * p + 1];
    }
  }
  if (a.scalars[1] == 0.f) {  // KL early stop (or already stopped): no optimizer step
    if (blockIdx.x == 0 && tid == 0 && a.scalars[2] != 0.f) *a.stop_flag = 1;
    return;
  }
  const int step = (tid == 0) ? *a.step : 0;
  float q = 0.f;
  {   // this thread's entries of the per-block squares (k = tid, tid + 256, ...), loads batched, adds in index order
    float x[4];
    for (int k0 = tid; k0 < a.nblk; k0 += 4 * 256) {
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = (k0 + 256 * u < a.nblk) ? a.blocksq[k0 + 256 * u] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k0 + 256 * u < a.nblk) q += x[u];
    }
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off, 64);
  if ((tid & 63) == 0) sh[tid >> 6] = q;
  __syncthreads();
  if (tid == 0) {
    const float total = sqrtf((sh[0] + sh[1]) + (sh[2] + sh[3]));
    ks = adam_scalars(total, a.max_norm, step, a.lr, a.beta1, a.beta2);
    if (blockIdx.x == 0 && a.stats_out) a.stats_out[6] = total;
  }
  __syncthreads();
  if (!live) return;
  const AdamScalars k = ks;
  float m, v;
  const float pn = adam_apply(gr, k, a.beta1, a.beta2, a.eps, m0, v0, p0, &m, &v);
  a.m[p] = m;
  a.v[p] = v;
  a.params[p] = pn;
  if (a.wimage) wimage_put_at(a.wimage, i0, i1, pn);                    // the split gradient kernel's pre-split weight fragments
}

__global__ __launch_bounds__(256) void weight_image_kernel(const float* params, unsigned short* image, const int* map, int P) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < P) wimage_put(image, map, p, params[p]);
}
// debug: number of image elements (all three planes) that differ from what `params` split to
__global__ __launch_bounds__(256) void weight_image_check_kernel(const float* params, const unsigned short* image, const int* map, int P,
                                                                 int* mismatches) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  __bf16 h, m, l;
  split1(params[p], h, m, l);
  const unsigned short want[3] = {__builtin_bit_cast(unsigned short, h), __builtin_bit_cast(unsigned short, m),
                                  __builtin_bit_cast(unsigned short, l)};
  int bad = 0;
  for (int k = 0; k < 2; ++k) {
    const int pos = map[2 * p + k];
    if (pos < 0) continue;
    for (int q = 0; q < 3; ++q) bad += image[pos + q * WIMG_PLANE] != want[q];
  }
  if (bad) atomicAdd(mismatches, bad);
}
hipError_t launch_weight_image_check(const float* params, const unsigned short* image, const int* map, int P, int* mismatches,
                                     hipStream_t s) {
  hipLaunchKernelGGL(weight_image_check_kernel, dim3((P + 255) / 256), dim3(256), 0, s, params, image, map, P, mismatches);
  return hipGetLastError();
}
hipError_t launch_weight_image(const float* params, unsigned short* image, const int* map, int P, hipStream_t s) {
  hipLaunchKernelGGL(weight_image_kernel, dim3((P + 255) / 256), dim3(256), 0, s, params, image, map, P);
  return hipGetLastError();
}
hipError_t launch_ppo_adam(const AdamArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(ppo_adam_kernel, dim3((a.P + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

__global__ void epoch_advance_kernel(unsigned long long* p) { *p += 1ull; }
hipError_t launch_epoch_advance(unsigned long long* p, hipStream_t s) {
  hipLaunchKernelGGL(epoch_advance_kernel, dim3(1), dim3(1), 0, s, p);
  return hipGetLastError();
}
__global__ void set_int_kernel(int* p, int v) { *p = v; }
hipError_t launch_set_int(int* p, int v, hipStream_t s) {
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, s, p, v);
  return hipGetLastError();
}

}  // namespace ph

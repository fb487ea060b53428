// ADAP's context term as a fused-update variant: the reference's ADAP learner
// (pantheonrl/algos/adap/adap_learn.py:229-371) is PPO.train() whose per-minibatch loss gains
//     context_loss_coeff * get_context_kl_loss(...)                                        (adap_learn.py:313-320)
// where (adap/util.py:97-131) up to num_state_samples states of the minibatch (th.randperm) are re-evaluated under
// num_context_samples freshly sampled contexts -- the observation rows carry the rollout's context in their last
// context_size components and AdapPolicy feeds features ++ context to the ordinary MlpExtractor (adap/policies.py:104-119),
// so the network is the MlpPolicy of width D = obs + context -- and the loss is the mean over context pairs (a before b)
// of mean_s exp(-KL(pi(.|s,a) || pi(.|s,b))).
//
// Here that term is ONE small launch per minibatch next to the PPO gradient launch: workgroup w takes ADAP_ROWS / C of the
// sampled states, stages the policy network's weights in LDS in one batch of loads (they are L2-resident: the gradient launch
// reads them too) while one wave walks sample -> minibatch order -> buffer row, and treats the states' C context rows as ONE
// 16-row M tile of v_mfma_f32_16x16x4_f32 for the forward products, the two backward products and the weight gradients
// (K = the 16 rows).  Between them thread (row, logit) forms the pairwise KL terms of its state on 16-lane DPP rows.  The
// gradients are written, already scaled by coeff / (pairs * states), to the workgroup's own slab of the policy-side
// parameters; ppo_reduce_kernel adds the slabs to the PPO gradient in a fixed order before the norm, so the clip and the Adam
// step see the gradient of the whole loss exactly as the reference's single backward() does.
#include "ph_launch.h"

namespace ph {

constexpr int ALD = HID + 1;   // padded leading dimension of the 64-wide activation tiles
constexpr int W2LD = HID + 4;  // leading dimension of W2 in LDS: rows stay 16-byte aligned, row-strided float4 reads spread over banks

// floats of one workgroup's gradient slab: the policy network's parameters [0, vf_W1) and the action head [act_W, val_W)
__host__ __device__ inline int adap_slab_len(const ph_layout& lay) { return lay.vf_W1 + (lay.val_W - lay.act_W); }
int adap_slab_floats(const ph_layout& lay) { return adap_slab_len(lay); }

int adap_workgroups(int n_ctx, int n_states) {
  const int spw = ADAP_ROWS / n_ctx;
  return (n_states + spw - 1) / spw;
}

// LDS floats of one workgroup
static size_t adap_lds_floats(const NetDims& nd, int n_ctx, int ctx_size) {
  const int spw = ADAP_ROWS / n_ctx, npairs = n_ctx * (n_ctx - 1) / 2;
  return (size_t)nd.F * HID + (size_t)HID * W2LD + (size_t)((HID * nd.L + 3) & ~3)   // W1, W2 (padded rows), act_W
         + (size_t)((2 * HID + nd.L + 3) & ~3)                                        // b1, b2, act_b
         + (size_t)ADAP_ROWS * (nd.F + 1)      // xs
         + (size_t)4 * ADAP_ROWS * ALD         // h1s h2s dz1s dz2s
         + (size_t)ADAP_ROWS * (nd.L + 1) + (size_t)2 * ADAP_ROWS * (nd.L + 1 > 16 ? nd.L + 1 : 16)  // zs; lps, pbs (>= 16 per row)
         + (size_t)n_ctx * ctx_size            // contexts
         + (size_t)(spw * npairs > 16 ? spw * npairs : 16) + (size_t)spw * npairs * nd.A   // per-pair exp(-KL) (>= 16 slots), per-(pair, component) KL
         + 32                                  // rowphys
         + (size_t)((nd.A + 4) & ~3);          // prefix sums of the action components
}
size_t adap_lds_bytes(const NetDims& nd, int n_ctx, int ctx_size) { return adap_lds_floats(nd, n_ctx, ctx_size) * sizeof(float); }

// all-reduce over the 16 lanes of a DPP row (row_ror 1, 2, 4, 8): every lane ends with the same value
template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_row<0x121>(v);
  v += dpp_row<0x122>(v);
  v += dpp_row<0x124>(v);
  return v + dpp_row<0x128>(v);
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_row<0x121>(v));
  v = fmaxf(v, dpp_row<0x122>(v));
  v = fmaxf(v, dpp_row<0x124>(v));
  return fmaxf(v, dpp_row<0x128>(v));
}

// pair index -> (i, j), i < j, in itertools.combinations order
__device__ __forceinline__ void pair_of(int pr, int C, int& i, int& j) {
  i = 0;
  int left = pr;
  while (left >= C - 1 - i) {
    left -= C - 1 - i;
    ++i;
  }
  j = i + 1 + left;
}
__device__ __forceinline__ int pair_index(int i, int j, int C) { return i * (2 * C - i - 1) / 2 + (j - i - 1); }

__global__ __launch_bounds__(256) void adap_context_kernel(AdapArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if (*a.stop_flag != 0) return;   // KL early stop already raised: the reduce launch ignores everything
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  const int F = nd.F, L = nd.L, FP = F + 1, LP = L + 1, A = nd.A;
  const int C = a.n_ctx, cs = a.ctx_size, spw = ADAP_ROWS / C, npairs = C * (C - 1) / 2;
  const int tid = threadIdx.x;
  const int s0 = blockIdx.x * spw;
  const int ns = (a.n_states - s0 < spw) ? a.n_states - s0 : spw;   // states of this workgroup
  const int R = ns * C;                                              // live rows; row r = (state sl, context i) = sl * C + i

  // the policy network's weights, staged once: every later phase reads LDS (a dependent chain of ~10 short phases would
  // otherwise pay an L2 round trip per unrolled batch of loads in each of them)
  float* w1s = smem;                      // [F][HID]
  float* w2s = w1s + F * HID;             // [HID][W2LD]
  float* aws = w2s + HID * W2LD;          // [HID][L]
  float* bs = aws + ((HID * L + 3) & ~3); // b1 [HID] | b2 [HID] | act_b [L]
  float* xs = bs + ((2 * HID + L + 3) & ~3);   // [ADAP_ROWS][FP]
  float* h1s = xs + ADAP_ROWS * FP;       // [ADAP_ROWS][ALD]
  float* h2s = h1s + ADAP_ROWS * ALD;
  float* dz1s = h2s + ADAP_ROWS * ALD;
  float* dz2s = dz1s + ADAP_ROWS * ALD;
  float* zs = dz2s + ADAP_ROWS * ALD;     // [ADAP_ROWS][LP] logits -> dL/dlogits
  float* lps = zs + ADAP_ROWS * LP;       // log-probabilities (per action component)
  float* pbs = lps + ADAP_ROWS * (LP > 16 ? LP : 16);   // probabilities (rows of >= 16 slots: the small-head path pads)
  float* cxs = pbs + ADAP_ROWS * (LP > 16 ? LP : 16);   // [C][cs] sampled contexts
  float* tvs = cxs + C * cs;              // [spw][npairs] exp(-KL)
  float* kls = tvs + (spw * npairs > 16 ? spw * npairs : 16);   // [spw][npairs][A] KL of every action component
  int* rowphys = (int*)(kls + spw * npairs * A);   // [spw] buffer row of every sampled state
  int* aoff = rowphys + 32;                        // [A + 1] first logit of every action component

  PH_STAMP(a.prof, 0);
  // Two dependent chains start the kernel: weights -> LDS (one round trip) and sample -> minibatch order -> buffer row (two).
  // Waves 0-2 take the first, wave 3 the second; the observation gather by all four waves follows.
  if (tid < 192) {
    const float4* g1 = reinterpret_cast<const float4*>(a.params + lay.pi_W1);
    const float4* g2 = reinterpret_cast<const float4*>(a.params + lay.pi_W2);
    for (int e0 = 0; e0 < HID * HID / 4; e0 += 6 * 192) {                    // W2: 1024 float4, rows padded to W2LD
      float4 w2r[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int e = e0 + tid + 192 * i;
        w2r[i] = g2[e < HID * HID / 4 ? e : 0];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int e = e0 + tid + 192 * i;
        if (e < HID * HID / 4) *reinterpret_cast<float4*>(w2s + (e >> 4) * W2LD + 4 * (e & 15)) = w2r[i];
      }
    }
#pragma unroll 4
    for (int e = tid; e < F * (HID / 4); e += 192) reinterpret_cast<float4*>(w1s)[e] = g1[e];
    for (int e = tid; e < HID * L; e += 192) aws[e] = a.params[lay.act_W + e];
    if (tid < HID) bs[tid] = a.params[lay.pi_b1 + tid];
    else if (tid < 2 * HID) bs[tid] = a.params[lay.pi_b2 + tid - HID];
    for (int e = tid; e < L; e += 192) bs[2 * HID + e] = a.params[lay.act_b + e];
    for (int e = tid; e <= A; e += 192) aoff[e] = nd.act_off[e];
  } else {
    const int ln = tid - 192;
    // ---- the samples: teacher-forced, or drawn here from the keyed streams ----
    const uint64_t key = epoch_key((a.seed ^ 0xADA9C0DEull) + (a.epoch ? *a.epoch : 0ull), (int)a.mbi);
    if (ln < C) {
      float* c = cxs + ln * cs;
      if (a.contexts) {
        for (int k = 0; k < cs; ++k) c[k] = a.contexts[ln * cs + k];
      } else if (a.sampler == PH_CTX_NATURAL_NUMBERS) {    // util.py:80-89: (num, 1) integers in [0, ctx_size); ctx_size == 1 here
        int v = (int)(philox_uniform(key, 1ull, (uint32_t)ln, 0u) * (float)cs);
        v = v >= cs ? cs - 1 : v;
        for (int k = 0; k < cs; ++k) c[k] = k == 0 ? (float)v : 0.f;
      } else if (a.sampler == PH_CTX_CATEGORICAL) {        // util.py:70-77
        int hot = (int)(philox_uniform(key, 1ull, (uint32_t)ln, 0u) * (float)cs);
        hot = hot >= cs ? cs - 1 : hot;
        for (int k = 0; k < cs; ++k) c[k] = k == hot ? 1.f : 0.f;
      } else {
        float ss = 0.f;
        for (int k = 0; k < cs; ++k) {
          const float u = philox_uniform(key, 1ull, (uint32_t)ln, (uint32_t)k);
          const float v = a.sampler == PH_CTX_POSITIVE_SQUARE ? u : u * 2.f - 1.f;   // util.py:54-67
          c[k] = v;
          ss += v * v;
        }
        if (a.sampler == PH_CTX_L2) {                       // util.py:42-51: scaled onto the unit sphere
          const float nrm = sqrtf(ss);
          for (int k = 0; k < cs; ++k) c[k] = c[k] / nrm;
        }
      }
      if (blockIdx.x == 0 && a.used_contexts)
        for (int k = 0; k < cs; ++k) a.used_contexts[ln * cs + k] = c[k];
    }
    if (ln >= 16 && ln < 16 + spw) {
      const int sl = ln - 16;
      int row = 0;
      if (sl < ns) {
        // th.randperm(B)[:num_state_samples] (util.py:106): explicit positions, or the head of a keyed permutation of [0, nb)
        const int q = a.state_idx ? a.state_idx[s0 + sl] : (int)feistel_perm((uint32_t)(s0 + sl), (uint32_t)a.nb, a.nb_hb, key);
        if (a.used_state_idx) a.used_state_idx[s0 + sl] = q;
        row = env_major_to_phys(a.idx[q], a.T, a.E);
      }
      rowphys[sl] = row;
    }
  }
  __syncthreads();
  PH_STAMP(a.prof, 2);
  // ---- X: the state's own components, then context i (policies.py:111-117) ----
  for (int e = tid; e < ADAP_ROWS * F; e += 256) {
    const int r = e / F, f = e - r * F;
    float v = 0.f;
    if (r < R) {
      const int sl = r / C, i = r - sl * C;
      v = f < F - cs ? a.rb_obs[(size_t)rowphys[sl] * nd.D + f] : cxs[i * cs + (f - (F - cs))];
    }
    xs[r * FP + f] = v;
  }
  __syncthreads();

  // The 16 (state, context) rows are one M tile of v_mfma_f32_16x16x4_f32: lane (c = lane & 15, g = lane >> 4) supplies
  // A[c][k0 + g] and B[k0 + g][col0 + c] and receives D[4g + r][col0 + c].  Wave w owns the 16 output columns 16w .. 16w + 15
  // of every 64-wide product (two accumulation chains in flight), so a layer is 16 MFMAs per wave instead of ~70 VALU FMAs
  // behind as many LDS reads per thread.
  const int wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4, col = 16 * wave + c;
  const float* AW = aws;
  PH_STAMP(a.prof, 3);
  // ---- H1 = tanh(X W1 + b1) ----
  {
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < F; k0 += 64) {   // 16 k-steps per batch: all 32 operand reads in flight, then the 16 products
      float av[16], bv[16];
#pragma unroll
      for (int s4 = 0; s4 < 16; ++s4) {    // clamped index + select instead of a branch around the read
        const int k = k0 + 4 * s4 + g, kc = k < F ? k : F - 1;
        const float x = xs[c * FP + kc], w = w1s[kc * HID + col];
        av[s4] = k < F ? x : 0.f;
        bv[s4] = w;
      }
#pragma unroll
      for (int s4 = 0; s4 < 16; s4 += 2) {
        e = mma16<false>(av[s4], bv[s4], e, lane);
        o = mma16<false>(av[s4 + 1], bv[s4 + 1], o, lane);
      }
    }
    const float bb = bs[col];
#pragma unroll
    for (int r = 0; r < 4; ++r) h1s[(4 * g + r) * ALD + col] = fast_tanh(e[r] + o[r] + bb);
  }
  __syncthreads();
  // ---- H2 = tanh(H1 W2 + b2) ----
  {
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    float av[16], bv[16];
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      av[s4] = h1s[c * ALD + 4 * s4 + g];
      bv[s4] = w2s[(4 * s4 + g) * W2LD + col];
    }
#pragma unroll
    for (int s4 = 0; s4 < 16; s4 += 2) {
      e = mma16<false>(av[s4], bv[s4], e, lane);
      o = mma16<false>(av[s4 + 1], bv[s4 + 1], o, lane);
    }
    const float bb = bs[HID + col];
#pragma unroll
    for (int r = 0; r < 4; ++r) h2s[(4 * g + r) * ALD + col] = fast_tanh(e[r] + o[r] + bb);
  }
  __syncthreads();
  PH_STAMP(a.prof, 4);
  // ---- logits = H2 act_W + act_b: 16 logits per wave ----
  for (int n0 = 16 * wave; n0 < L; n0 += 64) {
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    const bool cok = n0 + c < L;
    const int nc = cok ? n0 + c : L - 1;
    float av[16], bv[16];
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      av[s4] = h2s[c * ALD + 4 * s4 + g];
      bv[s4] = AW[(4 * s4 + g) * L + nc];   // columns >= L compute a duplicate of column L - 1 that is never stored
    }
#pragma unroll
    for (int s4 = 0; s4 < 16; s4 += 2) {
      e = mma16<false>(av[s4], bv[s4], e, lane);
      o = mma16<false>(av[s4 + 1], bv[s4 + 1], o, lane);
    }
    if (cok) {
      const float bb = bs[2 * HID + n0 + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) zs[(4 * g + r) * LP + n0 + c] = e[r] + o[r] + bb;
    }
  }
  __syncthreads();
  PH_STAMP(a.prof, 5);
  const float wgt = a.coef / (float)(npairs * a.n_states);
  if (A == 1 && L <= 16) {
    // One Discrete head of at most 16 logits (every BASELINE action space but Liar's Dice): thread (row = tid / 16, slot
    // = tid % 16) owns one logit of one (state, context) row, so a row is a 16-lane DPP row and its max / sum / KL are four
    // row_ror steps.  Slots >= L hold lp = 0, p = 0 and nothing below is predicated.  After ONE exchange through LDS every
    // thread walks the other contexts j of its state:
    //   i < j (pair (i, j), KL(i || j)):  dz -= w T p_i ((lp_i - lp_j) - KL)      i > j (pair (j, i), KL(j || i)):  dz -= w T (p_i - p_j)
    // with T = exp(-KL): two barriers and all 256 lanes busy instead of five phases with a handful of active lanes each.
    const int r = tid >> 4, q = tid & 15;
    const bool rowon = r < R;
    const float z = q < L ? zs[r * LP + (q < L ? q : 0)] : -3.0e38f;
    const float mx = row16_max(z);
    const float se = row16_sum(__expf(z - mx));      // slots >= L add exp(-huge) = 0
    const float lse = mx + __logf(se);
    const float lp = q < L ? z - lse : 0.f, pq = q < L ? __expf(z - lse) : 0.f;
    lps[r * 16 + q] = lp;
    pbs[r * 16 + q] = pq;
    __syncthreads();
    const int sl = r / C, ci = r - sl * C;
    float dz = 0.f, tsum = 0.f;
    for (int j = 0; j < C; ++j) {
      const int ro = ((rowon ? sl * C + j : r) << 4) + q;
      const float lpj = lps[ro], pj = pbs[ro];
      const bool first = ci < j;   // this row is the pair's first distribution
      const float d = lp - lpj;
      const float kl = row16_sum(first ? pq * d : -pj * d);
      const float tv = j == ci ? 0.f : __expf(-kl);
      tsum += first ? tv : 0.f;
      dz -= tv * (first ? pq * (d - kl) : pq - pj);
    }
    if (q < L) zs[r * LP + q] = rowon ? dz * wgt : 0.f;
    if (q == 0) tvs[r] = rowon ? tsum : 0.f;     // (tvs holds spw * npairs >= 1 floats per state: 16 fit for every C >= 2)
    __syncthreads();
    if (tid == 0) {   // this workgroup's share of sum_s sum_pairs exp(-KL), in a fixed order
      float v = 0.f;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) v += tvs[rr];
      a.loss_part[blockIdx.x] = v;
    }
  } else {
    // ---- log-softmax of every action component, one thread per (row, component) ----
    for (int e = tid; e < ADAP_ROWS * A; e += 256) {
      const int rr = e / A, comp = e - rr * A;
      const int lo = aoff[comp], n = aoff[comp + 1] - lo;
      const float* z = zs + rr * LP + lo;
      float mx = -3.0e38f;
      for (int c = 0; c < n; ++c) mx = fmaxf(mx, z[c]);
      float se = 0.f;
      for (int c = 0; c < n; ++c) se += __expf(z[c] - mx);
      const float lse = mx + __logf(se);
      for (int c = 0; c < n; ++c) {
        const float lq = z[c] - lse;
        lps[rr * LP + lo + c] = lq;
        pbs[rr * LP + lo + c] = __expf(lq);
      }
    }
    __syncthreads();
    // ---- KL(pi_i || pi_j) of every (state, pair i < j, component) [torch kl_divergence(Categorical, Categorical)] ----
    for (int e = tid; e < ns * npairs * A; e += 256) {
      const int comp = e % A, t = e / A, sl = t / npairs, pr = t - sl * npairs;
      int i, j;
      pair_of(pr, C, i, j);
      const int lo = aoff[comp], n = aoff[comp + 1] - lo;
      const float* li = lps + (sl * C + i) * LP + lo;
      const float* lj = lps + (sl * C + j) * LP + lo;
      const float* pi = pbs + (sl * C + i) * LP + lo;
      float kl = 0.f;
      for (int c = 0; c < n; ++c) kl = __builtin_fmaf(pi[c], li[c] - lj[c], kl);
      kls[t * A + comp] = kl;
    }
    __syncthreads();
    for (int t = tid; t < ns * npairs; t += 256) {   // util.py:128: exp(-KL), the MultiCategorical KL is the components' sum
      float kl = 0.f;
      for (int comp = 0; comp < A; ++comp) kl += kls[t * A + comp];
      tvs[t] = __expf(-kl);
    }
    __syncthreads();
    if (tid < 64) {   // this workgroup's share of sum_s sum_pairs exp(-KL): strided partial sums, then a fixed-order wave fold
      float v = 0.f;
      for (int t = tid; t < ns * npairs; t += 64) v += tvs[t];
  #pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      if (tid == 0) a.loss_part[blockIdx.x] = v;
    }
    // ---- dL/dlogits.  L = w sum_{s, i<j} T_ij(s), T = exp(-KL_ij), w = coeff / (pairs * states):
    //        d KL_ij / d z_i[c] = p_i[c] ((lp_i[c] - lp_j[c]) - KL_ij^comp),   d KL_ij / d z_j[c] = p_j[c] - p_i[c] ----
    for (int e = tid; e < ADAP_ROWS * L; e += 256) {
      const int rr = e / L, c = e - rr * L;
      float d = 0.f;
      if (rr < R) {
        const int sl = rr / C, i = rr - sl * C;
        int comp = 0;
        while (aoff[comp + 1] <= c) ++comp;
        const float lpi = lps[rr * LP + c], ppi = pbs[rr * LP + c];
        for (int j = 0; j < C; ++j) {
          if (j == i) continue;
          const int ro = (sl * C + j) * LP + c;
          if (i < j) {
            const int t = sl * npairs + pair_index(i, j, C);
            d -= tvs[t] * ppi * ((lpi - lps[ro]) - kls[t * A + comp]);
          } else {
            const int t = sl * npairs + pair_index(j, i, C);
            d -= tvs[t] * (ppi - pbs[ro]);
          }
        }
        d *= wgt;
      }
      zs[rr * LP + c] = d;
    }
  }
  __syncthreads();
  PH_STAMP(a.prof, 7);
  // ---- dZ2 = (dlogits act_W^T) * (1 - H2^2): K = L in steps of 4 ----
  {
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < L; k0 += 4) {
      const int k = k0 + g;
      const float a0 = k < L ? zs[c * LP + k] : 0.f, b0 = k < L ? AW[col * L + k] : 0.f;
      e = mma16<false>(a0, b0, e, lane);
    }
    float hv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = h2s[(4 * g + r) * ALD + col];
#pragma unroll
    for (int r = 0; r < 4; ++r) dz2s[(4 * g + r) * ALD + col] = e[r] * (1.0f - hv[r] * hv[r]);
  }
  __syncthreads();
  // ---- dZ1 = (dZ2 W2^T) * (1 - H1^2) ----
  {
    f32x4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
    float av[16], bv[16];
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
      av[s4] = dz2s[c * ALD + 4 * s4 + g];
      bv[s4] = w2s[col * W2LD + 4 * s4 + g];
    }
#pragma unroll
    for (int s4 = 0; s4 < 16; s4 += 2) {
      e = mma16<false>(av[s4], bv[s4], e, lane);
      o = mma16<false>(av[s4 + 1], bv[s4 + 1], o, lane);
    }
    float hv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) hv[r] = h1s[(4 * g + r) * ALD + col];
#pragma unroll
    for (int r = 0; r < 4; ++r) dz1s[(4 * g + r) * ALD + col] = (e[r] + o[r]) * (1.0f - hv[r] * hv[r]);
  }
  __syncthreads();
  PH_STAMP(a.prof, 8);
  // ---- weight gradients (rows >= R carry dlogits = 0, hence zeros all the way down); the slab holds the policy network's
  //      share only: [pi_W1 pi_b1 pi_W2 pi_b2 | act_W act_b] (the value side takes no part).  dW = In^T dOut over the 16 rows:
  //      K = 16 is four MFMAs per 16 x 16 tile of dW; wave w owns output columns 16w .. 16w + 15 and walks the row tiles ----
  float* out = a.extra + (size_t)blockIdx.x * adap_slab_len(lay);
  const int head0 = lay.vf_W1;   // slab offset of act_W
  {
    float b1v[4], b2v[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      b1v[s4] = dz1s[(4 * s4 + g) * ALD + col];
      b2v[s4] = dz2s[(4 * s4 + g) * ALD + col];
    }
    for (int m0 = 0; m0 < F; m0 += 64) {            // dW1[f][j] = sum_r X[r][f] dZ1[r][j], four 16-row tiles per batch
      float av[16];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        const int m = m0 + 16 * t4 + c, mc = m < F ? m : F - 1;   // rows >= F: duplicates that are never stored
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) av[4 * t4 + s4] = xs[(4 * s4 + g) * FP + mc];
      }
      f32x4 e[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        e[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) e[t4] = mma16<false>(av[4 * t4 + s4], b1v[s4], e[t4], lane);
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = m0 + 16 * t4 + 4 * g + r;
          if (m < F) out[lay.pi_W1 + m * HID + col] = e[t4][r];
        }
      }
    }
    {                                               // dW2[k][j] = sum_r H1[r][k] dZ2[r][j]
      float av[16];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) av[4 * t4 + s4] = h1s[(4 * s4 + g) * ALD + 16 * t4 + c];
      }
      f32x4 e[4];
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
        e[t4] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) e[t4] = mma16<false>(av[4 * t4 + s4], b2v[s4], e[t4], lane);
      }
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[lay.pi_W2 + (16 * t4 + 4 * g + r) * HID + col] = e[t4][r];
      }
    }
    for (int n0 = 0; n0 < L; n0 += 16) {            // d act_W[k][c'] = sum_r H2[r][k] dlogits[r][c']: wave w owns rows 16w ..
      const bool cok = n0 + c < L;
      f32x4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
        e = mma16<false>(h2s[(4 * s4 + g) * ALD + col], cok ? zs[(4 * s4 + g) * LP + n0 + c] : 0.f, e, lane);
      if (cok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[head0 + (16 * wave + 4 * g + r) * L + n0 + c] = e[r];
      }
    }
  }
  PH_STAMP(a.prof, 9);
  if (tid < 2 * HID) {   // b1, b2
    const float* dz = tid < HID ? dz1s : dz2s;
    const int j = tid & (HID - 1);
    float v[ADAP_ROWS], sum = 0.f;
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) v[rr] = dz[rr * ALD + j];
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) sum += v[rr];
    out[(tid < HID ? lay.pi_b1 : lay.pi_b2) + j] = sum;
  } else {
    for (int q = tid - 2 * HID; q < L; q += 128) {   // act_b[c]
      float sum = 0.f;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) sum += zs[rr * LP + q];
      out[head0 + HID * L + q] = sum;
    }
  }
  PH_STAMP(a.prof, 10);
}

hipError_t launch_adap_context(const AdapArgs& a, int nwg, hipStream_t s) {
  const size_t lds = adap_lds_bytes(a.nd, a.n_ctx, a.ctx_size);
  static bool opted[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 48 * 1024 && dev >= 0 && dev < 64 && !opted[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)adap_context_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    opted[dev] = true;
  }
  hipLaunchKernelGGL(adap_context_kernel, dim3(nwg), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace ph

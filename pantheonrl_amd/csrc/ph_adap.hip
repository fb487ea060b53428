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
// reads them too), runs the states' C context rows through the network, forms the pairwise terms and back-propagates them; every parameter's partial derivative is computed by one
// thread from LDS operands and written, already scaled by coeff / (pairs * states), to the workgroup's own slab in the
// canonical parameter order.  ppo_reduce_kernel adds the slabs to the PPO gradient in a fixed order before the norm, so the
// clip and the Adam step see the gradient of the whole loss exactly as the reference's single backward() does.
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
         + (size_t)3 * ADAP_ROWS * (nd.L + 1)  // zs lps pbs
         + (size_t)n_ctx * ctx_size            // contexts
         + (size_t)spw * npairs * (nd.A + 1)   // per-pair exp(-KL) and per-(pair, action component) KL
         + 32                                  // rowphys
         + (size_t)((nd.A + 4) & ~3);          // prefix sums of the action components
}
size_t adap_lds_bytes(const NetDims& nd, int n_ctx, int ctx_size) { return adap_lds_floats(nd, n_ctx, ctx_size) * sizeof(float); }

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
  const int F = nd.F, L = nd.L, FP = F + 1, LP = L + 1, P = lay.P, A = nd.A;
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
  float* pbs = lps + ADAP_ROWS * LP;      // probabilities
  float* cxs = pbs + ADAP_ROWS * LP;      // [C][cs] sampled contexts
  float* tvs = cxs + C * cs;              // [spw][npairs] exp(-KL)
  float* kls = tvs + spw * npairs;        // [spw][npairs][A] KL of every action component
  int* rowphys = (int*)(kls + spw * npairs * A);   // [spw] buffer row of every sampled state
  int* aoff = rowphys + 32;                        // [A + 1] first logit of every action component

  PH_STAMP(a.prof, 0);
  {
    const float4* g1 = reinterpret_cast<const float4*>(a.params + lay.pi_W1);
    const float4* g2 = reinterpret_cast<const float4*>(a.params + lay.pi_W2);
    float4 w2r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w2r[i] = g2[tid + 256 * i];                  // HID * HID / 4 = 1024 float4
#pragma unroll 4
    for (int e = tid; e < F * (HID / 4); e += 256) reinterpret_cast<float4*>(w1s)[e] = g1[e];
    for (int e = tid; e < HID * L; e += 256) aws[e] = a.params[lay.act_W + e];
    if (tid < HID) bs[tid] = a.params[lay.pi_b1 + tid];
    else if (tid < 2 * HID) bs[tid] = a.params[lay.pi_b2 + tid - HID];
    for (int e = tid; e < L; e += 256) bs[2 * HID + e] = a.params[lay.act_b + e];
    for (int e = tid; e <= A; e += 256) aoff[e] = nd.act_off[e];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i, k = e >> 4, c4 = e & 15;
      *reinterpret_cast<float4*>(w2s + k * W2LD + 4 * c4) = w2r[i];
    }
  }
  PH_STAMP(a.prof, 1);
  // ---- the samples: teacher-forced, or drawn here from the keyed streams ----
  const uint64_t key = epoch_key((a.seed ^ 0xADA9C0DEull) + (a.epoch ? *a.epoch : 0ull), (int)a.mbi);
  if (tid < C) {
    float* c = cxs + tid * cs;
    if (a.contexts) {
      for (int k = 0; k < cs; ++k) c[k] = a.contexts[tid * cs + k];
    } else if (a.sampler == PH_CTX_CATEGORICAL) {        // util.py:70-77
      int hot = (int)(philox_uniform(key, 1ull, (uint32_t)tid, 0u) * (float)cs);
      hot = hot >= cs ? cs - 1 : hot;
      for (int k = 0; k < cs; ++k) c[k] = k == hot ? 1.f : 0.f;
    } else {
      float ss = 0.f;
      for (int k = 0; k < cs; ++k) {
        const float u = philox_uniform(key, 1ull, (uint32_t)tid, (uint32_t)k);
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
      for (int k = 0; k < cs; ++k) a.used_contexts[tid * cs + k] = c[k];
  }
  if (tid >= 64 && tid < 64 + spw) {
    const int sl = tid - 64;
    int row = 0;
    if (sl < ns) {
      // th.randperm(B)[:num_state_samples] (util.py:106): explicit positions, or the head of a keyed permutation of [0, nb)
      const int q = a.state_idx ? a.state_idx[s0 + sl] : (int)feistel_perm((uint32_t)(s0 + sl), (uint32_t)a.nb, a.nb_hb, key);
      if (a.used_state_idx) a.used_state_idx[s0 + sl] = q;
      row = env_major_to_phys(a.idx[q], a.T, a.E);
    }
    rowphys[sl] = row;
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

  const int r = tid >> 4, cg = tid & 15;   // thread (row, group of 4 hidden units)
  const float* AW = aws;
  PH_STAMP(a.prof, 3);
  // ---- H1 = tanh(X W1 + b1) ----
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* x = xs + r * FP;
#pragma unroll 8
    for (int k = 0; k < F; ++k) {
      const float xv = x[k];
      const float4 w = *reinterpret_cast<const float4*>(w1s + k * HID + 4 * cg);
      acc[0] = __builtin_fmaf(xv, w.x, acc[0]);
      acc[1] = __builtin_fmaf(xv, w.y, acc[1]);
      acc[2] = __builtin_fmaf(xv, w.z, acc[2]);
      acc[3] = __builtin_fmaf(xv, w.w, acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h1s[r * ALD + 4 * cg + j] = fast_tanh(acc[j] + bs[4 * cg + j]);
  }
  __syncthreads();
  // ---- H2 = tanh(H1 W2 + b2) ----
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* h = h1s + r * ALD;
#pragma unroll 8
    for (int k = 0; k < HID; ++k) {
      const float hv = h[k];
      const float4 w = *reinterpret_cast<const float4*>(w2s + k * W2LD + 4 * cg);
      acc[0] = __builtin_fmaf(hv, w.x, acc[0]);
      acc[1] = __builtin_fmaf(hv, w.y, acc[1]);
      acc[2] = __builtin_fmaf(hv, w.z, acc[2]);
      acc[3] = __builtin_fmaf(hv, w.w, acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h2s[r * ALD + 4 * cg + j] = fast_tanh(acc[j] + bs[HID + 4 * cg + j]);
  }
  __syncthreads();
  PH_STAMP(a.prof, 4);
  // ---- logits = H2 act_W + act_b ----
  for (int c = cg; c < L; c += 16) {
    float z = bs[2 * HID + c];
#pragma unroll 8
    for (int k = 0; k < HID; ++k) z = __builtin_fmaf(h2s[r * ALD + k], AW[k * L + c], z);
    zs[r * LP + c] = z;
  }
  __syncthreads();
  PH_STAMP(a.prof, 5);
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
  PH_STAMP(a.prof, 6);
  // ---- dL/dlogits.  L = w sum_{s, i<j} T_ij(s), T = exp(-KL_ij), w = coeff / (pairs * states):
  //        d KL_ij / d z_i[c] = p_i[c] ((lp_i[c] - lp_j[c]) - KL_ij^comp),   d KL_ij / d z_j[c] = p_j[c] - p_i[c] ----
  const float wgt = a.coef / (float)(npairs * a.n_states);
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
  __syncthreads();
  PH_STAMP(a.prof, 7);
  // ---- dZ2 = (dlogits act_W^T) * (1 - H2^2); thread (row, units cg + 16 j): row-strided LDS reads spread over the banks ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = cg + 16 * j;
    float d = 0.f;
    for (int c = 0; c < L; ++c) d = __builtin_fmaf(zs[r * LP + c], AW[k * L + c], d);
    const float hv = h2s[r * ALD + k];
    dz2s[r * ALD + k] = d * (1.0f - hv * hv);
  }
  __syncthreads();
  // ---- dZ1 = (dZ2 W2^T) * (1 - H1^2) ----
  {
    float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int c = 0; c < HID; c += 4) {
      const float z0 = dz2s[r * ALD + c], z1 = dz2s[r * ALD + c + 1], z2 = dz2s[r * ALD + c + 2], z3 = dz2s[r * ALD + c + 3];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 w = *reinterpret_cast<const float4*>(w2s + (cg + 16 * j) * W2LD + c);
        d[j] = __builtin_fmaf(z0, w.x, d[j]);
        d[j] = __builtin_fmaf(z1, w.y, d[j]);
        d[j] = __builtin_fmaf(z2, w.z, d[j]);
        d[j] = __builtin_fmaf(z3, w.w, d[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = cg + 16 * j;
      const float hv = h1s[r * ALD + k];
      dz1s[r * ALD + k] = d[j] * (1.0f - hv * hv);
    }
  }
  __syncthreads();
  PH_STAMP(a.prof, 8);
  // ---- every parameter's derivative by one thread (rows >= R carry dlogits = 0, hence zeros all the way down); the slab
  //      holds the policy network's share only: [pi_W1 pi_b1 pi_W2 pi_b2 | act_W act_b] (the value side takes no part) ----
  float* out = a.extra + (size_t)blockIdx.x * adap_slab_len(lay);
  const int head0 = lay.vf_W1;   // slab offset of act_W
  {
    // Thread (column j, wave f0) owns dW[f][j] for f = f0, f0 + 4, ...: the 16 row values of a column f are the same for the
    // whole wave, so lanes 0..15 fetch them with ONE LDS read and every FMA takes its row through v_readlane (an SGPR operand)
    // instead of 16 broadcast LDS reads per entry; two entries at a time keep two accumulation chains in flight.
    const int j = tid & (HID - 1), f0 = tid >> 6, rl = tid & (ADAP_ROWS - 1);
    float d[ADAP_ROWS];
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) d[rr] = dz1s[rr * ALD + j];
    for (int f = f0; f < F; f += 8) {
      const bool two = f + 4 < F;
      const float xa = xs[rl * FP + f], xb = xs[rl * FP + (two ? f + 4 : f)];
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) {
        sa = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xa), rr)), d[rr], sa);
        sb = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xb), rr)), d[rr], sb);
      }
      out[lay.pi_W1 + f * HID + j] = sa;
      if (two) out[lay.pi_W1 + (f + 4) * HID + j] = sb;
    }
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) d[rr] = dz2s[rr * ALD + j];
#pragma unroll 2
    for (int k = f0; k < HID; k += 8) {
      const float xa = h1s[rl * ALD + k], xb = h1s[rl * ALD + k + 4];
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) {
        sa = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xa), rr)), d[rr], sa);
        sb = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xb), rr)), d[rr], sb);
      }
      out[lay.pi_W2 + k * HID + j] = sa;
      out[lay.pi_W2 + (k + 4) * HID + j] = sb;
    }
  }
  PH_STAMP(a.prof, 9);
  if (tid < 2 * HID) {   // b1, b2
    const float* dz = tid < HID ? dz1s : dz2s;
    const int j = tid & (HID - 1);
    float s = 0.f;
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) s += dz[rr * ALD + j];
    out[(tid < HID ? lay.pi_b1 : lay.pi_b2) + j] = s;
  }
  for (int q = tid; q < HID * L + L; q += 256) {   // act_W[k][c], then act_b[c]
    float s = 0.f;
    if (q < HID * L) {
      const int k = q / L, c = q - k * L;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s = __builtin_fmaf(h2s[rr * ALD + k], zs[rr * LP + c], s);
    } else {
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s += zs[rr * LP + (q - HID * L)];
    }
    out[head0 + q] = s;
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

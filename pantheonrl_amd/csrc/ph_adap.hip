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
// sampled states, runs their C context rows through the policy network (weights straight from L2, where the gradient launch
// keeps them), forms the pairwise terms and back-propagates them; every parameter's partial derivative is computed by one
// thread from LDS operands and written, already scaled by coeff / (pairs * states), to the workgroup's own slab in the
// canonical parameter order.  ppo_reduce_kernel adds the slabs to the PPO gradient in a fixed order before the norm, so the
// clip and the Adam step see the gradient of the whole loss exactly as the reference's single backward() does.
#include "ph_launch.h"

namespace ph {

constexpr int ALD = HID + 1;   // padded leading dimension of the 64-wide activation tiles

int adap_workgroups(int n_ctx, int n_states) {
  const int spw = ADAP_ROWS / n_ctx;
  return (n_states + spw - 1) / spw;
}

// LDS floats of one workgroup
static size_t adap_lds_floats(const NetDims& nd, int n_ctx, int ctx_size) {
  const int spw = ADAP_ROWS / n_ctx, npairs = n_ctx * (n_ctx - 1) / 2;
  return (size_t)ADAP_ROWS * (nd.F + 1)        // xs
         + (size_t)4 * ADAP_ROWS * ALD         // h1s h2s dz1s dz2s
         + (size_t)3 * ADAP_ROWS * (nd.L + 1)  // zs lps pbs
         + (size_t)n_ctx * ctx_size            // contexts
         + (size_t)spw * npairs * (nd.A + 1)   // per-pair exp(-KL) and per-(pair, action component) KL
         + 32;                                 // rowphys
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

  float* xs = smem;                       // [ADAP_ROWS][FP]
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
  const float* W1 = a.params + lay.pi_W1;
  const float* W2 = a.params + lay.pi_W2;
  const float* AW = a.params + lay.act_W;
  // ---- H1 = tanh(X W1 + b1) ----
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* x = xs + r * FP;
#pragma unroll 8
    for (int k = 0; k < F; ++k) {
      const float xv = x[k];
      const float4 w = *reinterpret_cast<const float4*>(W1 + (size_t)k * HID + 4 * cg);
      acc[0] = __builtin_fmaf(xv, w.x, acc[0]);
      acc[1] = __builtin_fmaf(xv, w.y, acc[1]);
      acc[2] = __builtin_fmaf(xv, w.z, acc[2]);
      acc[3] = __builtin_fmaf(xv, w.w, acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h1s[r * ALD + 4 * cg + j] = fast_tanh(acc[j] + a.params[lay.pi_b1 + 4 * cg + j]);
  }
  __syncthreads();
  // ---- H2 = tanh(H1 W2 + b2) ----
  {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* h = h1s + r * ALD;
#pragma unroll 8
    for (int k = 0; k < HID; ++k) {
      const float hv = h[k];
      const float4 w = *reinterpret_cast<const float4*>(W2 + (size_t)k * HID + 4 * cg);
      acc[0] = __builtin_fmaf(hv, w.x, acc[0]);
      acc[1] = __builtin_fmaf(hv, w.y, acc[1]);
      acc[2] = __builtin_fmaf(hv, w.z, acc[2]);
      acc[3] = __builtin_fmaf(hv, w.w, acc[3]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) h2s[r * ALD + 4 * cg + j] = fast_tanh(acc[j] + a.params[lay.pi_b2 + 4 * cg + j]);
  }
  __syncthreads();
  // ---- logits = H2 act_W + act_b ----
  for (int c = cg; c < L; c += 16) {
    float z = a.params[lay.act_b + c];
#pragma unroll 8
    for (int k = 0; k < HID; ++k) z = __builtin_fmaf(h2s[r * ALD + k], AW[k * L + c], z);
    zs[r * LP + c] = z;
  }
  __syncthreads();
  // ---- log-softmax of every action component, one thread per (row, component) ----
  for (int e = tid; e < ADAP_ROWS * A; e += 256) {
    const int rr = e / A, comp = e - rr * A;
    const int lo = nd.act_off[comp], n = nd.act_off[comp + 1] - lo;
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
    const int lo = nd.act_off[comp], n = nd.act_off[comp + 1] - lo;
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
  if (tid == 0) {   // this workgroup's share of sum_s sum_pairs exp(-KL), in a fixed order
    float s = 0.f;
    for (int t = 0; t < ns * npairs; ++t) s += tvs[t];
    a.loss_part[blockIdx.x] = s;
  }
  // ---- dL/dlogits.  L = w sum_{s, i<j} T_ij(s), T = exp(-KL_ij), w = coeff / (pairs * states):
  //        d KL_ij / d z_i[c] = p_i[c] ((lp_i[c] - lp_j[c]) - KL_ij^comp),   d KL_ij / d z_j[c] = p_j[c] - p_i[c] ----
  const float wgt = a.coef / (float)(npairs * a.n_states);
  for (int e = tid; e < ADAP_ROWS * L; e += 256) {
    const int rr = e / L, c = e - rr * L;
    float d = 0.f;
    if (rr < R) {
      const int sl = rr / C, i = rr - sl * C;
      int comp = 0;
      while (nd.act_off[comp + 1] <= c) ++comp;
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
  // ---- dZ2 = (dlogits act_W^T) * (1 - H2^2) ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 4 * cg + j;
    float d = 0.f;
    for (int c = 0; c < L; ++c) d = __builtin_fmaf(zs[r * LP + c], AW[k * L + c], d);
    const float hv = h2s[r * ALD + k];
    dz2s[r * ALD + k] = d * (1.0f - hv * hv);
  }
  __syncthreads();
  // ---- dZ1 = (dZ2 W2^T) * (1 - H1^2) ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 4 * cg + j;
    float d = 0.f;
    const float* wrow = W2 + (size_t)k * HID;
#pragma unroll 4
    for (int c = 0; c < HID; c += 4) {
      const float4 w = *reinterpret_cast<const float4*>(wrow + c);
      d = __builtin_fmaf(dz2s[r * ALD + c], w.x, d);
      d = __builtin_fmaf(dz2s[r * ALD + c + 1], w.y, d);
      d = __builtin_fmaf(dz2s[r * ALD + c + 2], w.z, d);
      d = __builtin_fmaf(dz2s[r * ALD + c + 3], w.w, d);
    }
    const float hv = h1s[r * ALD + k];
    dz1s[r * ALD + k] = d * (1.0f - hv * hv);
  }
  __syncthreads();
  // ---- every parameter's derivative by one thread (rows >= R carry dlogits = 0, hence zeros all the way down) ----
  float* out = a.extra + (size_t)blockIdx.x * P;
  {
    const int j = tid & (HID - 1), f0 = tid >> 6;
    float d[ADAP_ROWS];
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) d[rr] = dz1s[rr * ALD + j];
    for (int f = f0; f < F; f += 4) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s = __builtin_fmaf(xs[rr * FP + f], d[rr], s);
      out[lay.pi_W1 + f * HID + j] = s;
    }
#pragma unroll
    for (int rr = 0; rr < ADAP_ROWS; ++rr) d[rr] = dz2s[rr * ALD + j];
    for (int k = f0; k < HID; k += 4) {
      float s = 0.f;
#pragma unroll
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s = __builtin_fmaf(h1s[rr * ALD + k], d[rr], s);
      out[lay.pi_W2 + k * HID + j] = s;
    }
  }
  for (int p = tid; p < P; p += 256) {
    float s = 0.f;
    if (p >= lay.pi_W1 && p < lay.pi_b1) continue;   // done above
    if (p >= lay.pi_W2 && p < lay.pi_b2) continue;
    if (p >= lay.pi_b1 && p < lay.pi_b1 + HID) {
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s += dz1s[rr * ALD + (p - lay.pi_b1)];
    } else if (p >= lay.pi_b2 && p < lay.pi_b2 + HID) {
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s += dz2s[rr * ALD + (p - lay.pi_b2)];
    } else if (p >= lay.act_W && p < lay.act_W + HID * L) {
      const int q = p - lay.act_W, k = q / L, c = q - k * L;
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s = __builtin_fmaf(h2s[rr * ALD + k], zs[rr * LP + c], s);
    } else if (p >= lay.act_b && p < lay.act_b + L) {
      for (int rr = 0; rr < ADAP_ROWS; ++rr) s += zs[rr * LP + (p - lay.act_b)];
    }                                                  // the value network and value head take no part in the context term
    out[p] = s;
  }
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

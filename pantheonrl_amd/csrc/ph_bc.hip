// Behavioural cloning on the engine (SURVEY.md 8f rank 4 tail): the reference's BC (pantheonrl/algos/bc.py:180-366) trains a
// FeedForward32Policy -- SB3 ActorCriticPolicy with net_arch = [32, 32], i.e. ONE shared 32-32 tanh trunk feeding action_net and
// value_net (pantheonrl/common/util.py:114-123) -- by supervised learning on (observation, action) pairs:
//     loss = -mean(log pi(a|o)) - ent_weight * mean(H[pi(.|o)]) + l2_weight * sum(w^2) / 2          (bc.py:291-303)
// with torch.optim.Adam defaults and DataLoader batches of 32 (bc.py:115, 262-267): one optimizer step per 32 rows.
//
// A run is therefore a chain of thousands of tiny dependent steps (~0.3 MFLOP each).  One launch per step would spend its time
// on launch boundaries; spreading one step over the chip would spend it on inter-workgroup hand-offs.  Here ONE persistent
// workgroup runs the whole chain inside a single launch: the parameter vector lives in LDS from the first minibatch to the
// last, every gradient entry is computed by the thread that owns the parameter (no atomics, no reduction pass) and that thread
// applies torch's Adam update on the spot; the moments stream through L2.  A 32x32 output tile on f32 MFMA would
// occupy one wave for 2 K cycles at the VALU's own f32 rate (guide section 3); the same tile as plain FMAs is spread over all
// four SIMDs of the CU, so the products here are VALU loops on LDS operands.
#include "ph_launch.h"

namespace ph {

constexpr int BH = PH_BC_HIDDEN;       // hidden width of the shared trunk
constexpr int BR = 32;                 // rows per tile (the reference's batch size)
constexpr int BLD = BH + 1;            // padded leading dimension of the 32-wide activation tiles

struct BcArgs {
  NetDims nd;            // obs_kind, D, F, A, L, obs_off, act_off (lay / Lp / nchunk unused)
  ph_bc_layout lay;
  float* params;         // (P) in/out
  float* adam_m;         // (P)
  float* adam_v;         // (P)
  int* step;             // optimizer steps applied so far (device), advanced by the number of minibatches
  const float* obs;      // (N, D) dataset
  const float* acts;     // (N, A) dataset, integer-valued
  const int* order;      // (n_epochs, N) visiting order (DataLoader(shuffle=True) draws one permutation per epoch)
  int N, batch, n_epochs, max_batches;
  float lr, beta1, beta2, eps, ent_weight, l2_weight;
  float* stats;          // (total minibatches, PH_BC_NSTAT) or null
};

__device__ __forceinline__ float bc_tanh(float x) { return fast_tanh(x); }

// block-wide sum of one float per thread (256 threads), result broadcast; red = 8 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bc_train_kernel(BcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDims& nd = a.nd;
  const ph_bc_layout& lay = a.lay;
  const int F = nd.F, L = nd.L, P = lay.P, FP = F + 1;
  float* ps = smem;                 // (P) parameters, resident for the whole run
  float* xs = ps + ((P + 3) & ~3);  // [BR][FP] features of the tile
  float* h1s = xs + BR * FP;        // [BR][BLD]
  float* h2s = h1s + BR * BLD;      // [BR][BLD]
  float* dz1s = h2s + BR * BLD;     // [BR][BLD]
  float* dz2s = dz1s + BR * BLD;    // [BR][BLD]
  float* zs = dz2s + BR * BLD;      // [BR][L + 1] logits -> dL/dlogits
  float* red = zs + BR * (L + 1);   // [8]
  int* rowidx = (int*)(red + 8);    // [BR] dataset row, -1 = padding
  float* gs = (float*)(rowidx + BR);   // (P) gradient of the minibatch when it spans several 32-row tiles
  int* rowact = (int*)(gs + P);        // [BR][A] expert actions of the tile
  __shared__ float bcorr[2];        // lr / (1 - beta1^t), sqrt(1 - beta2^t)

  const int tid = threadIdx.x;
  for (int p = tid; p < P; p += 256) ps[p] = a.params[p];
  int step0 = *a.step;
  __syncthreads();

  const int per_epoch = (a.N + a.batch - 1) / a.batch;
  int total = a.n_epochs * per_epoch;
  if (a.max_batches > 0 && a.max_batches < total) total = a.max_batches;
  // beta^t of Adam's bias corrections as running products in fp64 (thread 0; one pow at the start instead of two per step)
  double b1t = pow((double)a.beta1, (double)step0), b2t = pow((double)a.beta2, (double)step0);

#if defined(PH_BC_PROF)
  long long prof[16] = {0}, last = clock64();
#define BC_STAMP(i) do { if (tid == 0) { const long long now = clock64(); prof[i] += now - last; last = now; } } while (0)
#else
#define BC_STAMP(i) do { } while (0)
#endif
  for (int mb = 0; mb < total; ++mb) {
    const int ep = mb / per_epoch, b = mb - ep * per_epoch;
    const int start = b * a.batch;
    const int nb = (a.N - start < a.batch) ? a.N - start : a.batch;
    const float inv_nb = 1.0f / (float)nb;
    float s_lp = 0.f, s_h = 0.f, s_pt = 0.f;   // batch sums of log-prob, entropy, prob of the true action (threads < BR)

    for (int t0 = 0; t0 < nb; t0 += BR) {
      // ---- tile rows and features ----
      if (tid < BR) rowidx[tid] = (t0 + tid < nb) ? a.order[(size_t)ep * a.N + start + t0 + tid] : -1;
      __syncthreads();
      for (int e = tid; e < BR * nd.A; e += 256) {
        const int row = rowidx[e / nd.A];
        rowact[e] = row >= 0 ? (int)a.acts[(size_t)row * nd.A + (e % nd.A)] : 0;
      }
      if (nd.obs_kind == PH_SPACE_BOX) {
        for (int e = tid; e < BR * F; e += 256) {
          const int r = e / F, f = e - r * F, row = rowidx[r];
          xs[r * FP + f] = row >= 0 ? a.obs[(size_t)row * nd.D + f] : 0.f;
        }
      } else {
        for (int e = tid; e < BR * F; e += 256) xs[(e / F) * FP + (e % F)] = 0.f;
        __syncthreads();
        for (int e = tid; e < BR * nd.D; e += 256) {
          const int r = e / nd.D, comp = e - r * nd.D, row = rowidx[r];
          if (row < 0) continue;
          const int lo = nd.obs_off[comp], n = nd.obs_off[comp + 1] - lo;
          int x = (int)a.obs[(size_t)row * nd.D + comp];
          x = x < 0 ? 0 : (x >= n ? n - 1 : x);
          xs[r * FP + lo + x] = 1.f;
        }
      }
      __syncthreads();
      BC_STAMP(0);
      const int r = tid >> 3, cg = tid & 7;   // thread (row, group of 4 hidden units)
      // ---- H1 = tanh(X W1 + b1) ----
      {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* w = ps + lay.W1 + 4 * cg;
        const float* x = xs + r * FP;
#pragma unroll 8
        for (int k = 0; k < F; ++k) {
          const float xv = x[k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(xv, w[k * BH + j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) h1s[r * BLD + 4 * cg + j] = bc_tanh(acc[j] + ps[lay.b1 + 4 * cg + j]);
      }
      __syncthreads();
      BC_STAMP(1);
      // ---- H2 = tanh(H1 W2 + b2) ----
      {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* w = ps + lay.W2 + 4 * cg;
#pragma unroll 8
        for (int k = 0; k < BH; ++k) {
          const float hv = h1s[r * BLD + k];
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(hv, w[k * BH + j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) h2s[r * BLD + 4 * cg + j] = bc_tanh(acc[j] + ps[lay.b2 + 4 * cg + j]);
      }
      __syncthreads();
      BC_STAMP(2);
      // ---- logits = H2 act_W + act_b ----
      for (int c = cg; c < L; c += 8) {
        float z = ps[lay.act_b + c];
#pragma unroll 8
        for (int k = 0; k < BH; ++k) z = __builtin_fmaf(h2s[r * BLD + k], ps[lay.act_W + k * L + c], z);
        zs[r * (L + 1) + c] = z;
      }
      __syncthreads();
      BC_STAMP(3);
      // ---- per row: log-prob of the expert action, entropy, dL/dlogits (one lane per row; sums over action components) ----
      if (tid < BR) {
        float* z = zs + tid * (L + 1);
        const int row = rowidx[tid];
        if (row >= 0) {
          float lp = 0.f, ent = 0.f;
          for (int comp = 0; comp < nd.A; ++comp) {
            const int lo = nd.act_off[comp], n = nd.act_off[comp + 1] - lo;
            float mx = -3.0e38f;
            for (int c = 0; c < n; ++c) mx = fmaxf(mx, z[lo + c]);
            float se = 0.f;
            for (int c = 0; c < n; ++c) se += __expf(z[lo + c] - mx);
            const float lse = mx + __logf(se);
            int act = rowact[tid * nd.A + comp];
            act = act < 0 ? 0 : (act >= n ? n - 1 : act);
            float h = 0.f;
            for (int c = 0; c < n; ++c) {
              const float lq = z[lo + c] - lse;
              h -= __expf(lq) * lq;
            }
            lp += z[lo + act] - lse;
            ent += h;
            // d/dz_c [ -(1/nb) log p_a - (w/nb) H ] = -(1/nb)([c = a] - p_c) + (w/nb) p_c ((z_c - lse) + H)
            for (int c = 0; c < n; ++c) {
              const float lq = z[lo + c] - lse, pc = __expf(lq);
              z[lo + c] = -inv_nb * (((c == act) ? 1.f : 0.f) - pc) + a.ent_weight * inv_nb * pc * (lq + h);
            }
          }
          s_lp += lp;
          s_h += ent;
          s_pt += __expf(lp);
        } else {
          for (int c = 0; c < L; ++c) z[c] = 0.f;
        }
      }
      __syncthreads();
      BC_STAMP(4);
      // ---- dZ2 = (dlogits act_W^T) * (1 - H2^2) ----
      {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * cg + j;
          float d = 0.f;
          for (int c = 0; c < L; ++c) d = __builtin_fmaf(zs[r * (L + 1) + c], ps[lay.act_W + k * L + c], d);
          const float hv = h2s[r * BLD + k];
          dz2s[r * BLD + k] = d * (1.0f - hv * hv);
        }
      }
      __syncthreads();
      BC_STAMP(5);
      // ---- dZ1 = (dZ2 W2^T) * (1 - H1^2) ----
      {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * cg + j;
          float d = 0.f;
#pragma unroll 8
          for (int c = 0; c < BH; ++c) d = __builtin_fmaf(dz2s[r * BLD + c], ps[lay.W2 + k * BH + c], d);
          const float hv = h1s[r * BLD + k];
          dz1s[r * BLD + k] = d * (1.0f - hv * hv);
        }
      }
      __syncthreads();
      BC_STAMP(6);
      // ---- owner-computes gradients: every parameter's sum over the tile's rows, by the thread that will update it ----
      {
        // W1[f][j] and W2[k][j]: thread tid owns column j = tid & 31 of rows f = (tid >> 5) + 8 i.  The 32 values of
        // dZ[:, j] are read once into registers; each entry is then 32 independent LDS reads and one FMA chain.
        const int j = tid & (BH - 1), f0 = tid >> 5;
        float d[BR];
#pragma unroll
        for (int rr = 0; rr < BR; ++rr) d[rr] = dz1s[rr * BLD + j];
        for (int f = f0; f < F; f += 8) {
          float x[BR];
#pragma unroll
          for (int rr = 0; rr < BR; ++rr) x[rr] = xs[rr * FP + f];
          float s = 0.f;
#pragma unroll
          for (int rr = 0; rr < BR; ++rr) s = __builtin_fmaf(x[rr], d[rr], s);
          const int p = lay.W1 + f * BH + j;
          gs[p] = (t0 == 0) ? s : gs[p] + s;
        }
#pragma unroll
        for (int rr = 0; rr < BR; ++rr) d[rr] = dz2s[rr * BLD + j];
        for (int k = f0; k < BH; k += 8) {
          float x[BR];
#pragma unroll
          for (int rr = 0; rr < BR; ++rr) x[rr] = h1s[rr * BLD + k];
          float s = 0.f;
#pragma unroll
          for (int rr = 0; rr < BR; ++rr) s = __builtin_fmaf(x[rr], d[rr], s);
          const int p = lay.W2 + k * BH + j;
          gs[p] = (t0 == 0) ? s : gs[p] + s;
        }
      }
      BC_STAMP(7);
      for (int p = lay.b1 + tid; p < P; p += 256) {   // biases and the head: the remaining few hundred entries
        if (p >= lay.W2 && p < lay.b2) continue;       // W2 done above
        float s = 0.f;
        if (p < lay.W2) {                       // b1[j]
          const int j = p - lay.b1;
#pragma unroll 8
          for (int rr = 0; rr < BR; ++rr) s += dz1s[rr * BLD + j];
        } else if (p < lay.act_W) {             // b2[j]
          const int j = p - lay.b2;
#pragma unroll 8
          for (int rr = 0; rr < BR; ++rr) s += dz2s[rr * BLD + j];
        } else if (p < lay.act_b) {             // act_W[k][c]
          const int q = p - lay.act_W, k = q / L, c = q - k * L;
#pragma unroll 8
          for (int rr = 0; rr < BR; ++rr) s = __builtin_fmaf(h2s[rr * BLD + k], zs[rr * (L + 1) + c], s);
        } else if (p < lay.val_W) {             // act_b[c]
          const int c = p - lay.act_b;
#pragma unroll 8
          for (int rr = 0; rr < BR; ++rr) s += zs[rr * (L + 1) + c];
        }                                       // value_net receives no gradient from the BC loss (only the l2 term)
        gs[p] = (t0 == 0) ? s : gs[p] + s;      // each entry is touched by its owner only
      }
      __syncthreads();
    }

    BC_STAMP(8);
    // ---- statistics (before the update, like the reference's stats_dict) and Adam ----
    float sq = 0.f;
    for (int p = tid; p < P; p += 256) sq = __builtin_fmaf(ps[p], ps[p], sq);
    const float l2_norm = 0.5f * block_sum(sq, red, tid);
    const float mean_lp = block_sum(tid < BR ? s_lp : 0.f, red, tid) * inv_nb;
    const float mean_h = block_sum(tid < BR ? s_h : 0.f, red, tid) * inv_nb;
    const float mean_pt = block_sum(tid < BR ? s_pt : 0.f, red, tid) * inv_nb;
    if (tid == 0) {
      b1t *= (double)a.beta1;
      b2t *= (double)a.beta2;
      bcorr[0] = (float)((double)a.lr / (1.0 - b1t));
      bcorr[1] = (float)sqrt(1.0 - b2t);
      if (a.stats) {
        float* st = a.stats + (size_t)mb * PH_BC_NSTAT;
        const float neglogp = -mean_lp, ent_loss = -a.ent_weight * mean_h, l2_loss = a.l2_weight * l2_norm;
        st[0] = neglogp;
        st[1] = mean_h;
        st[2] = ent_loss;
        st[3] = mean_pt;
        st[4] = l2_norm;
        st[5] = l2_loss;
        st[6] = neglogp + ent_loss + l2_loss;
        st[7] = (float)nb;
      }
    }
    __syncthreads();
    BC_STAMP(9);
    const float step_size = bcorr[0], bc2s = bcorr[1];
#pragma unroll 4
    for (int p = tid; p < P; p += 256) {
      const float w = ps[p];
      const float gr = gs[p] + a.l2_weight * w;
      const float m0 = a.adam_m[p], v0 = a.adam_v[p];
      const float m = m0 + (gr - m0) * (1.0f - a.beta1);           // exp_avg.lerp_(grad, 1 - beta1)
      const float v = v0 * a.beta2 + (1.0f - a.beta2) * gr * gr;   // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
      a.adam_m[p] = m;
      a.adam_v[p] = v;
      ps[p] = w - step_size * (m / (sqrtf(v) / bc2s + a.eps));     // param.addcdiv_(exp_avg, denom, -step_size)
    }
    __syncthreads();
  }
  BC_STAMP(10);
  for (int p = tid; p < P; p += 256) a.params[p] = ps[p];
  if (tid == 0) *a.step = step0 + total;
#if defined(PH_BC_PROF)
  if (tid == 0 && a.stats)
    for (int i = 0; i < 16; ++i) a.stats[i] = (float)((double)prof[i] / (double)total);
#endif
}

size_t bc_train_lds_bytes(int F, int L, int P, int A) {
  return sizeof(float) * (size_t)(((P + 3) & ~3) + BR * (F + 1) + 4 * BR * BLD + BR * (L + 1) + 8 + BR + P + BR * A);
}

struct BcTrainLaunch {
  NetDims nd;
  ph_bc_layout lay;
};

hipError_t launch_bc_train(const NetDims& nd, const ph_bc_layout& lay, float* params, float* adam_m, float* adam_v, int* step,
                           const float* obs, const float* acts, const int* order, int N, int batch, int n_epochs,
                           int max_batches, const ph_bc_hyper& hp, float* stats, hipStream_t s) {
  BcArgs a;
  a.nd = nd;
  a.lay = lay;
  a.params = params;
  a.adam_m = adam_m;
  a.adam_v = adam_v;
  a.step = step;
  a.obs = obs;
  a.acts = acts;
  a.order = order;
  a.N = N;
  a.batch = batch;
  a.n_epochs = n_epochs;
  a.max_batches = max_batches;
  a.lr = hp.learning_rate;
  a.beta1 = hp.adam_beta1;
  a.beta2 = hp.adam_beta2;
  a.eps = hp.adam_eps;
  a.ent_weight = hp.ent_weight;
  a.l2_weight = hp.l2_weight;
  a.stats = stats;
  const size_t lds = bc_train_lds_bytes(nd.F, nd.L, lay.P, nd.A);
  static size_t allowed[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (lds > allowed[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)bc_train_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed[dev] = lds;
  }
  hipLaunchKernelGGL(bc_train_kernel, dim3(1), dim3(256), lds, s, a);
  return hipGetLastError();
}

// ---- forward of the shared-trunk policy: one lane per row, weights staged per workgroup ------------------------------------------
struct BcFwdArgs {
  NetDims nd;
  ph_bc_layout lay;
  const float* params;
  const float* obs;              // (n, D)
  int n;
  const unsigned char* mask;     // (n, L) or null
  const float* uniforms;         // (n, A) or null
  const float* given;            // (n, A) or null: evaluate these actions
  uint64_t seed, counter;
  int deterministic;
  int* act_i32;
  float* values;
  float* logp;
  float* entropy;
  float* logits;
};

__global__ __launch_bounds__(64) void bc_forward_kernel(BcFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDims& nd = a.nd;
  const ph_bc_layout& lay = a.lay;
  const int F = nd.F, L = nd.L, P = lay.P;
  float* ps = smem;
  for (int p = threadIdx.x; p < P; p += 64) ps[p] = a.params[p];
  __syncthreads();
  const int row = blockIdx.x * 64 + threadIdx.x;
  if (row >= a.n) return;
  float h1[BH], h2[BH];
#pragma unroll
  for (int j = 0; j < BH; ++j) h1[j] = ps[lay.b1 + j];
  if (nd.obs_kind == PH_SPACE_BOX) {
    for (int k = 0; k < F; ++k) {
      const float x = a.obs[(size_t)row * nd.D + k];
#pragma unroll
      for (int j = 0; j < BH; ++j) h1[j] = __builtin_fmaf(x, ps[lay.W1 + k * BH + j], h1[j]);
    }
  } else {   // one-hot rows: the dense layer's k-ordered sum with the zero terms left out
    for (int comp = 0; comp < nd.D; ++comp) {
      const int lo = nd.obs_off[comp], n = nd.obs_off[comp + 1] - lo;
      int x = (int)a.obs[(size_t)row * nd.D + comp];
      x = x < 0 ? 0 : (x >= n ? n - 1 : x);
#pragma unroll
      for (int j = 0; j < BH; ++j) h1[j] += ps[lay.W1 + (lo + x) * BH + j];
    }
  }
#pragma unroll
  for (int j = 0; j < BH; ++j) h1[j] = bc_tanh(h1[j]);
#pragma unroll
  for (int j = 0; j < BH; ++j) h2[j] = ps[lay.b2 + j];
#pragma unroll
  for (int k = 0; k < BH; ++k) {
#pragma unroll
    for (int j = 0; j < BH; ++j) h2[j] = __builtin_fmaf(h1[k], ps[lay.W2 + k * BH + j], h2[j]);
  }
  float v = ps[lay.val_b];
#pragma unroll
  for (int j = 0; j < BH; ++j) {
    h2[j] = bc_tanh(h2[j]);
    v = __builtin_fmaf(h2[j], ps[lay.val_W + j], v);
  }
  if (a.values) a.values[row] = v;
  float lp_sum = 0.f, ent_sum = 0.f;
  for (int comp = 0; comp < nd.A; ++comp) {
    const int lo = nd.act_off[comp], n = nd.act_off[comp + 1] - lo;
    float z[PH_MAX_LOGITS];
    float mx = -3.0e38f;
    int arg = 0;
    for (int c = 0; c < n; ++c) {
      float s = ps[lay.act_b + lo + c];
#pragma unroll
      for (int k = 0; k < BH; ++k) s = __builtin_fmaf(h2[k], ps[lay.act_W + k * L + lo + c], s);
      if (a.mask && a.mask[(size_t)row * L + lo + c] == 0) s -= 30.0f;   // modular/policies.py:330-333
      if (a.logits) a.logits[(size_t)row * L + lo + c] = s;
      z[c] = s;
      if (s > mx) {
        mx = s;
        arg = c;
      }
    }
    float se = 0.f;
    for (int c = 0; c < n; ++c) se += __expf(z[c] - mx);
    const float lse = mx + __logf(se);
    int act = arg;
    if (a.given) {
      act = (int)a.given[(size_t)row * nd.A + comp];
      act = act < 0 ? 0 : (act >= n ? n - 1 : act);
    } else if (!a.deterministic) {
      const float u = a.uniforms ? a.uniforms[(size_t)row * nd.A + comp] : philox_uniform(a.seed, a.counter, (uint32_t)row, (uint32_t)comp);
      float cdf = 0.f;
      act = n - 1;
      for (int c = 0; c < n; ++c) {
        cdf += __expf(z[c] - lse);
        if (u < cdf) {
          act = c;
          break;
        }
      }
    }
    float h = 0.f;
    for (int c = 0; c < n; ++c) {
      const float lq = z[c] - lse;
      h -= __expf(lq) * lq;
    }
    lp_sum += z[act] - lse;
    ent_sum += h;
    if (a.act_i32) a.act_i32[(size_t)row * nd.A + comp] = act;
  }
  if (a.logp) a.logp[row] = lp_sum;
  if (a.entropy) a.entropy[row] = ent_sum;
}

hipError_t launch_bc_forward(const NetDims& nd, const ph_bc_layout& lay, const float* params, const float* obs, int n,
                             const unsigned char* mask, const float* uniforms, const float* given, uint64_t seed,
                             uint64_t counter, int deterministic, int* act_i32, float* values, float* logp, float* entropy,
                             float* logits, hipStream_t s) {
  BcFwdArgs a;
  a.nd = nd;
  a.lay = lay;
  a.params = params;
  a.obs = obs;
  a.n = n;
  a.mask = mask;
  a.uniforms = uniforms;
  a.given = given;
  a.seed = seed;
  a.counter = counter;
  a.deterministic = deterministic;
  a.act_i32 = act_i32;
  a.values = values;
  a.logp = logp;
  a.entropy = entropy;
  a.logits = logits;
  const size_t lds = sizeof(float) * (size_t)lay.P;
  static size_t allowed[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  if (lds > allowed[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)bc_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed[dev] = lds;
  }
  hipLaunchKernelGGL(bc_forward_kernel, dim3((n + 63) / 64), dim3(64), lds, s, a);
  return hipGetLastError();
}

}  // namespace ph

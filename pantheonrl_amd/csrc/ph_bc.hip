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

// debug: -DPH_BC_PROF makes thread 0 accumulate the shader clock per phase (scripts/bc_phase_profile.py)
#if defined(PH_BC_PROF)
#define BC_PROF_DECL long long prof[16] = {0}, last = clock64()
#define BC_STAMP(i) do { if (tid == 0) { const long long now = clock64(); prof[i] += now - last; last = now; } } while (0)
#else
#define BC_PROF_DECL do { } while (0)
#define BC_STAMP(i) do { } while (0)
#endif

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

  BC_PROF_DECL;
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


// ---- the same chain with the five products on MFMA tiles ----------------------------------------------------------------------
// The VALU loops above are LDS-latency bound (scripts/bc_phase_profile.py: 59 k cycles per 32-row step, of which 19 k are the
// moments' round trip + the next rows' dependent gather, 9.6 k dZ1, 9.3 k the weight-gradient columns).  Here
//   * X.W1, H1.W2, H2.act_W and dZ2.W2^T are split over the four waves along K (each wave 32x32 v_mfma_f32_32x32x2_f32 partial
//     tiles over a quarter of K, summed through LDS): one MFMA instruction does 2 048 MACs for two operand reads per lane;
//   * dW1 / dW2 / d act_W are whole 32x32 tiles over the 32 rows, dealt to the waves round-robin, stored to the gradient
//     buffer straight from the accumulators;
//   * the dataset rows of the NEXT tile (index -> observation / action gather, two dependent HBM round trips) are fetched into
//     registers while the current tile computes; the index of the tile after that rides one step further ahead;
//   * Adam's moments live in LDS beside the parameters when they fit (else they stream through L2).
// Box or one-hot observations with D <= 128 stored components, heads up to 64 logits; other shapes take the kernel above.
constexpr int BC_XR = 16;   // raw observation values prefetched per thread: 32 rows * D / 256

struct BcTile {   // which rows of the visiting order a tile covers
  int ep, start, t0, nb;
};

// the four split-K partial tiles of element o, all reads issued together (4 unrolled elements: 16 reads in flight)
#define BC_PART4(dst, base)                                                                                        \
  do {                                                                                                             \
    float pa_[4], pb_[4], pc_[4], pd_[4];                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                             \
      const int o_ = (base) + j_;                                                                                  \
      pa_[j_] = part[o_];                                                                                          \
      pb_[j_] = part[BR * BLD + o_];                                                                               \
      pc_[j_] = part[2 * BR * BLD + o_];                                                                           \
      pd_[j_] = part[3 * BR * BLD + o_];                                                                           \
    }                                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)(dst)[j_] = (pa_[j_] + pb_[j_]) + (pc_[j_] + pd_[j_]);         \
  } while (0)

template <bool moments_in_lds>
__global__ __launch_bounds__(256) void bc_train_mfma_kernel(BcArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDims& nd = a.nd;
  const ph_bc_layout& lay = a.lay;
  const int F = nd.F, L = nd.L, P = lay.P, D = nd.D;
  const int Fpad = (F + 31) & ~31, XP = Fpad + 1;       // K of layer 1 padded so that a quarter of it is a multiple of 8
  const int nLt = (L + 31) >> 5, ZP = 32 * nLt + 1;     // logits in nLt column tiles
  const int Lk = (L + 7) & ~7;                          // K of dZ2 = dlogits . act_W^T
  const int P4 = (P + 3) & ~3;
  float* ps = smem;                       // (P) parameters, resident for the whole run
  float* gs = ps + P4;                    // (P) gradient of the minibatch
  float* ms = gs + P4;                    // (P) exp_avg     } only when moments_in_lds
  float* vs = ms + (moments_in_lds ? P4 : 0);   // (P) exp_avg_sq
  float* xs = vs + (moments_in_lds ? P4 : 0);   // [BR][XP] features, columns F..Fpad-1 zero
  float* h1s = xs + BR * XP;              // [BR][BLD]
  float* h2s = h1s + BR * BLD;
  float* dz1s = h2s + BR * BLD;
  float* dz2s = dz1s + BR * BLD;
  float* zs = dz2s + BR * BLD;            // [BR][ZP] logits -> dL/dlogits (columns L..Lk-1 zero)
  float* part = zs + BR * ZP;             // [4][BR][BLD] split-K partial tiles
  float* red = part + 4 * BR * BLD;       // [4][4] block reduction
  int* rowidx = (int*)(red + 16);         // [BR] dataset row of the current tile, -1 = padding
  int* rownext = rowidx + BR;             // [BR] dataset row of the next tile
  int* rowact = rownext + BR;             // [BR][A]
  __shared__ float bcorr[2];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lh = lane >> 5;
  for (int p = tid; p < P; p += 256) {
    ps[p] = a.params[p];
    if (moments_in_lds) {
      ms[p] = a.adam_m[p];
      vs[p] = a.adam_v[p];
    }
  }
  for (int e = tid; e < BR * XP; e += 256) xs[e] = 0.f;
  for (int e = tid; e < BR * ZP; e += 256) zs[e] = 0.f;
  const int step0 = *a.step;
  const int per_epoch = (a.N + a.batch - 1) / a.batch;
  int total = a.n_epochs * per_epoch;
  if (a.max_batches > 0 && a.max_batches < total) total = a.max_batches;
  double b1t = pow((double)a.beta1, (double)step0), b2t = pow((double)a.beta2, (double)step0);
  BC_PROF_DECL;

  // flat tile sequence over (minibatch, 32-row tile)
  auto tile_of = [&](int mb, int t0) -> BcTile {
    BcTile t;
    t.ep = mb / per_epoch;
    const int b = mb - t.ep * per_epoch;
    t.start = b * a.batch;
    t.nb = (a.N - t.start < a.batch) ? a.N - t.start : a.batch;
    t.t0 = t0;
    return t;
  };
  auto advance = [&](int& mb, int& t0) {   // -> the tile after (mb, t0); mb == total when the run is over
    const BcTile t = tile_of(mb, t0);
    if (t0 + BR < t.nb) t0 += BR;
    else { mb += 1; t0 = 0; }
  };
  auto index_of = [&](int mb, int t0) -> int {   // dataset row this thread (tid < BR) serves in that tile, -1 = none
    if (tid >= BR || mb >= total) return -1;
    const BcTile t = tile_of(mb, t0);
    return (t0 + tid < t.nb) ? a.order[(size_t)t.ep * a.N + t.start + t0 + tid] : -1;
  };
  const int n_raw = BR * D;               // raw observation values of a tile, n_raw <= 256 * BC_XR
  float xr[BC_XR];
  int ar = 0;
  auto gather = [&](const int* rows) {    // issue the loads of a tile's rows; nothing here waits for them
#pragma unroll
    for (int i = 0; i < BC_XR; ++i) {
      const int e = tid + 256 * i;
      xr[i] = 0.f;
      if (e < n_raw) {
        const int r = e / D, row = rows[r];
        if (row >= 0) xr[i] = a.obs[(size_t)row * D + (e - r * D)];
      }
    }
    ar = 0;
    if (tid < BR * nd.A) {
      const int row = rows[tid / nd.A];
      if (row >= 0) ar = (int)a.acts[(size_t)row * nd.A + (tid % nd.A)];
    }
  };

  // pipeline fill: rows of tile 0 in registers, index of tile 1 in a register
  if (tid < BR) rownext[tid] = index_of(0, 0);
  __syncthreads();
  gather(rownext);
  int mb_i = 0, t0_i = 0;                 // the tile whose index sits in idx_ahead
  advance(mb_i, t0_i);
  int idx_ahead = index_of(mb_i, t0_i);

  for (int mb = 0; mb < total; ++mb) {
    const BcTile cur = tile_of(mb, 0);
    const int nb = cur.nb;
    const float inv_nb = 1.0f / (float)nb;
    float s_lp = 0.f, s_h = 0.f, s_pt = 0.f;
    if (tid == 255) {   // this step's bias corrections (fp64, off the critical path: consumed after the statistics barrier)
      b1t *= (double)a.beta1;
      b2t *= (double)a.beta2;
      bcorr[0] = (float)((double)a.lr / (1.0 - b1t));
      bcorr[1] = (float)sqrt(1.0 - b2t);
    }

    for (int t0 = 0; t0 < nb; t0 += BR) {
      // ---- P0: the prefetched rows land in LDS; the next tile's rows and the index after that are requested ----
      if (tid < BR) rowidx[tid] = rownext[tid];
      if (tid < BR * nd.A) rowact[tid] = ar;
      if (nd.obs_kind == PH_SPACE_BOX) {
#pragma unroll
        for (int i = 0; i < BC_XR; ++i) {
          const int e = tid + 256 * i;
          if (e < n_raw) xs[(e / D) * XP + (e % D)] = xr[i];
        }
      } else {
        for (int e = tid; e < BR * F; e += 256) xs[(e / F) * XP + (e % F)] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BC_XR; ++i) {
          const int e = tid + 256 * i;
          if (e < n_raw) {
            const int r = e / D, comp = e - r * D;
            if (rownext[r] >= 0) {
              const int lo = nd.obs_off[comp], n = nd.obs_off[comp + 1] - lo;
              int x = (int)xr[i];
              x = x < 0 ? 0 : (x >= n ? n - 1 : x);
              xs[r * XP + lo + x] = 1.f;
            }
          }
        }
      }
      __syncthreads();
      if (tid < BR) rownext[tid] = idx_ahead;      // rows of the next tile (index fetched one step ago)
      __syncthreads();
      gather(rownext);                             // consumed at the next P0
      advance(mb_i, t0_i);
      idx_ahead = index_of(mb_i, t0_i);            // consumed one step later
      BC_STAMP(0);

      const int r = tid >> 3, cg = tid & 7;        // thread (row, group of 4 hidden units) of the VALU epilogues
      // ---- P1: H1 = tanh(X W1 + b1): split-K partial tiles, then the sum ----
      {
        f32x16 acc = {0};
        acc = tile_mma<false, false, false>(xs, XP, ps + lay.W1, BH, 0, 0, wave * (Fpad >> 2), Fpad >> 2, acc, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) part[(wave * BR + drow(q, lh)) * BLD + li] = acc[q];
      }
      __syncthreads();
      {
        float sum4[4], b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = ps[lay.b1 + 4 * cg + j];
        BC_PART4(sum4, r * BLD + 4 * cg);
#pragma unroll
        for (int j = 0; j < 4; ++j) h1s[r * BLD + 4 * cg + j] = bc_tanh(sum4[j] + b4[j]);
      }
      __syncthreads();
      BC_STAMP(1);
      // ---- P2: H2 = tanh(H1 W2 + b2) ----
      {
        f32x16 acc = {0};
        acc = tile_mma<false, false, false>(h1s, BLD, ps + lay.W2, BH, 0, 0, wave * 8, 8, acc, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) part[(wave * BR + drow(q, lh)) * BLD + li] = acc[q];
      }
      __syncthreads();
      {
        float sum4[4], b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b4[j] = ps[lay.b2 + 4 * cg + j];
        BC_PART4(sum4, r * BLD + 4 * cg);
#pragma unroll
        for (int j = 0; j < 4; ++j) h2s[r * BLD + 4 * cg + j] = bc_tanh(sum4[j] + b4[j]);
      }
      __syncthreads();
      BC_STAMP(2);
      // ---- P3: logits = H2 act_W + act_b (columns >= L of a tile are garbage and never read) ----
      for (int nt = 0; nt < nLt; ++nt) {
        f32x16 acc = {0};
        acc = tile_mma<false, false, false>(h2s, BLD, ps + lay.act_W, L, 0, 32 * nt, wave * 8, 8, acc, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) part[(wave * BR + drow(q, lh)) * BLD + li] = acc[q];
        __syncthreads();
        {
          float sum4[4];
          BC_PART4(sum4, r * BLD + 4 * cg);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = 32 * nt + 4 * cg + j;
            if (c < L) zs[r * ZP + c] = sum4[j] + ps[lay.act_b + c];
          }
        }
        __syncthreads();
      }
      BC_STAMP(3);
      // ---- P4: per row: log-prob of the expert action, entropy, dL/dlogits ----
      if (tid < BR) {
        float* z = zs + tid * ZP;
        if (rowidx[tid] >= 0) {
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
        for (int c = L; c < Lk; ++c) z[c] = 0.f;      // K padding of the next product
      }
      __syncthreads();
      BC_STAMP(4);
      // ---- P5: dZ2 = (dlogits act_W^T) * (1 - H2^2): K = Lk is small, one wave ----
      if (wave == 0) {
        f32x16 acc = {0};
        acc = tile_mma<false, true, false>(zs, ZP, ps + lay.act_W, L, 0, 0, 0, Lk, acc, lane);
        float hv[16];   // all reads, then all writes (the compiler serialises read - wait - write pairs it cannot prove disjoint)
#pragma unroll
        for (int q = 0; q < 16; ++q) hv[q] = h2s[drow(q, lh) * BLD + li];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 16; ++q) dz2s[drow(q, lh) * BLD + li] = acc[q] * (1.0f - hv[q] * hv[q]);
      }
      __syncthreads();
      BC_STAMP(5);
      // ---- P6: dZ1 = (dZ2 W2^T) * (1 - H1^2) ----
      {
        f32x16 acc = {0};
        acc = tile_mma<false, true, false>(dz2s, BLD, ps + lay.W2, BH, 0, 0, wave * 8, 8, acc, lane);
#pragma unroll
        for (int q = 0; q < 16; ++q) part[(wave * BR + drow(q, lh)) * BLD + li] = acc[q];
      }
      __syncthreads();
      {
        float hv[4], pa[4], pb[4], pc[4], pd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int o = r * BLD + 4 * cg + j;
          hv[j] = h1s[o];
          pa[j] = part[o];
          pb[j] = part[BR * BLD + o];
          pc[j] = part[2 * BR * BLD + o];
          pd[j] = part[3 * BR * BLD + o];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) dz1s[r * BLD + 4 * cg + j] = ((pa[j] + pb[j]) + (pc[j] + pd[j])) * (1.0f - hv[j] * hv[j]);
      }
      __syncthreads();
      BC_STAMP(6);
      // ---- P7: weight gradients as whole tiles over the 32 rows, dealt to the waves; biases by single lanes ----
      {
        const int nW1 = Fpad >> 5, ntiles = nW1 + 1 + nLt;
        const bool first = t0 == 0;
        for (int ti = wave; ti < ntiles; ti += 4) {
          f32x16 acc = {0};
          if (ti < nW1) {                       // dW1[f][j] = sum_r X[r][f] dZ1[r][j]
            acc = tile_mma<true, false, false>(xs, XP, dz1s, BLD, 32 * ti, 0, 0, BR, acc, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int f = 32 * ti + drow(q, lh);
              if (f < F) {
                const int p = lay.W1 + f * BH + li;
                gs[p] = first ? acc[q] : gs[p] + acc[q];
              }
            }
          } else if (ti == nW1) {               // dW2[k][j] = sum_r H1[r][k] dZ2[r][j]
            acc = tile_mma<true, false, false>(h1s, BLD, dz2s, BLD, 0, 0, 0, BR, acc, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int p = lay.W2 + drow(q, lh) * BH + li;
              gs[p] = first ? acc[q] : gs[p] + acc[q];
            }
          } else {                              // d act_W[k][c] = sum_r H2[r][k] dlogits[r][c]
            const int nt = ti - nW1 - 1;
            acc = tile_mma<true, false, false>(h2s, BLD, zs, ZP, 0, 32 * nt, 0, BR, acc, lane);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int c = 32 * nt + li;
              if (c < L) {
                const int p = lay.act_W + drow(q, lh) * L + c;
                gs[p] = first ? acc[q] : gs[p] + acc[q];
              }
            }
          }
        }
        if (tid < 2 * BH + L) {                 // b1 | b2 | act_b: column sums
          const float* src = tid < BH ? dz1s + tid : (tid < 2 * BH ? dz2s + (tid - BH) : zs + (tid - 2 * BH));
          const int ld = tid < 2 * BH ? BLD : ZP;
          float sum = 0.f;
#pragma unroll 8
          for (int rr = 0; rr < BR; ++rr) sum += src[rr * ld];
          const int p = tid < BH ? lay.b1 + tid : (tid < 2 * BH ? lay.b2 + (tid - BH) : lay.act_b + (tid - 2 * BH));
          gs[p] = first ? sum : gs[p] + sum;
        }
        if (first && tid < BH + 1) gs[lay.val_W + tid] = 0.f;   // value_net: no gradient from the BC loss (only l2)
      }
      __syncthreads();
      BC_STAMP(7);
    }

    // ---- P8: statistics (before the update, like the reference's stats_dict) and Adam ----
    float sq = 0.f;
    for (int p = tid; p < P; p += 256) sq = __builtin_fmaf(ps[p], ps[p], sq);
    float v4[4] = {sq, tid < BR ? s_lp : 0.f, tid < BR ? s_h : 0.f, tid < BR ? s_pt : 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v4[k] += __shfl_down(v4[k], off, 64);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) red[wave * 4 + k] = v4[k];
    }
    __syncthreads();
    if (tid == 0) {
      float t4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) t4[k] = (red[k] + red[4 + k]) + (red[8 + k] + red[12 + k]);
      if (a.stats) {
        float* st = a.stats + (size_t)mb * PH_BC_NSTAT;
        const float l2_norm = 0.5f * t4[0], mean_lp = t4[1] * inv_nb, mean_h = t4[2] * inv_nb;
        const float neglogp = -mean_lp, ent_loss = -a.ent_weight * mean_h, l2_loss = a.l2_weight * l2_norm;
        st[0] = neglogp;
        st[1] = mean_h;
        st[2] = ent_loss;
        st[3] = t4[3] * inv_nb;
        st[4] = l2_norm;
        st[5] = l2_loss;
        st[6] = neglogp + ent_loss + l2_loss;
        st[7] = (float)nb;
      }
    }
    __syncthreads();
    BC_STAMP(8);
    // torch's update with the two divisions and the square root on the hardware's 1-ulp v_rcp_f32 / v_sqrt_f32 (the IEEE
    // sequences cost ~60 instructions per parameter here -- the Adam loop was the longest phase of a step); the resulting
    // last-bit differences are far inside what 1e-3-sized Adam steps do to a near-zero gradient entry anyway
    const float step_size = bcorr[0], inv_bc2s = __builtin_amdgcn_rcpf(bcorr[1]);
#pragma unroll 4
    for (int p = tid; p < P; p += 256) {
      const float w = ps[p];
      const float gr = gs[p] + a.l2_weight * w;
      float m0, v0;
      if constexpr (moments_in_lds) {
        m0 = ms[p];
        v0 = vs[p];
      } else {
        m0 = a.adam_m[p];
        v0 = a.adam_v[p];
      }
      const float m = m0 + (gr - m0) * (1.0f - a.beta1);
      const float v = v0 * a.beta2 + (1.0f - a.beta2) * gr * gr;
      if constexpr (moments_in_lds) {
        ms[p] = m;
        vs[p] = v;
      } else {
        a.adam_m[p] = m;
        a.adam_v[p] = v;
      }
      ps[p] = w - step_size * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) * inv_bc2s + a.eps);
    }
    __syncthreads();
    BC_STAMP(9);
  }
  for (int p = tid; p < P; p += 256) {
    a.params[p] = ps[p];
    if (moments_in_lds) {
      a.adam_m[p] = ms[p];
      a.adam_v[p] = vs[p];
    }
  }
  if (tid == 0) *a.step = step0 + total;
#if defined(PH_BC_PROF)
  if (tid == 0 && a.stats)
    for (int i = 0; i < 16; ++i) a.stats[i] = (float)((double)prof[i] / (double)total);
#endif
}

static size_t bc_mfma_lds_bytes(int F, int L, int P, int A, bool moments) {
  const int Fpad = (F + 31) & ~31, XP = Fpad + 1, nLt = (L + 31) >> 5, ZP = 32 * nLt + 1, P4 = (P + 3) & ~3;
  return sizeof(float) * (size_t)(P4 * (moments ? 4 : 2) + BR * XP + 4 * BR * BLD + BR * ZP + 4 * BR * BLD + 16 + 2 * BR + BR * A);
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
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev = (dev >= 0 && dev < 64) ? dev : 0;
  static int use_mfma = -1;
  if (use_mfma < 0) {
    const char* e = getenv("PH_BC_MFMA");
    use_mfma = (e && e[0] == '0') ? 0 : 1;
  }
  if (use_mfma && BR * nd.D <= 256 * BC_XR && BR * nd.A <= 256 && nd.L <= 64 &&
      bc_mfma_lds_bytes(nd.F, nd.L, lay.P, nd.A, false) <= 160 * 1024) {
    const bool moments = bc_mfma_lds_bytes(nd.F, nd.L, lay.P, nd.A, true) <= 160 * 1024;
    const size_t lds = bc_mfma_lds_bytes(nd.F, nd.L, lay.P, nd.A, moments);
    static size_t allowed_m[2][64] = {{0}};
    if (lds > allowed_m[moments][dev]) {
      hipError_t e = moments ? hipFuncSetAttribute((const void*)bc_train_mfma_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                             : hipFuncSetAttribute((const void*)bc_train_mfma_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      allowed_m[moments][dev] = lds;
    }
    if (moments) hipLaunchKernelGGL(bc_train_mfma_kernel<true>, dim3(1), dim3(256), lds, s, a);
    else hipLaunchKernelGGL(bc_train_mfma_kernel<false>, dim3(1), dim3(256), lds, s, a);
    return hipGetLastError();
  }
  const size_t lds = bc_train_lds_bytes(nd.F, nd.L, lay.P, nd.A);
  static size_t allowed[64] = {0};
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
  float* zs = smem + ((P + 3) & ~3) + threadIdx.x;   // this lane's logits of one action component: zs[c * 64] (LDS, not scratch)
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
    auto z = [&](int c) -> float& { return zs[c * 64]; };
    float mx = -3.0e38f;
    int arg = 0;
    for (int c = 0; c < n; ++c) {
      float s = ps[lay.act_b + lo + c];
#pragma unroll
      for (int k = 0; k < BH; ++k) s = __builtin_fmaf(h2[k], ps[lay.act_W + k * L + lo + c], s);
      if (a.mask && a.mask[(size_t)row * L + lo + c] == 0) s -= 30.0f;   // modular/policies.py:330-333
      if (a.logits) a.logits[(size_t)row * L + lo + c] = s;
      z(c) = s;
      if (s > mx) {
        mx = s;
        arg = c;
      }
    }
    float se = 0.f;
    for (int c = 0; c < n; ++c) se += __expf(z(c) - mx);
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
        cdf += __expf(z(c) - lse);
        if (u < cdf) {
          act = c;
          break;
        }
      }
    }
    float h = 0.f;
    for (int c = 0; c < n; ++c) {
      const float lq = z(c) - lse;
      h -= __expf(lq) * lq;
    }
    lp_sum += z(act) - lse;
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
  const size_t lds = sizeof(float) * ((((size_t)lay.P + 3) & ~(size_t)3) + 64 * (size_t)PH_MAX_LOGITS);
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

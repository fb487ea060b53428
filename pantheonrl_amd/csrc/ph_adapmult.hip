// AdapPolicyMult on the device (pantheonrl/algos/adap/policies.py:136-283): AdapPolicy with MultModel as its extractor.  The
// stored observation row is features ++ context (adap_learn.py:448-452); per net (policies.py:239-264)
//     x      = tanh(W1 o + b1)                       o = the row WITHOUT its context
//     x_a    = tanh(Ws x + bs)     64 -> 64 C         viewed as (64, C): element [j][c] is output j C + c   (policies.py:245 / 258)
//     y      = x + x_a @ ctx                                                                               (policies.py:246 / 259)
//     latent = tanh(W2 y + b2)
// then action_net / value_net as in MlpPolicy.  This is NOT the 64-64 MLP the engine's fused kernels are built around, and ADAP is
// off the bench path (SURVEY.md 2 marks it out of scope as a feature), so the network is laid out as a CHAIN OF SMALL LAUNCHES over
// dense [rows][width] intermediates in HBM -- one dense layer, one elementwise step or one weight-gradient product per launch, each
// a few lines whose arithmetic can be read off (the products as f32 MFMA tiles) -- not as a fused tile kernel: correctness and the reference's semantics first
// (forward, PPO minibatch gradient, ADAP's context term; oracle: oracle/sb3_oracle.py AdapMultPolicyOracle).  tanh / exp / log are
// the engine's own definitions (ph_device.h), shared with every other kernel, so the rollout's log-probabilities and the
// update's agree as they do for MlpPolicy.  Sampling, log-prob and the fused RolloutBuffer.add are ph_rowtail.h's row tails.
#include "ph_launch.h"
#include "ph_rowtail.h"

namespace ph {

// ---- primitives -----------------------------------------------------------------------------------------------------------------
// The three products below run on the matrix pipe as 32x32 v_mfma_f32_32x32x2_f32 tiles on LDS operands (ph_device.h: tile_mma,
// one tile per wave of a 64 x 64 block) -- bit for bit the k-ordered fmaf chain per output element, i.e. the
// numbers of the register-tile fmaf loops these kernels were first written as.
// Y[r][n] = act(b[n] + sum_k X[r * ldx + k] W[k * N + n]), r < rows, n < N, K <= 64; act: 0 none, 1 tanh.  A block owns 64 rows x
// 64 columns; X (transposed to [k][row]) and W's columns pass through LDS once, zero-padded to 64; k ascending per entry.
__global__ __launch_bounds__(256) void am_dense_kernel(const float* __restrict__ X, int ldx, int K, const float* __restrict__ W,
                                                       const float* __restrict__ b, int N, float* __restrict__ Y, int rows, int act) {
  __shared__ __attribute__((aligned(16))) float xt[64][68], ws[64][68];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int r0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int rl = e >> 6, k = e & 63;   // consecutive lanes read consecutive k of one row of X
    xt[k][rl] = (r0 + rl < rows && k < K) ? X[(size_t)(r0 + rl) * ldx + k] : 0.f;
    const int kk = e >> 6, c = e & 63;   // ... and consecutive columns of one row of W
    ws[kk][c] = (kk < K && n0 + c < N) ? W[(size_t)kk * N + n0 + c] : 0.f;
  }
  __syncthreads();
  const int mt = wave >> 1, nt = wave & 1, li = lane & 31, lh = lane >> 5;
  const int n = n0 + nt * 32 + li;
  const float bv = n < N ? b[n] : 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = bv;
  acc = tile_mma<true, false, false>(&xt[0][0], 68, &ws[0][0], 68, mt * 32, nt * 32, 0, 64, acc, lane);   // rows k >= K are zero
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = r0 + mt * 32 + drow(r, lh);
      if (row < rows) Y[(size_t)row * N + n] = act ? fast_tanh(acc[r]) : acc[r];
    }
  }
}
static hipError_t dense(const float* X, int ldx, int K, const float* W, const float* b, int N, float* Y, int rows, int act,
                        hipStream_t s) {
  hipLaunchKernelGGL(am_dense_kernel, dim3((rows + 63) / 64, (N + 63) / 64), dim3(256), 0, s, X, ldx, K, W, b, N, Y, rows, act);
  return hipGetLastError();
}

// y[r][j] = x[r][j] + sum_c ctx[r][c] xa[r][j C + c]; ctx = the last C components of row r of X
__global__ void am_contract_kernel(const float* __restrict__ x, const float* __restrict__ xa, const float* __restrict__ X, int ldx,
                                   int Fo, int C, float* __restrict__ y, int rows) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)rows * HID) return;
  const size_t r = e / HID;
  const int j = (int)(e - r * HID);
  float v = x[e];
  for (int c = 0; c < C; ++c) v = fmaf(X[r * ldx + Fo + c], xa[r * HID * C + (size_t)j * C + c], v);
  y[e] = v;
}

// dX[r][k] (+)= sum_n dY[r][n] W[k * N + n]   (dX = dY W^T), K = 64.  64 rows per block; dY and W pass through LDS in chunks of
// 64 columns (W's chunk transposed to [n][k]), n in ascending order per entry; one 32 x 32 output tile per wave.
__global__ __launch_bounds__(256) void am_dense_dx_kernel(const float* __restrict__ dY, const float* __restrict__ W, int N, int K,
                                                          float* __restrict__ dX, int rows, int accumulate) {
  __shared__ __attribute__((aligned(16))) float ds[64][68], wt[64][68];   // [row][n], [n][k]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int mt = wave >> 1, nt = wave & 1, li = lane & 31, lh = lane >> 5;
  const int r0 = blockIdx.x * 64, k = nt * 32 + li;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + mt * 32 + drow(r, lh);
    acc[r] = (accumulate && row < rows) ? dX[(size_t)row * K + k] : 0.f;
  }
  for (int n0 = 0; n0 < N; n0 += 64) {
    const int nn = (N - n0 < 64) ? N - n0 : 64;
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) {
      const int rl = e >> 6, n = e & 63;
      ds[rl][n] = (r0 + rl < rows && n < nn) ? dY[(size_t)(r0 + rl) * N + n0 + n] : 0.f;
      const int kk = e >> 6;                 // consecutive lanes read consecutive n of one row kk of W
      wt[n][kk] = (n < nn) ? W[(size_t)kk * N + n0 + n] : 0.f;
    }
    __syncthreads();
    acc = tile_mma<false, false, false>(&ds[0][0], 68, &wt[0][0], 68, mt * 32, nt * 32, 0, 64, acc, lane);   // columns n >= nn are zero
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = r0 + mt * 32 + drow(r, lh);
    if (row < rows) dX[(size_t)row * K + k] = acc[r];
  }
}
static hipError_t dense_dx(const float* dY, const float* W, int N, int K, float* dX, int rows, int accumulate, hipStream_t s) {
  hipLaunchKernelGGL(am_dense_dx_kernel, dim3((rows + 63) / 64), dim3(256), 0, s, dY, W, N, K, dX, rows, accumulate);
  return hipGetLastError();
}

// out[e] = d[e] * (1 - a[e]^2)   (tanh backward; in place is fine)
__global__ void am_tanh_bwd_kernel(const float* __restrict__ d, const float* __restrict__ a, float* __restrict__ out, size_t n) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) out[e] = d[e] * (1.0f - a[e] * a[e]);
}
// dza[r][j C + c] = dy[r][j] ctx[r][c] (1 - xa[r][j C + c]^2)
__global__ void am_contract_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xa, const float* __restrict__ X,
                                       int ldx, int Fo, int C, float* __restrict__ dza, int rows) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t W = (size_t)HID * C;
  if (e >= (size_t)rows * W) return;
  const size_t r = e / W;
  const int n = (int)(e - r * W), j = n / C, c = n - j * C;
  const float a = xa[e];
  dza[e] = dy[r * HID + j] * X[r * ldx + Fo + c] * (1.0f - a * a);
}

// slab k (rows [k * per, (k + 1) * per) of the minibatch): slab[woff + kk * N + n] = sum_r X[r * ldx + kk] dY[r * N + n] and
// slab[boff + n] = sum_r dY[r * N + n].  A block owns a 64 (kk) x 64 (n) tile of the product for its slab, a wave one 32 x 32 tile of
// it; X and dY pass through LDS 32 rows at a time; every entry adds its rows in ascending order (a fixed summation order).
__global__ __launch_bounds__(256) void am_dense_dw_kernel(const float* __restrict__ X, int ldx, int K, const float* __restrict__ dY,
                                                          int N, float* __restrict__ slabs, int slab_len, int woff, int boff, int rows,
                                                          int per) {
  __shared__ __attribute__((aligned(16))) float xs[32][68], ds[32][68];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int mt = wave >> 1, nt = wave & 1, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 64;
  const int r0 = blockIdx.x * per, r1 = (r0 + per < rows) ? r0 + per : rows;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;   // tid < 64: column n0 + tid of dY
  for (int rb = r0; rb < r1; rb += 32) {
    __syncthreads();
    for (int e = tid; e < 32 * 64; e += 256) {
      const int rl = e >> 6, c = e & 63;
      const bool live = rb + rl < r1;
      xs[rl][c] = (live && c < K) ? X[(size_t)(rb + rl) * ldx + c] : 0.f;
      ds[rl][c] = (live && n0 + c < N) ? dY[(size_t)(rb + rl) * N + n0 + c] : 0.f;
    }
    __syncthreads();
    acc = tile_mma<true, false, false>(&xs[0][0], 68, &ds[0][0], 68, mt * 32, nt * 32, 0, 32, acc, lane);   // rows past r1 are zero
    if (tid < 64) {
#pragma unroll 8
      for (int rl = 0; rl < 32; ++rl) bsum += ds[rl][tid];
    }
  }
  float* slab = slabs + (size_t)blockIdx.x * slab_len;
  const int n = n0 + nt * 32 + li;
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = mt * 32 + drow(r, lh);
      if (kk < K) slab[woff + kk * N + n] = acc[r];
    }
  }
  if (tid < 64 && n0 + tid < N) slab[boff + n0 + tid] = bsum;
}
static hipError_t dense_dw(const float* X, int ldx, int K, const float* dY, int N, float* slabs, int nslab, int slab_len, int woff,
                           int boff, int rows, hipStream_t s) {
  const int per = (rows + nslab - 1) / nslab;
  hipLaunchKernelGGL(am_dense_dw_kernel, dim3(nslab, (N + 63) / 64), dim3(256), 0, s, X, ldx, K, dY, N, slabs, slab_len, woff, boff,
                     rows, per);
  return hipGetLastError();
}

// ---- one net, forward and backward ---------------------------------------------------------------------------------------------
static int net_off(const ph_adapmult_layout& L, int net, int which) {   // which: 0 W1 1 b1 2 Ws 3 bs 4 W2 5 b2
  const int pi[6] = {L.pi_W1, L.pi_b1, L.pi_Ws, L.pi_bs, L.pi_W2, L.pi_b2};
  const int vf[6] = {L.vf_W1, L.vf_b1, L.vf_Ws, L.vf_bs, L.vf_W2, L.vf_b2};
  return net == 0 ? pi[which] : vf[which];
}
hipError_t am_forward_net(const ph_adapmult_layout& L, const float* params, int net, const float* X, int ldx, int rows,
                          const AmWork& w, hipStream_t s) {
  const int C = L.C, Fo = L.Fo;
  hipError_t e;
  if ((e = dense(X, ldx, Fo, params + net_off(L, net, 0), params + net_off(L, net, 1), HID, w.x, rows, 1, s)) != hipSuccess) return e;
  if ((e = dense(w.x, HID, HID, params + net_off(L, net, 2), params + net_off(L, net, 3), HID * C, w.xa, rows, 1, s)) != hipSuccess)
    return e;
  {
    const size_t n = (size_t)rows * HID;
    hipLaunchKernelGGL(am_contract_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w.x, w.xa, X, ldx, Fo, C, w.y, rows);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if ((e = dense(w.y, HID, HID, params + net_off(L, net, 4), params + net_off(L, net, 5), HID, w.h, rows, 1, s)) != hipSuccess) return e;
  if (net == 0) return dense(w.h, HID, HID, params + L.act_W, params + L.act_b, L.L, w.z, rows, 0, s);   // logits [rows][L]
  return dense(w.h, HID, HID, params + L.val_W, params + L.val_b, 1, w.v, rows, 0, s);                     // value [rows]
}
// from w.dz [rows][L] (net 0) or w.dv [rows] (net 1) and the forward's intermediates: every parameter of the net and its head
// gets its entry in each of the nslab slabs (canonical parameter order; the other net's region is written by its own call)
// head_w_off / head_b_off: where the head's entries sit in a slab (canonical: act_W / act_b or val_W / val_b; the context term's
// slabs hold the policy-side parameters followed by the action head)
hipError_t am_backward_net(const ph_adapmult_layout& L, const float* params, int net, const float* X, int ldx, int rows,
                           const AmWork& w, float* slabs, int nslab, int slab_len, int head_w_off, int head_b_off, hipStream_t s) {
  const int C = L.C, Fo = L.Fo;
  const float* dout = net == 0 ? w.dz : w.dv;
  const int No = net == 0 ? L.L : 1;
  const float* Wo = params + (net == 0 ? L.act_W : L.val_W);
  hipError_t e;
  // head: d act_W = h^T dz, d act_b = column sums;  dh = dz act_W^T
  if ((e = dense_dw(w.h, HID, HID, dout, No, slabs, nslab, slab_len, head_w_off, head_b_off, rows, s)) != hipSuccess) return e;
  if ((e = dense_dx(dout, Wo, No, HID, w.dzh, rows, 0, s)) != hipSuccess) return e;
  const size_t n64 = (size_t)rows * HID, nC = (size_t)rows * HID * C;
  hipLaunchKernelGGL(am_tanh_bwd_kernel, dim3((unsigned)((n64 + 255) / 256)), dim3(256), 0, s, w.dzh, w.h, w.dzh, n64);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  // branch_2: dW2 = y^T dzh;  dy = dzh W2^T
  if ((e = dense_dw(w.y, HID, HID, w.dzh, HID, slabs, nslab, slab_len, net_off(L, net, 4), net_off(L, net, 5), rows, s)) != hipSuccess)
    return e;
  if ((e = dense_dx(w.dzh, params + net_off(L, net, 4), HID, HID, w.dy, rows, 0, s)) != hipSuccess) return e;
  // scaling: dza = dy ctx (1 - xa^2);  dWs = x^T dza;  dx = dy + dza Ws^T (into w.dy)
  hipLaunchKernelGGL(am_contract_bwd_kernel, dim3((unsigned)((nC + 255) / 256)), dim3(256), 0, s, w.dy, w.xa, X, ldx, Fo, C, w.dza,
                     rows);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = dense_dw(w.x, HID, HID, w.dza, HID * C, slabs, nslab, slab_len, net_off(L, net, 2), net_off(L, net, 3), rows, s)) !=
      hipSuccess)
    return e;
  if ((e = dense_dx(w.dza, params + net_off(L, net, 2), HID * C, HID, w.dy, rows, 1, s)) != hipSuccess) return e;
  // branch_1: dz1 = dx (1 - x^2);  dW1 = o^T dz1
  hipLaunchKernelGGL(am_tanh_bwd_kernel, dim3((unsigned)((n64 + 255) / 256)), dim3(256), 0, s, w.dy, w.x, w.dy, n64);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return dense_dw(X, ldx, Fo, w.dy, HID, slabs, nslab, slab_len, net_off(L, net, 0), net_off(L, net, 1), rows, s);
}

// ---- rollout step: row tails on the logits / values of am_forward_net ----------------------------------------------------------
__global__ __launch_bounds__(256) void am_act_kernel(FwdArgs a, const float* __restrict__ z, const float* __restrict__ v) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const NetDims& nd = a.nd;
  if (g < a.n) {
    float zr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) zr[k] = k < nd.L ? z[(size_t)g * nd.L + k] : 0.f;
    discrete8_row_tail(a, nd, g, zr, fwd_counter(a));   // mask offset, sampling / given action, log-prob, entropy, buffer row
    value_row_tail(a, g, v[g]);
  }
  if (a.rb_obs) {   // RolloutBuffer.add copies the observation (the full row: features ++ context)
    const size_t total = (size_t)a.n * nd.D;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x)
      a.rb_obs[e] = a.obs[e];
  }
}
hipError_t launch_am_act(const FwdArgs& a, const float* z, const float* v, hipStream_t s) {
  hipLaunchKernelGGL(am_act_kernel, dim3((a.n + 255) / 256), dim3(256), 0, s, a, z, v);
  return hipGetLastError();
}

// ---- the minibatch: gather, the two losses ---------------------------------------------------------------------------------------
// dense copies of the minibatch's rows and per-row scalars (idx = env-major indices, SB3's swap_and_flatten order)
__global__ void am_gather_kernel(AmGather g) {
  const int i = blockIdx.x;
  if (i >= g.nb) return;
  const int phys = env_major_to_phys(g.idx[i], g.T, g.E);
  for (int d = threadIdx.x; d < g.D; d += blockDim.x) g.xg[(size_t)i * g.D + d] = g.rb_obs[(size_t)phys * g.D + d];
  if (threadIdx.x == 0) {
    float adv = g.rb_adv[phys];
    if (g.norm_adv && g.nb > 1) adv = (adv - g.advstats[0]) / (g.advstats[1] + 1e-8f);   // adap_learn.py:269-271
    g.adv[i] = adv;
    g.act[i] = g.rb_act[phys];
    g.oldlp[i] = g.rb_logp[phys];
    g.ret[i] = g.rb_ret[phys];
    g.oldv[i] = g.rb_val[phys];
  }
}
hipError_t launch_am_gather(const AmGather& g, hipStream_t s) {
  hipLaunchKernelGGL(am_gather_kernel, dim3(g.nb), dim3(64), 0, s, g);
  return hipGetLastError();
}

// policy side of the PPO loss (adap_learn.py:253-320 without the context term; the arithmetic of ppo_grad_kernel's one-lane-per-row
// loss phase): dz = dL/dlogits, per-block partial statistics {policy loss, -, entropy loss, clip fraction, approximate KL}
__global__ __launch_bounds__(256) void am_loss_pi_kernel(const float* __restrict__ z, int L, const float* __restrict__ act,
                                                         const float* __restrict__ oldlp, const float* __restrict__ adv, int nb,
                                                         float clip, float ent_coef, float* __restrict__ dz,
                                                         float* __restrict__ statpart, int per) {
  __shared__ float sh[4][NSTATP];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * per, r1 = (r0 + per < nb) ? r0 + per : nb;
  const float inv_nb = 1.0f / (float)nb;
  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;
  for (int r = r0 + tid; r < r1; r += 256) {
    float zr[8];
    float m = -3.0e38f;
    for (int k = 0; k < L; ++k) {
      zr[k] = z[(size_t)r * L + k];
      m = fmaxf(m, zr[k]);
    }
    float se = 0.f;
    for (int k = 0; k < L; ++k) se += fast_exp(zr[k] - m);
    const float lse = m + fast_log(se);
    int a = (int)act[r];
    a = a < 0 ? 0 : (a >= L ? L - 1 : a);
    float ent = 0.f;
    for (int k = 0; k < L; ++k) {
      const float lp = zr[k] - lse;
      ent -= fast_exp(lp) * lp;
    }
    const float logp = zr[a] - lse;
    const float av = adv[r], lr = logp - oldlp[r], ratio = fast_exp(lr);
    const float lo_c = 1.0f - clip, hi_c = 1.0f + clip;
    const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
    const float pl1 = av * ratio, pl2 = av * rc;
    // torch.min backward: the smaller branch gets the gradient, ties split 1/2 + 1/2; clamp passes it iff lo <= ratio <= hi
    const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
    const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
    const float g_lp = -inv_nb * av * ratio * gate;   // dL/dlogp
    const float g_en = -ent_coef * inv_nb;             // dL/dH
    st[0] += -fminf(pl1, pl2);
    st[2] += -ent;
    st[3] += (fabsf(ratio - 1.0f) > clip) ? 1.f : 0.f;
    st[4] += (ratio - 1.0f) - lr;
    for (int k = 0; k < L; ++k) {
      const float lp = zr[k] - lse, p = fast_exp(lp);
      const float dlogp = ((k == a) ? 1.f : 0.f) - p;
      const float dent = -p * (lp + ent);
      dz[(size_t)r * L + k] = g_lp * dlogp + g_en * dent;
    }
  }
  // block sums in a fixed order: lanes by shuffles, waves through LDS
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) {
    float v = st[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) sh[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid < NSTATP) statpart[(size_t)blockIdx.x * NSTATP + tid] = (sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid]);
}
// value side: dv = dL/dv, partial statistic {-, value loss}
__global__ __launch_bounds__(256) void am_loss_vf_kernel(const float* __restrict__ v, const float* __restrict__ ret,
                                                         const float* __restrict__ oldv, int nb, float clip_vf, float vf_coef,
                                                         float* __restrict__ dv, float* __restrict__ statpart, int per) {
  __shared__ float sh[4];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * per, r1 = (r0 + per < nb) ? r0 + per : nb;
  const float inv_nb = 1.0f / (float)nb;
  float s1 = 0.f;
  for (int r = r0 + tid; r < r1; r += 256) {
    const float val = v[r], old = oldv[r];
    float vp = val, pass = 1.f;
    if (clip_vf >= 0.f) {   // adap_learn.py:288-298
      const float dlt = val - old;
      pass = (dlt >= -clip_vf && dlt <= clip_vf) ? 1.f : 0.f;
      vp = old + fminf(fmaxf(dlt, -clip_vf), clip_vf);
    }
    const float err = vp - ret[r];
    s1 += err * err;
    dv[r] = vf_coef * 2.0f * err * inv_nb * pass;
  }
  for (int off = 32; off > 0; off >>= 1) s1 += __shfl_down(s1, off, 64);
  if ((tid & 63) == 0) sh[tid >> 6] = s1;
  __syncthreads();
  if (tid < NSTATP) statpart[(size_t)blockIdx.x * NSTATP + tid] = tid == 1 ? (sh[0] + sh[1]) + (sh[2] + sh[3]) : 0.f;
}
hipError_t launch_am_loss(const AmWork& w, int L, int nb, const ph_ppo_hyper& hp, float* statpart, int nslab, int net,
                          hipStream_t s) {
  const int per = (nb + nslab - 1) / nslab;
  if (net == 0)
    hipLaunchKernelGGL(am_loss_pi_kernel, dim3(nslab), dim3(256), 0, s, w.z, L, w.act, w.oldlp, w.adv, nb, hp.clip_range, hp.ent_coef,
                       w.dz, statpart, per);
  else
    hipLaunchKernelGGL(am_loss_vf_kernel, dim3(nslab), dim3(256), 0, s, w.v, w.ret, w.oldv, nb, hp.clip_range_vf, hp.vf_coef, w.dv,
                       statpart + (size_t)nslab * NSTATP, per);
  return hipGetLastError();
}

// ---- ADAP's context term (adap/util.py:97-131) for this network --------------------------------------------------------------------
// rows (state s, context i) = the state's features ++ sampled context i, S * Cs of them; samples as in adap_context_kernel
// (teacher-forced, or Philox / the head of a keyed Feistel permutation of the minibatch)
__global__ __launch_bounds__(256) void am_ctx_rows_kernel(AmCtx a) {
  __shared__ float cxs[ADAP_ROWS * 8];
  __shared__ int rowphys[256];
  const int tid = threadIdx.x, Cs = a.n_ctx, cs = a.ctx_size;
  const uint64_t key = epoch_key((a.seed ^ 0xADA9C0DEull) + (a.epoch ? *a.epoch : 0ull), (int)a.mbi);
  if (tid < Cs) {
    float* c = cxs + tid * cs;
    if (a.contexts) {
      for (int k = 0; k < cs; ++k) c[k] = a.contexts[tid * cs + k];
    } else if (a.sampler == PH_CTX_NATURAL_NUMBERS) {
      int v = (int)(philox_uniform(key, 1ull, (uint32_t)tid, 0u) * (float)cs);
      v = v >= cs ? cs - 1 : v;
      for (int k = 0; k < cs; ++k) c[k] = k == 0 ? (float)v : 0.f;
    } else if (a.sampler == PH_CTX_CATEGORICAL) {
      int hot = (int)(philox_uniform(key, 1ull, (uint32_t)tid, 0u) * (float)cs);
      hot = hot >= cs ? cs - 1 : hot;
      for (int k = 0; k < cs; ++k) c[k] = k == hot ? 1.f : 0.f;
    } else {
      float ss = 0.f;
      for (int k = 0; k < cs; ++k) {
        const float u = philox_uniform(key, 1ull, (uint32_t)tid, (uint32_t)k);
        const float v = a.sampler == PH_CTX_POSITIVE_SQUARE ? u : u * 2.f - 1.f;
        c[k] = v;
        ss += v * v;
      }
      if (a.sampler == PH_CTX_L2) {
        const float nrm = sqrtf(ss);
        for (int k = 0; k < cs; ++k) c[k] = c[k] / nrm;
      }
    }
    if (a.used_contexts)
      for (int k = 0; k < cs; ++k) a.used_contexts[tid * cs + k] = c[k];
  }
  for (int sl = tid; sl < a.n_states; sl += 256) {
    const int q = a.state_idx ? a.state_idx[sl] : (int)feistel_perm((uint32_t)sl, (uint32_t)a.nb, a.nb_hb, key);
    if (a.used_state_idx) a.used_state_idx[sl] = q;
    rowphys[sl] = env_major_to_phys(a.idx[q], a.T, a.E);
  }
  __syncthreads();
  const int R = a.n_states * Cs, D = a.D, Fo = D - cs;
  for (int e = tid; e < R * D; e += 256) {
    const int r = e / D, f = e - r * D, sl = r / Cs, i = r - sl * Cs;
    a.rows[e] = f < Fo ? a.rb_obs[(size_t)rowphys[sl] * D + f] : cxs[i * cs + (f - Fo)];
  }
}
// one thread per state: log-softmax of its Cs rows, KL(pi_i || pi_j) of every pair i < j in itertools.combinations order,
// T = exp(-KL); loss share = sum of T; dz[i] -= w T p_i ((lp_i - lp_j) - KL), dz[j] -= w T (p_j - p_i), w = coef / (pairs * states)
__global__ void am_ctx_loss_kernel(const float* __restrict__ z, int L, int n_states, int Cs, float wgt, float* __restrict__ dz,
                                   float* __restrict__ loss_part) {
  const int sl = blockIdx.x * blockDim.x + threadIdx.x;
  if (sl >= n_states) return;
  float tsum = 0.f;
  const float* zs = z + (size_t)sl * Cs * L;
  float* ds = dz + (size_t)sl * Cs * L;
  for (int e = 0; e < Cs * L; ++e) ds[e] = 0.f;
  auto lse_of = [&](int i) {
    float m = -3.0e38f;
    for (int k = 0; k < L; ++k) m = fmaxf(m, zs[i * L + k]);
    float se = 0.f;
    for (int k = 0; k < L; ++k) se += __expf(zs[i * L + k] - m);
    return m + __logf(se);
  };
  for (int i = 0; i < Cs; ++i) {
    const float li = lse_of(i);
    for (int j = i + 1; j < Cs; ++j) {
      const float lj = lse_of(j);
      float kl = 0.f;
      for (int k = 0; k < L; ++k) {
        const float lpi = zs[i * L + k] - li, lpj = zs[j * L + k] - lj;
        kl += __expf(lpi) * (lpi - lpj);
      }
      const float tv = __expf(-kl);
      tsum += tv;
      for (int k = 0; k < L; ++k) {
        const float lpi = zs[i * L + k] - li, lpj = zs[j * L + k] - lj;
        const float pi = __expf(lpi), pj = __expf(lpj);
        ds[i * L + k] -= wgt * tv * pi * ((lpi - lpj) - kl);
        ds[j * L + k] -= wgt * tv * (pj - pi);
      }
    }
  }
  loss_part[sl] = tsum;
}
hipError_t launch_am_ctx_rows(const AmCtx& a, hipStream_t s) {
  hipLaunchKernelGGL(am_ctx_rows_kernel, dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_am_ctx_loss(const float* z, int L, int n_states, int Cs, float wgt, float* dz, float* loss_part, hipStream_t s) {
  hipLaunchKernelGGL(am_ctx_loss_kernel, dim3((n_states + 63) / 64), dim3(64), 0, s, z, L, n_states, Cs, wgt, dz, loss_part);
  return hipGetLastError();
}

}  // namespace ph

// ppo_grad_split_oh_kernel: the PPO minibatch gradient for ONE-HOT observations (Discrete / MultiDiscrete: Liar's Dice, BASELINE
// config 2 -- pantheonrl/envs/liargym/liar.py:18-19: 30 components, F = 270 features; action space MultiDiscrete [7, 12]) with
// every product on the bf16 matrix pipe at float32 accuracy: ppo_grad_split_kernel's recipe (ph_ppo_split.hip; SB3 PPO.train()
// inner loop, pantheonrl/common/agents.py:155, arithmetic pantheonrl/algos/adap/adap_learn.py:253-344) widened to several feature
// chunks and to heads of up to 32 logits in up to four action components.
//
// What the one-hot structure buys: a feature is 0 or 1, i.e. exactly a bf16, so X needs ONE plane -- the first layer's two
// products (X W1 forward, X^T dZ1 backward; 60 % of the exact-f32 general kernel's time on this shape) are three bf16 MFMA terms
// per 16x16x32 block instead of the sixteen-times-slower f32 instruction.  The tile's X -- all NCH chunks, 64 rows x 64 NCH
// features, 8 KB a chunk -- is built in LDS once per tile: the row's four lanes zero its NCH granule rows and drop a 1.0 at each
// of its D hot features (2-byte stores), and both products read that ONE layout (ds_read_b128 along the features for X W1,
// ds_read_b64_tr_b16 across the rows for X^T dZ1).  W1's fragments arrive pre-split from the weight image the optimizer keeps
// (ph_split.h), one chunk ahead of the product that uses them; nothing of W1 ever sits in LDS.
//
// The head is MFMA work too (19 logits x 64 units per row are too many for the four-lanes-per-row FMA loops of the small-head
// kernel): H2 is stored as planes [row][unit] straight from S2's accumulators (1 - H2^2 stays in registers for dZ2);
//   z^T   = act_W^T H2^T     A = head fragments from the image (logit x unit), B = H2 rows read along the units: wave w gets the
//                            logits of ITS 16 rows, lane (row, kg) holding logits 16 lb + 4 kg + r -- a row's softmax is
//                            in-lane work plus a reduction over the four lanes {j, j+16, j+32, j+48} (v_permlane16/32_swap);
//   dL/dz -> planes [row][32 logits] (the zero padding of logits >= L included: the products contract over all 32);
//   d act_W = H2^T dz        A = H2 read across the rows (tr), B = dz read across the rows (tr): wave w owns units 16 w ..;
//   dH2^T = act_W dz^T       A = head fragments (unit x logit), B = dz rows read along the logits; the result has S2's
//                            register layout, so dZ2 = dH2 (1 - H2^2) needs no exchange and is stored as planes over the
//                            wave's own columns of H2 (its d act_W product was their last reader).
// The value net runs the same code with one "logit" (the value) and the value loss in place of the surrogate.
//
// Grid (nWG, 2 nets), 64-row tiles, four waves, wave w = output columns 16 w .. 16 w + 15 of every product.  dW2, d b2, d b1, the
// head gradients accumulate in registers across a workgroup's tiles; dW1's NCH x 4 blocks go straight to the slab (first tile:
// stored; later tiles: re-loaded as the accumulator -- a minibatch of this shape class is one tile per workgroup: 8 192 rows =
// 128 tiles per net).  Slabs are in accumulator order (grad_slab_map_split_oh); 103 KB of LDS at NCH = 5: one workgroup per CU.
#include "ph_split_tile.h"

namespace ph {

namespace {
constexpr int OH_LBMAX = 2;                  // 16-logit blocks the dz planes and the head fragments are laid out for
constexpr int DZL_ROW = 64;                  // bytes of one dz plane row: 32 logits (bf16)
constexpr int DZL_PLANE = 64 * DZL_ROW;      // one dz plane

__host__ __device__ constexpr int oh_nfrag(int nch) { return 8 * nch + 16 + 2 * OH_LBMAX + 4; }
__host__ __device__ constexpr int oh_frag_w1(int nch, int wave, int ch, int c) { return (wave * nch + ch) * 2 + c; }
__host__ __device__ constexpr int oh_frag_w2f(int nch, int wave, int c) { return 8 * nch + wave * 2 + c; }
__host__ __device__ constexpr int oh_frag_w2b(int nch, int wave, int c) { return 8 * nch + 8 + wave * 2 + c; }
__host__ __device__ constexpr int oh_frag_hz(int nch, int lb, int c) { return 8 * nch + 16 + lb * 2 + c; }
__host__ __device__ constexpr int oh_frag_hd(int nch, int wave) { return 8 * nch + 16 + 2 * OH_LBMAX + wave; }
// slab of one (workgroup, net): [dW2 4096][dW1 nch x 4096][d b1 64][d b2 64][d head W OH_LBMAX x 1024][d head b 32]
__host__ __device__ constexpr int oh_rs_w1(int ch) { return 4096 * (1 + ch); }
__host__ __device__ constexpr int oh_rs_b1(int nch) { return 4096 * (1 + nch); }
__host__ __device__ constexpr int oh_rs_b2(int nch) { return oh_rs_b1(nch) + 64; }
__host__ __device__ constexpr int oh_rs_hw(int nch) { return oh_rs_b2(nch) + 64; }
__host__ __device__ constexpr int oh_rs_hb(int nch) { return oh_rs_hw(nch) + OH_LBMAX * 1024; }
__host__ __device__ constexpr int oh_rs_net(int nch) { return oh_rs_hb(nch) + 16 * OH_LBMAX; }

// dz planes [row][32 logits]: granule g (8 logits) of row a at granule g ^ ((a >> 2) & 3) -- sixteen rows read along the
// logits then fall on distinct bank groups, and the swizzle is constant over the four rows a transposing read spans
__device__ __forceinline__ int dzl_swz(int a) { return (a >> 2) & 3; }
}  // namespace

// X fragment of one (block, half chunk): one plane for one-hot observations, three for Box observations
template <bool BOX>
struct XFrag {
  bf16x8 p[BOX ? 3 : 1];
};
// acc[b] += X_b W for the four row (or feature) blocks b, the chains interleaved; small terms first
template <bool BOX>
__device__ __forceinline__ void xmma4(f32x4 (&acc)[4], const XFrag<BOX> (&x)[4], const Frag3& w) {
  if constexpr (BOX) {
    constexpr int TX[6] = {0, 2, 1, 0, 1, 0}, TW[6] = {2, 0, 1, 1, 0, 0};   // mma6's term order
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = mfma16(x[b].p[TX[t]], w.p[TW[t]], acc[b]);
  } else {
#pragma unroll
    for (int p = 2; p >= 0; --p)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = mfma16(x[b].p[0], w.p[p], acc[b]);
  }
}

// BOX: the same kernel for Box observations of 1 .. 4 feature chunks that the single-chunk / small-head kernels do not take (the
// reference's Overcooked + ADAP pairing: 62 + 3 = 65 features; MultiDiscrete heads): X is three planes per chunk, split here from
// the float32 rows (once per tile; no plane image), and its two products are six terms like every other.
template <int NCH, int LB, bool BOX>
__global__ __launch_bounds__(256, 1) void ppo_grad_split_oh_kernel(GradArgs a) {
  const int stop_now = __builtin_nontemporal_load(a.stop_flag);
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  char* smem = reinterpret_cast<char*>(smem_f);
  constexpr int R = 64;
  constexpr int XPL = BOX ? 3 : 1, XCH = XPL * PL_BYTES;   // planes and bytes of one chunk of X
  constexpr int XOH = 0, H1T = NCH * XCH, H2B = H1T + PB_BYTES, DZL = H2B + PB_BYTES, SC = DZL + 3 * DZL_PLANE;
  float* b1s = reinterpret_cast<float*>(smem + SC);   // [64]
  float* b2s = b1s + HID;                             // [64]
  float* hbs = b2s + HID;                             // [32] head bias (policy: act_b, zero beyond L | value: val_b)
  float* radv = hbs + 32;                             // [R] normalised advantage | return
  float* rold = radv + R;                             // [R] old log-prob | old value
  int* ract = reinterpret_cast<int*>(rold + R);       // [R][4] action of every component
  int* rowphys = ract + 4 * R;                        // [R]
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  constexpr int NFRAG = oh_nfrag(NCH);
  constexpr int RS_NETW = oh_rs_net(NCH);

  const int net = blockIdx.y;
  const int oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
  const float inv_nb = 1.0f / (float)a.nb;
  const int lbn = net == 0 ? LB : 1;                  // 16-logit blocks of this net's head
  const int f_last = nd.F - 64 * (NCH - 1);           // features of the last chunk (1 .. 64)
  const bool norm = net == 0 && a.norm_adv && a.nb > 1;
  const float* advp = a.advstats ? a.advstats : a.params;
  const float adv0 = __builtin_nontemporal_load(advp), adv1 = __builtin_nontemporal_load(advp + 1);
  const float adv_mean = norm ? adv0 : 0.f;
  const float adv_den = norm ? adv1 + 1e-8f : 1.f;
  // this wave's fragments of the weight image: fragment f, plane p at uint4 index ((net NFRAG + f) 3 + p) 64 + lane
  const uint4* const wimg = reinterpret_cast<const uint4*>(a.wimage) + (size_t)net * NFRAG * 3 * 64 + (threadIdx.x & 63);
  auto ld_frag = [&](int f) -> Frag3 {
    Frag3 r;
#pragma unroll
    for (int p = 0; p < 3; ++p) r.p[p] = __builtin_bit_cast(bf16x8, wimg[(f * 3 + p) * 64]);
    return r;
  };

  // ---- a tile's rows: lane (row = tid / 4, q = tid % 4) gathers the row's scalars and components q, q + 4, .. of its observation ----
  constexpr int MAXC = 16;                              // components per lane: D <= 64
  // what the gather LOADS, untouched: nothing here waits for a load, so a gather issued under another phase's products (the next
  // tile's, at the start of S7) costs that phase nothing; the arithmetic on the values happens at the commit
  struct RowGather {
    int phys;
    float x[MAXC];      // component q + 4 i of the observation (one-hot observations), as stored
    int lo[MAXC], n[MAXC];   // its first feature and its number of categories (n = 0: no such component)
    float s0, s1;       // q == 0: advantage | return, old log-prob | old value
    float act0, act1;   // q == 1: action components 0, 1;  q == 2: components 2, 3 (as stored)
  };
  auto load_phys = [&](int tile) -> int {   // the row's place in the rollout buffer (first of the gather's two dependent trips)
    const int gi = tile * R + ((int)threadIdx.x >> 2);
    return gi < a.nb ? (a.idx_phys ? a.idx_phys[gi] : minibatch_row(a, gi)) : -1;
  };
  auto gather_rows = [&](int phys) -> RowGather {
    const int tid = threadIdx.x, q = tid & 3;
    RowGather g;
    g.phys = phys;
    g.s0 = g.s1 = g.act0 = g.act1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      g.x[i] = 0.f;
      g.lo[i] = g.n[i] = 0;
    }
    if (g.phys >= 0) {
      const size_t ph_row = (size_t)g.phys;
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        const int comp = q + 4 * i;
        if (!BOX && comp < nd.D) {
          g.lo[i] = nd.obs_off[comp];
          g.n[i] = nd.obs_off[comp + 1] - g.lo[i];
          g.x[i] = a.rb_obs[ph_row * nd.D + comp];
        }
      }
      if (q == 0) {
        g.s0 = net == 0 ? a.rb_adv[ph_row] : a.rb_ret[ph_row];
        g.s1 = net == 0 ? a.rb_logp[ph_row] : a.rb_val[ph_row];
      } else if (net == 0 && q <= 2) {
        const int c0 = 2 * (q - 1);
        if (c0 < nd.A) g.act0 = a.rb_act[ph_row * nd.A + c0];
        if (c0 + 1 < nd.A) g.act1 = a.rb_act[ph_row * nd.A + c0 + 1];
      }
    }
    return g;
  };
  // BOX: the row's float32 features instead of hot positions -- lane q takes granules q and q + 4 (eight features each) of every chunk
  struct BoxRow {
    float x[BOX ? NCH : 1][2][8];
  };
  auto gather_box = [&](int phys) -> BoxRow {
    BoxRow bx;
    const int q = threadIdx.x & 3;
#pragma unroll
    for (int ch = 0; ch < (BOX ? NCH : 1); ++ch)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int f = 64 * ch + 8 * (q + 4 * gi) + e;
          bx.x[ch][gi][e] = (BOX && phys >= 0 && f < nd.F) ? a.rb_obs[(size_t)phys * nd.D + f] : 0.f;
        }
    return bx;
  };
  // X of the tile in LDS: the row's four lanes zero its NCH plane rows, then set the hot features (same wave: in order)
  auto zero_rows = [&]() {   // independent of the gather: issued while its loads travel
    if constexpr (BOX) return;   // (every granule of a Box row is written by the commit)
    const int tid = threadIdx.x, row = tid >> 2, q = tid & 3;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      uint4* xrow = reinterpret_cast<uint4*>(smem + XOH + ch * XCH + row * PL_ROW);
      xrow[q] = zero;
      xrow[q + 4] = zero;
    }
  };
  auto commit_rows = [&](const RowGather& g, const BoxRow& bx) {
    const int tid = threadIdx.x, row = tid >> 2, q = tid & 3;
    [[maybe_unused]] const SplitSel ssel = split_sel();
    wave_lds_sync();
    const int sw = pl_swz(row);
    if constexpr (BOX) {   // split the row's features into their three planes: one 16-byte store per plane and granule
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          Frag3 f3;
          split8(ssel, bx.x[ch][gi], f3);
          st_planes8(smem, XOH + ch * XCH + row * PL_ROW + ((((q + 4 * gi) ^ sw)) << 4), f3);
        }
    } else {
#pragma unroll
      for (int i = 0; i < MAXC; ++i) {
        if (g.n[i] > 0) {   // a live row's component: its category, clamped, is the hot feature
          int v = (int)g.x[i];
          v = v < 0 ? 0 : (v >= g.n[i] ? g.n[i] - 1 : v);
          const int f = g.lo[i] + v;
          const int ch = f >> 6, k = f & 63;
          *reinterpret_cast<unsigned short*>(smem + XOH + ch * XCH + row * PL_ROW + ((((k >> 3) ^ sw)) << 4) + 2 * (k & 7)) = 0x3F80;   // bf16 1.0
        }
      }
    }
    if (q == 0) {
      rowphys[row] = g.phys;
      radv[row] = (norm && g.phys >= 0) ? (g.s0 - adv_mean) / adv_den : g.s0;
      rold[row] = g.s1;
    } else if (q <= 2) {
      ract[row * 4 + 2 * (q - 1)] = (int)g.act0;
      ract[row * 4 + 2 * (q - 1) + 1] = (int)g.act1;
    }
  };

  // ---- prologue ----
  // W1's fragments of every chunk (this wave's 16 columns: NCH x 2 x 3 sixteen-byte loads per lane, L2-resident image) go out
  // first -- they depend on nothing, and behind the row gather's two dependent round trips they were a third one in front of S1
  // Chunks of W1 fragments in flight: a ring of two, refilled inside S1 (chunk ch + 2 requested when chunk ch's products are
  // issued).  All NCH up front is as fast alone (20.3 vs 20.0 us) but takes the kernel from 427 to 499 VGPRs -- and at 427 a wave of
  // the OTHER learner's reduce / Adam kernels (64 VGPRs) still fits on every SIMD beside this kernel's one wave, so the two
  // learners' updates overlap: 4.2 -> 4.0 ms per Liar's Dice iteration (profiles/r05_z_liar_grad_w1_slots_ab.txt).
#ifndef PH_OH_W1_SLOTS
#define PH_OH_W1_SLOTS 2
#endif
  constexpr int W1S = NCH < PH_OH_W1_SLOTS ? NCH : PH_OH_W1_SLOTS;
  Frag3 W1f[W1S][2];
#pragma unroll
  for (int ch = 0; ch < W1S; ++ch)
#pragma unroll
    for (int c = 0; c < 2; ++c) W1f[ch][c] = ld_frag(oh_frag_w1(NCH, threadIdx.x >> 6, ch, c));
  // ... then everything else that depends on nothing (biases, the action components' logit ranges), then the row gather; the
  // rows' X planes are zeroed while the loads travel
  float bias1 = 0.f, bias2 = 0.f, hb = 0.f;
  {
    const int tid = threadIdx.x;
    if (tid < HID) {
      bias1 = a.params[oB1 + tid];
      bias2 = a.params[oB2 + tid];
    }
    if (tid < 32) hb = net == 0 ? (tid < nd.L ? a.params[lay.act_b + tid] : 0.f) : (tid == 0 ? a.params[lay.val_b] : 0.f);
  }
  int alo[4], ahi[4];   // policy: component c's logits are [alo[c], ahi[c])
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    alo[c] = (net == 0 && c < nd.A) ? nd.act_off[c] : 0;
    ahi[c] = (net == 0 && c < nd.A) ? nd.act_off[c + 1] : 0;
  }
  __builtin_amdgcn_sched_barrier(0);   // (left to itself the scheduler sinks these loads to their first use, behind the gather)
  PH_STAMP(a.prof, 8);
  if (stop_now) return;
  zero_rows();
  RowGather rg = gather_rows(load_phys(blockIdx.x));
  BoxRow bxr = gather_box(rg.phys);
  PH_STAMP(a.prof, 9);
  {
    const int tid = threadIdx.x;
    if (tid < HID) {
      b1s[tid] = bias1;
      b2s[tid] = bias2;
    }
    if (tid < 32) hbs[tid] = hb;
  }

  f32x4 gW2[4];
  f32x4 gB1 = {0.f, 0.f, 0.f, 0.f}, gB2 = {0.f, 0.f, 0.f, 0.f};
  f32x4 gHW[OH_LBMAX], gHB[OH_LBMAX];
#pragma unroll
  for (int b = 0; b < 4; ++b) gW2[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int lb = 0; lb < OH_LBMAX; ++lb) gHW[lb] = gHB[lb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;

  float* const rslab = a.slabs + ((size_t)blockIdx.x * 2 + net) * RS_NETW;
  bool first = true;
  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x, first = false) {
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63, wave = tid >> 6;
    const SplitSel ssel = split_sel();   // the split's two selector words (ph_split_tile.h), rebuilt per tile
    const int j = lane & 15, kg = lane >> 4;
    const int unit = 16 * wave + j;   // this lane's column of every 16x16 result in the [k][unit] orientation
    const bool has_next = tile + (int)gridDim.x < a.ntiles;

    // ---- T0: rows -> X (one-hot plane), scalars ----
    if (!first) {
#pragma unroll
      for (int ch = 0; ch < W1S; ++ch)
#pragma unroll
        for (int c = 0; c < 2; ++c) W1f[ch][c] = ld_frag(oh_frag_w1(NCH, wave, ch, c));
      zero_rows();   // (the tile's rows were gathered under the previous tile's last phase)
    }
    if (first) PH_STAMP(a.prof, 10);
    commit_rows(rg, bxr);
    if (first) PH_STAMP(a.prof, 11);
    lds_barrier();
    if (first) PH_STAMP(a.prof, 1);

    // per-lane operand offsets (ph_ppo_split.hip): plain reads, transposing reads, C-layout stores
    const int pb0 = plain_base(j, kg, 0), pb1 = plain_base(j, kg, 1);
    int trb[4], csb[4];
    int tlo_w, thi_w;   // transposing read of columns 16 wave .. +15 of a [row][unit] plane buffer (contraction over its rows)
    {
      const int a0 = 8 * kg + (j >> 2);
      const int g0 = ((j & 3) >> 1) ^ pl_swz(a0), row0 = a0 * PL_ROW + 8 * (j & 1);
      const int g1 = (kg >> 1) ^ pl_swz(unit), row1 = unit * PL_ROW + 8 * (kg & 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        trb[k] = row0 + ((g0 ^ (2 * k)) << 4);
        csb[k] = row1 + ((g1 ^ (2 * k)) << 4);
      }
      tlo_w = row0 + ((g0 ^ (2 * wave)) << 4);
      thi_w = row0 + ((g0 ^ (2 * (wave ^ 1))) << 4);
    }
    // [row][unit] plane store of a swapped-role result (lane = row 16 b + j, registers = units 16 wave + 4 kg + r): 8 bytes a plane
    const int rst = j * PL_ROW + (((2 * wave + (kg >> 1)) ^ pl_swz(j)) << 4) + 8 * (kg & 1);

    // ---- S1: H1 = tanh(X W1 + b1) -> H1T planes; 1 - H1^2 kept for dZ1 ----
    f32x4 d1[4];
    {
      f32x4 acc[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // a block's X fragment: rows 16 b + i, features 32 c + 8 kg .. of chunk ch (one plane, or three)
      auto x_plain = [&](int ch, int c, int b) -> XFrag<BOX> {
        XFrag<BOX> f;
#pragma unroll
        for (int p = 0; p < XPL; ++p) f.p[p] = ld_plain1(smem, c == 0 ? pb0 : pb1, XOH + ch * XCH + p * PL_BYTES + b * 16 * PL_ROW);
        return f;
      };
      XFrag<BOX> xa[4], xb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        xa[b] = x_plain(0, 0, b);
        xb[b] = x_plain(0, 1, b);
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        xmma4<BOX>(acc, xa, W1f[ch % W1S][0]);   // four independent accumulator chains, interleaved
        xmma4<BOX>(acc, xb, W1f[ch % W1S][1]);
        if (ch + W1S < NCH) {   // this slot of the ring is free: chunk ch + W1S's fragments
#pragma unroll
          for (int c = 0; c < 2; ++c) W1f[ch % W1S][c] = ld_frag(oh_frag_w1(NCH, wave, ch + W1S, c));
        }
        if (ch + 1 < NCH) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            xa[b] = x_plain(ch + 1, 0, b);
            xb[b] = x_plain(ch + 1, 1, b);
          }
        }
      }
      const float bb = b1s[unit];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fast_tanh(acc[b][r] + bb);
          d1[b][r] = 1.0f - v[r] * v[r];
        }
        bf16x4 p[3];
        split4(ssel, v, p);
        st_planes4(smem, H1T + csb[b], p);
      }
    }
    Frag3 W2f[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) W2f[c] = ld_frag(oh_frag_w2f(NCH, wave, c));
    lds_barrier();
    if (first) PH_STAMP(a.prof, 2);

    // the next tile's rows: their places in the buffer are requested here, the rows themselves at the start of S7 (under its
    // products), and they are committed at the next tile's T0 -- a workgroup that walks several tiles pays the gather's two
    // dependent round trips once
    int phys_next = -1;
    if (has_next) phys_next = load_phys(tile + (int)gridDim.x);
    // ---- S2: H2 = tanh(H1 W2 + b2) -> H2 planes [row][unit] (roles swapped: result lane = row, registers = units) ----
    // (every fragment set is requested one phase ahead of its product: the image sits in L2, a microsecond away)
    Frag3 HZf[OH_LBMAX][2];
#pragma unroll
    for (int lb = 0; lb < OH_LBMAX; ++lb)
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (lb < LB) HZf[lb][c] = ld_frag(oh_frag_hz(NCH, lb, c));
    f32x4 d2[4];
    {
      const float4 bb = *reinterpret_cast<const float4*>(b2s + 16 * wave + 4 * kg);
      const float bbv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const Frag3 xa = ld_tr(smem, trb[b], trb[b ^ 1], H1T, H1T + 4 * PL_ROW);
        const Frag3 xb = ld_tr(smem, trb[b], trb[b ^ 1], H1T + 32 * PL_ROW, H1T + 36 * PL_ROW);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma6(W2f[0], xa, acc);
        acc = mma6(W2f[1], xb, acc);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fast_tanh(acc[r] + bbv[r]);
          d2[b][r] = 1.0f - v[r] * v[r];
        }
        bf16x4 p[3];
        split4(ssel, v, p);
        st_planes4(smem, H2B + b * 16 * PL_ROW + rst, p);
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 3);

    // ---- HZ: head forward of this wave's 16 rows, loss, dL/dz -> dz planes ----
    const Frag3 hd = ld_frag(oh_frag_hd(NCH, wave));   // A of dH2^T (HD): act_W rows (units) 16 w + i, logits 8 kg ..
    {
      const int row = 16 * wave + j;
      const bool valid = rowphys[row] >= 0;
      const float live = valid ? 1.f : 0.f;
      float z[OH_LBMAX][4];
#pragma unroll
      for (int lb = 0; lb < OH_LBMAX; ++lb)
#pragma unroll
        for (int r = 0; r < 4; ++r) z[lb][r] = 0.f;
      {
        const Frag3 h0 = ld_plain(smem, pb0, H2B + wave * 16 * PL_ROW), h1 = ld_plain(smem, pb1, H2B + wave * 16 * PL_ROW);
#pragma unroll
        for (int lb = 0; lb < OH_LBMAX; ++lb) {
          if (lb < LB && lb < lbn) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = mma6(HZf[lb][0], h0, acc);
            acc = mma6(HZf[lb][1], h1, acc);
            const float4 hb = *reinterpret_cast<const float4*>(hbs + 16 * lb + 4 * kg);
            z[lb][0] = acc[0] + hb.x;
            z[lb][1] = acc[1] + hb.y;
            z[lb][2] = acc[2] + hb.z;
            z[lb][3] = acc[3] + hb.w;
          }
        }
      }
      float dz[OH_LBMAX][4];
#pragma unroll
      for (int lb = 0; lb < OH_LBMAX; ++lb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dz[lb][r] = 0.f;
      if (net == 0) {
        float P[OH_LBMAX][4], lp[OH_LBMAX][4], hce[OH_LBMAX][4], isa[OH_LBMAX][4];
#pragma unroll
        for (int lb = 0; lb < OH_LBMAX; ++lb)
#pragma unroll
          for (int r = 0; r < 4; ++r) P[lb][r] = lp[lb][r] = hce[lb][r] = isa[lb][r] = 0.f;
        float logp = 0.f, ent = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (c >= nd.A) continue;
          const int lo = alo[c], hi = ahi[c];
          int act = ract[row * 4 + c];
          act = act < 0 ? 0 : (act >= hi - lo ? hi - lo - 1 : act);
          const int kact = lo + act;
          float m = -3.0e38f;
#pragma unroll
          for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * lb + 4 * kg + r;
              m = (k >= lo && k < hi) ? fmaxf(m, z[lb][r]) : m;
            }
          m = kg_max(m);
          float e[OH_LBMAX][4];
          float se = 0.f;
#pragma unroll
          for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * lb + 4 * kg + r;
              e[lb][r] = (k >= lo && k < hi) ? fast_exp(z[lb][r] - m) : 0.f;
              se += e[lb][r];
            }
          se = kg_sum(se);
          const float lse = m + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
          float hc = 0.f, zact = 0.f;
#pragma unroll
          for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * lb + 4 * kg + r;
              if (k >= lo && k < hi) {
                const float p = e[lb][r] * inv, l = z[lb][r] - lse;
                P[lb][r] = p;
                lp[lb][r] = l;
                hc -= p * l;
                zact += (k == kact) ? z[lb][r] : 0.f;
                isa[lb][r] = (k == kact) ? 1.f : 0.f;
              }
            }
          hc = kg_sum(hc);
          zact = kg_sum(zact);
#pragma unroll
          for (int lb = 0; lb < LB; ++lb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * lb + 4 * kg + r;
              hce[lb][r] = (k >= lo && k < hi) ? hc : hce[lb][r];
            }
          logp += zact - lse;
          ent += hc;
        }
        const float adv = radv[row];
        const float lr = logp - rold[row];
        const float ratio = fast_exp(lr);
        const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float pl1 = adv * ratio, pl2 = adv * rc;
        const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
        const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);   // torch.min / clamp backward
        const float g_lp = -inv_nb * adv * ratio * gate * live;
        const float g_en = -a.ent_coef * inv_nb * live;
        if (valid && kg == 0) {
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
        }
#pragma unroll
        for (int lb = 0; lb < LB; ++lb)
#pragma unroll
          for (int r = 0; r < 4; ++r)   // logits outside every component (padding): P = isa = 0 -> dz = 0
            dz[lb][r] = g_lp * (isa[lb][r] - P[lb][r]) + g_en * (-P[lb][r] * (lp[lb][r] + hce[lb][r]));
      } else {
        // value: the one "logit" is v (lane kg == 0, register 0 of block 0)
        const float v = z[0][0];
        const float retn = radv[row], oldv = rold[row];
        float vp = v, pass = 1.f;
        if (a.clip_vf >= 0.f) {
          const float dlt = v - oldv;
          pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
          vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
        }
        const float err = vp - retn;
        if (valid && kg == 0) st[1] += err * err;
        dz[0][0] = (valid && kg == 0) ? a.vf_coef * 2.0f * err * inv_nb * pass : 0.f;
      }
      const int sw = dzl_swz(row);
#pragma unroll
      for (int lb = 0; lb < OH_LBMAX; ++lb) {
        bf16x4 p[3];
        split4(ssel, dz[lb], p);
#pragma unroll
        for (int q = 0; q < 3; ++q)
          *reinterpret_cast<bf16x4*>(smem + DZL + q * DZL_PLANE + row * DZL_ROW + (((2 * lb + (kg >> 1)) ^ sw) << 4) + 8 * (kg & 1)) = p[q];
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 4);

    // ---- HD: d head weights (this wave's 16 units), d head bias, dH2 -> dZ2 planes over this wave's columns of H2 ----
    Frag3 W2b[2];   // B of dH1 (S6a)
#pragma unroll
    for (int c = 0; c < 2; ++c) W2b[c] = ld_frag(oh_frag_w2b(NCH, wave, c));
    {
      // dz read across the rows: lane (t = j, kg) addresses the 8-byte piece (row a + t / 4, logits 16 lb + 4 (t % 4) ..)
      auto dz_tr = [&](int lb, int c) -> Frag3 {
        Frag3 f;
        const int a0 = 32 * c + 8 * kg, t = j;
        const int o_lo = (a0 + (t >> 2)) * DZL_ROW + (((2 * lb + ((t & 3) >> 1)) ^ dzl_swz(a0)) << 4) + 8 * (t & 1);
        const int o_hi = (a0 + 4 + (t >> 2)) * DZL_ROW + (((2 * lb + ((t & 3) >> 1)) ^ dzl_swz(a0 + 4)) << 4) + 8 * (t & 1);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const bf16x4 lo4 = ld_tr4(smem, DZL + p * DZL_PLANE + o_lo), hi4 = ld_tr4(smem, DZL + p * DZL_PLANE + o_hi);
          f.p[p] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        return f;
      };
      {
        const Frag3 h0 = ld_tr(smem, tlo_w, thi_w, H2B, H2B + 4 * PL_ROW);                    // A: H2^T, units 16 w + i, rows 8 kg ..
        const Frag3 h1 = ld_tr(smem, tlo_w, thi_w, H2B + 32 * PL_ROW, H2B + 36 * PL_ROW);     //    rows 32 + 8 kg ..
#pragma unroll
        for (int lb = 0; lb < OH_LBMAX; ++lb) {
          if (lb < lbn) {
            const Frag3 z0 = dz_tr(lb, 0), z1 = dz_tr(lb, 1);
            gHW[lb] = mma6(h0, z0, gHW[lb]);
            gHW[lb] = mma6(h1, z1, gHW[lb]);
            if (wave == lb) {
              gHB[lb] = mma_ones(z0, gHB[lb]);
              gHB[lb] = mma_ones(z1, gHB[lb]);
            }
          }
        }
      }
      wave_lds_sync();   // this wave's H2 columns are consumed (its own reads above were their last): dZ2 goes over them
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        Frag3 zb;   // B: dz row 16 b + j, logits 8 kg .. +7
        const int rowb = 16 * b + j;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          zb.p[p] = *reinterpret_cast<const bf16x8*>(smem + DZL + p * DZL_PLANE + rowb * DZL_ROW + ((kg ^ dzl_swz(rowb)) << 4));
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma6(hd, zb, acc);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[r] * d2[b][r];
        bf16x4 p[3];
        split4(ssel, v, p);
        st_planes4(smem, H2B + b * 16 * PL_ROW + rst, p);
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 5);

    // ---- S6a: dW2 += H1^T dZ2, d b2; dH1 = dZ2 W2^T -> dZ1 (registers) ----
    f32x4 dh1[4];
    {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const Frag3 dzf = ld_tr(smem, tlo_w, thi_w, H2B + 32 * c * PL_ROW, H2B + (32 * c + 4) * PL_ROW);   // B: dZ2 columns 16 w .., rows 32 c + 8 kg ..
        gB2 = mma_ones(dzf, gB2);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const Frag3 h1 = ld_plain(smem, c == 0 ? pb0 : pb1, H1T + b * 16 * PL_ROW);   // A: H1T rows (input units) 16 b + i
          gW2[b] = mma6(h1, dzf, gW2[b]);
        }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)   // A: dZ2 rows 16 b + i, units 32 c + 8 kg ..
          acc = mma6(ld_plain(smem, c == 0 ? pb0 : pb1, H2B + b * 16 * PL_ROW), W2b[c], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) dh1[b][r] = acc[r] * d1[b][r];
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 6);

    // ---- S6b: dZ1 -> DZ1T planes over H1T (this wave's 16 units = the only rows its dW1 product reads) ----
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      float v[4] = {dh1[b][0], dh1[b][1], dh1[b][2], dh1[b][3]};
      bf16x4 p[3];
      split4(ssel, v, p);
      st_planes4(smem, H1T + csb[b], p);
    }
    wave_lds_sync();

    // ---- S7: dW1 = X^T dZ1 chunk by chunk, block by block, straight to the slab; d b1 ----
    if (has_next) {
      rg = gather_rows(phys_next);
      bxr = gather_box(phys_next);
    }
    {
      const Frag3 dz0 = ld_plain(smem, pb0, H1T + wave * 16 * PL_ROW);   // B: DZ1T row (unit) 16 w + j, rows 8 kg .. / 32 + 8 kg ..
      const Frag3 dz1 = ld_plain(smem, pb1, H1T + wave * 16 * PL_ROW);
      gB1 = mma_ones(dz0, gB1);
      gB1 = mma_ones(dz1, gB1);
      // chunk by chunk: the four blocks' operands are read together (and the next chunk's before this chunk's stores), the four
      // accumulator chains run interleaved, then the four 16-byte stores leave
      // A: features 64 ch + 16 b + i (lane), tile rows 32 c + 8 kg .. (contraction): transposing reads of X's plane(s)
      auto x_tr = [&](int ch, int c, int b) -> XFrag<BOX> {
        XFrag<BOX> f;
#pragma unroll
        for (int p = 0; p < XPL; ++p)
          f.p[p] = ld_tr1(smem, trb[b], trb[b ^ 1], XOH + ch * XCH + p * PL_BYTES + 32 * c * PL_ROW, XOH + ch * XCH + p * PL_BYTES + (32 * c + 4) * PL_ROW);
        return f;
      };
      XFrag<BOX> xa[4], xb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        xa[b] = x_tr(0, 0, b);
        xb[b] = x_tr(0, 1, b);
      }
#pragma unroll
      for (int ch = 0; ch < NCH; ++ch) {
        float* const dst = rslab + oh_rs_w1(ch) + (wave * 4 * 64 + lane) * 4;
        f32x4 g[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          g[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (!first) g[b] = *reinterpret_cast<const f32x4*>(dst + b * 256);
        }
        // blocks of the last chunk beyond F are padding: no parameter behind them (the slab map says -1), so they are not stored --
        // the slab stores are what bounds this phase (27 MB per launch at NCH = 5).  Their products still run: a branch around
        // them splits the basic block and with it the interleaving of the four chains (measured: + 1.4 k cycles).
        const int nb_live = (ch + 1 < NCH) ? 4 : (f_last + 15) >> 4;
        xmma4<BOX>(g, xa, dz0);
        xmma4<BOX>(g, xb, dz1);
        if (ch + 1 < NCH) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            xa[b] = x_tr(ch + 1, 0, b);
            xb[b] = x_tr(ch + 1, 1, b);
          }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (b >= nb_live) continue;
          if (has_next) *reinterpret_cast<f32x4*>(dst + b * 256) = g[b];   // re-read by this workgroup's next tile
          else st_slab16(dst + b * 256, g[b]);
        }
      }
    }
    lds_barrier();  // X / H1T / row scalars are free for the next tile
    if (first) PH_STAMP(a.prof, 7);
  }
  PH_STAMP(a.prof, 12);

  // ---- epilogue: register accumulators -> slab, statistics ----
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int b = 0; b < 4; ++b) st_slab16(rslab + ((wave * 4 + b) * 64 + lane) * 4, gW2[b]);
#pragma unroll
    for (int lb = 0; lb < OH_LBMAX; ++lb) st_slab16(rslab + oh_rs_hw(NCH) + ((lb * 4 + wave) * 64 + lane) * 4, gHW[lb]);
    if (lane < 16) {   // every row of a ones product is the column sum
      rslab[oh_rs_b1(NCH) + 16 * wave + lane] = gB1[0];
      rslab[oh_rs_b2(NCH) + 16 * wave + lane] = gB2[0];
      if (wave == 0) rslab[oh_rs_hb(NCH) + lane] = gHB[0][0];
      if (wave == 1) rslab[oh_rs_hb(NCH) + 16 + lane] = gHB[1][0];
    }
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) {
      float v = st[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      st[k] = v;
    }
    float* part = smem_f;  // [4 waves][NSTATP] over X
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NSTATP; ++k) part[wave * NSTATP + k] = st[k];
    }
    lds_barrier();
    if (tid < NSTATP)
      a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + tid] =
          ((part[tid] + part[NSTATP + tid]) + part[2 * NSTATP + tid]) + part[3 * NSTATP + tid];
  }
  PH_STAMP(a.prof, 13);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static size_t grad_split_oh_lds_bytes(int nch, bool box) {
  return (size_t)nch * (box ? 3 : 1) * PL_BYTES + 2 * PB_BYTES + 3 * DZL_PLANE + sizeof(float) * (HID + HID + 32 + 64 + 64 + 4 * 64 + 64);
}

// one-hot observations of up to five feature chunks and D <= 64 components, or Box observations of up to four chunks that the
// single-chunk / small-head kernels do not take; up to four action components and 32 logits
// (PH_GRAD_SPLIT_OH=0 switches the kernel off: the exact-f32 general kernel then takes the shape)
bool grad_split_oh_eligible(const NetDims& nd) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_GRAD_SPLIT_OH");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled || nd.gauss || nd.A < 1 || nd.A > 4 || nd.L > 16 * OH_LBMAX || nd.nchunk < 1 || grad_fast_eligible(nd)) return false;
  if (nd.obs_kind == PH_SPACE_BOX) return nd.nchunk <= 4;
  return nd.nchunk <= 5 && nd.D <= 64;
}
int grad_split_oh_slab_len(const NetDims& nd) { return 2 * oh_rs_net(nd.nchunk); }
int grad_split_oh_wimage_elems(const NetDims& nd) { return 2 * oh_nfrag(nd.nchunk) * 3 * WIMG_PLANE; }

template <int NCH, int LB, bool BOX>
static hipError_t launch_split_oh_inst(const GradArgs& a, int nwg, hipStream_t s) {
  const size_t lds = grad_split_oh_lds_bytes(NCH, BOX);
  static bool allowed_dev[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& allowed = allowed_dev[(dev >= 0 && dev < 64) ? dev : 0];
  if (!allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_split_oh_kernel<NCH, LB, BOX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  hipLaunchKernelGGL((ppo_grad_split_oh_kernel<NCH, LB, BOX>), dim3(nwg, 2), dim3(256), lds, s, a);
  return hipGetLastError();
}
template <int NCH>
static hipError_t launch_split_oh_nch(const GradArgs& a, int nwg, hipStream_t s) {
  const bool box = a.nd.obs_kind == PH_SPACE_BOX;
  if constexpr (NCH <= 4) {
    if (box) return a.nd.L <= 16 ? launch_split_oh_inst<NCH, 1, true>(a, nwg, s) : launch_split_oh_inst<NCH, 2, true>(a, nwg, s);
  } else {
    if (box) return hipErrorInvalidValue;
  }
  return a.nd.L <= 16 ? launch_split_oh_inst<NCH, 1, false>(a, nwg, s) : launch_split_oh_inst<NCH, 2, false>(a, nwg, s);
}
hipError_t launch_ppo_grad_split_oh(const GradArgs& a, int nwg, hipStream_t s) {
  switch (a.nd.nchunk) {
    case 1: return launch_split_oh_nch<1>(a, nwg, s);
    case 2: return launch_split_oh_nch<2>(a, nwg, s);
    case 3: return launch_split_oh_nch<3>(a, nwg, s);
    case 4: return launch_split_oh_nch<4>(a, nwg, s);
    case 5: return launch_split_oh_nch<5>(a, nwg, s);
    default: break;
  }
  return hipErrorInvalidValue;
}

// parameter -> weight-image elements (plane 0) of the fragments ppo_grad_split_oh_kernel loads: [P][2], -1 = none
void grad_weight_image_map_oh(const ph_layout& lay, int nch, int* map) {
  for (int i = 0; i < 2 * lay.P; ++i) map[i] = -1;
  const int nfrag = oh_nfrag(nch);
  auto elem = [&](int net, int frag, int lane, int e) { return ((net * nfrag + frag) * 3 + 0) * WIMG_PLANE + lane * 8 + e; };
  auto put = [&](int p, int idx) {
    if (map[2 * p] < 0) map[2 * p] = idx;
    else map[2 * p + 1] = idx;
  };
  for (int net = 0; net < 2; ++net) {
    const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2;
    // W1[f][n]: B of X W1 -- chunk f / 64, contraction index k = f % 64: fragment (wave n / 16, chunk, k / 32), lane (k / 8 % 4) 16 + n % 16, slot k % 8
    for (int f = 0; f < lay.F; ++f)
      for (int n = 0; n < HID; ++n) {
        const int k = f & 63;
        put(oW1 + f * HID + n, elem(net, oh_frag_w1(nch, n >> 4, f >> 6, k >> 5), ((k >> 3) & 3) * 16 + (n & 15), k & 7));
      }
    for (int k = 0; k < HID; ++k)
      for (int n = 0; n < HID; ++n) {
        // W2f: W2[k][n] as (operand row = unit n, contraction over inputs k); W2b: (column = input unit k, contraction over outputs n)
        put(oW2 + k * HID + n, elem(net, oh_frag_w2f(nch, n >> 4, k >> 5), ((k >> 3) & 3) * 16 + (n & 15), k & 7));
        put(oW2 + k * HID + n, elem(net, oh_frag_w2b(nch, k >> 4, n >> 5), ((n >> 3) & 3) * 16 + (k & 15), n & 7));
      }
    // head: W[u][l] (policy: act_W [64][L]; value: val_W [64], l = 0)
    const int Lh = net == 0 ? lay.L : 1, oHW = net == 0 ? lay.act_W : lay.val_W;
    for (int u = 0; u < HID; ++u)
      for (int l = 0; l < Lh; ++l) {
        // HZ: operand row = logit l (block l / 16), contraction over units u;  HD: operand row = unit u, contraction over logits l
        put(oHW + u * Lh + l, elem(net, oh_frag_hz(nch, l >> 4, u >> 5), ((u >> 3) & 3) * 16 + (l & 15), u & 7));
        put(oHW + u * Lh + l, elem(net, oh_frag_hd(nch, u >> 4), ((l >> 3) & 3) * 16 + (u & 15), l & 7));
      }
  }
}

// slab position -> parameter index (-1 = padding) for the kernel's accumulator order: [2 nets][oh_rs_net(nch)]
void grad_slab_map_split_oh(const ph_layout& lay, int nch, int* map) {
  const int rsn = oh_rs_net(nch);
  for (int net = 0; net < 2; ++net) {
    int* m = map + net * rsn;
    for (int i = 0; i < rsn; ++i) m[i] = -1;
    const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
    const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
    for (int s = 0; s < HID * HID; ++s) {
      // position ((wave 4 + blk) 64 + lane) 4 + r holds element (k = 16 blk + 4 (lane / 16) + r, col = 16 wave + lane % 16)
      const int wave = s >> 10, blk = (s >> 8) & 3, lane = (s >> 2) & 63, r = s & 3;
      const int k = 16 * blk + 4 * (lane >> 4) + r, col = 16 * wave + (lane & 15);
      m[s] = oW2 + k * HID + col;
      for (int ch = 0; ch < nch; ++ch)
        if (64 * ch + k < lay.F) m[oh_rs_w1(ch) + s] = oW1 + (64 * ch + k) * HID + col;
    }
    for (int i = 0; i < HID; ++i) {
      m[oh_rs_b1(nch) + i] = oB1 + i;
      m[oh_rs_b2(nch) + i] = oB2 + i;
    }
    // head: position ((lb 4 + wave) 64 + lane) 4 + r holds d W[unit 16 wave + 4 (lane / 16) + r][logit 16 lb + lane % 16]
    const int Lh = net == 0 ? lay.L : 1, oHW = net == 0 ? lay.act_W : lay.val_W, oHB = net == 0 ? lay.act_b : lay.val_b;
    for (int lb = 0; lb < OH_LBMAX; ++lb)
      for (int s = 0; s < 1024; ++s) {
        const int wave = s >> 8, lane = (s >> 2) & 63, r = s & 3;
        const int u = 16 * wave + 4 * (lane >> 4) + r, l = 16 * lb + (lane & 15);
        if (l < Lh) m[oh_rs_hw(nch) + lb * 1024 + s] = oHW + u * Lh + l;
      }
    for (int l = 0; l < Lh && l < 16 * OH_LBMAX; ++l) m[oh_rs_hb(nch) + l] = oHB + l;
  }
}

}  // namespace ph

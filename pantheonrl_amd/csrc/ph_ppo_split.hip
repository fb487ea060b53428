// ppo_grad_split_kernel: the PPO minibatch gradient of ppo_grad_fast_kernel (same shape class, same loss arithmetic, same
// slab / reduce / Adam pipeline; SB3 PPO.train() inner loop, pantheonrl/common/agents.py:155, arithmetic SURVEY.md A.3) with
// every 64x64x64 product on the bf16 matrix pipe at float32 accuracy.
//
// gfx950 has no fast f32 matrix instruction: v_mfma_f32_32x32x2_f32 runs at the f32 VECTOR rate (1/16 of bf16) and occupies the
// SIMD's vector lanes, so in ppo_grad_fast_kernel the MFMA, VALU and LDS issue cycles add up.  Here every f32
// operand x is carried as three bf16 planes  x = h + m + l  (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 3 x 8 = 24
// significand bits, the subtraction residues are exact) and a product is six v_mfma_f32_16x16x32_bf16 terms accumulated in
// f32: hh + hm + mh + mm + hl + lh (the dropped ml / lm / ll terms are <= 2^-24 relative, the size of one f32 rounding).
// Measured against float64 (scripts/ubench/split_bf16_probe.hip, 64x64x64, profiles/r03_split_bf16_probe.txt) the six-term
// product has a LOWER rms error than the exact-f32 fmaf chain (7.6e-7 vs 1.2e-6 on N(0,1) operands) because the matrix pipe sums
// the 32 products of an instruction before it rounds.  Six bf16 MFMAs cost 6/16 of the f32 MFMA they replace AND run on the
// matrix pipe beside the vector ALU instead of on it.
//
// Decomposition: grid (nWG, 2 nets), 64-row tiles, four waves; wave w owns hidden units / output columns [16w, 16w+16) of
// every product (N side) and all 64 rows or inputs (M side: four 16x16 blocks).  Its three weight operands (W1, W2 for the
// forward pass, W2 by rows for dH1) arrive as ready-made MFMA B fragments from an image of the parameters that the optimizer
// kernel keeps split (ph_split.h): no weight ever sits in LDS and none is split here.  W1's fragments stay in registers
// (24 VGPRs); W2's two sets are fetched (L2) inside the phase before the one that uses them, which keeps the kernel at 219
// VGPRs -- two of its waves and one wave of another learner's reduce / Adam kernel share a SIMD.
// Activations live in LDS only as bf16 planes, each in ONE layout; the product that contracts over
// the plane's row index reads it with ds_read_b64_tr_b16 (the hardware 4x4 transpose), the one that contracts over the
// contiguous index with ds_read_b128:
//     X    [row][feature]   16-byte copies of the plane image    S1 (plain)   dW1 (tr)          rows gathered from GradArgs.ximg
//     H1T  [unit][row]      S1 epilogue (4 rows = one b64)      S2 (tr)      dW2 (plain)       -> DZ1T [unit][row] (dW1, plain)
//     DZ2  [row][plane][unit]  head phase (8 units = one b128)  dH1 (plain)  dW2 (tr)          row r overlays H2's row r (f32, head only)
// The observation rows are split ONCE per train() call (obs_planes_kernel -> GradArgs.ximg: they do not change across the
// call's epochs and minibatches), and the five per-row scalars arrive packed in minibatch order (adv_stats_kernel ->
// GradArgs.rec_pi / rec_vf): a tile's gather is one 16-byte record per row and six 16-byte plane granules per lane -- no
// index arithmetic, no split, no 4-byte gathers in the 40 launches.
// Bias gradients are MFMAs with an all-ones A operand on the B fragments already in registers; layer 1's bias rides as
// feature 63 (FOLD) as in the f32 kernel.  3 x 24 KB of planes + 5.8 KB of head state = 79.7 KB -> two workgroups per CU.
#include "ph_split_tile.h"

// Settled by same-box A/B (CHANGELOG round 3 / 4), no longer switches:
//  * W1's fragments are fetched from the (L2-resident) weight image at the top of every tile, under the row commit, instead of
//    staying in registers from the prologue on: 24 registers less across the tile walk, which keeps room on every SIMD for a
//    wave of another learner's reduce / Adam kernel;
//  * the 32 KB of weight-gradient accumulators of a workgroup go to its slab as write-through (sc1) stores that leave while other
//    workgroups still compute (MI355X_MICROARCH.md, "publish-large") -- plain stores stay dirty in the XCD's L2 and the
//    end-of-kernel release writes 17.9 MB back behind the last workgroup;
//  * they leave as soon as they are final -- dW2 (and d b2) after the LAST tile's dW2 product, under that tile's dH1 / dZ1 / dW1
//    work, and dW1 block by block inside the last dW1 product (which therefore runs block-outer) -- instead of as one burst per
//    workgroup after the last barrier, when all 512 workgroups finish together;
//  * dH2 = dz act_W^T as packed FMAs over ADJACENT logits of one weight row.

namespace ph {

struct SplitRowMeta {
  int phys;
  float adv, old, act;
};

// A tile's observation rows as plane granules in registers: wave w stages rows 16w .. 16w+15 of the tile; lane = (row 8i + lane/8,
// logical granule lane % 8) for i = 0, 1 and every plane: 8 consecutive lanes read one 128-byte plane row of the image.
struct XRows {
  uint4 v[6];   // [i][plane]
  // physv: lane r < 16 holds the physical buffer row of tile row 16w + r (negative = dead row -> the image's zero row)
  __device__ __forceinline__ void issue(int physv, const uint4* ximg, int zero_row, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = 8 * i + (lane >> 3);
      int p = __builtin_amdgcn_ds_bpermute(4 * r, physv);
      p = p < 0 ? zero_row : p;
      const uint4* src = ximg + (size_t)p * XIMG_ROW_U4 + (lane & 7);
#pragma unroll
      for (int q = 0; q < 3; ++q) v[i * 3 + q] = ld_nt16(src + q * 8);
    }
  }
  __device__ __forceinline__ void commit(char* x, int wave, int lane) const {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 16 * wave + 8 * i + (lane >> 3);
      const int off = row * PL_ROW + (((lane & 7) ^ pl_swz(row)) << 4);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4*>(x + off + q * PL_BYTES) = v[i * 3 + q];
    }
  }
};

template <int NK, bool FOLD>
__global__ __launch_bounds__(256, 2) void ppo_grad_split_kernel(GradArgs a) {
  // the stop flag (target_kl early stop) is LOADED first and tested after the first tile's loads are in flight: a dependent
  // round trip at the top of the kernel delays everything behind it, and nothing before the test writes global memory
  const int stop_now = __builtin_nontemporal_load(a.stop_flag);
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  char* smem = reinterpret_cast<char*>(smem_f);
  constexpr int R = 64;
  // The dZ2 buffer is ROW-interleaved -- row r holds its three planes back to back (384 bytes) -- and H2 (f32, 256 bytes a row)
  // lives in the same row slots: a wave's dZ2 rows then overlay exactly the H2 rows that only this wave still reads, and the
  // head phase (forward/loss, d head weights, dZ2 commit) runs without a workgroup barrier inside.
  constexpr int DZ_ROW = 3 * PL_ROW, DZ_PL = PL_ROW;
  constexpr int XT = 0, H1T = PB_BYTES, DZ2 = 2 * PB_BYTES;   // byte offsets of the plane buffers
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  // H2[row][unit] (f32): 16-byte granule u/4 of row r at granule (u/4) ^ h2_swz(r) of the row slot -- the S2 epilogue's and the
  // head phase's 16-byte accesses are then (nearly) conflict-free (scripts/lds_swizzle_search.py); the swizzle depends on row
  // bits 0..1 only, which are compile-time constants where the d act_W loop walks rows
  auto h2_swz = [](int r) -> int { return (r & 1) | ((r & 2) ? 12 : 0); };
  auto h2_at = [&](int r, int gran) -> float* { return reinterpret_cast<float*>(smem + DZ2 + r * DZ_ROW + ((gran ^ h2_swz(r)) << 4)); };
  float* hw = smem_f + 3 * PB_BYTES / 4;        // policy: act_W as [64][8] skewed (head_row) | value: val_W [64]
  float* dzs = hw + HW_FLOATS;                  // policy: dL/dlogits [R][8] | value: dL/dv [R]
  float* b1s = dzs + R * 8;                     // [64]
  float* b2s = b1s + HID;                       // [64]
  float* hbs = b2s + HID;                       // act_b [8] | val_b
  float* radv = hbs + 16;                       // [R]
  float* rold = radv + R;                       // [R]
  float* ract = rold + R;                       // [R]
  int* rowphys = (int*)(ract + R);              // [R]

  const int net = blockIdx.y;
  const int oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
  const float inv_nb = 1.0f / (float)a.nb;
  const int nk = nd.L;

  const bool norm = net == 0 && a.norm_adv && a.nb > 1;
  // unconditional loads (a null table reads two parameters instead): no branch, no wait here -- first use is the first tile's T0
  const float* advp = a.advstats ? a.advstats : a.params;
  const float adv0 = __builtin_nontemporal_load(advp), adv1 = __builtin_nontemporal_load(advp + 1);
  const float adv_mean = norm ? adv0 : 0.f;
  const float adv_den = norm ? adv1 + 1e-8f : 1.f;

  // lane i < 16 of wave w serves row 16*w + i: its record {physical row, advantage | return, old log-prob | old value, action}
  const uint4* recs = net == 0 ? a.rec_pi : a.rec_vf;
  auto row_record = [&](int tile, int wave, int lane) -> SplitRowMeta {
    const int gi = tile * R + wave * 16 + lane;
    SplitRowMeta m;
    m.phys = -1;
    m.adv = m.old = m.act = 0.f;
    if (lane < 16 && gi < a.nb) {
      const uint4 r = ld_nt16(recs + gi);
      m.phys = (int)r.x;
      m.adv = __uint_as_float(r.y);
      m.old = __uint_as_float(r.z);
      m.act = __uint_as_float(r.w);
    }
    return m;
  };

  // ---- prologue: this wave's W1 fragments (from the image), head weights, the first tile's rows ----
  Frag3 W1f[2];   // (W2 for the forward pass is fetched inside S1 for S2; the third set, W2 by rows for dH1, is fetched from the image inside S6a: 24 registers less across the tile)
  XRows xt;
  SplitRowMeta meta;
  {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // issue order: the first tile's row records, then every weight of this wave (independent of them), then -- once the
    // records are back -- the observation planes of the rows they name
    meta = row_record(blockIdx.x, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    // this wave's weight fragments arrive already split (ph_split.h: ppo_adam_kernel keeps the image in step with params)
    float bias1 = 0.f, bias2 = 0.f, hv0 = 0.f, hv1 = 0.f, hb = 0.f;
    if (tid < HID) {
      bias1 = a.params[oB1 + tid];
      bias2 = a.params[oB2 + tid];
    }
    if (net == 0) {
      const int j0 = tid >> 3, k = tid & 7;
      if (k < nk) {
        hv0 = a.params[lay.act_W + j0 * nk + k];
        hv1 = a.params[lay.act_W + (j0 + 32) * nk + k];
      }
      if (tid < 8) hb = (tid < nk) ? a.params[lay.act_b + tid] : -3.0e38f;
    } else {
      if (tid < HID) hv0 = a.params[lay.val_W + tid];
      if (tid == 0) hb = a.params[lay.val_b];
    }
    __builtin_amdgcn_sched_barrier(0);
    PH_STAMP(a.prof, 8);
    PH_STAMP(a.prof, 9);
    xt.issue(meta.phys, a.ximg, a.ximg_zero_row, lane);
    __builtin_amdgcn_sched_barrier(0);
    PH_STAMP(a.prof, 10);
    if (stop_now) return;
    PH_STAMP(a.prof, 11);
    if (tid < HID) {
      b1s[tid] = bias1;
      b2s[tid] = bias2;
    }
    if (net == 0) {
      hw[head_row(tid >> 3) + (tid & 7)] = hv0;
      hw[head_row((tid >> 3) + 32) + (tid & 7)] = hv1;
      if (tid < 8) hbs[tid] = hb;
    } else {
      if (tid < HID) hw[tid] = hv0;
      if (tid == 0) hbs[0] = hb;
    }
  }

  f32x4 gW1[4], gW2[4];
  f32x4 gB1 = {0.f, 0.f, 0.f, 0.f}, gB2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    gW1[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    gW2[b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float gh0 = 0.f;              // value: d val_W[lane] partial over this wave's rows
  float ghr[NK];                // policy: d act_W[lane][k] partial over this wave's rows
#pragma unroll
  for (int k = 0; k < NK; ++k) ghr[k] = 0.f;
  float ghb = 0.f;
  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;

  float* const rslab = a.slabs + ((size_t)blockIdx.x * 2 + net) * RS_NET;
  bool slabs_out = false;   // dW1 / dW2 / d b2 already stored by the last tile
  bool first = true;
  // every workgroup owns at least one tile (launch_ppo_grad_split refuses a grid wider than the tile count): a loop entered
  // unconditionally -- with a guarded loop the accumulators are zeroed for the skip path and COPIED into the loop's registers
  int tile = blockIdx.x;
  do {
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63, wave = tid >> 6;
    const SplitSel ssel = split_sel();   // the split's two selector words (ph_split_tile.h), rebuilt per tile
    const int j = lane & 15, kg = lane >> 4;
    const bool has_next = tile + (int)gridDim.x < a.ntiles;
    const int unit = 16 * wave + j;   // this lane's column of every 16x16 result

    {   // W1 by (feature, unit): B of S1, six L2-resident 16-byte loads per lane, in flight under the row commit
      const uint4* img = reinterpret_cast<const uint4*>(a.wimage) + (size_t)((net * 4 + wave) * 18) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) W1f[c].p[p] = __builtin_bit_cast(bf16x8, img[(c * 3 + p) * 64]);
    }
    // ---- T0: this tile's rows land in LDS as planes ----
    if (lane < 16) {
      const int row = wave * 16 + lane;
      rowphys[row] = meta.phys;
      radv[row] = (norm && meta.phys >= 0) ? (meta.adv - adv_mean) / adv_den : meta.adv;
      rold[row] = meta.old;
      ract[row] = meta.act;
    }
    if (first) PH_STAMP(a.prof, 14);
    xt.commit(smem + XT, wave, lane);
    if (first) PH_STAMP(a.prof, 15);
    lds_barrier();
    if (first) PH_STAMP(a.prof, 1);

    // per-lane operand offsets of this tile walk.  pl_swz is linear, so the (chunk, half, block) part of an address is a
    // compile-time displacement plus an XOR of the granule index with 2 * (block ^ half): four per-lane bases serve the sixteen
    // transposing-read addresses of a product, four more the C-layout stores, two the plain reads -- no address arithmetic
    // inside the products
    const int pb0 = plain_base(j, kg, 0), pb1 = plain_base(j, kg, 1);
    int trb[4], csb[4];
    {
      const int a0 = 8 * kg + (j >> 2);
      const int g0 = ((j & 3) >> 1) ^ pl_swz(a0), row0 = a0 * PL_ROW + 8 * (j & 1);
      const int g1 = (kg >> 1) ^ pl_swz(unit), row1 = unit * PL_ROW + 8 * (kg & 1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        trb[k] = row0 + ((g0 ^ (2 * k)) << 4);
        csb[k] = row1 + ((g1 ^ (2 * k)) << 4);
      }
    }
    // transposing fragment of plane rows 32c + 8kg .. +7 (contraction index), columns 16*mblk .. +15 (lane index)
    auto ld_trf = [&](int buf, int c, int mblk) -> Frag3 {
      return ld_tr(smem, trb[mblk], trb[mblk ^ 1], buf + 32 * c * PL_ROW, buf + (32 * c + 4) * PL_ROW);
    };

    f32x4 d1[4];   // 1 - H1^2 of this lane's 16 elements (rows 16*blk + 4*kg + r, column `unit`): kept for dZ1
    // ---- S1: H1 = tanh(X W1 (+ b1)) -> H1T planes.  Block-outer: the epilogue of block b (VALU) runs under the MFMAs of b + 1 ----
    Frag3 W2f[2];   // W2 by (input, unit): B of S2, fetched under S1's products (L2-resident image)
    {
      const uint4* img = reinterpret_cast<const uint4*>(a.wimage) + (size_t)((net * 4 + wave) * 18 + 6) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) W2f[c].p[p] = __builtin_bit_cast(bf16x8, img[(c * 3 + p) * 64]);
    }
    {
      const float bb = FOLD ? 0.f : b1s[unit];
      // the next block's fragments are read BEFORE this block's epilogue stores: the compiler cannot move an LDS read across an
      // LDS store it cannot prove disjoint, so in plain loop order every block's reads waited behind the previous block's
      // stores (a deeper pipeline -- MFMAs of block b + 1 issued before the epilogue of block b, reads two blocks ahead -- measured
      // slower: 26.05 vs 25.65 us, at 231 instead of 219 VGPRs)
      Frag3 xa = ld_plain(smem, pb0, XT), xb = ld_plain(smem, pb1, XT);   // A: X rows 16b + i, features 32c + 8kg ..
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma6(xa, W1f[0], acc);
        acc = mma6(xb, W1f[1], acc);
        if (b < 3) {
          xa = ld_plain(smem, pb0, XT + (b + 1) * 16 * PL_ROW);
          xb = ld_plain(smem, pb1, XT + (b + 1) * 16 * PL_ROW);
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fast_tanh(FOLD ? acc[r] : acc[r] + bb);
          d1[b][r] = 1.0f - v[r] * v[r];
        }
        // the split consumes v in place (v_dot2c accumulates into its input): d1 first, or every v is copied for it
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(d1[b][0]), "v"(d1[b][1]), "v"(d1[b][2]), "v"(d1[b][3]));
        bf16x4 p[3];
        split4(ssel, v, p);
        st_planes4(smem, H1T + csb[b], p);
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 2);

    // ---- S2: H2 = tanh(H1 W2 + b2) -> H2 (f32).  Operand roles swapped (A = W2 fragments): the result tile is H2^T, i.e.
    //      lane = row 16*blk + j, registers = units 16*wave + 4*kg + r -> one 16-byte store per block ----
    SplitRowMeta meta_next = meta;
    if (has_next) meta_next = row_record(tile + gridDim.x, wave, lane);   // next tile's rows, committed at its T0
    {
      const float4 bb = *reinterpret_cast<const float4*>(b2s + 16 * wave + 4 * kg);
      Frag3 xa = ld_trf(H1T, 0, 0), xb = ld_trf(H1T, 1, 0);   // read-ahead as in S1
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma6(W2f[0], xa, acc);
        acc = mma6(W2f[1], xb, acc);
        if (b < 3) {
          xa = ld_trf(H1T, 0, b + 1);
          xb = ld_trf(H1T, 1, b + 1);
        }
        *reinterpret_cast<float4*>(h2_at(16 * b + j, 4 * wave + kg)) =
            make_float4(fast_tanh(acc[0] + bb.x), fast_tanh(acc[1] + bb.y), fast_tanh(acc[2] + bb.z), fast_tanh(acc[3] + bb.w));
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 3);

    // ---- SH-a: head forward, loss, dL/dhead; dZ2 = dH2 * (1 - H2^2) stays in registers; four lanes per row ----
    float dzv[16];
    const int hr = tid >> 2, hq = tid & 3;
    {
      const int r = hr, q = hq;
      const bool valid = rowphys[r] >= 0;
      float h[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // head_unit(q, m) = 8q + (m & 7) + 32 (m >> 3): two runs of eight units
        const float4 v = *reinterpret_cast<const float4*>(h2_at(r, 2 * q + (g & 1) + 8 * (g >> 1)));
        h[4 * g] = v.x; h[4 * g + 1] = v.y; h[4 * g + 2] = v.z; h[4 * g + 3] = v.w;
      }
      if (net == 0) {
        float z[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) z[k] = 0.f;
        for_head_rows<NK>(hw, q, [&](int m, const float4& w0, const float4& w1) {
          const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int k = 0; k < NK; ++k) z[k] = __builtin_fmaf(h[m], wk[k], z[k]);
        });
        float pr[NK];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          z[k] = quad_sum(z[k]) + hbs[k];
          mx = fmaxf(mx, z[k]);
        }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          pr[k] = fast_exp(z[k] - mx);
          se += pr[k];
        }
        const float lse = mx + fast_log(se), inv = __builtin_amdgcn_rcpf(se);
        int act = (int)ract[r];
        act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
        float ent = 0.f, zact = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          pr[k] *= inv;
          ent -= pr[k] * (z[k] - lse);
          zact = (k == act) ? z[k] : zact;
        }
        const float logp = zact - lse;
        const float adv = radv[r];
        const float lr = logp - rold[r];
        const float ratio = fast_exp(lr);
        const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float pl1 = adv * ratio, pl2 = adv * rc;
        const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
        const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
        const float live = valid ? 1.f : 0.f;
        const float g_lp = -inv_nb * adv * ratio * gate * live;
        const float g_en = -a.ent_coef * inv_nb * live;
        if (valid && q == 0) {
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
        }
        float dz[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dz[k] = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          const float dlogp = ((k == act) ? 1.f : 0.f) - pr[k];
          const float dent = -pr[k] * ((z[k] - lse) + ent);
          dz[k] = g_lp * dlogp + g_en * dent;
        }
        if (q == 0) {
          float4* o = reinterpret_cast<float4*>(dzs + r * 8);
          o[0] = make_float4(dz[0], dz[1], dz[2], dz[3]);
          if constexpr (NK > 4) o[1] = make_float4(dz[4], dz[5], dz[6], dz[7]);
        }
        // dH2[m] = sum_k dz[k] act_W[m][k] as packed FMAs over ADJACENT logits of one weight row (the register pairs a 16-byte
        // LDS read delivers): left to itself the vectoriser pairs two ROWS instead and pays a v_mov per packed operand
        typedef float hp2 __attribute__((ext_vector_type(2)));
        hp2 dzp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dzp[k] = (hp2){dz[2 * k], dz[2 * k + 1]};      // dz[k >= NK] = 0
        for_head_rows<NK>(hw, q, [&](int m, const float4& w0, const float4& w1) {
          hp2 acc = (hp2){w0.x, w0.y} * dzp[0];
          if constexpr (NK > 2) acc = __builtin_elementwise_fma((hp2){w0.z, w0.w}, dzp[1], acc);
          if constexpr (NK > 4) acc = __builtin_elementwise_fma((hp2){w1.x, w1.y}, dzp[2], acc);
          if constexpr (NK > 6) acc = __builtin_elementwise_fma((hp2){w1.z, w1.w}, dzp[3], acc);
          // the horizontal add as ONE scalar v_add_f32: left to the vectoriser, two rows' sums become a packed add fed by three
          // v_movs that transpose the pairs (2 instructions per row instead of 1)
          float hsum;
          asm("v_add_f32 %0, %1, %2" : "=v"(hsum) : "v"(acc.x), "v"(acc.y));
          dzv[m] = hsum * (1.0f - h[m] * h[m]);
        });
      } else {
        float wv[16];
        float v = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          wv[m] = hw[head_unit(q, m)];
          v = __builtin_fmaf(h[m], wv[m], v);
        }
        v = quad_sum(v) + hbs[0];
        const float retn = radv[r], oldv = rold[r];
        float vp = v, pass = 1.f;
        if (a.clip_vf >= 0.f) {
          const float dlt = v - oldv;
          pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
          vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
        }
        const float err = vp - retn;
        const float dv = valid ? a.vf_coef * 2.0f * err * inv_nb * pass : 0.f;
        if (valid && q == 0) st[1] += err * err;
        if (q == 0) dzs[r] = dv;
#pragma unroll
        for (int m = 0; m < 16; ++m) dzv[m] = dv * wv[m] * (1.0f - h[m] * h[m]);
      }
    }
    wave_lds_sync();   // dzs rows of this wave are written and read by this wave only
    if (first) PH_STAMP(a.prof, 4);

    // ---- SH-b: d head weights / d head bias over this wave's 16 rows (H2 rows of this wave, still in LDS) ----
    if (net == 0) {
      // unit `lane` of row r sits at dword ((lane >> 2) ^ h2_swz(r)) * 4 + (lane & 3) of the row slot; h2_swz(r) only depends on r & 3
      int lofs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) lofs[i] = (((lane >> 2) ^ h2_swz(i)) << 2) | (lane & 3);
      const float* hp = reinterpret_cast<const float*>(smem + DZ2 + wave * 16 * DZ_ROW);
      const float* dp = dzs + wave * 16 * 8;
#pragma unroll 1
      for (int r0 = 0; r0 < 16; r0 += 4, hp += 4 * (DZ_ROW / 4), dp += 4 * 8) {
        float hv[4];
        float4 da[4], db[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          hv[i] = hp[i * (DZ_ROW / 4) + lofs[i]];
          da[i] = *reinterpret_cast<const float4*>(dp + i * 8);
          if constexpr (NK > 4) db[i] = *reinterpret_cast<const float4*>(dp + i * 8 + 4);
          else db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float dk[8] = {da[i].x, da[i].y, da[i].z, da[i].w, db[i].x, db[i].y, db[i].z, db[i].w};
#pragma unroll
          for (int k = 0; k < NK; ++k) ghr[k] = __builtin_fmaf(hv[i], dk[k], ghr[k]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lane < NK) ghb += lds_sum16(dzs + wave * 16 * 8 + lane, 8);
    } else {
      float hv[16], dv[16];
      // (wave * 16 + i) * DZ_ROW written as a per-wave base plus a compile-time row displacement: from the sum the compiler makes an
      // OR (wave * 16 has no low bits), multiplies that, and every row's address becomes arithmetic instead of an offset field
      const char* hrow = smem + DZ2 + wave * (16 * DZ_ROW);
      const float* drow = dzs + wave * 16;
      int col4[4];   // h2_swz(i) depends on i & 3 only: four per-lane column offsets (bytes)
#pragma unroll
      for (int i = 0; i < 4; ++i) col4[i] = ((((lane >> 2) ^ h2_swz(i)) << 2) | (lane & 3)) * 4;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        hv[i] = *reinterpret_cast<const float*>(hrow + i * DZ_ROW + col4[i & 3]);
        dv[i] = drow[i];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        gh0 = __builtin_fmaf(hv[i], dv[i], gh0);
        ghb += dv[i];
      }
    }
    wave_lds_sync();   // this wave's H2 rows are consumed; its dZ2 rows go on top of them

    // ---- SH-c: dZ2 -> planes, row-interleaved [row][plane][unit], over this wave's H2 rows ----
    {
      const int sw = pl_swz(hr);
#pragma unroll
      for (int g = 0; g < 2; ++g) {   // units 8q .. 8q+7 (granule q) and 32 + 8q .. (granule 4 + q)
        Frag3 f;
        split8(ssel, dzv + 8 * g, f);
        st_planes8<DZ_PL>(smem, DZ2 + hr * DZ_ROW + (((4 * g + hq) ^ sw) << 4), f);
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 5);

    // ---- S6a: dW2 += H1^T dZ2, d b2 ; dH1 = dZ2 W2^T ----
    f32x4 dh1[4];
    {
      // B of dW2: dZ2 columns 16w .. +15 (this wave's units), contraction over rows 32c + 8kg ..: transposing reads of the
      // row-interleaved buffer (row stride DZ_ROW, planes DZ_PL apart)
      const int a0 = 8 * kg + (j >> 2), g0 = ((j & 3) >> 1) ^ pl_swz(a0), row0 = a0 * DZ_ROW + 8 * (j & 1);
      const int tlo = row0 + ((g0 ^ (2 * wave)) << 4), thi = row0 + ((g0 ^ (2 * (wave ^ 1))) << 4);
      const int db0 = j * DZ_ROW + ((kg ^ pl_swz(j)) << 4), db1 = j * DZ_ROW + (((4 + kg) ^ pl_swz(j)) << 4);
      // W2 by rows (B of dH1): six L2-resident 16-byte loads per lane, issued here and consumed after the dW2 product
      Frag3 W2b[2];
      {
        const uint4* img = reinterpret_cast<const uint4*>(a.wimage) + (size_t)((net * 4 + wave) * 18 + 12) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) W2b[c].p[p] = __builtin_bit_cast(bf16x8, img[(c * 3 + p) * 64]);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const Frag3 dz = ld_tr<DZ_PL>(smem, tlo, thi, DZ2 + 32 * c * DZ_ROW, DZ2 + (32 * c + 4) * DZ_ROW);
        gB2 = mma_ones(dz, gB2);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const Frag3 h1 = ld_plain(smem, c == 0 ? pb0 : pb1, H1T + b * 16 * PL_ROW);   // A: H1T rows (input units) 16b + i
          gW2[b] = mma6(h1, dz, gW2[b]);
        }
      }
      if (!has_next) {   // dW2 and d b2 are final: their 16 KB leave under the rest of this tile
        float* const w2o = rslab + RS_W2 + (wave * 4 * 64 + lane) * 4;   // block b: + 1 KB, as an immediate
        st_slab16<0>(w2o, gW2[0]);
        st_slab16<1024>(w2o, gW2[1]);
        st_slab16<2048>(w2o, gW2[2]);
        st_slab16<3072>(w2o, gW2[3]);
        if (lane < 16) rslab[RS_B2 + 16 * wave + lane] = gB2[0];   // every row of the ones product is the column sum
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)   // A: dZ2 rows 16b + i, units 32c + 8kg ..
          acc = mma6(ld_plain<DZ_PL>(smem, c == 0 ? db0 : db1, DZ2 + b * 16 * DZ_ROW), W2b[c], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) dh1[b][r] = acc[r] * d1[b][r];   // dZ1 (registers; stored after the barrier)
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

    // ---- S7: dW1 += X^T dZ1 (d b1 rides as feature 63, or as a ones product).  The next tile's rows are gathered underneath. ----
    meta = meta_next;
    if (has_next) xt.issue(meta.phys, a.ximg, a.ximg_zero_row, lane);
    {
      // block-outer: dW1 block b (features 16b .. +15) is final after its two chunks, and on the last tile it leaves at once
      const Frag3 dz0 = ld_plain(smem, pb0, H1T + wave * 16 * PL_ROW);   // B: DZ1T row (unit) 16w + j, rows 8kg .. / 32 + 8kg ..
      const Frag3 dz1 = ld_plain(smem, pb1, H1T + wave * 16 * PL_ROW);
      if constexpr (!FOLD) {
        gB1 = mma_ones(dz0, gB1);
        gB1 = mma_ones(dz1, gB1);
      }
      float* const w1o = rslab + RS_W1 + (wave * 4 * 64 + lane) * 4;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        gW1[b] = mma6(ld_trf(XT, 0, b), dz0, gW1[b]);   // A: features 16b + i (lane), tile rows 8kg .. (contraction): transposing reads of X
        gW1[b] = mma6(ld_trf(XT, 1, b), dz1, gW1[b]);
        if (!has_next) {
          if (b == 0) st_slab16<0>(w1o, gW1[0]);
          else if (b == 1) st_slab16<1024>(w1o, gW1[1]);
          else if (b == 2) st_slab16<2048>(w1o, gW1[2]);
          else st_slab16<3072>(w1o, gW1[3]);
        }
      }
      if (!has_next) {
        if constexpr (!FOLD)
          if (lane < 16) rslab[RS_B1 + 16 * wave + lane] = gB1[0];
        slabs_out = true;
      }
    }
    lds_barrier();  // XT / H1T / row scalars are free for the next tile
    if (first) PH_STAMP(a.prof, 7);
    tile += gridDim.x;
    first = false;
  } while (tile < a.ntiles);
  PH_STAMP(a.prof, 12);

  // ---- epilogue: accumulators -> slab (once), cross-wave sums in a fixed order ----
  {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!slabs_out) {
      float* const w2o = rslab + RS_W2 + (wave * 4 * 64 + lane) * 4;
      float* const w1o = rslab + RS_W1 + (wave * 4 * 64 + lane) * 4;
      st_slab16<0>(w2o, gW2[0]); st_slab16<1024>(w2o, gW2[1]); st_slab16<2048>(w2o, gW2[2]); st_slab16<3072>(w2o, gW2[3]);
      st_slab16<0>(w1o, gW1[0]); st_slab16<1024>(w1o, gW1[1]); st_slab16<2048>(w1o, gW1[2]); st_slab16<3072>(w1o, gW1[3]);
      if (lane < 16) {   // every row of the ones products is the column sum
        rslab[RS_B2 + 16 * wave + lane] = gB2[0];
        if constexpr (!FOLD) rslab[RS_B1 + 16 * wave + lane] = gB1[0];
      }
    }
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) {
      float v = st[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      st[k] = v;
    }
    float* part = smem_f;  // [3 + NK][4 waves][64] over XT
    part[(0 * 4 + wave) * 64 + lane] = gh0;
    part[(1 * 4 + wave) * 64 + lane] = ghb;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NSTATP; ++k) part[(2 * 4 + wave) * 64 + k] = st[k];
    }
    if (net == 0) {
#pragma unroll
      for (int k = 0; k < NK; ++k) part[((3 + k) * 4 + wave) * 64 + lane] = ghr[k];
    }
    lds_barrier();
    auto wsum = [&](int which, int idx) {
      return ((part[(which * 4 + 0) * 64 + idx] + part[(which * 4 + 1) * 64 + idx]) + part[(which * 4 + 2) * 64 + idx]) +
             part[(which * 4 + 3) * 64 + idx];
    };
    if (net == 1 && tid < HID) rslab[RS_HW + tid] = wsum(0, tid);
    if (net == 0) {
#pragma unroll
      for (int k0 = 0; k0 < NK; k0 += 4) {
        const int k = k0 + (tid >> 6), jj = tid & 63;
        if (k < NK) rslab[RS_HW + jj * 8 + k] = wsum(3 + k, jj);
      }
    }
    if (net == 0 && tid < 8) rslab[RS_HB + tid] = wsum(1, tid);
    if (net == 1 && tid == 0) rslab[RS_HB] = wsum(1, 0);
    if (tid < NSTATP) a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + tid] = wsum(2, tid);
  }
  PH_STAMP(a.prof, 13);
}

#ifndef PH_SPLIT_LDS_PAD
#define PH_SPLIT_LDS_PAD 0   // occupancy experiment only (scripts/build_variants.sh): > 640 leaves ONE workgroup per CU
#endif
static size_t grad_split_lds_bytes() {
  return (size_t)3 * PB_BYTES + sizeof(float) * (size_t)(HW_FLOATS + 64 * 8 + 2 * HID + 16 + 3 * 64 + 64) + PH_SPLIT_LDS_PAD;
}

// the split kernel takes Box observations of the fast kernel's shape class (PH_GRAD_SPLIT=0 switches it off)
bool grad_split_eligible(const NetDims& nd) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_GRAD_SPLIT");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  return enabled && grad_fast_eligible(nd) && nd.obs_kind == PH_SPACE_BOX;
}

template <int NK, bool FOLD>
static hipError_t launch_split_inst(const GradArgs& a, int nwg, hipStream_t s) {
  const size_t lds = grad_split_lds_bytes();
  static bool allowed_dev[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& allowed = allowed_dev[(dev >= 0 && dev < 64) ? dev : 0];
  if (!allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_split_kernel<NK, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  hipLaunchKernelGGL((ppo_grad_split_kernel<NK, FOLD>), dim3(nwg, 2), dim3(256), lds, s, a);
  return hipGetLastError();
}
template <int NK>
static hipError_t launch_split_nk(const GradArgs& a, int nwg, hipStream_t s) {
  return grad_fast_fold(a.nd) ? launch_split_inst<NK, true>(a, nwg, s) : launch_split_inst<NK, false>(a, nwg, s);
}

hipError_t launch_ppo_grad_split(const GradArgs& a, int nwg, hipStream_t s) {
  if (nwg < 1 || nwg > a.ntiles) return hipErrorInvalidValue;   // the kernel's tile walk is entered unconditionally (grad_plan: nwg <= ntiles)
  switch (a.nd.L) {
    case 1: return launch_split_nk<1>(a, nwg, s);
    case 2: return launch_split_nk<2>(a, nwg, s);
    case 3: return launch_split_nk<3>(a, nwg, s);
    case 4: return launch_split_nk<4>(a, nwg, s);
    case 5: return launch_split_nk<5>(a, nwg, s);
    case 6: return launch_split_nk<6>(a, nwg, s);
    case 7: return launch_split_nk<7>(a, nwg, s);
    default: break;
  }
  return launch_split_nk<8>(a, nwg, s);
}

// parameter -> weight-image elements (plane 0) of the fragments ppo_grad_split_kernel loads: [P][2], -1 = none
void grad_weight_image_map(const ph_layout& lay, bool fold, int* map) {
  for (int i = 0; i < 2 * lay.P; ++i) map[i] = -1;
  auto elem = [](int net, int wave, int set, int c, int lane, int e) {
    return (((((net * 4 + wave) * 3 + set) * 2 + c) * 3 + 0) * 64 + lane) * 8 + e;
  };
  auto put = [&](int p, int idx) {
    if (map[2 * p] < 0) map[2 * p] = idx;
    else map[2 * p + 1] = idx;
  };
  for (int net = 0; net < 2; ++net) {
    const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
    const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2;
    for (int k = 0; k < HID; ++k) {
      for (int n = 0; n < HID; ++n) {
        // B fragment element of (contraction index k, column n): wave n/16, lane (k/8 % 4) * 16 + n % 16, chunk k/32, slot k % 8
        const int by_col = elem(net, n >> 4, 0, k >> 5, ((k >> 3) & 3) * 16 + (n & 15), k & 7);
        if (k < lay.F) put(oW1 + k * HID + n, by_col);
        else if (fold && k == HID - 1) put(oB1 + n, by_col);                               // W1f: feature 63 is b1
        put(oW2 + k * HID + n, by_col + (elem(0, 0, 1, 0, 0, 0) - elem(0, 0, 0, 0, 0, 0))); // W2f: W2[k][n], contraction over inputs k
        // W2b: W2[row k][out n] as (column = input unit k, contraction over outputs n)
        put(oW2 + k * HID + n, elem(net, k >> 4, 2, n >> 5, ((n >> 3) & 3) * 16 + (k & 15), n & 7));
      }
    }
  }
}

// slab position -> parameter index for the split kernel's accumulator order: position ((wave*4 + blk)*64 + lane)*4 + r holds
// element (k = 16*blk + 4*(lane>>4) + r, col = 16*wave + (lane & 15)) of dW2 / dW1
void grad_slab_map_split(const ph_layout& lay, int* map, bool fold) {
  const int F = lay.F, L = lay.L;
  for (int net = 0; net < 2; ++net) {
    int* m = map + net * RS_NET;
    for (int i = 0; i < RS_NET; ++i) m[i] = -1;
    const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
    const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
    for (int s = 0; s < HID * HID; ++s) {
      const int wave = s >> 10, blk = (s >> 8) & 3, lane = (s >> 2) & 63, r = s & 3;
      const int k = 16 * blk + 4 * (lane >> 4) + r, col = 16 * wave + (lane & 15);
      m[RS_W2 + s] = oW2 + k * HID + col;
      if (k < F) m[RS_W1 + s] = oW1 + k * HID + col;
      else if (fold && k == HID - 1) m[RS_W1 + s] = oB1 + col;
    }
    for (int i = 0; i < HID; ++i) {
      if (!fold) m[RS_B1 + i] = oB1 + i;
      m[RS_B2 + i] = oB2 + i;
    }
    if (net == 0) {
      for (int j = 0; j < HID; ++j)
        for (int k = 0; k < L && k < 8; ++k) m[RS_HW + j * 8 + k] = lay.act_W + j * L + k;
      for (int k = 0; k < L && k < 8; ++k) m[RS_HB + k] = lay.act_b + k;
    } else {
      for (int j = 0; j < HID; ++j) m[RS_HW + j] = lay.val_W + j;
      m[RS_HB] = lay.val_b;
    }
  }
}

}  // namespace ph

// ppo_grad_split8_kernel: ppo_grad_split_kernel (ph_ppo_split.hip -- same arithmetic, same LDS plane layouts, same weight image,
// same slab order; SB3 PPO.train() inner loop, pantheonrl/common/agents.py:155, arithmetic pantheonrl/algos/adap/adap_learn.py:253-344)
// with EIGHT waves per 64-row tile instead of four: four waves per SIMD at two workgroups per CU.
//
// Why (round 6, profiles/r06_n_onewg_ab.txt): with ONE four-wave workgroup per CU the launch takes 33 us, with two 24.7 -- a wave of
// the four-wave kernel runs its matrix, vector and LDS work almost end to end (in-order issue: 34 k cycles alone for 28 k of pipe
// time), and the second wave of a SIMD hides only a third of that.  A tile's 72 KB of planes bound the ROWS in flight per CU (128),
// not the waves: here wave (cw, hf) owns output columns 16 cw .. +15 of every product, like the four-wave kernel's wave cw, but only
// the 16-row (or 16-input) blocks 2 hf and 2 hf + 1 of it -- half the matrix instructions, half the epilogue work, half the
// accumulators per wave, and four waves per SIMD to put beside each other.  The head phase runs eight lanes per row (eight hidden
// units each).  Costs: the two halves of a column block fetch the same weight fragments (the image's L2 traffic doubles), one more
// workgroup barrier per tile (DZ1T is written by both halves before either reads its unit rows), per-wave fixed work twice.
// The slab positions, the statistics record and every reduction order that reaches a result bit -- except the cross-wave sums of
// the head-weight gradients, eight partials instead of four -- are the four-wave kernel's.
#include "ph_split_tile.h"

#ifndef PH_SPLIT8_HEAD_G
#define PH_SPLIT8_HEAD_G 2
#endif

namespace ph {

namespace {

struct RowMeta8 {
  int phys;
  float adv, old, act;
};

// A tile's observation rows as plane granules in registers: wave W stages rows 8W .. 8W+7; lane = (row 8W + lane/8, logical
// granule lane % 8) for every plane: 8 consecutive lanes read one 128-byte plane row of the image.
struct XRows8 {
  uint4 v[3];
  // physv: lane r < 8 holds the physical buffer row of tile row 8W + r (negative = dead row -> the image's zero row)
  __device__ __forceinline__ void issue(int physv, const uint4* ximg, int zero_row, int lane) {
    int p = __builtin_amdgcn_ds_bpermute(4 * (lane >> 3), physv);
    p = p < 0 ? zero_row : p;
    const uint4* src = ximg + (size_t)p * XIMG_ROW_U4 + (lane & 7);
#pragma unroll
    for (int q = 0; q < 3; ++q) v[q] = ld_nt16(src + q * 8);
  }
  __device__ __forceinline__ void commit(char* x, int wave, int lane) const {
    const int row = 8 * wave + (lane >> 3);
    const int off = row * PL_ROW + (((lane & 7) ^ pl_swz(row)) << 4);
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<uint4*>(x + off + q * PL_BYTES) = v[q];
  }
};

// sum over the eight lanes of a row group; every lane ends with the bitwise-identical result (quad sums are symmetric, the
// half-row mirror pairs lane i with lane 7 - i of the other quad)
__device__ __forceinline__ float oct_sum(float v) {
  v = quad_sum(v);
  v += dpp_quad<0x141>(v);  // row_half_mirror
  return v;
}

// sum of 8 LDS values p[i*stride], all reads issued before the adds (fixed tree order)
__device__ __forceinline__ float lds_sum8(const float* p, int stride) {
  float t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = p[i * stride];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int w = 4; w > 0; w >>= 1) {
#pragma unroll
    for (int i = 0; i < w; ++i) t[i] += t[i + w];
  }
  return t[0];
}

// Calls f(m, w0, w1) for the 8 head-weight rows of lane (q, hh) -- units 8 q + 32 hh + m, at hw[72 q + 288 hh + 8 m] (ph_head.h:
// head_row) --, two rows per group, the next group's reads issued before the current group's arithmetic.
template <int NK = 8, class F>
__device__ __forceinline__ void for_head_rows8(const float* hwq, F&& f) {
  constexpr int G = PH_SPLIT8_HEAD_G;   // rows per group; no read-ahead: four waves per SIMD cover the LDS round trip, registers do not
#pragma unroll
  for (int g = 0; g < 8 / G; ++g) {
    float4 wa[G], wb[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
      const float4* w = reinterpret_cast<const float4*>(hwq + 8 * (G * g + i));
      wa[i] = w[0];
      if constexpr (NK > 6) wb[i] = w[1];
      else if constexpr (NK > 4) {
        const float2 t = *reinterpret_cast<const float2*>(hwq + 8 * (G * g + i) + 4);
        wb[i] = make_float4(t.x, t.y, 0.f, 0.f);
      } else wb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < G; ++i) f(G * g + i, wa[i], wb[i]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace

#ifndef PH_SPLIT8_WAVES_PER_EU
#define PH_SPLIT8_WAVES_PER_EU 4
#endif

template <int NK, bool FOLD>
__global__ __launch_bounds__(512, PH_SPLIT8_WAVES_PER_EU) void ppo_grad_split8_kernel(GradArgs a) {
  const int stop_now = __builtin_nontemporal_load(a.stop_flag);
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  char* smem = reinterpret_cast<char*>(smem_f);
  constexpr int R = 64;
  constexpr int DZ_ROW = 3 * PL_ROW, DZ_PL = PL_ROW;
  constexpr int XT = 0, H1T = PB_BYTES, DZ2 = 2 * PB_BYTES;   // byte offsets of the plane buffers
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
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
  const float* advp = a.advstats ? a.advstats : a.params;
  const float adv0 = __builtin_nontemporal_load(advp), adv1 = __builtin_nontemporal_load(advp + 1);
  const float adv_mean = norm ? adv0 : 0.f;
  const float adv_den = norm ? adv1 + 1e-8f : 1.f;

  // lane i < 8 of wave W serves row 8 W + i: its record {physical row, advantage | return, old log-prob | old value, action}
  const uint4* recs = net == 0 ? a.rec_pi : a.rec_vf;
  auto row_record = [&](int tile, int wave, int lane) -> RowMeta8 {
    const int gi = tile * R + wave * 8 + lane;
    RowMeta8 m;
    m.phys = -1;
    m.adv = m.old = m.act = 0.f;
    if (lane < 8 && gi < a.nb) {
      const uint4 r = ld_nt16(recs + gi);
      m.phys = (int)r.x;
      m.adv = __uint_as_float(r.y);
      m.old = __uint_as_float(r.z);
      m.act = __uint_as_float(r.w);
    }
    return m;
  };

  // ---- prologue: head weights, the first tile's rows ----
  XRows8 xt;
  RowMeta8 meta;
  {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    meta = row_record(blockIdx.x, wave, lane);
    __builtin_amdgcn_sched_barrier(0);
    float bias = 0.f, hv0 = 0.f, hb = 0.f;
    if (tid < 2 * HID) bias = a.params[(tid < HID ? oB1 : oB2 - HID) + tid];
    if (net == 0) {
      const int j0 = tid >> 3, k = tid & 7;
      if (k < nk) hv0 = a.params[lay.act_W + j0 * nk + k];
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
    if (tid < 2 * HID) b1s[tid] = bias;   // b1s | b2s are adjacent
    if (net == 0) {
      hw[head_row(tid >> 3) + (tid & 7)] = hv0;
      if (tid < 8) hbs[tid] = hb;
    } else {
      if (tid < HID) hw[tid] = hv0;
      if (tid == 0) hbs[0] = hb;
    }
  }

  f32x4 gW1[2], gW2[2];
  f32x4 gB1 = {0.f, 0.f, 0.f, 0.f}, gB2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int b = 0; b < 2; ++b) {
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
  int tile = blockIdx.x;   // every workgroup owns at least one tile (the launcher refuses a grid wider than the tile count)
  do {
    int tidv = threadIdx.x;
    asm volatile("" : "+v"(tidv));
    const int tid = tidv, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cw = wave & 3, hf = wave >> 2;
    const SplitSel ssel = split_sel();
    const int j = lane & 15, kg = lane >> 4;
    const bool has_next = tile + (int)gridDim.x < a.ntiles;
    const int unit = 16 * cw + j;   // this lane's column of every 16x16 result

    Frag3 W1f[2];
    {   // W1 by (feature, unit): B of S1, six L2-resident 16-byte loads per lane, in flight under the row commit
      const uint4* img = reinterpret_cast<const uint4*>(a.wimage) + (size_t)((net * 4 + cw) * 18) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) W1f[c].p[p] = __builtin_bit_cast(bf16x8, img[(c * 3 + p) * 64]);
    }
    // ---- T0: this tile's rows land in LDS as planes ----
    if (lane < 8) {
      const int row = wave * 8 + lane;
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

    // per-lane operand offsets, this wave's half (blocks 2 hf, 2 hf + 1) folded in: block bb of the half = + bb * 16 rows as an
    // immediate for the plain reads; the transposing reads and the C-layout stores address a block through an XOR of the granule
    // index with 2 * (block ^ half), so the half's two blocks need two bases each
    const int pb0 = plain_base(j, kg, 0), pb1 = plain_base(j, kg, 1);
    const int pb0h = pb0 + hf * 32 * PL_ROW, pb1h = pb1 + hf * 32 * PL_ROW;
    int trb[2], csb[2];
    {
      const int a0 = 8 * kg + (j >> 2);
      const int g0 = ((j & 3) >> 1) ^ pl_swz(a0), row0 = a0 * PL_ROW + 8 * (j & 1);
      const int g1 = (kg >> 1) ^ pl_swz(unit), row1 = unit * PL_ROW + 8 * (kg & 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        trb[k] = row0 + ((g0 ^ (2 * (2 * hf + k))) << 4);
        csb[k] = row1 + ((g1 ^ (2 * (2 * hf + k))) << 4);
      }
    }
    // transposing fragment of plane rows 32c + 8kg .. +7 (contraction index), columns 16*(2 hf + bb) .. +15 (lane index)
    auto ld_trf = [&](int buf, int c, int bb) -> Frag3 {
      return ld_tr(smem, trb[bb], trb[bb ^ 1], buf + 32 * c * PL_ROW, buf + (32 * c + 4) * PL_ROW);
    };

    f32x4 d1[2];   // 1 - H1^2 of this lane's 8 elements (rows 16*b + 4*kg + r, column `unit`): kept for dZ1
    // ---- S1: H1 = tanh(X W1 (+ b1)) -> H1T planes ----
    {
      const float bb1 = FOLD ? 0.f : b1s[unit];
      Frag3 xa = ld_plain(smem, pb0h, XT), xb = ld_plain(smem, pb1h, XT);   // A: X rows 16b + i, features 32c + 8kg ..
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma6(xa, W1f[0], acc);
        acc = mma6(xb, W1f[1], acc);
        if (bb < 1) {
          xa = ld_plain(smem, pb0h, XT + 16 * PL_ROW);
          xb = ld_plain(smem, pb1h, XT + 16 * PL_ROW);
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fast_tanh(FOLD ? acc[r] : acc[r] + bb1);
          d1[bb][r] = 1.0f - v[r] * v[r];
        }
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(d1[bb][0]), "v"(d1[bb][1]), "v"(d1[bb][2]), "v"(d1[bb][3]));
        bf16x4 p[3];
        split4(ssel, v, p);
        st_planes4(smem, H1T + csb[bb], p);
      }
    }
    Frag3 W2f[2];   // W2 by (input, unit): B of S2 (requested behind S1's products: W1's registers are free)
    {
      const uint4* img = reinterpret_cast<const uint4*>(a.wimage) + (size_t)((net * 4 + cw) * 18 + 6) * 64 + lane;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p) W2f[c].p[p] = __builtin_bit_cast(bf16x8, img[(c * 3 + p) * 64]);
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 2);

    // ---- S2: H2 = tanh(H1 W2 + b2) -> H2 (f32).  Operand roles swapped (A = W2 fragments): the result tile is H2^T, i.e.
    //      lane = row 16*b + j, registers = units 16*cw + 4*kg + r -> one 16-byte store per block ----
    RowMeta8 meta_next = meta;
    if (has_next) meta_next = row_record(tile + gridDim.x, wave, lane);   // next tile's rows, committed at its T0
    {
      const float4 bb2 = *reinterpret_cast<const float4*>(b2s + 16 * cw + 4 * kg);
      float* const h2o = h2_at(32 * hf + j, 4 * cw + kg);   // h2_swz depends on row bits 0..1 = j's
      Frag3 xa = ld_trf(H1T, 0, 0), xb = ld_trf(H1T, 1, 0);
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mma6(W2f[0], xa, acc);
        acc = mma6(W2f[1], xb, acc);
        if (bb < 1) {
          xa = ld_trf(H1T, 0, 1);
          xb = ld_trf(H1T, 1, 1);
        }
        *reinterpret_cast<float4*>(h2o + bb * 16 * (DZ_ROW / 4)) =
            make_float4(fast_tanh(acc[0] + bb2.x), fast_tanh(acc[1] + bb2.y), fast_tanh(acc[2] + bb2.z), fast_tanh(acc[3] + bb2.w));
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 3);

    // ---- SH-a: head forward, loss, dL/dhead; dZ2 = dH2 * (1 - H2^2) stays in registers; eight lanes per row ----
    float dzv[8];
    const int hr = tid >> 3, hq = tid & 3, hh = (tid >> 2) & 1;   // row, and units 8 hq + 32 hh .. +7
    {
      const int r = hr;
      const bool valid = rowphys[r] >= 0;
      float h[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(h2_at(r, 2 * hq + g + 8 * hh));
        h[4 * g] = v.x; h[4 * g + 1] = v.y; h[4 * g + 2] = v.z; h[4 * g + 3] = v.w;
      }
      if (net == 0) {
        const float* hwq = hw + 72 * hq + 288 * hh;
        float z[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) z[k] = 0.f;
        for_head_rows8<NK>(hwq, [&](int m, const float4& w0, const float4& w1) {
          const float wk[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
          for (int k = 0; k < NK; ++k) z[k] = __builtin_fmaf(h[m], wk[k], z[k]);
        });
        float pr[NK];
        float mx = -3.0e38f;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          z[k] = oct_sum(z[k]) + hbs[k];
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
        if (valid && (tid & 7) == 0) {
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
        if ((tid & 7) == 0) {
          float4* o = reinterpret_cast<float4*>(dzs + r * 8);
          o[0] = make_float4(dz[0], dz[1], dz[2], dz[3]);
          if constexpr (NK > 4) o[1] = make_float4(dz[4], dz[5], dz[6], dz[7]);
        }
        typedef float hp2 __attribute__((ext_vector_type(2)));
        hp2 dzp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) dzp[k] = (hp2){dz[2 * k], dz[2 * k + 1]};      // dz[k >= NK] = 0
        for_head_rows8<NK>(hwq, [&](int m, const float4& w0, const float4& w1) {
          hp2 acc = (hp2){w0.x, w0.y} * dzp[0];
          if constexpr (NK > 2) acc = __builtin_elementwise_fma((hp2){w0.z, w0.w}, dzp[1], acc);
          if constexpr (NK > 4) acc = __builtin_elementwise_fma((hp2){w1.x, w1.y}, dzp[2], acc);
          if constexpr (NK > 6) acc = __builtin_elementwise_fma((hp2){w1.z, w1.w}, dzp[3], acc);
          float hsum;
          asm("v_add_f32 %0, %1, %2" : "=v"(hsum) : "v"(acc.x), "v"(acc.y));
          dzv[m] = hsum * (1.0f - h[m] * h[m]);
        });
      } else {
        float wv[8];
        float v = 0.f;
        const float* hwq = hw + 8 * hq + 32 * hh;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          wv[m] = hwq[m];
          v = __builtin_fmaf(h[m], wv[m], v);
        }
        v = oct_sum(v) + hbs[0];
        const float retn = radv[r], oldv = rold[r];
        float vp = v, pass = 1.f;
        if (a.clip_vf >= 0.f) {
          const float dlt = v - oldv;
          pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
          vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
        }
        const float err = vp - retn;
        const float dv = valid ? a.vf_coef * 2.0f * err * inv_nb * pass : 0.f;
        if (valid && (tid & 7) == 0) st[1] += err * err;
        if ((tid & 7) == 0) dzs[r] = dv;
#pragma unroll
        for (int m = 0; m < 8; ++m) dzv[m] = dv * wv[m] * (1.0f - h[m] * h[m]);
      }
    }
    wave_lds_sync();   // dzs rows of this wave are written and read by this wave only
    if (first) PH_STAMP(a.prof, 4);

    // ---- SH-b: d head weights / d head bias over this wave's 8 rows (H2 rows of this wave, still in LDS) ----
    if (net == 0) {
      int lofs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) lofs[i] = (((lane >> 2) ^ h2_swz(i)) << 2) | (lane & 3);
      const float* hp = reinterpret_cast<const float*>(smem + DZ2 + wave * 8 * DZ_ROW);
      const float* dp = dzs + wave * 8 * 8;
#pragma unroll 1
      for (int r0 = 0; r0 < 8; r0 += 4, hp += 4 * (DZ_ROW / 4), dp += 4 * 8) {
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
      if (lane < NK) ghb += lds_sum8(dzs + wave * 8 * 8 + lane, 8);
    } else {
      float hv[8], dv[8];
      const char* hrow = smem + DZ2 + wave * (8 * DZ_ROW);
      const float* drow = dzs + wave * 8;
      int col4[4];   // h2_swz(i) depends on i & 3 only: four per-lane column offsets (bytes)
#pragma unroll
      for (int i = 0; i < 4; ++i) col4[i] = ((((lane >> 2) ^ h2_swz(i)) << 2) | (lane & 3)) * 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        hv[i] = *reinterpret_cast<const float*>(hrow + i * DZ_ROW + col4[i & 3]);
        dv[i] = drow[i];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        gh0 = __builtin_fmaf(hv[i], dv[i], gh0);
        ghb += dv[i];
      }
    }
    wave_lds_sync();   // this wave's H2 rows are consumed; its dZ2 rows go on top of them

    // ---- SH-c: dZ2 -> planes, row-interleaved [row][plane][unit], over this wave's H2 rows ----
    {
      Frag3 f;
      split8(ssel, dzv, f);   // units 8 hq + 32 hh .. +7 = granule 4 hh + hq
      st_planes8<DZ_PL>(smem, DZ2 + hr * DZ_ROW + (((4 * hh + hq) ^ pl_swz(hr)) << 4), f);
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 5);

    // ---- S6a: dW2 += H1^T dZ2 (this half's two input blocks), d b2 ; dH1 = dZ2 W2^T (this half's two row blocks) ----
    f32x4 dh1[2];
    {
      const int a0 = 8 * kg + (j >> 2), g0 = ((j & 3) >> 1) ^ pl_swz(a0), row0 = a0 * DZ_ROW + 8 * (j & 1);
      const int tlo = row0 + ((g0 ^ (2 * cw)) << 4), thi = row0 + ((g0 ^ (2 * (cw ^ 1))) << 4);
      const int db0 = (32 * hf + j) * DZ_ROW + ((kg ^ pl_swz(j)) << 4), db1 = (32 * hf + j) * DZ_ROW + (((4 + kg) ^ pl_swz(j)) << 4);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const Frag3 dz = ld_tr<DZ_PL>(smem, tlo, thi, DZ2 + 32 * c * DZ_ROW, DZ2 + (32 * c + 4) * DZ_ROW);
        if (hf == 0) gB2 = mma_ones(dz, gB2);
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const Frag3 h1 = ld_plain(smem, c == 0 ? pb0h : pb1h, H1T + bb * 16 * PL_ROW);   // A: H1T rows (input units) 16b + i
          gW2[bb] = mma6(h1, dz, gW2[bb]);
        }
      }
      // W2 by rows (B of dH1): six L2-resident 16-byte loads per lane
      Frag3 W2b[2];
      {
        const uint4* img = reinterpret_cast<const uint4*>(a.wimage) + (size_t)((net * 4 + cw) * 18 + 12) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int p = 0; p < 3; ++p) W2b[c].p[p] = __builtin_bit_cast(bf16x8, img[(c * 3 + p) * 64]);
      }
      if (!has_next) {   // dW2 and d b2 are final: they leave under the rest of this tile
        float* const w2o = rslab + RS_W2 + ((cw * 4 + 2 * hf) * 64 + lane) * 4;   // block bb: + 1 KB, as an immediate
        st_slab16<0>(w2o, gW2[0]);
        st_slab16<1024>(w2o, gW2[1]);
        if (hf == 0 && lane < 16) rslab[RS_B2 + 16 * cw + lane] = gB2[0];   // every row of the ones product is the column sum
      }
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 2; ++c)   // A: dZ2 rows 16b + i, units 32c + 8kg ..
          acc = mma6(ld_plain<DZ_PL>(smem, c == 0 ? db0 : db1, DZ2 + bb * 16 * DZ_ROW), W2b[c], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) dh1[bb][r] = acc[r] * d1[bb][r];   // dZ1 (registers; stored after the barrier)
      }
    }
    lds_barrier();
    if (first) PH_STAMP(a.prof, 6);

    // ---- S6b: dZ1 -> DZ1T planes over H1T: this half's two row blocks of unit row 16 cw + j.  BOTH halves' blocks make the unit
    //      rows the dW1 product reads: a workgroup barrier (the four-wave kernel's waves own their unit rows alone) ----
#pragma unroll
    for (int bb = 0; bb < 2; ++bb) {
      float v[4] = {dh1[bb][0], dh1[bb][1], dh1[bb][2], dh1[bb][3]};
      bf16x4 p[3];
      split4(ssel, v, p);
      st_planes4(smem, H1T + csb[bb], p);
    }
    meta = meta_next;
    if (has_next) xt.issue(meta.phys, a.ximg, a.ximg_zero_row, lane);   // the next tile's rows, gathered under S7
    lds_barrier();

    // ---- S7: dW1 += X^T dZ1 (this half's two feature blocks; d b1 rides as feature 63, or as a ones product) ----
    {
      const Frag3 dz0 = ld_plain(smem, pb0, H1T + cw * 16 * PL_ROW);   // B: DZ1T row (unit) 16 cw + j, rows 8kg .. / 32 + 8kg ..
      const Frag3 dz1 = ld_plain(smem, pb1, H1T + cw * 16 * PL_ROW);
      if constexpr (!FOLD) {
        if (hf == 0) {
          gB1 = mma_ones(dz0, gB1);
          gB1 = mma_ones(dz1, gB1);
        }
      }
      float* const w1o = rslab + RS_W1 + ((cw * 4 + 2 * hf) * 64 + lane) * 4;
#pragma unroll
      for (int bb = 0; bb < 2; ++bb) {
        gW1[bb] = mma6(ld_trf(XT, 0, bb), dz0, gW1[bb]);   // A: features 16b + i (lane), tile rows 8kg .. (contraction)
        gW1[bb] = mma6(ld_trf(XT, 1, bb), dz1, gW1[bb]);
        if (!has_next) {
          if (bb == 0) st_slab16<0>(w1o, gW1[0]);
          else st_slab16<1024>(w1o, gW1[1]);
        }
      }
      if (!has_next) {
        if constexpr (!FOLD)
          if (hf == 0 && lane < 16) rslab[RS_B1 + 16 * cw + lane] = gB1[0];
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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cw = wave & 3, hf = wave >> 2;
    if (!slabs_out) {
      float* const w2o = rslab + RS_W2 + ((cw * 4 + 2 * hf) * 64 + lane) * 4;
      float* const w1o = rslab + RS_W1 + ((cw * 4 + 2 * hf) * 64 + lane) * 4;
      st_slab16<0>(w2o, gW2[0]); st_slab16<1024>(w2o, gW2[1]);
      st_slab16<0>(w1o, gW1[0]); st_slab16<1024>(w1o, gW1[1]);
      if (hf == 0 && lane < 16) {   // every row of the ones products is the column sum
        rslab[RS_B2 + 16 * cw + lane] = gB2[0];
        if constexpr (!FOLD) rslab[RS_B1 + 16 * cw + lane] = gB1[0];
      }
    }
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) {
      float v = st[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      st[k] = v;
    }
    float* part = smem_f;  // [3 + NK][8 waves][64] over XT
    part[(0 * 8 + wave) * 64 + lane] = gh0;
    part[(1 * 8 + wave) * 64 + lane] = ghb;
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < NSTATP; ++k) part[(2 * 8 + wave) * 64 + k] = st[k];
    }
    if (net == 0) {
#pragma unroll
      for (int k = 0; k < NK; ++k) part[((3 + k) * 8 + wave) * 64 + lane] = ghr[k];
    }
    lds_barrier();
    auto wsum = [&](int which, int idx) {
      float s = part[(which * 8 + 0) * 64 + idx];
#pragma unroll
      for (int w = 1; w < 8; ++w) s += part[(which * 8 + w) * 64 + idx];
      return s;
    };
    if (net == 1 && tid < HID) rslab[RS_HW + tid] = wsum(0, tid);
    if (net == 0) {
      const int k = tid >> 6, jj = tid & 63;
      if (k < NK) rslab[RS_HW + jj * 8 + k] = wsum(3 + k, jj);
    }
    if (net == 0 && tid < 8) rslab[RS_HB + tid] = wsum(1, tid);
    if (net == 1 && tid == 0) rslab[RS_HB] = wsum(1, 0);
    if (tid < NSTATP) a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + tid] = wsum(2, tid);
  }
  PH_STAMP(a.prof, 13);
}

template <int NK, bool FOLD>
static hipError_t launch_split8_inst(const GradArgs& a, int nwg, size_t lds, hipStream_t s) {
  static bool allowed_dev[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  bool& allowed = allowed_dev[(dev >= 0 && dev < 64) ? dev : 0];
  if (!allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_split8_kernel<NK, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  hipLaunchKernelGGL((ppo_grad_split8_kernel<NK, FOLD>), dim3(nwg, 2), dim3(512), lds, s, a);
  return hipGetLastError();
}
template <int NK>
static hipError_t launch_split8_nk(const GradArgs& a, int nwg, size_t lds, hipStream_t s) {
  return grad_fast_fold(a.nd) ? launch_split8_inst<NK, true>(a, nwg, lds, s) : launch_split8_inst<NK, false>(a, nwg, lds, s);
}

// same contract as launch_ppo_grad_split (ph_ppo_split.hip), which routes here
hipError_t launch_ppo_grad_split8(const GradArgs& a, int nwg, size_t lds, hipStream_t s) {
  if (nwg < 1 || nwg > a.ntiles) return hipErrorInvalidValue;
  switch (a.nd.L) {
    case 1: return launch_split8_nk<1>(a, nwg, lds, s);
    case 2: return launch_split8_nk<2>(a, nwg, lds, s);
    case 3: return launch_split8_nk<3>(a, nwg, lds, s);
    case 4: return launch_split8_nk<4>(a, nwg, lds, s);
    case 5: return launch_split8_nk<5>(a, nwg, lds, s);
    case 6: return launch_split8_nk<6>(a, nwg, lds, s);
    case 7: return launch_split8_nk<7>(a, nwg, lds, s);
    default: break;
  }
  return launch_split8_nk<8>(a, nwg, lds, s);
}

}  // namespace ph

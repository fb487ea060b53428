// ppo_grad_rp_kernel ("row-parallel"): the PPO minibatch gradient (SB3 PPO.train() inner loop, pantheonrl/common/agents.py:155;
// arithmetic from SURVEY.md A.3 and the in-tree copy pantheonrl/algos/adap/adap_learn.py:253-344) for Box observations
// with F <= 64 features and a Discrete head with <= 8 logits (every BASELINE config but Liar's Dice and RPS).
//
// Grid (nWG, 2 nets), 512 threads = 8 waves, one workgroup per CU (two waves per SIMD), 128 minibatch rows per step.
//   P1  A WAVE owns 16 of the rows: forward, loss and the head's backward run on v_mfma_f32_16x16x4_f32 tiles whose
//       activations never leave the wave (registers + a wave-private 2 x [16][64] LDS scratch that turns the C/D fragment
//       layout into A/B operand layouts).  No workgroup barrier inside: waves drift apart, so one wave's tanh / softmax
//       VALU work overlaps the MFMAs of the other wave on its SIMD (barrier-phased kernels keep co-resident waves in
//       lockstep: all in MFMA, then all in VALU).
//   P2  dW2 = H1^T dZ2 over ALL 128 rows: each wave owns two 16x16 output tiles (x two row halves = four independent MFMA
//       chains) and reads every wave's scratch; then its own rows' dH1 = dZ2 W2^T, dZ1.
//   P3  dW1 = X^T dZ1 likewise.
// Four LDS-only barriers per 128-row step separate P1 / P2 / dZ1 write-back / P3; the weight-gradient tiles a wave owns stay
// in 32 registers over all steps and go to the workgroup's slab once.  (Keeping the full 64x64 gradients per wave --
// no barriers at all -- needs 144 accumulator registers per lane and spills at two waves per SIMD.)
//
// LDS matrices are [rows][64] with the column XOR-swizzled by swz(row): conflict-free for the three fragment access
// shapes used here -- "16 rows x 2 columns" (A operands), "2 rows x 16 columns" (B / transposed-A operands) and the C/D
// shape "rows 4g+r x 16 columns".
#include "ph_launch.h"

namespace ph {

template <int CTRL>
__device__ __forceinline__ float dpp_row(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// all-reduce over the 16 lanes of a DPP row (= the 16 columns of one row group); every lane gets the identical bits
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_row<0x128>(v);  // row_ror:8
  v += dpp_row<0x124>(v);  // row_ror:4
  v += dpp_row<0x122>(v);  // row_ror:2
  v += dpp_row<0x121>(v);  // row_ror:1
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_row<0x128>(v));
  v = fmaxf(v, dpp_row<0x124>(v));
  v = fmaxf(v, dpp_row<0x122>(v));
  v = fmaxf(v, dpp_row<0x121>(v));
  return v;
}

// acc[t] += sum_s A(s) x B(s, t) over 16 k-steps with the next step's five operands fetched ahead of the current step's four
// MFMAs; the sched_barriers pin that shape (left alone, the scheduler hoists all 80 operand reads and spills).
template <bool VALU, class FA, class FB>
__device__ __forceinline__ void mma_1x4(f32x4 (&acc)[4], FA&& fa, FB&& fb, int lane) {
  float a0 = fa(0), b0[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) b0[t] = fb(0, t);
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    float a1 = 0.f, b1[4] = {0.f, 0.f, 0.f, 0.f};
    if (s + 1 < 16) {
      a1 = fa(s + 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) b1[t] = fb(s + 1, t);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mma16<VALU>(a0, b0[t], acc[t], lane);
    __builtin_amdgcn_sched_barrier(0);
    a0 = a1;
#pragma unroll
    for (int t = 0; t < 4; ++t) b0[t] = b1[t];
  }
}
constexpr int RP_WAVES = 8, RP_NT = RP_WAVES * 64, RP_ROWS = RP_WAVES * 16;
constexpr int DZ_LD = 18;    // wave-private dL/dlogits [16][18]
constexpr int WOT_LD = 80;   // act_W^T [8][80]
// LDS map (floats)
constexpr int RP_W1 = 0, RP_W2 = RP_W1 + 4096, RP_XP = RP_W2 + 4096, RP_WO = RP_XP + RP_ROWS * 64, RP_WOT = RP_WO + 64 * 16,
              RP_B1 = RP_WOT + 8 * WOT_LD, RP_B2 = RP_B1 + 64, RP_HB = RP_B2 + 64, RP_META = RP_HB + 16,
              RP_WAVE0 = RP_META + 4 * RP_ROWS;
constexpr int RP_HP1 = 0, RP_HP2 = 1024, RP_DZ = 2048, RP_WAVE_SZ = RP_DZ + 16 * DZ_LD;
constexpr int RP_LDS_FLOATS = RP_WAVE0 + RP_WAVES * RP_WAVE_SZ;
constexpr int RP_RED_FLOATS = RP_WAVES * 32 * 64;   // epilogue fold buffer
constexpr int RP_TOTAL_FLOATS = RP_LDS_FLOATS > RP_RED_FLOATS ? RP_LDS_FLOATS : RP_RED_FLOATS;

__device__ __forceinline__ void rp_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct RpMeta {
  int phys;
  float adv, old, act;
};

// NET is a template parameter so that each net's register allocation carries only its own head state
template <bool VALU, int NET>
__device__ __forceinline__ void rp_body(const GradArgs& a) {
  PH_STAMP(a.prof, 0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const NetDims& nd = a.nd;
  const ph_layout& lay = nd.lay;
  float* W1s = smem + RP_W1;
  float* W2s = smem + RP_W2;
  float* XP = smem + RP_XP;     // [128][64] swizzled observation tile
  float* Wos = smem + RP_WO;    // policy: act_W [64][16] (cols >= L zero) | value: val_W [64]
  float* WoT = smem + RP_WOT;   // policy: act_W^T [8][WOT_LD]
  float* b1s = smem + RP_B1;
  float* b2s = smem + RP_B2;
  float* hbs = smem + RP_HB;    // policy: act_b [16] (-3e38 beyond L) | value: val_b
  int* mphys = (int*)(smem + RP_META);          // [128] physical buffer row, -1 = padding
  float* madv = smem + RP_META + RP_ROWS;       // [128] normalised advantage | returns
  float* mold = smem + RP_META + 2 * RP_ROWS;   // [128] old log-prob | old value
  float* mact = smem + RP_META + 3 * RP_ROWS;   // [128] action index

  constexpr int net = NET;
  const int oW1 = net == 0 ? lay.pi_W1 : lay.vf_W1, oB1 = net == 0 ? lay.pi_b1 : lay.vf_b1;
  const int oW2 = net == 0 ? lay.pi_W2 : lay.vf_W2, oB2 = net == 0 ? lay.pi_b2 : lay.vf_b2;
  float* slab = a.slabs + (size_t)blockIdx.x * lay.P;
  const float inv_nb = 1.0f / (float)a.nb;
  const int nk = nd.L;
  const uint64_t perm_key = a.idx ? 0ull : epoch_key(a.perm_seed + (a.epoch ? *a.epoch : 0ull), a.perm_epoch);
  const bool norm = net == 0 && a.norm_adv && a.nb > 1;
  const float adv_mean = norm ? a.advstats[0] : 0.f;
  const float adv_den = norm ? a.advstats[1] + 1e-8f : 1.f;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;

  // rows of a step are gathered by all 8 waves: lane i < 16 of wave w serves row w + 8i (so that a wave-instruction of the
  // observation gather is one contiguous row); the index load, the scalar gathers and the row gather are issued in
  // different phases and consumed at the next step's start
  auto row_index = [&](int step) -> int {
    const int gi = step * RP_ROWS + wave + 8 * lane;
    if (lane >= 16 || gi >= a.nb) return -1;
    return a.idx ? a.idx[gi] : (int)feistel_perm((uint32_t)(a.mb_start + gi), a.perm_n, a.perm_hb, perm_key);
  };
  auto row_scalars = [&](int n) -> RpMeta {
    RpMeta m;
    m.phys = -1;
    m.adv = m.old = m.act = 0.f;
    if (n >= 0) {
      m.phys = env_major_to_phys(n, a.T, a.E);
      if (net == 0) {
        m.adv = a.rb_adv[m.phys];
        m.old = a.rb_logp[m.phys];
        m.act = a.rb_act[m.phys];
      } else {
        m.adv = a.rb_ret[m.phys];
        m.old = a.rb_val[m.phys];
      }
    }
    return m;
  };
  auto load_x = [&](int physv, float (&xr)[16]) {   // raw loads; masking happens at commit
    const int f = lane < nd.F ? lane : 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int p = __builtin_amdgcn_readlane(physv, i);
      xr[i] = a.rb_obs[(size_t)(p < 0 ? 0 : p) * nd.D + f];
    }
  };

  // ---- prologue: weights -> LDS, first step's rows in flight ----
  RpMeta meta = row_scalars(blockIdx.x < a.ntiles ? row_index(blockIdx.x) : -1);
  float xr[16];
  load_x(meta.phys, xr);
  {
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // 64x64 floats = 1024 float4 per matrix, 512 threads
      const int q = tid + RP_NT * i, k = q >> 4, c4 = (q & 15) << 2;
      const float4 w1 = (k < nd.F) ? *reinterpret_cast<const float4*>(a.params + oW1 + (size_t)k * HID + c4)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 w2 = *reinterpret_cast<const float4*>(a.params + oW2 + (size_t)k * HID + c4);
      const int sw = swz(k);   // bit 0 of sw is clear: column pairs (2p, 2p+1) stay adjacent and ordered
      *reinterpret_cast<float2*>(W1s + k * 64 + (c4 ^ sw)) = make_float2(w1.x, w1.y);
      *reinterpret_cast<float2*>(W1s + k * 64 + ((c4 + 2) ^ sw)) = make_float2(w1.z, w1.w);
      *reinterpret_cast<float2*>(W2s + k * 64 + (c4 ^ sw)) = make_float2(w2.x, w2.y);
      *reinterpret_cast<float2*>(W2s + k * 64 + ((c4 + 2) ^ sw)) = make_float2(w2.z, w2.w);
    }
    if (tid < HID) {
      b1s[tid] = a.params[oB1 + tid];
      b2s[tid] = a.params[oB2 + tid];
    }
    if (net == 0) {
      for (int e = tid; e < 64 * 16; e += RP_NT) {
        const int j = e >> 4, k = e & 15;
        Wos[e] = (k < nk) ? a.params[lay.act_W + j * nk + k] : 0.f;
      }
      for (int e = tid; e < 8 * 64; e += RP_NT) {
        const int k = e >> 6, j = e & 63;
        WoT[k * WOT_LD + j] = (k < nk) ? a.params[lay.act_W + j * nk + k] : 0.f;
      }
      // padded logits get a -3e38 "bias": they drop out of softmax, entropy and every gradient with no special cases
      if (tid < 16) hbs[tid] = (tid < nk) ? a.params[lay.act_b + tid] : -3.0e38f;
    } else {
      if (tid < HID) Wos[tid] = a.params[lay.val_W + tid];
      if (tid == 0) hbs[0] = a.params[lay.val_b];
    }
  }

  float* wbuf = smem + RP_WAVE0 + wave * RP_WAVE_SZ;
  float* HP1 = wbuf + RP_HP1;   // [16][64] swizzled: H1 -> dZ1   (rows 16*wave .. +15 of the step)
  float* HP2 = wbuf + RP_HP2;   // [16][64] swizzled: H2 -> dZ2
  float* DZs = wbuf + RP_DZ;    // [16][DZ_LD] dL/dlogits
  const int row0 = 16 * wave;

  // LDS operand addressing.  B-type element (row 4s+g, col 16t+c) of a swizzled [..][64] matrix sits at
  // ((bq ^ 16t) ^ KS(s)) + 256 s with KS(s) = (8s & 62) ^ 16 (s & 1) a compile-time constant; A-type element
  // (row R, col 4s+g) at (R*64 + (g ^ swz(R))) ^ 4s: one v_xor per operand.
#define RP_KS(s) ((((8 * (s)) & 62) ^ (16 * ((s) & 1))))
#define RP_B(buf, s, t) (buf)[((bq ^ (16 * (t))) ^ RP_KS(s)) + 256 * (s)]
#define RP_A(buf, s) (buf)[aq ^ (4 * (s))]
  // the same fragment of ANOTHER wave's scratch: k-step S = 0..31 over the 128 rows of the step
#define RP_BW(off, S, t) smem[RP_WAVE0 + ((S) >> 2) * RP_WAVE_SZ + (off) + (((bq ^ (16 * (t))) ^ RP_KS((S) & 3)) + 256 * ((S) & 3))]

  // the two output tiles (m, n0), (m, n0 + 1) of dW1 / dW2 this wave owns, each as two row-half chains
  const int tm = wave >> 1, tn0 = 2 * (wave & 1);
  f32x4 G1[2][2], G2[2][2], gWo[4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      G1[h][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      G2[h][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
  for (int m = 0; m < 4; ++m) gWo[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  float gb1[4] = {0.f, 0.f, 0.f, 0.f}, gb2[4] = {0.f, 0.f, 0.f, 0.f};
  float gvw[4] = {0.f, 0.f, 0.f, 0.f};   // value net: d val_W[16t + c] partial over this lane's rows
  float ghb = 0.f;                       // policy: d act_b[c] partial | value: d val_b partial
  float st[NSTATP];
#pragma unroll
  for (int k = 0; k < NSTATP; ++k) st[k] = 0.f;

#if defined(RP_PRIO)
  if (wave >= 4) __builtin_amdgcn_s_setprio(RP_PRIO);   // A/B: static priority for the younger half of the workgroup
#endif
  bool first = true;
  for (int step = blockIdx.x; step < a.ntiles; step += gridDim.x, first = false) {
    const bool has_next = step + (int)gridDim.x < a.ntiles;
    // Lane coordinates are re-derived per step from an opaque copy of the lane id: every operand address below is a
    // loop-invariant function of it, and left visible the compiler hoists hundreds of them out of the step loop and spills.
    int lane_o = threadIdx.x & 63;
    asm volatile("" : "+v"(lane_o));
    const int lane = lane_o, c = lane & 15, g = lane >> 4;
    const int bq = g * 64 + (c ^ (2 * g) ^ (16 * (g & 1)));
    const int aq = c * 64 + (g ^ swz(c));
    const int aqx = (row0 + c) * 64 + (g ^ swz(row0 + c));
    // ---- T0: this step's rows (gathered during the previous step / the prologue) land in LDS ----
    if (lane < 16) {
      const int row = wave + 8 * lane;
      mphys[row] = meta.phys;
      madv[row] = (norm && meta.phys >= 0) ? (meta.adv - adv_mean) / adv_den : meta.adv;
      mold[row] = meta.old;
      mact[row] = meta.act;
    }
    {
      const bool fok = lane < nd.F;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int p = __builtin_amdgcn_readlane(meta.phys, i);
        XP[sidx(wave + 8 * i, lane)] = (p >= 0 && fok) ? xr[i] : 0.f;
      }
    }
    rp_barrier();
    if (first) PH_STAMP(a.prof, 1);
    const int n_next = has_next ? row_index(step + gridDim.x) : -1;   // index load: consumed at P2

    // ================= P1: wave-private forward + head (rows row0 .. row0+15) =================
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_1x4<VALU>(acc, [&](int s) { return XP[aqx ^ (4 * s)]; }, [&](int s, int t) { return RP_B(W1s, s, t); }, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = b1s[16 * t + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) HP1[sidx(4 * g + r, 16 * t + c)] = fast_tanh(acc[t][r] + b);
    }
    if (first) PH_STAMP(a.prof, 2);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    mma_1x4<VALU>(acc, [&](int s) { return RP_A(HP1, s); }, [&](int s, int t) { return RP_B(W2s, s, t); }, lane);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float b = b2s[16 * t + c];
#pragma unroll
      for (int r = 0; r < 4; ++r) HP2[sidx(4 * g + r, 16 * t + c)] = fast_tanh(acc[t][r] + b);
    }
    if (first) PH_STAMP(a.prof, 3);

    // head: loss, dL/dhead, dZ2 = dH2 * (1 - H2^2) in place over H2
    float dz2[4][4];
    if constexpr (net == 0) {
      f32x4 zacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 16; ++s) zacc = mma16<VALU>(RP_A(HP2, s), Wos[(4 * s + g) * 16 + c], zacc, lane);
      const float bias = hbs[c];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lrow = 4 * g + r, row = row0 + lrow;
        const float z = zacc[r] + bias;                       // padded columns: -3e38
        const float mx = row_max16(z);
        const float p = fast_exp(z - mx);
        const float se = row_sum16(p);
        const float lse = mx + fast_log(se);
        const float pr = p * __builtin_amdgcn_rcpf(se);
        const float ent = -row_sum16(pr * (z - lse));
        int act = (int)mact[row];
        act = act < 0 ? 0 : (act >= nk ? nk - 1 : act);
        const float logp = row_sum16(c == act ? z : 0.f) - lse;
        const float adv = madv[row];
        const float lr = logp - mold[row];
        const float ratio = fast_exp(lr);
        const float lo_c = 1.0f - a.clip, hi_c = 1.0f + a.clip;
        const float rc = fminf(fmaxf(ratio, lo_c), hi_c);
        const float pl1 = adv * ratio, pl2 = adv * rc;
        // torch.min backward: the smaller branch gets the gradient, ties split 1/2 + 1/2; clamp passes it iff lo <= ratio <= hi
        const float inr = (ratio >= lo_c && ratio <= hi_c) ? 1.f : 0.f;
        const float gate = (pl1 < pl2) ? 1.f : ((pl1 > pl2) ? inr : 0.5f + 0.5f * inr);
        const bool valid = mphys[row] >= 0;
        const float live = valid ? 1.f : 0.f;
        const float g_lp = -inv_nb * adv * ratio * gate * live;
        const float g_en = -a.ent_coef * inv_nb * live;
        if (valid && c == 0) {
          st[0] += -fminf(pl1, pl2);
          st[2] += -ent;
          st[3] += (fabsf(ratio - 1.0f) > a.clip) ? 1.f : 0.f;
          st[4] += (ratio - 1.0f) - lr;
        }
        const float dlogp = ((c == act) ? 1.f : 0.f) - pr;
        const float dent = -pr * ((z - lse) + ent);
        const float dzv = g_lp * dlogp + g_en * dent;
        DZs[lrow * DZ_LD + c] = dzv;
        ghb += dzv;
      }
      // d act_W += H2^T dz over this wave's rows (M = 64 hidden units: 4 tiles, N = 16 logit columns, K = 16 rows)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bv = DZs[(4 * s + g) * DZ_LD + c];
#pragma unroll
        for (int m = 0; m < 4; ++m) gWo[m] = mma16<VALU>(RP_B(HP2, s, m), bv, gWo[m], lane);
      }
      // dH2 = dz act_W^T  (K = 8 logits: 2 steps)
      f32x4 dh[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) dh[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float av = DZs[c * DZ_LD + 4 * s + g];
#pragma unroll
        for (int t = 0; t < 4; ++t) dh[t] = mma16<VALU>(av, WoT[(4 * s + g) * WOT_LD + 16 * t + c], dh[t], lane);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float h = HP2[sidx(4 * g + r, 16 * t + c)];   // H2 again, in the C/D layout
          dz2[t][r] = dh[t][r] * (1.0f - h * h);
        }
    } else {
      float h2[4][4], vw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        vw[t] = Wos[16 * t + c];
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[t][r] = HP2[sidx(4 * g + r, 16 * t + c)];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + 4 * g + r;
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) v = __builtin_fmaf(h2[t][r], vw[t], v);
        v = row_sum16(v) + hbs[0];
        const float retn = madv[row], oldv = mold[row];
        float vp = v, pass = 1.f;
        if (a.clip_vf >= 0.f) {
          const float dlt = v - oldv;
          pass = (dlt >= -a.clip_vf && dlt <= a.clip_vf) ? 1.f : 0.f;
          vp = oldv + fminf(fmaxf(dlt, -a.clip_vf), a.clip_vf);
        }
        const float err = vp - retn;
        const bool valid = mphys[row] >= 0;
        const float dv = valid ? a.vf_coef * 2.0f * err * inv_nb * pass : 0.f;
        if (valid && c == 0) st[1] += err * err;
        if (c == 0) ghb += dv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          gvw[t] = __builtin_fmaf(h2[t][r], dv, gvw[t]);
          dz2[t][r] = dv * vw[t] * (1.0f - h2[t][r] * h2[t][r]);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        HP2[sidx(4 * g + r, 16 * t + c)] = dz2[t][r];
        gb2[t] += dz2[t][r];
      }
    }
    rp_barrier();   // B1: every wave's H1 and dZ2 are in its scratch
    if (first) PH_STAMP(a.prof, 4);

    // ================= P2: dW2 tiles over all 128 rows, then this wave's dH1 / dZ1 =================
    RpMeta meta_next = row_scalars(n_next);          // next step's scalar gathers + row gather, consumed at its T0
    if (has_next) load_x(meta_next.phys, xr);
    {
      float a0[2], b0[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a0[h] = RP_BW(RP_HP1, 16 * h, tm);
        b0[h][0] = RP_BW(RP_HP2, 16 * h, tn0);
        b0[h][1] = RP_BW(RP_HP2, 16 * h, tn0 + 1);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a1[2] = {0.f, 0.f}, b1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (j + 1 < 16) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            a1[h] = RP_BW(RP_HP1, 16 * h + j + 1, tm);
            b1[h][0] = RP_BW(RP_HP2, 16 * h + j + 1, tn0);
            b1[h][1] = RP_BW(RP_HP2, 16 * h + j + 1, tn0 + 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int n = 0; n < 2; ++n) G2[h][n] = mma16<VALU>(a0[h], b0[h][n], G2[h][n], lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a0[h] = a1[h];
          b0[h][0] = b1[h][0];
          b0[h][1] = b1[h][1];
        }
      }
    }
    if (first) PH_STAMP(a.prof, 5);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    {
      int wq[4];   // B[k = j][n = k'] = W2[k'][j]: row 16t + c of W2s, column 4s + g -- an A-type access
#pragma unroll
      for (int t = 0; t < 4; ++t) wq[t] = (16 * t + c) * 64 + (g ^ swz(16 * t + c));
      mma_1x4<VALU>(acc, [&](int s) { return RP_A(HP2, s); }, [&](int s, int t) { return W2s[wq[t] ^ (4 * s)]; }, lane);
    }
    float dz1[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float h = HP1[sidx(4 * g + r, 16 * t + c)];
        dz1[t][r] = acc[t][r] * (1.0f - h * h);
        gb1[t] += dz1[t][r];
      }
    }
    rp_barrier();   // B2: nobody reads H1 any more
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) HP1[sidx(4 * g + r, 16 * t + c)] = dz1[t][r];
    rp_barrier();   // B3: every wave's dZ1 is in its scratch
    if (first) PH_STAMP(a.prof, 6);

    // ================= P3: dW1 tiles over all 128 rows =================
    {
#define RP_BX(S, t) XP[((bq ^ (16 * (t))) ^ RP_KS(S)) + 256 * (S)]
      float a0[2], b0[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        a0[h] = RP_BX(16 * h, tm);
        b0[h][0] = RP_BW(RP_HP1, 16 * h, tn0);
        b0[h][1] = RP_BW(RP_HP1, 16 * h, tn0 + 1);
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a1[2] = {0.f, 0.f}, b1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        if (j + 1 < 16) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            a1[h] = RP_BX(16 * h + j + 1, tm);
            b1[h][0] = RP_BW(RP_HP1, 16 * h + j + 1, tn0);
            b1[h][1] = RP_BW(RP_HP1, 16 * h + j + 1, tn0 + 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int n = 0; n < 2; ++n) G1[h][n] = mma16<VALU>(a0[h], b0[h][n], G1[h][n], lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          a0[h] = a1[h];
          b0[h][0] = b1[h][0];
          b0[h][1] = b1[h][1];
        }
      }
    }
    meta = meta_next;
    rp_barrier();   // B4: XP, the scratch buffers and the row scalars are free for the next step
    if (first) PH_STAMP(a.prof, 7);
  }
  PH_STAMP(a.prof, 12);

  // ---- epilogue: owned weight-gradient tiles -> slab; per-wave partials folded across the 8 waves in a fixed order ----
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 16 * tm + 4 * g + r, col = 16 * (tn0 + n) + c;
      slab[oW2 + k * HID + col] = G2[0][n][r] + G2[1][n][r];
      if (k < nd.F) slab[oW1 + (size_t)k * HID + col] = G1[0][n][r] + G1[1][n][r];
    }
  }
  {
    // lane-level pre-reduction of the bias partials over the 4 row groups g (columns live in lanes c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      gb1[t] += __shfl_xor(gb1[t], 16, 64);
      gb1[t] += __shfl_xor(gb1[t], 32, 64);
      gb2[t] += __shfl_xor(gb2[t], 16, 64);
      gb2[t] += __shfl_xor(gb2[t], 32, 64);
      gvw[t] += __shfl_xor(gvw[t], 16, 64);
      gvw[t] += __shfl_xor(gvw[t], 32, 64);
    }
    ghb += __shfl_xor(ghb, 16, 64);
    ghb += __shfl_xor(ghb, 32, 64);
#pragma unroll
    for (int k = 0; k < NSTATP; ++k) {
      float v = st[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
      st[k] = v;
    }
    float* red = smem;   // [wave][reg 0..31][lane]; the last rp_barrier of the loop freed every LDS region
    // registers 0..15: gWo tiles; 16..19 gb1; 20..23 gb2; 24..27 gvw; 28 ghb; 29 stats (lane k < NSTATP)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 32 + m * 4 + r) * 64 + lane] = gWo[m][r];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      red[(wave * 32 + 16 + t) * 64 + lane] = gb1[t];
      red[(wave * 32 + 20 + t) * 64 + lane] = gb2[t];
      red[(wave * 32 + 24 + t) * 64 + lane] = gvw[t];
    }
    red[(wave * 32 + 28) * 64 + lane] = ghb;
    {
      float sv = 0.f;
#pragma unroll
      for (int k = 0; k < NSTATP; ++k) {
        const float tot = __shfl(st[k], 0, 64);   // outside the select: a cross-lane read needs its source lane active
        sv = (lane == k) ? tot : sv;
      }
      red[(wave * 32 + 29) * 64 + lane] = sv;
    }
    rp_barrier();
    for (int e = tid; e < 30 * 64; e += RP_NT) {
      const int reg = e >> 6, l = e & 63;
      float v = red[e];
#pragma unroll
      for (int w = 1; w < RP_WAVES; ++w) v += red[w * 32 * 64 + e];
      if (reg < 16) {
        if (net == 0) {
          const int m = reg >> 2, r = reg & 3;
          const int j = 16 * m + 4 * (l >> 4) + r, k = l & 15;
          if (k < nk) slab[lay.act_W + j * nk + k] = v;
        }
      } else if (reg < 20) {
        if (l < 16) slab[oB1 + 16 * (reg - 16) + l] = v;
      } else if (reg < 24) {
        if (l < 16) slab[oB2 + 16 * (reg - 20) + l] = v;
      } else if (reg < 28) {
        if (net == 1 && l < 16) slab[lay.val_W + 16 * (reg - 24) + l] = v;
      } else if (reg == 28) {
        if (net == 0) {
          if (l < nk) slab[lay.act_b + l] = v;
        } else if (l == 0) {
          slab[lay.val_b] = v;
        }
      } else {
        if (l < NSTATP) a.statpart[((size_t)net * gridDim.x + blockIdx.x) * NSTATP + l] = v;
      }
    }
  }
  PH_STAMP(a.prof, 13);
}

template <bool VALU>
__global__ __launch_bounds__(RP_NT, 1) void ppo_grad_rp_kernel(GradArgs a) {
  if (*a.stop_flag) return;
  if (blockIdx.y == 0) rp_body<VALU, 0>(a);
  else rp_body<VALU, 1>(a);
}

bool grad_rp_eligible(const NetDims& nd) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("PH_GRAD_RP");   // opt-in: measured 44.5 us vs 42.2 us for ppo_grad_fast_kernel at the bench shape
    enabled = (e && e[0] == '1') ? 1 : 0;
  }
  return enabled && nd.obs_kind == PH_SPACE_BOX && nd.nchunk == 1 && nd.A == 1 && nd.L <= 8;
}

// rows are walked in steps of 128 (8 waves x 16); one workgroup per CU and net
void grad_rp_plan(int nb, int num_cu, int* nsteps, int* nwg) {
  const int steps = (nb + RP_ROWS - 1) / RP_ROWS;
  int w = num_cu / 2;
  if (w < 1) w = 1;
  *nsteps = steps;
  *nwg = steps < w ? steps : w;
}

template <bool VALU>
static hipError_t launch_rp_variant(const GradArgs& a, int nwg, hipStream_t s) {
  const size_t lds = sizeof(float) * (size_t)RP_TOTAL_FLOATS;
  static bool allowed = false;  // > 64 KiB of dynamic LDS is opt-in, once per kernel (kept out of graph capture)
  if (!allowed) {
    hipError_t e = hipFuncSetAttribute((const void*)ppo_grad_rp_kernel<VALU>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    allowed = true;
  }
  hipLaunchKernelGGL((ppo_grad_rp_kernel<VALU>), dim3(nwg, 2), dim3(RP_NT), lds, s, a);
  return hipGetLastError();
}

hipError_t launch_ppo_grad_rp(const GradArgs& a, int nwg, int gemm_mode, hipStream_t s) {
  return gemm_mode != 0 ? launch_rp_variant<true>(a, nwg, s) : launch_rp_variant<false>(a, nwg, s);
}

}  // namespace ph

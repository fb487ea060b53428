// Device-side building blocks shared by the gfx950 kernels (policy forward, PPO gradient, GAE).
// CDNA4 only: 64-lane wavefronts, v_mfma_f32_32x32x2_f32, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pantheon_hip.h"

namespace ph {

constexpr int HID = PH_HIDDEN;   // hidden width of both MLPs
constexpr int LDH = HID + 1;     // LDS leading dimension of every 64-wide matrix: odd => the strided
                                 // (transposed-operand) ds_read_b32 pattern below is bank-conflict free
constexpr int NSTATP = 8;        // per-workgroup partial-stat record (sums, not means)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Row (inside a 32x32 C/D tile) held by accumulator register r of a lane in half h = lane>>5.
// (cdna_hip_programming.md section 3: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).)
__device__ __forceinline__ int drow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---- float32 transcendental helpers ----------------------------------------------------------------------------------
// ocml's tanhf/expf/logf are correctly rounded but cost ~300 cycles per wave-instruction-equivalent here (exec-masked
// range branches); the epilogues of every layer are tanh, so they dominated the kernels.  These versions are
// branch-free and stay within a few ulp (rel. error <= |x|*1.2e-7 on exp), far inside the 2e-5 parity tolerance of
// the logits.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
// tanh(x) = 1 - 2 / (e^{2x} + 1): two quarter-rate ops and three plain ones, absolute error <= 2e-7 over the whole range (the
// relative error grows for |x| < 1e-3, where the absolute one is ~1e-8).  f32 MFMA and VALU instructions of a SIMD do not
// overlap on gfx950 (scripts/ubench, profiles/r02_ubench_*.txt), so every VALU op of the layer epilogues is paid in full: the
// round-1 form (odd polynomial below |x| = 0.3, this expression above: 15 plain ops) cost 1 us per gradient launch more.
__device__ __forceinline__ float fast_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * (2.0f * 1.44269504088896340736f));
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// One 32x32 output tile  acc += A[m0:m0+32, k0:k0+klen] * B[k0:k0+klen, n0:n0+32]  with both operands in
// LDS.  A(m,k) = TA ? A[k*lda+m] : A[m*lda+k];  B(k,n) = TB ? B[n*ldb+k] : B[k*ldb+n].
// MFMA path: v_mfma_f32_32x32x2_f32, lane l supplies A(m0+(l&31), k+(l>>5)) and B(k+(l>>5), n0+(l&31)).
// klen must be a multiple of 8: operands are fetched four MFMAs ahead (8 k-values per group) so the LDS latency of
// group g+1 hides under the 256 cycles the matrix pipe spends on group g.
// The hardware result is bit-for-bit the k-ordered fmaf chain (guide section 3), which is what the VALU
// path below computes with plain v_fma_f32 in the same accumulator layout -- a drop-in cross-check.
template <bool TA, bool TB, bool VALU>
__device__ __forceinline__ f32x16 tile_mma(const float* A, int lda, const float* B, int ldb, int m0, int n0,
                                           int k0, int klen, f32x16 acc, int lane = -1) {
  if (lane < 0) lane = threadIdx.x & 63;
  const int i = lane & 31, h = lane >> 5;
  if constexpr (!VALU) {
    const float* ap = TA ? (A + (k0 + h) * lda + m0 + i) : (A + (m0 + i) * lda + k0 + h);
    const float* bp = TB ? (B + (n0 + i) * ldb + k0 + h) : (B + (k0 + h) * ldb + n0 + i);
    const int astep = TA ? 2 * lda : 2;
    const int bstep = TB ? 2 : 2 * ldb;
    float a0[4], b0[4], a1[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0[u] = ap[u * astep];
      b0[u] = bp[u * bstep];
    }
    for (int s = 0; s < klen; s += 8) {
      ap += 4 * astep;
      bp += 4 * bstep;
      if (s + 8 < klen) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a1[u] = ap[u * astep];
          b1[u] = bp[u * bstep];
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ds_reads ahead of this group's MFMAs
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0[u] = a1[u];
        b0[u] = b1[u];
      }
    }
  } else {
    const int col = n0 + i;
    for (int k = k0; k < k0 + klen; ++k) {  // per output element: the same k-ordered fmaf chain as the MFMA
      const float b = TB ? B[col * ldb + k] : B[k * ldb + col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + drow(r, h);
        const float a = TA ? A[k * lda + row] : A[row * lda + k];
        acc[r] = __builtin_fmaf(a, b, acc[r]);
      }
    }
  }
  return acc;
}

// ---- 16x16x4 f32 MFMA tile step ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
// D[4g+r][c] += sum_k A[c'][k] B[k][c]: lane (c = lane&15, g = lane>>4) supplies a = A[c][g], b = B[g][c].
// VALU restatement (gemm_mode 1): the same lanes' operands fetched with ds_bpermute, k-ordered fmaf chain.
template <bool VALU>
__device__ __forceinline__ f32x4 mma16(float a, float b, f32x4 acc, int lane) {
  if constexpr (!VALU) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  } else {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float bk = __shfl(b, c + 16 * k, 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(__shfl(a, 4 * g + r + 16 * k, 64), bk, acc[r]);
    }
    return acc;
  }
}

// ---- XOR-swizzled [rows][64] LDS matrices for the 16x16x4 fragments ----------------------------------------------------
// element (row, col) sits at row*64 + (col ^ swz(row)): conflict-free for "16 rows x 2 columns" (A operands), "2 rows x 16
// columns" (B / transposed-A operands) and the C/D shape "rows 4g+r x 16 columns" (see ph_ppo_rp.hip)
__device__ __forceinline__ int swz(int row) { return ((2 * row) & 62) ^ (16 * ((row ^ (row >> 2)) & 1)); }
__device__ __forceinline__ int sidx(int row, int col) { return row * 64 + (col ^ swz(row)); }

// ---- counter-based RNG: Philox4x32-10 --------------------------------------------------------------------
__device__ __host__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
// four uniforms in [0,1) with 24 random bits each for (seed, counter, row, block): all four words of one Philox call
__device__ __host__ inline void philox_uniform4(uint64_t seed, uint64_t counter, uint32_t row, uint32_t block, float (&u)[4]) {
  uint32_t c[4] = {row, block, (uint32_t)counter, (uint32_t)(counter >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}
// uniform in [0,1) with 24 random bits for (seed, counter, row, component)
__device__ __host__ inline float philox_uniform(uint64_t seed, uint64_t counter, uint32_t row, uint32_t comp) {
  uint32_t c[4] = {row, comp, (uint32_t)counter, (uint32_t)(counter >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

// ---- keyed pseudo-random permutation of [0,n): 4-round balanced Feistel + cycle walking ----------------------
// Replaces np.random.permutation(T*E) (SB3 RolloutBuffer.get) when the caller passes no explicit index array:
// no index buffer in HBM, no sort.  hb = half-width in bits, 2^(2*hb) >= n.
__device__ __host__ inline uint32_t feistel_round_fn(uint32_t r, uint64_t key, uint32_t round) {
  uint32_t x = r * 0x9E3779B1u + (uint32_t)key + round * 0x85EBCA6Bu;
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  x ^= (uint32_t)(key >> 32);
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
  return x;
}
// `hb` packs the two half widths: left a = hb & 0xff bits, right b = hb >> 8 bits, a + b = ceil(log2 n) (feistel_half_bits).
// Round 4: an ALTERNATING network over exactly ceil(log2 n) bits -- L ^= F(R), R ^= F(L), L ^= F(R), R ^= F(L): every round is
// an involution-like XOR of one half with a function of the other, hence a bijection for ANY pair of widths -- instead of the
// balanced network over 2 ceil(bits / 2) bits it replaces: the walk domain is < 2 n instead of < 4 n, and for n a power of two
// (the bench: 2^17 rows) there is NO cycle walk at all, where a 64-lane wave used to run ~7 iterations (the slowest lane's).
__device__ __host__ inline uint32_t feistel_perm(uint32_t i, uint32_t n, uint32_t hb, uint64_t key) {
  const uint32_t la = hb & 0xffu, lb = hb >> 8;
  const uint32_t mask_l = (1u << la) - 1u, mask_r = (1u << lb) - 1u;
  uint32_t x = i;
  do {
    uint32_t l = x >> lb, r = x & mask_r;
    l ^= feistel_round_fn(r, key, 0u) & mask_l;
    r ^= feistel_round_fn(l, key, 1u) & mask_r;
    l ^= feistel_round_fn(r, key, 2u) & mask_l;
    r ^= feistel_round_fn(l, key, 3u) & mask_r;
    x = (l << lb) | r;
  } while (x >= n);
  return x;
}
__host__ inline uint32_t feistel_half_bits(uint32_t n) {
  uint32_t bits = 1;
  while ((1ull << bits) < (uint64_t)n) ++bits;
  if (bits < 2) bits = 2;                       // both halves at least one bit wide
  const uint32_t la = bits / 2, lb = bits - la;
  return la | (lb << 8);
}

// debug phase stamp: lane 0 of the workgroup records the shader clock (no-op when prof == nullptr)
#define PH_STAMP(prof, slot)                                                                               \
  do {                                                                                                     \
    if ((prof) != nullptr && threadIdx.x == 0)                                                             \
      (prof)[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (slot)] = (long long)clock64();          \
  } while (0)

// ---- spec resolved for kernels ---------------------------------------------------------------------------------
struct NetDims {
  int obs_kind;     // PH_SPACE_*
  int D, F, A, L;   // stored obs len, features, stored action len, logits
  int Lp;           // logits padded to a multiple of 32 (MFMA N tile)
  int head16;       // 1: at most 4 action components of at most 16 logits each (the gradient kernel's per-component head phase)
  int nchunk;       // ceil(F / 64) feature chunks of the first layer
  const int* obs_off;  // device: prefix sums of obs nvec (D+1) for the one-hot path, else nullptr
  const int* act_off;  // device: prefix sums of act nvec (A+1)
  const int* slab_map; // device: slab position -> parameter index when the spec runs on register-order slabs, else nullptr
  const int* slab_map_split;  // device: the same for ppo_grad_split_kernel's accumulator order, or nullptr (spec not eligible)
  const int* wimage_map;      // device: parameter -> weight-image elements of the split kernel (ph_split.h), or nullptr
  int split;           // this call's gradient kernel under gemm_mode 2 (slab_map then IS slab_map_split): 0 none (exact f32), 1 ppo_grad_split_kernel
                       // (Box observations <= 64 features, one small Discrete head), 2 ppo_grad_split_oh_kernel (one-hot observations)
  int split_kind;      // which of the two the SPEC is eligible for (0 neither): what `split` becomes when gemm_mode 2 is asked for
  int slab_len_split;  // floats per gradient slab in that kernel's accumulator order
  int wimage_elems;    // bf16 elements of its weight fragment image
  int spec_id;         // the spec-cache entry this record was resolved from (> 0; identifies a spec across calls)
  int gauss;           // 1: Box action space -- DiagGaussian head (SB3 DiagGaussianDistribution): the A head outputs are the means, log_std[A]
                       // sits behind val_b in the parameter vector (lay.val_b + 1); 0 (also of a zeroed record): the categorical family
  ph_layout lay;
};

// ---- global -> LDS staging, split into issue (all global loads of a lane, no waits) and commit (LDS stores) ------------
// A workgroup issues every staging load of a phase back to back and commits afterwards, so a phase costs ONE memory
// latency; loads issued before an MFMA phase and committed after it cost none.

// X tile: features [c*64, c*64+64) of the R rows whose physical row index is rowphys[r] (-1 = padding -> zeros).
// Box: straight copy, one 256-byte row segment per wave-instruction.  Discrete family: one-hot (built at commit).
template <int R, int NT>
struct XStage {
  static constexpr int ITERS = R * HID / NT;
  float v[ITERS];
  const int* feat_lds = nullptr;   // [R][D] hot feature row of every (row, component) of the tile, or null
  // Discrete family: the one-hot position of every (row, component) of the tile, once per tile -- the feature chunks are then
  // built without touching global memory (a chunk built straight from the observations costs a dependent round trip per
  // 256 elements, per chunk, in the forward pass and again for dW1).  All loads of a batch of 8 are in flight together.
  __device__ __forceinline__ void build_feat(int* feat, const int* rowphys, const float* obs, const NetDims& nd, int tid) {
    const int total = R * nd.D;
    for (int e0 = 0; e0 < total; e0 += 8 * NT) {
      int fv[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + tid + NT * i;
        fv[i] = -(1 << 20);
        if (e < total) {
          const int r = e / nd.D, comp = e - r * nd.D;
          const int ph_row = rowphys[r];
          if (ph_row >= 0) {
            const int lo = nd.obs_off[comp], n = nd.obs_off[comp + 1] - lo;
            int x = (int)obs[(size_t)ph_row * nd.D + comp];
            x = x < 0 ? 0 : (x >= n ? n - 1 : x);
            fv[i] = lo + x;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + tid + NT * i;
        if (e < total) feat[e] = fv[i];
      }
    }
    feat_lds = feat;
  }
  // feature -> observation component table for commit_onehot: fcomp[f] for f < nchunk * 64 (features >= F map to component 0,
  // whose hot position can never equal them).  Two global loads per thread, then LDS writes; caller barriers afterwards.
  __device__ __forceinline__ static void build_fcomp(int* fcomp, const NetDims& nd, int tid) {
    for (int comp = tid; comp < nd.D; comp += NT) {
      const int lo = nd.obs_off[comp], hi = nd.obs_off[comp + 1];
      for (int f = lo; f < hi; ++f) fcomp[f] = comp;
    }
    for (int f = nd.F + tid; f < nd.nchunk * HID; f += NT) fcomp[f] = 0;
  }
  // One-hot chunk c straight from the tile's hot positions: lane column kk = feature c*64 + kk belongs to ONE component, so
  // X[r][kk] = (hot position of that component in row r == feature).  No zero fill, no scatter, no barrier inside.
  __device__ __forceinline__ void commit_onehot(float* dst, const int* fcomp, const NetDims& nd, int c, int tid) const {
    const int kk = tid & 63, f = c * HID + kk;
    const int* fr = feat_lds + fcomp[f];
    int hot[ITERS];   // all reads first: the compiler cannot tell that dst and the hot positions never alias
#pragma unroll
    for (int i = 0; i < ITERS; ++i) hot[i] = fr[((tid + NT * i) >> 6) * nd.D];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < ITERS; ++i) dst[((tid + NT * i) >> 6) * LDH + kk] = (hot[i] == f) ? 1.f : 0.f;
  }
  __device__ __forceinline__ void issue(const int* rowphys, const float* obs, const NetDims& nd, int c, int tid = -1) {
    if (nd.obs_kind != PH_SPACE_BOX) return;
    if (tid < 0) tid = threadIdx.x;
    const int kk = tid & 63, f = c * HID + kk;
    const bool fok = f < nd.F;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int ph_row = rowphys[(tid + NT * i) >> 6];
      v[i] = (ph_row >= 0 && fok) ? obs[(size_t)ph_row * nd.D + f] : 0.f;
    }
  }
  // caller: dst must be free (barrier before), and a barrier must follow before dst is read
  __device__ __forceinline__ void commit(float* dst, const int* rowphys, const float* obs, const NetDims& nd, int c, int tid = -1) {
    if (tid < 0) tid = threadIdx.x;
    if (nd.obs_kind == PH_SPACE_BOX) {
      const int kk = tid & 63;
#pragma unroll
      for (int i = 0; i < ITERS; ++i) dst[((tid + NT * i) >> 6) * LDH + kk] = v[i];
      return;
    }
    if (feat_lds) {   // hot feature rows precomputed once per tile (build_feat): the chunk is built from LDS alone
      const int base = c * HID;
      for (int e = tid; e < R * HID; e += NT) dst[(e >> 6) * LDH + (e & 63)] = 0.f;
      __syncthreads();
      for (int e = tid; e < R * nd.D; e += NT) {
        const int f = feat_lds[e] - base;   // padding rows hold a large negative value
        if (f >= 0 && f < HID) dst[(e / nd.D) * LDH + f] = 1.f;
      }
      return;
    }
    const int base = c * HID;
    for (int e = tid; e < R * HID; e += NT) dst[(e >> 6) * LDH + (e & 63)] = 0.f;
    __syncthreads();
    for (int e = tid; e < R * nd.D; e += NT) {
      const int r = e / nd.D, comp = e - r * nd.D;
      const int ph_row = rowphys[r];
      if (ph_row < 0) continue;
      const int lo = nd.obs_off[comp], n = nd.obs_off[comp + 1] - lo;
      int x = (int)obs[(size_t)ph_row * nd.D + comp];
      x = x < 0 ? 0 : (x >= n ? n - 1 : x);
      const int f = lo + x - base;
      if (f >= 0 && f < HID) dst[r * LDH + f] = 1.f;
    }
  }
};

// rows [row0, row0+64) of an input-major weight matrix W[nrows_total][64] -> dst[64][LDH]; rows >= nrows_total are
// zero.  16-byte global loads (W rows are 256-byte aligned relative to the 16-byte aligned parameter vector).
template <int NT>
struct WStage {
  static constexpr int ITERS = HID * HID / 4 / NT;
  float4 v[ITERS];
  // row63 != null: row 63 of the block comes from that 64-float vector instead (a bias riding as the last weight row)
  __device__ __forceinline__ void issue(const float* W, int row0, int nrows_total, int tid = -1, const float* row63 = nullptr) {
    if (tid < 0) tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int q = tid + NT * i;  // float4 index inside the 64x64 block
      const int k = row0 + (q >> 4);
      const float* src = (k < nrows_total) ? W + (size_t)k * HID + ((q & 15) << 2)
                                           : ((row63 && (q >> 4) == HID - 1) ? row63 + ((q & 15) << 2) : nullptr);
      v[i] = src ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void commit(float* dst, int tid = -1) {
    if (tid < 0) tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < ITERS; ++i) {
      const int q = tid + NT * i;
      float* d = dst + (q >> 4) * LDH + ((q & 15) << 2);
      d[0] = v[i].x; d[1] = v[i].y; d[2] = v[i].z; d[3] = v[i].w;
    }
  }
};

// act_W[64][L] -> dst[64][ldo] (columns >= L zero): lane tid owns hidden row j = tid % 64 and a column slice.
template <int NT, int LPMAX = PH_MAX_LOGITS>
struct WoStage {
  static constexpr int PARTS = NT / 64;       // column slices
  static constexpr int MAXC = LPMAX / PARTS;  // columns per lane, upper bound
  float v[MAXC];
  __device__ __forceinline__ void issue(const float* W, int L, int Lp, int tid = -1) {
    if (tid < 0) tid = threadIdx.x;
    const int j = tid & 63, c0 = (tid >> 6) * (Lp / PARTS), per = Lp / PARTS;
#pragma unroll
    for (int u = 0; u < MAXC; ++u) v[u] = (u < per && c0 + u < L) ? W[j * L + c0 + u] : 0.f;
  }
  __device__ __forceinline__ void commit(float* dst, int Lp, int ldo, int tid = -1) {
    if (tid < 0) tid = threadIdx.x;
    const int j = tid & 63, c0 = (tid >> 6) * (Lp / PARTS), per = Lp / PARTS;
#pragma unroll
    for (int u = 0; u < MAXC; ++u)
      if (u < per) dst[j * ldo + c0 + u] = v[u];
  }
};

}  // namespace ph

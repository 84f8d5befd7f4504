// flash_attn.hip -- fused attention core for gfx950: ViT windowed / global attention with decomposed relative-position
// bias and the two softmax(QK^T)V products of the bi-directional VL fusion (row a9).
//
//     out[i,:] = softmax_j( clamp(scale * q_i.k_j) + bias_h[i, j / kw] + bias_w[i, j % kw] + mask_j ) . v_j
//
// Nothing of size Nq x Nk is ever written to HBM (the reference materialises the fp32 score tensor: 1.07 GB per image per
// global block, backbone/vit.py:74-79; >= 4 tensors of (B*8, Nv, L) in fuse_helper.py:69-115).
//
// Structure (one workgroup = 4 waves = 128 queries of one (batch, head); wave w owns 32 queries):
//   * S^T = K . Q^T with v_mfma_f32_32x32x16 (A = K tile from LDS, B = Q fragments held in registers for the whole kernel).
//     "Swapped" product: lane l owns query (l & 31) and holds 16 scores per 32-key block, so the row max / row sum of the
//     online softmax are in-register reductions plus ONE cross-half exchange (lanes l <-> l+32).
//   * O^T = V^T . P^T: P is converted to 16 bit in registers and used directly as the B operand (its (half, j) register
//     order defines the k order of the contraction); the matching V^T A operand is fetched from the row-major V tile in
//     LDS with the hardware transpose read ds_read_b64_tr_b16 (csrc/mfma.h).  O^T keeps the query on the lane index,
//     so the per-row rescale by exp(m_old - m_new) is a plain per-lane multiply.
//   * K/V tiles are streamed HBM -> registers -> LDS, double-buffered in LDS with a register prefetch of the next tile
//     issued before the MFMA work of the current one (one barrier per tile).
//   * decomposed rel-pos bias: one key tile == one key ROW of the token grid (kw keys, padded to a multiple of 32 and
//     masked), so bias_h is a per-(query, tile) scalar and bias_w[i, :] is loaded ONCE into registers in the S^T
//     register order and reused by every tile.
//   * key validity (ragged tail, padded row, text mask) is folded into one 64-bit ballot per tile.
//   * block -> (batch*head, q-tile) mapping is XCD-aware: the q-tiles that share one head's K/V run on the same XCD so
//     K/V are fetched from HBM once and re-read from that XCD's L2.
// Numerics: scores, softmax statistics and both accumulators are fp32; q, k, v and the probabilities P enter the MFMAs
// as bf16 or fp16.
#include <stdlib.h>

#include "common.h"
#include <algorithm>

#include "mfma.h"

namespace hipie {

struct FAParams {
  const void* q; const void* k; const void* v; void* out;
  int B, H, Nq, Nk;
  long q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
  const float* bias_h; const float* bias_w; int kh, kw;
  const void* tab_h; const void* tab_w;          // FUSEREL: rel-pos tables (2*kh-1, HD), (2*kw-1, HD) in the operand dtype
  const uint8_t* key_mask;
  float scale, clamp, defer;
  int nqt, ntiles, swz, prio;
  int k_cs;                     // elements between consecutive 8-element chunks of a K row: 8, or 16 with HIPIE_K_HL8_HI (the hi halves of an HL8 row)
  int out_f32;                  // HIPIE_OUT_F32: `out` is fp32 (strides in fp32 elements) -- no 16-bit rounding of the attention output
};

constexpr float kLog2e = 1.4426950408889634f;

__device__ __forceinline__ float max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

constexpr float kDefer = 8.f;                 // log2 of the largest P the deferred running max lets through

// T: bf16_t | f16_t;  HD: head dim;  NB: 32-key blocks per tile;  QB: 32-query blocks per wave;  BIAS: decomposed rel-pos
// bias (tile == key row);  CLAMP: clamp scale*q.k to +-clamp.
// One wave per SIMD (4 waves, up to 512 registers each): with QB = 2 every K / V^T fragment fetched from LDS feeds two
// MFMAs and the two query blocks give the scheduler independent MFMA and softmax streams to overlap.
// FUSEREL: the decomposed rel-pos bias is computed in the prologue from the tables (add_decomposed_rel_pos,
// backbone/utils.py:96-125): bias_w[q, kx] = q . Rw[qx - kx + kw - 1] and bias_h[q, ky] = q . Rh[qy - ky + kh - 1] are
// two small MFMA products per wave (table rows x the Q fragments already in registers), so the (B*heads, N, kh + kw) fp32
// bias tensors are never written to or read from HBM.  Requires kw % 32 == 0 (a wave's 32 queries share one grid row).
template <typename T, int HD, int NB, int QB, int WAVES, bool BIAS, bool CLAMP, bool MASKED, bool FUSEREL = false>
__global__ __launch_bounds__(WAVES * 64, ((WAVES == 8 || HD <= 80) && QB == 1) ? 2 : 1) void flash_attn_kernel(const FAParams p) {
  constexpr int KT = 32 * NB;                  // keys per tile
  constexpr int KS = HD / 16;                  // k16 steps of QK^T
  constexpr int DB = (HD + 31) / 32;           // 32-row d blocks of O^T
  constexpr int KSTR = HD + 8;                 // K tile row stride (elements): 16-B aligned, conflict-free b128 reads
  // V tile row stride: covers the padded d blocks and makes the 4 key rows of one ds_read_b64_tr_b16 land on disjoint
  // 16-bank windows: (VSTR / 2) % 64 in {16, 48}  (bank = dword address % 64, each row of a 32-lane group spans 16 banks)
  constexpr int VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;
  constexpr int CPR = HD / 8;                  // 16-byte chunks per row
  constexpr int NCH = KT * CPR;                // chunks per tile
  constexpr int NT = WAVES * 64;
  constexpr int CPT = (NCH + NT - 1) / NT;     // chunks per thread
  constexpr int BUF = KT * (KSTR + VSTR);      // elements per LDS buffer
  constexpr int NW = (KT + 63) / 64;           // validity words per tile
  constexpr int QPW = 32 * QB;                 // queries per wave
  // HD = 80 pads O^T to 96 rows: row HD of the padded block is free, so a column of ones at V[:, HD] makes the PV MFMAs
  // accumulate the softmax denominator  l = sum_k P[k]  there (from the same 16-bit-rounded P as the numerator) -- the 32
  // per-tile VALU adds of the row sum disappear.
  // fp16 is the parity policy's operand type: there the denominator stays an fp32 sum of the unrounded probabilities
  constexpr bool LTRICK = (HD % 32) != 0 && sizeof(T) == 2 && !__is_same(T, f16_t);
  constexpr int L_ROW = HD % 32, L_HI = (L_ROW >> 2) & 1, L_REG = (L_ROW & 3) + 4 * (L_ROW >> 3);
  typedef typename Mfma32<T>::frag frag;
  typedef typename Mfma32<T>::half_frag hfrag;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;

  // ---- block -> (bh, q tile) ----
  int bh, qt;
  {
    const int id = blockIdx.x;
    if (p.swz) { bh = (id & 7) + 8 * ((id >> 3) / p.nqt); qt = (id >> 3) % p.nqt; }
    else { bh = id / p.nqt; qt = id % p.nqt; }
  }
  const int b = bh / p.H, h = bh % p.H;
  const T* Qg = reinterpret_cast<const T*>(p.q) + b * p.q_sb + h * p.q_sh;
  const T* Kg = reinterpret_cast<const T*>(p.k) + b * p.k_sb + h * p.k_sh;
  const T* Vg = reinterpret_cast<const T*>(p.v) + b * p.v_sb + h * p.v_sh;
  T* Og = reinterpret_cast<T*>(p.out) + b * p.o_sb + h * p.o_sh;
  const uint8_t* Mg = p.key_mask ? p.key_mask + (long)b * p.Nk : nullptr;

  int qi[QB], qc[QB];                          // this lane's queries
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    qi[qb] = qt * (WAVES * QPW) + wave * QPW + 32 * qb + li;
    qc[qb] = min(qi[qb], p.Nq - 1);
  }

  // ---- Q fragments (B operand): lane (q = li, half hi) holds Q[q][16 ks + 8 hi + j] ----
  frag qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      qf[qb][ks] = *reinterpret_cast<const frag*>(Qg + (long)qc[qb] * p.q_st + 16 * ks + 8 * hi);

  constexpr bool BWL = BIAS && (NB <= 2) && (WAVES == 8);
  float bw[QB][NB][16];
  if constexpr (FUSEREL) {
    static_assert(BIAS && !BWL && !MASKED, "FUSEREL: full-tile bias variant only");
    const T* TH = reinterpret_cast<const T*>(p.tab_h);
    const T* TW = reinterpret_cast<const T*>(p.tab_w);
    // G[j][q] = Rw[qx0 + j] . q for the 32 + kw - 1 table rows a 32-query block can touch; staged through LDS because the row
    // a lane needs (j = li + kw - 1 - kx) sits in another register / lane half of the MFMA output
    constexpr int GROWS = 32 * (NB + 1);
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * (GROWS * 32);
    float* bh_all = reinterpret_cast<float*>(smem_raw + (size_t)2 * BUF * sizeof(T));      // [kh][WAVES*QPW]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q0 = min(qt * (WAVES * QPW) + wave * QPW + 32 * qb, p.Nq - 1);
      const int qx0 = q0 % p.kw;                                  // the block's 32 queries: one grid row, columns qx0 .. qx0 + 31
#pragma unroll
      for (int jb = 0; jb < NB + 1; ++jb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int row = min(qx0 + 32 * jb + li, 2 * p.kw - 2);
        const T* ap = TW + (long)row * HD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = Mfma32<T>::mma(*reinterpret_cast<const frag*>(ap + 16 * ks), qf[qb][ks], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[(32 * jb + crow(r, hi)) * 32 + li] = acc[r];
      }
      __syncthreads();
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kx = min(32 * blk + crow(r, hi), p.kw - 1);
          bw[qb][blk][r] = stage[(li + (p.kw - 1) - kx) * 32 + li] * kLog2e;
        }
      __syncthreads();             // the stage is rewritten by the next query block / overlapped by bh_all below
    }
    // H[ky][q] = Rh[qy - ky + kh - 1] . q for every key row: the per-(query, tile) scalar of the main loop
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      const int q0 = min(qt * (WAVES * QPW) + wave * QPW + 32 * qb, p.Nq - 1);
      const int qy = q0 / p.kw;
      for (int kb = 0; kb < (p.kh + 31) / 32; ++kb) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int row = min(max(qy - (32 * kb + li) + p.kh - 1, 0), 2 * p.kh - 2);
        const T* ap = TH + (long)row * HD + 8 * hi;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) acc = Mfma32<T>::mma(*reinterpret_cast<const frag*>(ap + 16 * ks), qf[qb][ks], acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ky = 32 * kb + crow(r, hi);
          if (ky < p.kh) bh_all[ky * (WAVES * QPW) + wave * QPW + 32 * qb + li] = acc[r] * kLog2e;
        }
      }
    }
  }

  // zero LDS once: the pad columns of the V tile feed the padded d rows of O^T (discarded, but keep them NaN-free)
  for (int i = tid; i < 2 * BUF / 8; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if (LTRICK) {
    __syncthreads();
    for (int i = tid; i < 2 * KT; i += NT) smem[(i / KT) * BUF + KT * KSTR + (i % KT) * VSTR + HD] = (T)1.0f;   // never overwritten by tile stores
  }

  // ---- bias_w in S^T register order, pre-multiplied by log2(e) (the softmax runs in the exp2 domain) ----
  // bias_w rows: in LDS for the 8-wave workgroup (one workgroup per CU; frees 16*NB VGPRs so two waves fit per SIMD); in
  // registers for 4-wave workgroups, whose smaller LDS footprint lets TWO independent workgroups share a CU
  constexpr int BWS = 32 * NB + 4;            // bias_w LDS row stride (floats): 16-B aligned rows, conflict-free b128 reads
  // Inside the tile loop the ONLY vector-memory traffic is the prefetch of the next tile (K, V and the 4 x 32 x WAVES
  // bias_h values of that key row, which arrive transposed as (BH, kh, Nq) so they are one coalesced line): no other
  // s_waitcnt vmcnt can drain it early (vmcnt retires in order, and a loaded value carried across the loop back-edge
  // makes the compiler wait vmcnt(0)).  bias_w rows and the key mask are staged in LDS once.
  float* bh_lds = reinterpret_cast<float*>(smem_raw + (size_t)2 * BUF * sizeof(T));      // [2][WAVES*QPW]
  float* bw_lds = bh_lds + 2 * WAVES * QPW;                                              // [WAVES*QPW][BWS]
  uint8_t* mk_lds = reinterpret_cast<uint8_t*>(bw_lds + (BWL ? WAVES * QPW * BWS : 0));
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    if (BIAS && !BWL && !FUSEREL) {
      const float* bwp = p.bias_w + ((long)bh * p.Nq + qc[qb]) * p.kw;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) bw[qb][blk][r] = bwp[min(32 * blk + crow(r, hi), p.kw - 1)] * kLog2e;
    }
  }
  if (BWL) {
    for (int r = 0; r < QPW; ++r) {
      const int q = min(qt * (WAVES * QPW) + wave * QPW + r, p.Nq - 1);
      const float* srcw = p.bias_w + ((long)bh * p.Nq + q) * p.kw;
      for (int j = lane; j < KT; j += 64) bw_lds[(wave * QPW + r) * BWS + j] = (j < p.kw ? srcw[j] : 0.f) * kLog2e;
    }
  }
  if (MASKED && Mg != nullptr)
    for (int i = tid; i < p.Nk; i += NT) mk_lds[i] = Mg[i];
  // this thread's slot of the per-tile bias_h line
  const float* bhsrc = (BIAS && !FUSEREL) ? p.bias_h + (long)bh * p.kh * p.Nq + min(qt * (WAVES * QPW) + tid, p.Nq - 1) : nullptr;
  const bool bh_thread = BIAS && !FUSEREL && (tid < WAVES * QPW);
  float bhr = 0.f;
  const float c1 = p.scale * kLog2e;
  const float cl2 = p.clamp * kLog2e;

  // ---- tile streaming: HBM/L2 -> registers -> LDS.  Per-thread chunk coordinates are tile-invariant and hoisted; the
  //      register arrays are native vectors indexed with compile-time constants and written unconditionally (clamped
  //      duplicates for the tail) so that they stay in VGPRs and the loads stay in flight across the MFMA work. ----
  int koff[CPT], voff[CPT], lk[CPT], lv[CPT], crow_[CPT];
  bool cval[CPT];
  const int nkeys_full = BIAS ? p.kw : KT;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int idx = min(c * NT + tid, NCH - 1);
    const int row = idx / CPR, ch = idx - row * CPR;
    cval[c] = (c * NT + tid) < NCH;
    crow_[c] = row;
    const int rowc = min(row, nkeys_full - 1);             // padded rows re-read the last valid key (finite data, P = 0)
    koff[c] = rowc * (int)p.k_st + ch * p.k_cs;
    voff[c] = rowc * (int)p.v_st + ch * 8;
    lk[c] = row * KSTR + ch * 8;
    lv[c] = row * VSTR + ch * 8;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in VGPRs (HIP's uint4 struct does not)
  u32x4 kr[CPT], vr[CPT];
#define FA_LOAD_REGS(t_)                                                                              \
  {                                                                                                   \
    const long key0_ = BIAS ? (long)(t_) * p.kw : (long)(t_) * KT;                                    \
    const int nk_ = BIAS ? p.kw : min(KT, p.Nk - (int)key0_);                                         \
    const T* kb_ = Kg + key0_ * p.k_st;                                                               \
    const T* vb_ = Vg + key0_ * p.v_st;                                                               \
    if (bh_thread) bhr = bhsrc[(long)(t_) * p.Nq];                                                    \
    if (nk_ == nkeys_full) {                                                                          \
      _Pragma("unroll") for (int c = 0; c < CPT; ++c) {                                               \
        kr[c] = *reinterpret_cast<const u32x4*>(kb_ + koff[c]);                                       \
        vr[c] = *reinterpret_cast<const u32x4*>(vb_ + voff[c]);                                       \
      }                                                                                               \
    } else { /* ragged last tile (no-bias mode): clamp the row per chunk */                            \
      _Pragma("unroll") for (int c = 0; c < CPT; ++c) {                                               \
        const int rc_ = min(crow_[c], nk_ - 1);                                                       \
        const int ch8_ = lk[c] - crow_[c] * KSTR;                                                     \
        kr[c] = *reinterpret_cast<const u32x4*>(kb_ + (long)rc_ * p.k_st + (ch8_ >> 3) * p.k_cs);     \
        vr[c] = *reinterpret_cast<const u32x4*>(vb_ + (long)rc_ * p.v_st + ch8_);                     \
      }                                                                                               \
    }                                                                                                 \
  }
#define FA_STORE_LDS(buf_)                                                                            \
  {                                                                                                   \
    T* Ks_ = smem + (buf_) * BUF;                                                                     \
    T* Vs_ = Ks_ + KT * KSTR;                                                                         \
    if (bh_thread) bh_lds[(buf_) * (WAVES * QPW) + tid] = bhr * kLog2e;                                \
    _Pragma("unroll") for (int c = 0; c < CPT; ++c) {                                                 \
      if (cval[c]) {                                                                                  \
        *reinterpret_cast<u32x4*>(Ks_ + lk[c]) = kr[c];                                               \
        *reinterpret_cast<u32x4*>(Vs_ + lv[c]) = vr[c];                                               \
      }                                                                                               \
    }                                                                                                 \
  }

  // ---- online softmax state (exp2 domain) ----
  f32x16 O[QB][DB];
  float m_run[QB], l_run[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    m_run[qb] = -INFINITY;
    l_run[qb] = 0.f;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[qb][d][r] = 0.f;
  }

  int nt = p.ntiles;
  // MASKED is instantiated only where some key can be invalid: padded key rows (BIAS, kw % 32 != 0), a ragged last tile
  // or a key mask.  The full-tile instantiation contains no validity code at all.
  __syncthreads();                 // LDS zero fill / mask staging done
  if (MASKED && !BIAS && Mg != nullptr) {
    // key tiles after the last attended key contribute exp(-inf) = 0: skip them.  (The shipped eval setting pads every
    // caption to 4096 tokens, of which ~200 are real: the image->text attention then walks 7 tiles instead of 128.)
    int* s_last = reinterpret_cast<int*>(mk_lds + ((size_t)p.Nk + 15) / 16 * 16);
    if (tid == 0) *s_last = -1;
    __syncthreads();
    int last = -1;
    for (int i = tid; i < p.Nk; i += NT)
      if (mk_lds[i] != 0) last = i;
    if (last >= 0) atomicMax(s_last, last);
    __syncthreads();
    nt = max(1, min(nt, (*s_last + KT) / KT));
  }
  FA_LOAD_REGS(0);
  FA_STORE_LDS(0);
  __syncthreads();
  if (nt > 1) FA_LOAD_REGS(1);
  const int l16 = lane & 15, g1 = (lane >> 4) & 1;

  for (int t = 0; t < nt; ++t) {
    const T* Ks = smem + (t & 1) * BUF;
    const T* Vs = Ks + KT * KSTR;
    const int key0 = BIAS ? t * p.kw : t * KT;
    const int nkeys = BIAS ? p.kw : min(KT, p.Nk - key0);
    float bh_t[QB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      bh_t[qb] = BIAS ? bh_lds[(FUSEREL ? t : (t & 1)) * (WAVES * QPW) + wave * QPW + 32 * qb + li] : 0.f;
    }

    // ---- S^T = K . Q^T  (each K fragment feeds QB MFMAs) ----
    if (p.prio) __builtin_amdgcn_s_setprio(1);     // MFMA clusters at raised priority: the co-resident workgroup's VALU yields
    f32x16 S[QB][NB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[qb][blk][r] = 0.f;
    {
      // all K fragments of a chunk of k-steps are requested from LDS first and the MFMAs follow behind a scheduling
      // barrier: left to itself the compiler sinks every ds_read_b128 next to its MFMA (one register quad, lgkmcnt(0) before
      // each MFMA), which turns the QK^T phase into a chain of exposed LDS latencies
      const T* kbase = Ks + li * KSTR + 8 * hi;
      constexpr int KCH = (KS * NB <= 12) ? KS : 4;
#pragma unroll
      for (int k0 = 0; k0 < KS; k0 += KCH) {
        frag kf[NB][KCH];
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk)
            if (k0 + kk < KS) kf[blk][kk] = *reinterpret_cast<const frag*>(kbase + 32 * blk * KSTR + 16 * (k0 + kk));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < KCH; ++kk)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
              if (k0 + kk < KS) S[qb][blk] = Mfma32<T>::mma(kf[blk][kk], qf[qb][k0 + kk], S[qb][blk]);
      }
    }

    if (p.prio) __builtin_amdgcn_s_setprio(0);
    unsigned long long vw[NW];
    if (MASKED) {
      // validity words: bit i of word w <=> key (64 w + i) of this tile is attended to
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int kk = 64 * w + lane;
        bool ok = kk < nkeys;
        if (Mg != nullptr && ok) ok = mk_lds[key0 + kk] != 0;
        vw[w] = __ballot(ok);
      }
    }

#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      // ---- s' = log2e * (clamp(scale * qk) + bias_w); the tile-constant bias_h is folded into the exponent offset ----
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
          if (BWL) b4 = *reinterpret_cast<const f32x4*>(bw_lds + (wave * QPW + 32 * qb + li) * BWS + 32 * blk + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * g + e;
            float s = S[qb][blk][r] * c1;
            if (CLAMP) s = fminf(fmaxf(s, -cl2), cl2);
            if (BIAS) s += BWL ? b4[e] : bw[qb][blk][r];
            S[qb][blk][r] = s;
          }
        }
      }
      if (MASKED) {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int kk = 32 * blk + crow(r, hi);
            const bool ok = (vw[kk >> 6] >> (kk & 63)) & 1ull;
            S[qb][blk][r] = ok ? S[qb][blk][r] : -INFINITY;
          }
        }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = max3(mx, S[qb][blk][r], S[qb][blk][r + 1]);
      mx = fmaxf(mx, __shfl_xor(mx, 32)) + bh_t[qb];
      // deferred max: the reference point m_run only moves when some row's maximum grew by more than 2^kDefer (or is
      // still unset); otherwise P = exp2(s - m_run) <= 2^kDefer stays well inside fp32 / 16-bit range and the O rescale
      // (and its exp2) is skipped for the whole wave -- with 32 rows per wave a plain running max moves in most tiles.
      const bool grow = mx > m_run[qb] + p.defer;                        // m_run = -inf: true as soon as a key is valid
      if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
        const float m_new = fmaxf(m_run[qb], mx);
        const float m_ref = (m_new == -INFINITY) ? 0.f : m_new;         // all keys so far masked: keep exp2() finite
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_ref);  // m_run = -inf -> 0
        m_run[qb] = m_new;
        if (!LTRICK) l_run[qb] *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) O[qb][d][r] *= alpha;
      }
      const float off = bh_t[qb] - ((m_run[qb] == -INFINITY) ? 0.f : m_run[qb]);
      float lsum = 0.f;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(S[qb][blk][r] + off);
          S[qb][blk][r] = pv;
          if (!LTRICK) lsum += pv;
        }
      if (!LTRICK) l_run[qb] += lsum;
    }

    // ---- O^T += V^T . P^T  (each V^T fragment feeds QB MFMAs) ----
    if (p.prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        frag pf[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qb][j] = (T)S[qb][blk][8 * s + j];
        const int krow0 = 32 * blk + 16 * s + 4 * hi;      // keys of this lane-half for k slots 0..3 (and +8 for 4..7)
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          const T* a0 = Vs + (krow0 + (l16 >> 2)) * VSTR + 32 * d + 16 * g1 + 4 * (l16 & 3);
          const hfrag lo = Mfma32<T>::tr_read(a0);
          const hfrag hi4 = Mfma32<T>::tr_read(a0 + 8 * VSTR);
          frag vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi4[j]; }
#pragma unroll
          for (int qb = 0; qb < QB; ++qb) O[qb][d] = Mfma32<T>::mma(vf, pf[qb], O[qb][d]);
        }
      }
    }

    if (p.prio) __builtin_amdgcn_s_setprio(0);
    if (t + 1 < nt) FA_STORE_LDS((t + 1) & 1);
    __syncthreads();
    if (t + 2 < nt) FA_LOAD_REGS(t + 2);
  }

  // ---- epilogue ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l_tot = LTRICK ? __shfl(O[qb][DB - 1][L_REG], li + 32 * L_HI) : l_run[qb] + __shfl_xor(l_run[qb], 32);
    const float inv = 1.f / l_tot;
    if (qi[qb] < p.Nq && p.out_f32) {
      float* orow = reinterpret_cast<float*>(p.out) + b * p.o_sb + h * p.o_sh + (long)qi[qb] * p.o_st;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int d0 = 32 * d + 8 * rr + 4 * hi;
          if (d0 < HD)
            *reinterpret_cast<float4*>(orow + d0) = make_float4(O[qb][d][4 * rr] * inv, O[qb][d][4 * rr + 1] * inv, O[qb][d][4 * rr + 2] * inv,
                                                                O[qb][d][4 * rr + 3] * inv);
        }
      }
    } else if (qi[qb] < p.Nq) {
      T* orow = Og + (long)qi[qb] * p.o_st;
#pragma unroll
      for (int d = 0; d < DB; ++d) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int d0 = 32 * d + 8 * rr + 4 * hi;           // rows crow(4 rr + (0..3), hi) of block d
          if (d0 < HD) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            t4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (T)(O[qb][d][4 * rr + e] * inv);
            *reinterpret_cast<t4*>(orow + d0) = o4;
          }
        }
      }
    }
  }
}
#undef FA_LOAD_REGS
#undef FA_STORE_LDS

template <typename T, int HD, int NB, int QB, int WAVES, bool BIAS, bool CLAMP, bool MASKED, bool FUSEREL = false>
static int launch_fa(FAParams& p, hipStream_t st) {
  constexpr int KT = 32 * NB, DB = (HD + 31) / 32;
  constexpr int VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;
  size_t lds = (size_t)2 * KT * ((HD + 8) + VSTR) * sizeof(T);
  lds += (size_t)(FUSEREL ? p.kh : 2) * WAVES * 32 * QB * sizeof(float);
  if (FUSEREL && lds < (size_t)WAVES * 32 * (NB + 1) * 32 * sizeof(float)) lds = (size_t)WAVES * 32 * (NB + 1) * 32 * sizeof(float);
  if (BIAS && NB <= 2 && WAVES == 8) lds += (size_t)WAVES * 32 * QB * (32 * NB + 4) * sizeof(float);
  if (MASKED && p.key_mask != nullptr) lds += ((size_t)p.Nk + 15) / 16 * 16 + 16;     // mask bytes + the last-valid-key word
  if (lds > 160 * 1024) return set_err(HIPIE_EINVAL, "flash_attn: %zu bytes of LDS needed (Nk=%d) > 160 KiB", lds, p.Nk);
  p.nqt = (p.Nq + WAVES * 32 * QB - 1) / (WAVES * 32 * QB);
  p.ntiles = BIAS ? p.kh : (p.Nk + KT - 1) / KT;
  const unsigned grid = (unsigned)(p.nqt * p.B * p.H);
  auto kern = flash_attn_kernel<T, HD, NB, QB, WAVES, BIAS, CLAMP, MASKED, FUSEREL>;
  static size_t lds_set[64] = {0};     // per device: raise the dynamic-LDS limit once per instantiation (not a stream operation: keep it out of graph capture)
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || lds > lds_set[dev])) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = lds;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, p);
  return check_launch("flash_attn");
}

// NBN: key blocks per tile without bias (smaller for the wide head); W: waves per workgroup for the large-Nq case
template <typename T, int HD, int NBN, int W>
static int dispatch_nb(FAParams& p, hipStream_t st, bool wide) {
  if (p.tab_h != nullptr) {
    if constexpr (HD == 64 || HD == 80) {
      // QB = 2 (64 queries per wave, one wave per SIMD, 488 registers, no spills) was measured at 1.24 ms vs 0.99 ms for
      // QB = 1 x two waves per SIMD: a compiler-scheduled single wave does not overlap its own softmax with its MFMAs
      if (p.kw == 64 && p.kh >= 1 && p.kh <= 64) return launch_fa<T, HD, 2, 1, 4, true, false, false, true>(p, st);
    }
    return set_err(HIPIE_EINVAL, "vit_attn_fused: needs a 64-wide token grid with <= 64 rows and head_dim 64/80 (got %dx%d, hd %d)", p.kh, p.kw, HD);
  }
  if (p.bias_h != nullptr) {
    const int nb = (p.kw + 31) / 32;
    const bool full = (p.kw % 32) == 0;
    switch (nb) {
      case 1: return full ? launch_fa<T, HD, 1, 1, 4, true, false, false>(p, st) : launch_fa<T, HD, 1, 1, 4, true, false, true>(p, st);
      case 2:
        if (wide) return full ? launch_fa<T, HD, 2, 1, W, true, false, false>(p, st) : launch_fa<T, HD, 2, 1, W, true, false, true>(p, st);
        return full ? launch_fa<T, HD, 2, 1, 4, true, false, false>(p, st) : launch_fa<T, HD, 2, 1, 4, true, false, true>(p, st);
      case 3: return full ? launch_fa<T, HD, 3, 1, 4, true, false, false>(p, st) : launch_fa<T, HD, 3, 1, 4, true, false, true>(p, st);
      default: return set_err(HIPIE_EINVAL, "flash_attn: kw=%d > 96 unsupported with rel-pos bias", p.kw);
    }
  }
  const bool masked = (p.key_mask != nullptr) || (p.Nk % (32 * NBN) != 0);
  if (p.clamp > 0.f)
    return masked ? launch_fa<T, HD, NBN, 1, 4, false, true, true>(p, st) : launch_fa<T, HD, NBN, 1, 4, false, true, false>(p, st);
  return masked ? launch_fa<T, HD, NBN, 1, 4, false, false, true>(p, st) : launch_fa<T, HD, NBN, 1, 4, false, false, false>(p, st);
}

template <typename T>
static int dispatch_hd(FAParams& p, int hd, hipStream_t st, bool wide) {
  switch (hd) {
    case 32: return dispatch_nb<T, 32, 2, 4>(p, st, false);
    case 64: return dispatch_nb<T, 64, 2, 8>(p, st, wide);
    case 80: return dispatch_nb<T, 80, 2, 8>(p, st, wide);
    case 256: return dispatch_nb<T, 256, 1, 4>(p, st, false);
    default: return set_err(HIPIE_EINVAL, "flash_attn: head_dim %d unsupported (32, 64, 80, 256)", hd);
  }
}

static int flash_attn_impl(FAParams p, int hd, int dtype, void* stream) {
  HIPIE_REQUIRE(p.q && p.k && p.v && p.out, "flash_attn: null pointer");
  HIPIE_REQUIRE(p.B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "flash_attn: bad shape B=%d H=%d Nq=%d Nk=%d", p.B, p.H, p.Nq, p.Nk);
  HIPIE_REQUIRE((p.bias_h == nullptr) == (p.bias_w == nullptr), "flash_attn: bias_h and bias_w must be given together");
  if (p.bias_h || p.tab_h) HIPIE_REQUIRE(p.kh > 0 && p.kw > 0 && (long)p.kh * p.kw == p.Nk, "flash_attn: kh*kw=%d*%d != Nk=%d", p.kh, p.kw, p.Nk);
  const long strides[] = {p.q_sb, p.q_st, p.q_sh, p.k_sb, p.k_st, p.k_sh, p.v_sb, p.v_st, p.v_sh, p.o_sb, p.o_st, p.o_sh};
  for (long s : strides) HIPIE_REQUIRE(s % 8 == 0, "flash_attn: strides must be multiples of 8 elements (16 bytes), got %ld", s);
  HIPIE_REQUIRE((((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.out) & 15) == 0, "flash_attn: pointers must be 16-byte aligned");
  p.swz = ((p.B * p.H) % 8 == 0) ? 1 : 0;
  p.k_cs = (dtype & HIPIE_K_HL8_HI) ? 16 : 8;
  // the hi halves of an HL8 buffer are fp16: read as bf16 they would be garbage keys without any error
  HIPIE_REQUIRE(!(dtype & HIPIE_K_HL8_HI) || (dtype & 0xff) == HIPIE_F16, "flash_attn: HIPIE_K_HL8_HI needs fp16 operands (dtype %d)", dtype & 0xff);
  dtype &= ~HIPIE_K_HL8_HI;
  p.defer = ((dtype & ~HIPIE_OUT_F32) == HIPIE_F16) ? 0.f : kDefer;      // fp16 = parity policy: classic running max
  // diagnostic switches: read from the environment once per process
  { static const float defer_env = [] { const char* d = study_env("HIPIE_FA_DEFER"); return d ? (float)atof(d) : -1.f; }();
    if (defer_env >= 0.f) p.defer = defer_env; }
  { static int prio = -1; if (prio < 0) { const char* e = study_env("HIPIE_FA_PRIO"); prio = e ? atoi(e) : 1; } p.prio = prio; }   // +1.5 % (tools/bench_attn.py)
  // 8 waves (256 queries) per workgroup halve the K/V traffic per query but keep all waves of a CU in lockstep
  bool wide = false;       // measured: two independent 4-wave workgroups per CU (1.09 ms) beat one 8-wave workgroup (1.16 ms)
  static const int waves_env = [] { const char* e = study_env("HIPIE_FA_WAVES"); return (e && (e[0] == '4' || e[0] == '8')) ? e[0] - '0' : 0; }();
  if (waves_env) wide = (waves_env == 8);
  hipStream_t st = (hipStream_t)stream;
  p.out_f32 = (dtype & HIPIE_OUT_F32) ? 1 : 0;
  dtype &= ~HIPIE_OUT_F32;
  switch (dtype) {
    case HIPIE_F16: return dispatch_hd<f16_t>(p, hd, st, wide);
    case HIPIE_BF16: return dispatch_hd<bf16_t>(p, hd, st, wide);
    default: return set_err(HIPIE_EINVAL, "flash_attn: dtype must be HIPIE_F16 or HIPIE_BF16 (got %d)", dtype);
  }
}

}  // namespace hipie

extern "C" int hipie_flash_attn(const void* q, const void* k, const void* v, void* out, int B, int H, int Nq, int Nk,
                                int head_dim, int64_t q_sb, int64_t q_st, int64_t q_sh, int64_t k_sb, int64_t k_st,
                                int64_t k_sh, int64_t v_sb, int64_t v_st, int64_t v_sh, int64_t o_sb, int64_t o_st,
                                int64_t o_sh, const float* bias_h, const float* bias_w, int kh, int kw,
                                const uint8_t* key_mask, float scale, float clamp, int dtype, void* stream) {
  hipie::FAParams p{};
  p.q = q; p.k = k; p.v = v; p.out = out;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_sb = q_sb; p.q_st = q_st; p.q_sh = q_sh; p.k_sb = k_sb; p.k_st = k_st; p.k_sh = k_sh;
  p.v_sb = v_sb; p.v_st = v_st; p.v_sh = v_sh; p.o_sb = o_sb; p.o_st = o_st; p.o_sh = o_sh;
  p.bias_h = bias_h; p.bias_w = bias_w; p.kh = kh; p.kw = kw; p.key_mask = key_mask;
  p.scale = scale; p.clamp = clamp;
  return hipie::flash_attn_impl(p, head_dim, dtype, stream);
}

extern "C" int hipie_vit_attn(const void* qkv, const float* rel_h, const float* rel_w, void* out, int B, int gh, int gw,
                              int heads, int hd, float scale, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(qkv && out, "vit_attn: null pointer");
  const long N = (long)gh * gw, C = (long)heads * hd;
  const size_t es = 2;
  FAParams p{};
  p.q = qkv;
  p.k = (const char*)qkv + C * es;
  p.v = (const char*)qkv + 2 * C * es;
  p.out = out;
  p.B = B; p.H = heads; p.Nq = (int)N; p.Nk = (int)N;
  p.q_sb = p.k_sb = p.v_sb = N * 3 * C; p.q_st = p.k_st = p.v_st = 3 * C; p.q_sh = p.k_sh = p.v_sh = hd;
  p.o_sb = N * C; p.o_st = C; p.o_sh = hd;
  p.bias_h = rel_h; p.bias_w = rel_w; p.kh = gh; p.kw = gw; p.key_mask = nullptr;
  p.scale = scale; p.clamp = 0.f;
  return flash_attn_impl(p, hd, dtype, stream);
}

extern "C" int hipie_vit_attn_fused(const void* qkv, const void* tab_h, const void* tab_w, void* out, int B, int gh, int gw,
                                    int heads, int hd, float scale, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(qkv && out && tab_h && tab_w, "vit_attn_fused: null pointer");
  HIPIE_REQUIRE((((uintptr_t)tab_h | (uintptr_t)tab_w) & 15) == 0, "vit_attn_fused: tables must be 16-byte aligned");
  const long N = (long)gh * gw, C = (long)heads * hd;
  const size_t es = 2;
  FAParams p{};
  p.q = qkv;
  p.k = (const char*)qkv + C * es;
  p.v = (const char*)qkv + 2 * C * es;
  p.out = out;
  p.B = B; p.H = heads; p.Nq = (int)N; p.Nk = (int)N;
  p.q_sb = p.k_sb = p.v_sb = N * 3 * C; p.q_st = p.k_st = p.v_st = 3 * C; p.q_sh = p.k_sh = p.v_sh = hd;
  p.o_sb = N * C; p.o_st = C; p.o_sh = hd;
  p.tab_h = tab_h; p.tab_w = tab_w; p.kh = gh; p.kw = gw; p.key_mask = nullptr;
  p.scale = scale; p.clamp = 0.f;
  return flash_attn_impl(p, hd, dtype, stream);
}

namespace hipie {
int xattn_i2t_try(const void* q, const void* k, const void* vl, const uint8_t* mask, void* out, int B, int H, int Nv, int L, int hd,
                  long E, float clamp, int dtype, hipStream_t st);      // bi_xattn.hip
int xattn_t2i_try(const void* q, const void* k, const void* vv, void* out, float* ws, size_t ws_bytes, int B, int H, int Nv, int L,
                  int hd, long E, float clamp, int dtype, hipStream_t st);
size_t xattn_t2i_workspace(int B, int H, int Nv, int L, int hd);
}

extern "C" int64_t hipie_bi_xattn_workspace(int B, int H, int Nv, int L, int hd) {
  return (int64_t)hipie::xattn_t2i_workspace(B, H, Nv, L, hd);
}

extern "C" int hipie_bi_xattn(const void* q, const void* k, const void* vv, const void* vl, const uint8_t* text_mask,
                              void* out_v, void* out_l, int B, int H, int Nv, int L, int hd, float clamp, int dtype,
                              void* stream) {
  return hipie_bi_xattn_ws(q, k, vv, vl, text_mask, out_v, out_l, nullptr, 0, B, H, Nv, L, hd, clamp, dtype, stream);
}

extern "C" int hipie_bi_xattn_ws(const void* q, const void* k, const void* vv, const void* vl, const uint8_t* text_mask,
                                 void* out_v, void* out_l, void* workspace, int64_t workspace_bytes, int B, int H, int Nv, int L,
                                 int hd, float clamp, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(q && k && vv && vl && out_v && out_l, "bi_xattn: null pointer");
  const long E = (long)H * hd;
  // image rows attend over text: softmax over L with the text mask (fuse_helper.py:97-110)
  FAParams a{};
  a.q = q; a.k = k; a.v = vl; a.out = out_v;
  a.B = B; a.H = H; a.Nq = Nv; a.Nk = L;
  a.q_sb = (long)Nv * E; a.q_st = E; a.q_sh = hd;
  a.k_sb = a.v_sb = (long)L * E; a.k_st = a.v_st = E; a.k_sh = a.v_sh = hd;
  a.o_sb = (long)Nv * E; a.o_st = E; a.o_sh = hd;
  a.key_mask = text_mask; a.scale = 1.f; a.clamp = clamp;
  // the flash kernel for both directions: diagnostics, and fp32 outputs (the specialised kernels write the operand type)
  static const bool generic_env = study_env("HIPIE_XATTN_GENERIC") != nullptr;
  const bool generic_only = generic_env || (dtype & HIPIE_OUT_F32) != 0;
  int rc = generic_only ? 1 : xattn_i2t_try(q, k, vl, text_mask, out_v, B, H, Nv, L, hd, E, clamp, dtype, (hipStream_t)stream);
  if (rc == 1) rc = flash_attn_impl(a, hd, dtype, stream);     // shapes the specialised kernel does not cover (L <= 64: one tile; L > 224)
  if (rc != HIPIE_OK) return rc;
  // text rows attend over ALL image tokens: softmax over Nv of the transposed scores, no mask (fuse_helper.py:86-95)
  FAParams t{};
  t.q = k; t.k = q; t.v = vv; t.out = out_l;
  t.B = B; t.H = H; t.Nq = L; t.Nk = Nv;
  t.q_sb = (long)L * E; t.q_st = E; t.q_sh = hd;
  t.k_sb = t.v_sb = (long)Nv * E; t.k_st = t.v_st = E; t.k_sh = t.v_sh = hd;
  t.o_sb = (long)L * E; t.o_st = E; t.o_sh = hd;
  t.key_mask = nullptr; t.scale = 1.f; t.clamp = clamp;
  rc = generic_only ? 1 : xattn_t2i_try(q, k, vv, out_l, (float*)workspace, (size_t)std::max<int64_t>(workspace_bytes, 0), B, H, Nv, L, hd,
                                        E, clamp, dtype, (hipStream_t)stream);
  if (rc != 1) return rc;
  return flash_attn_impl(t, hd, dtype, stream);
}

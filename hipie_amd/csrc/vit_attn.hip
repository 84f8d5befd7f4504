// vit_attn.hip -- ViT attention with the decomposed relative-position bias computed in the kernel (gfx950): the global
// blocks (one key tile == one key row of the token grid) and the 14x14 windowed blocks (two key rows per tile) of
// hipie/backbone/vit.py:67-83 + hipie/backbone/utils.py:96-125.
//
//     out[i,:] = softmax_j( scale * q_i.k_j + q_i.Rh[yi - yj + kh - 1] + q_i.Rw[xi - xj + kw - 1] ) . v_j
//
// Contract of the operands (the host folds the two constants into the weights once, hipie_amd/modeling/vit.py):
//     q' = scale * log2(e) * q   (the q rows of the packed qkv tensor)        tables' = R / scale
// so that  q'.k + q'.Rh' + q'.Rw'  IS the logit in the exp2 domain: the MFMA output needs no multiply.
//
// Per key tile and wave (32 queries x 32*NB key slots), compared with the generic kernel of flash_attn.hip:
//   * the bias_w row of a lane is the C operand of the first QK^T MFMA (D = K.Q^T + C): no accumulator zeroing, no
//     scale multiply, no bias add -- and padded key slots carry C = -inf, so masking costs nothing either;
//   * bias_h is a per-(query, key row) scalar folded into the exponent offset (one LDS word per tile);
//   * the cross-half row maximum uses v_permlane32_swap (no LDS round trip on the critical path);
//   * the V^T fragments of PV step i+1 are requested from LDS before the MFMAs of step i (the first step's before the
//     softmax), so the ds_read_b64_tr_b16 latency sits behind matrix work instead of in front of every MFMA;
//   * FAST: row sums from a ones column of V in the padded d block and a deferred running max (as flash_attn.hip).
// The (B*heads, N, kh + kw) fp32 bias tensors of the reference never exist; nothing of size Nq x Nk reaches HBM.
// Workgroup = WAVES waves x 32 queries of one (batch, head); K/V tiles HBM -> registers -> LDS (double buffered), one
// barrier per tile; XCD-aware block -> head map (the q tiles of one head share an XCD's L2).
#include <stdlib.h>

#include "common.h"
#include "mfma.h"

namespace hipie {

// per-phase cycle totals of wave 0 / wave 4 of block 0 (timing build, HIPIE_VA_ABL=8): [wave/4][phase]
__device__ unsigned long long g_va_dbg[16];

struct VAParams {
  const void* qkv; void* out; const void* tab_h; const void* tab_w;
  int B, H, N, kh, kw;
  long sb, st;                  // qkv strides (elements): batch, token; q at +0, k at +C, v at +2C (C = H * HD)
  long o_sb, o_st;
  int nqt, swz, prio;
};

__device__ __forceinline__ float va_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

// keep a value alive without cost (timing ablations): an asm statement with a "v" operand must live in a __device__ function
// -- inside the __global__ body the host pass rejects the constraint and silently drops the kernel stub
template <typename V> __host__ __device__ __forceinline__ void va_keep(V v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" :: "v"(v));
#endif
}

// pin a value at this point of the instruction stream (the optimiser otherwise sinks a computation to its first use)
__host__ __device__ __forceinline__ void va_pin(float& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
}

// max over the two 32-lane halves (lane l <-> l ^ 32) without LDS: v_permlane32_swap exchanges the upper half of its first
// operand with the lower half of its second
__device__ __forceinline__ float va_xhalf_max(float x) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const unsigned int u = __builtin_bit_cast(unsigned int, x);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)r[0]), __builtin_bit_cast(float, (unsigned int)r[1]));
}

// T: bf16_t | f16_t;  HD: head dim (64 | 80);  NB: 32-slot key blocks per tile;  FAST: deferred max + MFMA row sums;
// R: key rows of the token grid per tile (1; 2 for the 14-wide windows);  KW: compile-time grid width when R > 1
// ABL: timing ablations (tools/bench_attn.py; results are garbage for ABL != 0): 1 no exp2, 2 no max / rescale, 3 no PV
// MFMAs, 4 no QK^T MFMAs, 5 no tile streaming (loads, LDS stores, barrier), 6 constant P (no add / exp2 / pack)
template <typename T, int HD, int NB, int WAVES, bool FAST, int R, int KW, int ABL>
__global__ __launch_bounds__(WAVES * 64, 2) void vit_attn_kernel(const VAParams p) {
  constexpr int KT = 32 * NB;                  // key slots per tile
  constexpr int KS = HD / 16;                  // k16 steps of QK^T
  constexpr int DB = (HD + 31) / 32;           // 32-row d blocks of O^T
  constexpr int KSTR = HD + 8;                 // K tile row stride (elements): 16-B aligned, conflict-free b128 reads
  constexpr int VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;   // see flash_attn.hip
  constexpr int CPR = HD / 8;                  // 16-byte chunks per row
  constexpr int NCH = KT * CPR;
  constexpr int NT = WAVES * 64;
  constexpr int CPT = (NCH + NT - 1) / NT;
  constexpr int BUF = KT * (KSTR + VSTR);      // elements per LDS buffer
  constexpr int QW = WAVES * 32;               // queries per workgroup
  constexpr bool LTRICK = FAST && (HD % 32) != 0;
  constexpr int L_ROW = HD % 32, L_HI = (L_ROW >> 2) & 1, L_REG = (L_ROW & 3) + 4 * (L_ROW >> 3);
  constexpr float DEFER = FAST ? 8.f : 0.f;
  static_assert(R == 1 || (KW > 0 && R * KW <= KT), "R key rows of KW keys must fit the tile");
  typedef typename Mfma32<T>::frag frag;
  typedef typename Mfma32<T>::half_frag hfrag;

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr size_t TILE_BYTES = ((size_t)2 * BUF * sizeof(T) > (size_t)WAVES * 4096) ? (size_t)2 * BUF * sizeof(T) : (size_t)WAVES * 4096;
  float* bh_all = reinterpret_cast<float*>(smem_raw + TILE_BYTES);                       // [kh][QW], log2 domain

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int kw = (R > 1) ? KW : p.kw, kh = p.kh;

  int bh, qt;
  {
    const int id = blockIdx.x;
    if (p.swz) { bh = (id & 7) + 8 * ((id >> 3) / p.nqt); qt = (id >> 3) % p.nqt; }
    else { bh = id / p.nqt; qt = id % p.nqt; }
  }
  const int b = bh / p.H, h = bh % p.H;
  const long C = (long)p.H * HD;
  const T* Qg = reinterpret_cast<const T*>(p.qkv) + b * p.sb + (long)h * HD;
  const T* Kg = Qg + C;
  const T* Vg = Qg + 2 * C;
  T* Og = reinterpret_cast<T*>(p.out) + b * p.o_sb + (long)h * HD;

  const int qi = qt * QW + wave * 32 + li;     // this lane's query (token index inside the image / window)
  const int qc = min(qi, p.N - 1);
  const int qy = qc / kw, qx = qc - qy * kw;
  const int nkt = R * kw;                      // keys per full tile

  // ---- Q fragments (B operand): lane (q = li, half hi) holds q'[16 ks + 8 hi + j] ----
  frag qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag*>(Qg + (long)qc * p.st + 16 * ks + 8 * hi);

  // ---- decomposed rel-pos bias (add_decomposed_rel_pos, backbone/utils.py:96-125) -----------------------------------
  // T[j][q] = tab[j] . q' for every table row j, 32 rows per MFMA block; each block is staged through a per-wave
  // (32 x 32) fp32 LDS tile because the row a lane needs sits in another register / lane half of the MFMA output:
  //   bias_w[q][kx] = Tw[qx - kx + kw - 1][q]  -> the lane's bw registers (C operand of the QK^T MFMAs; -inf on padded slots)
  //   bias_h[q][ky] = Th[qy - ky + kh - 1][q]  -> bh_all[ky][q] in LDS (one word per query and key row)
  f32x16 bw[NB];               // native vectors: each is the C operand (16 consecutive VGPRs) of a QK^T MFMA
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) bw[blk][r] = -INFINITY;
  {
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * (32 * 32);     // aliases the (still unused) tile buffers
    const T* TW = reinterpret_cast<const T*>(p.tab_w);
    const T* TH = reinterpret_cast<const T*>(p.tab_h);
    const int nbw = (2 * kw - 1 + 31) / 32, nbh = (2 * kh - 1 + 31) / 32;
    for (int jb = 0; jb < nbw; ++jb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const T* ap = TW + (long)min(32 * jb + li, 2 * kw - 2) * HD + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = Mfma32<T>::mma(*reinterpret_cast<const frag*>(ap + 16 * ks), qf[ks], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[crow(r, hi) * 32 + li] = acc[r];
      __syncthreads();
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int slot = 32 * blk + crow(r, hi);
          const int kx = (R > 1) ? slot % kw : slot;
          const int j = qx + kw - 1 - kx - 32 * jb;
          if (slot < R * kw && j >= 0 && j < 32) bw[blk][r] = stage[j * 32 + li];
        }
      __syncthreads();
    }
    for (int jb = 0; jb < nbh; ++jb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const T* ap = TH + (long)min(32 * jb + li, 2 * kh - 2) * HD + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = Mfma32<T>::mma(*reinterpret_cast<const frag*>(ap + 16 * ks), qf[ks], acc);
      // table row j = qy - ky + kh - 1  <=>  ky = qy + kh - 1 - j
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ky = qy + kh - 1 - (32 * jb + crow(r, hi));
        if (ky >= 0 && ky < kh) bh_all[ky * QW + wave * 32 + li] = acc[r];
      }
    }
    __syncthreads();             // every wave is done with its stage before the tile buffers are initialised
  }

  // zero the tile buffers once: the pad columns of the V tile feed the padded d rows of O^T (discarded, but NaN-free)
  for (int i = tid; i < 2 * BUF / 8; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if (LTRICK) {
    __syncthreads();
    for (int i = tid; i < 2 * KT; i += NT) smem[(i / KT) * BUF + KT * KSTR + (i % KT) * VSTR + HD] = (T)1.0f;   // never overwritten
  }

  // ---- tile streaming: HBM/L2 -> registers -> LDS (tile t = keys [t*R*kw, (t+1)*R*kw), clamped to the image) ----
  int koff[CPT], lk[CPT], lv[CPT], crow_[CPT], ch8_[CPT];
  bool cval[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int idx = min(c * NT + tid, NCH - 1);
    const int row = idx / CPR, ch = idx - row * CPR;
    cval[c] = (c * NT + tid) < NCH;
    crow_[c] = row;
    ch8_[c] = ch * 8;
    koff[c] = min(row, nkt - 1) * (int)p.st + ch * 8;     // padded slots re-read the tile's last valid key (finite data, P = 0)
    lk[c] = row * KSTR + ch * 8;
    lv[c] = row * VSTR + ch * 8;
  }
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 kr[CPT], vr[CPT];
#define VA_LOAD_REGS(t_)                                                                              \
  {                                                                                                   \
    const int key0_ = (t_) * nkt;                                                                     \
    const T* kb_ = Kg + (long)key0_ * p.st;                                                           \
    const T* vb_ = Vg + (long)key0_ * p.st;                                                           \
    if (R == 1 || key0_ + nkt <= p.N) {                                                               \
      _Pragma("unroll") for (int c = 0; c < CPT; ++c) {                                               \
        kr[c] = *reinterpret_cast<const u32x4*>(kb_ + koff[c]);                                       \
        vr[c] = *reinterpret_cast<const u32x4*>(vb_ + koff[c]);                                       \
      }                                                                                               \
    } else { /* ragged last tile (odd number of key rows): clamp to the last key of the image */      \
      _Pragma("unroll") for (int c = 0; c < CPT; ++c) {                                               \
        const long ro_ = (long)min(crow_[c], p.N - key0_ - 1) * p.st + ch8_[c];                       \
        kr[c] = *reinterpret_cast<const u32x4*>(kb_ + ro_);                                           \
        vr[c] = *reinterpret_cast<const u32x4*>(vb_ + ro_);                                           \
      }                                                                                               \
    }                                                                                                 \
  }
#define VA_STORE_LDS(buf_)                                                                            \
  {                                                                                                   \
    T* Ks_ = smem + (buf_) * BUF;                                                                     \
    T* Vs_ = Ks_ + KT * KSTR;                                                                         \
    _Pragma("unroll") for (int c = 0; c < CPT; ++c) {                                                 \
      if (cval[c]) {                                                                                  \
        *reinterpret_cast<u32x4*>(Ks_ + lk[c]) = kr[c];                                               \
        *reinterpret_cast<u32x4*>(Vs_ + lv[c]) = vr[c];                                               \
      }                                                                                               \
    }                                                                                                 \
  }

  // ---- online softmax state (exp2 domain) ----
  f32x16 O[DB];
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;

  const int nt = (kh + R - 1) / R;
  __syncthreads();                 // LDS initialised
  VA_LOAD_REGS(0);
  VA_STORE_LDS(0);
  __syncthreads();
  if (nt > 1) VA_LOAD_REGS(1);
  const int l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int vlane = (4 * hi + (l16 >> 2)) * VSTR + 16 * g1 + 4 * (l16 & 3);   // this lane's V^T read offset inside a (step, d) block

  for (int t = 0; t < nt; ++t) {
    const T* Ks = smem + (t & 1) * BUF;
    const T* Vs = Ks + KT * KSTR;
    float bh0 = bh_all[(R * t) * QW + wave * 32 + li], bh1 = 0.f;
    if (R > 1) bh1 = (R * t + 1 < kh) ? bh_all[(R * t + 1) * QW + wave * 32 + li] : -INFINITY;     // odd kh: the last tile has one row

    // ---- S^T = K . Q'^T + bias_w: all K fragments are requested first, the first k-step takes C = bias_w ----
    f32x16 S[NB];
    {
      const T* kbase = Ks + li * KSTR + 8 * hi;
      frag kf[NB][KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) kf[blk][ks] = *reinterpret_cast<const frag*>(kbase + 32 * blk * KSTR + 16 * ks);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        if (ABL == 4) { S[blk] = bw[blk]; va_keep(kf[blk][0]); va_keep(kf[blk][KS - 1]); }
        else S[blk] = Mfma32<T>::mma(kf[blk][0], qf[0], bw[blk]);
      }
#pragma unroll
      for (int ks = 1; ks < KS; ++ks)
#pragma unroll
        for (int blk = 0; blk < NB; ++blk)
          if (ABL != 4) S[blk] = Mfma32<T>::mma(kf[blk][ks], qf[ks], S[blk]);
    }
    // V^T fragments of the first PV step: in flight while the softmax statistics are computed
    hfrag vlo[DB], vhi[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d) {
      vlo[d] = Mfma32<T>::tr_read(Vs + vlane + 32 * d);
      vhi[d] = Mfma32<T>::tr_read(Vs + vlane + 32 * d + 8 * VSTR);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- row maximum (lane-local chain + one cross-half exchange); R > 1: add the key row's bias_h first ----
    if (R > 1) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[blk][r] += (32 * blk + crow(r, hi) < kw) ? bh0 : bh1;
    }
    float mx = -INFINITY;
    if (ABL == 2) mx = S[0][0];
    else {
      // first link in C++ over one element of EVERY score block: with R == 1 the chain below reads MFMA results directly, and hipcc places
      // the MFMA -> VALU wait states only for reads it can see (not inside the v_max3 asm statements)
      mx = __builtin_fmaxf(S[0][0], S[0][1]);            // a real VALU instruction also when NB == 1
#pragma unroll
      for (int blk = 1; blk < NB; ++blk) mx = __builtin_fmaxf(mx, S[blk][0]);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = va_max3(mx, S[blk][r], S[blk][r + 1]);
      mx = va_xhalf_max(mx);
    }
    mx += (R > 1 ? 0.f : bh0);
    // the reference point m_run only moves when some row's maximum grew by more than 2^DEFER (FAST) / at all (exact)
    const bool grow = mx > m_run + DEFER;
    if (ABL == 2) { if (t == 0) m_run = mx; }
    else if (__builtin_amdgcn_ballot_w64(grow) != 0ull) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // m_run = -inf -> 0 (every tile has a valid key)
      m_run = m_new;
      if (!LTRICK) l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
    }
    const float off = (R > 1 ? 0.f : bh0) - m_run;

    // ---- P = exp2(S + off);  O^T += V^T . P^T.  Step i = (blk, s): 8 key slots per lane half; the exps / packs of step
    //      i + 1 and the V^T reads of step i + 1 are issued in the shadow of the MFMAs of step i ----
    frag pf;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float pv = (ABL == 6) ? 0.5f : (ABL == 1) ? S[0][j] + off : __builtin_amdgcn_exp2f(S[0][j] + off);
      if (!LTRICK) l_run += pv;
      pf[j] = (T)pv;
    }
#pragma unroll
    for (int step = 0; step < 2 * NB; ++step) {
      const int blk = step >> 1, s = step & 1;
      frag vf[DB];
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[d][j] = vlo[d][j]; vf[d][4 + j] = vhi[d][j]; }
      frag pn = pf;
      if (step + 1 < 2 * NB) {
        const int nb_ = (step + 1) >> 1, ns_ = (step + 1) & 1;
        const T* vb = Vs + (32 * nb_ + 16 * ns_) * VSTR + vlane;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          vlo[d] = Mfma32<T>::tr_read(vb + 32 * d);
          vhi[d] = Mfma32<T>::tr_read(vb + 32 * d + 8 * VSTR);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = (ABL == 6) ? 0.5f : (ABL == 1) ? S[nb_][8 * ns_ + j] + off : __builtin_amdgcn_exp2f(S[nb_][8 * ns_ + j] + off);
          if (!LTRICK) l_run += pv;
          pn[j] = (T)pv;
        }
      }
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        if (ABL == 3) { va_keep(vf[d]); va_keep(pf); }
        else O[d] = Mfma32<T>::mma(vf[d], pf, O[d]);
      }
      if (ABL == 6) { va_keep(S[0]); va_keep(S[NB - 1]); }
      pf = pn;
      (void)blk; (void)s;
    }

    if (ABL != 5) {       // 7: no barrier, 8: no global loads, 9: no LDS stores
      if (ABL != 9 && t + 1 < nt) VA_STORE_LDS((t + 1) & 1);
      if (ABL != 7) __syncthreads();
      if (ABL != 8 && t + 2 < nt) VA_LOAD_REGS(t + 2);
    }
  }

  // ---- epilogue: O^T / l, 8-byte stores of 4 consecutive d ----
  {
    const float l_tot = LTRICK ? __shfl(O[DB - 1][L_REG], li + 32 * L_HI) : l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qi < p.N) {
      T* orow = Og + (long)qi * p.o_st;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int d0 = 32 * d + 8 * rr + 4 * hi;
          if (d0 < HD) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            t4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (T)(O[d][4 * rr + e] * inv);
            *reinterpret_cast<t4*>(orow + d0) = o4;
          }
        }
    }
  }
}
#undef VA_LOAD_REGS
#undef VA_STORE_LDS

template <typename T, int HD, int NB, int WAVES, bool FAST, int R, int KW, int ABL>
static int launch_va(VAParams& p, hipStream_t st) {
  constexpr int KT = 32 * NB, DB = (HD + 31) / 32;
  constexpr int VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;
  size_t lds = (size_t)2 * KT * ((HD + 8) + VSTR) * sizeof(T);
  const size_t stage = (size_t)WAVES * 32 * 32 * sizeof(float);
  if (lds < stage) lds = stage;
  lds += (size_t)p.kh * WAVES * 32 * sizeof(float);
  if (lds > 160 * 1024) return set_err(HIPIE_EINVAL, "vit_attn: %zu bytes of LDS needed (grid %dx%d) > 160 KiB", lds, p.kh, p.kw);
  p.nqt = (p.N + WAVES * 32 - 1) / (WAVES * 32);
  p.swz = ((p.B * p.H) % 8 == 0) ? 1 : 0;
  const unsigned grid = (unsigned)(p.nqt * p.B * p.H);
  auto kern = vit_attn_kernel<T, HD, NB, WAVES, FAST, R, KW, ABL>;
  if (lds > 64 * 1024) {     // the dynamic-LDS limit is a per-device attribute of the function (not a stream operation)
    static size_t lds_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || lds > lds_set[dev]) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (dev >= 0 && dev < 64) lds_set[dev] = lds;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, p);
  return check_launch("vit_attn");
}


// ---------------------------------------------------------------------------------------------------------------------
// Software-pipelined form for the global blocks (token grids 33..64 wide, >= 1024 tokens).
//
// What the hardware does (measured here with tools/ubench/overlap.hip and in-kernel cycle counters): the two waves of a SIMD
// do NOT hide one wave's MFMA burst behind the other's VALU / LDS burst (MFMA-only beside VALU-only = the sum of both); a
// wave hides vector work only behind its OWN MFMAs, a handful of instructions per MFMA, and every instruction of any kind
// (VALU, SALU, LDS, waitcnt, branch) costs an issue slot of a few cycles.  So (i) every MFMA of the loop is followed by a
// slice of independent work, and (ii) the instruction count per tile is cut to the bone:
//   phase 1   QK^T(t + 1) -> S_next  ((KS + 1) k-steps x NB MFMAs)   beside   P(t) = exp2(S_cur) -> 16 bit, steps 0 .. NSTEP-2
//   phase 2   PV(t)                  (NSTEP x DB MFMAs)              beside   last P step, row maxima of S_next, V^T reads
//   * the exponent offset off = bias_h(t) - m is ADDED BY THE MATRIX PIPE: one extra k-step multiplies a ones fragment (K side)
//     with the 16-bit pairs (bh_hi, bh_lo, m_hi, m_lo) in four k slots of the Q side.  bias_h is split into its pair once in
//     the prologue and kept PACKED in LDS, so the hot loop spends one ds_read_b32 per tile on it; m is kept on a 1/16 grid
//     so that its pair is exact.  Per score that leaves max3 (1/2) + exp2 + pack (1/2).
//   * m is the running reference of the PREVIOUS tile (deferred maximum: FAST moves it only when some score exceeds it by
//     2^8, the exact mode whenever a score exceeds it); the decision is one ballot over per-lane partial maxima, the
//     cross-half exchange and the O / S rescale live in the cold path.
//   * K / V tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction): no staging registers,
//     no ds_write, no exec-masked store blocks.  The padded LDS rows (K 176 B, V 192 B) are filled by the same DMA: a pad
//     chunk's lane simply points at a 16-byte constant (the V pad carries the ones column of the row-sum trick).
//   * K ring 3 deep, V ring 2 deep; the first K fragments of the next tile are fetched before the tile barrier.
__device__ __attribute__((aligned(32))) const unsigned short g_va_pad_bf16[16] = {0x3F80, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};     // [1.0, 0 x 7 | 0 x 8]
__device__ __attribute__((aligned(32))) const unsigned short g_va_pad_f16[16] = {0x3C00, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16 * lane] <- *gsrc.  Inline asm on purpose: the compiler tracks the builtin form as
// an LDS write and puts s_waitcnt vmcnt(0) in front of every later ds_read (a full L2 round trip per DMA, measured); here the
// completion is counted by hand -- vmcnt(0) before the tile barrier.  M0 carries the wave-uniform LDS byte address.
__device__ __forceinline__ void va_dma16(const void* gsrc, unsigned int lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
#endif
}

template <typename T> __device__ __forceinline__ unsigned int va_bits16(T x) { return (unsigned int)__builtin_bit_cast(unsigned short, x); }

// 16-bit pair (hi, lo) with hi + lo == x to ~2^-17 (bf16) / 2^-23 (f16) relative, packed [hi | lo << 16]
template <typename T> __device__ __forceinline__ unsigned int va_split2(float x) {
  const T h0 = (T)x;
  const T h1 = (T)(x - (float)h0);
  return va_bits16<T>(h0) | (va_bits16<T>(h1) << 16);
}

template <typename T, int HD, int NB, bool FAST, int ABL>
__global__ __launch_bounds__(512, 2) void vit_attn_sp_kernel(const VAParams p) {
  constexpr int WAVES = 8;
  constexpr int KT = 32 * NB;
  constexpr int KS = HD / 16;
  constexpr int DB = (HD + 31) / 32;
  constexpr int KSTR = HD + 8;                 // 11 (hd 80) / 9 (hd 64) 16-byte chunks per K row
  constexpr int VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;     // 12 chunks per V row for hd 64 / 80
  constexpr int CPR = HD / 8;                  // data chunks per row
  constexpr int KCH = KSTR / 8, VCH = VSTR / 8;            // chunks per padded LDS row
  constexpr int KBLK = (KT * KCH + 63) / 64, VBLK = (KT * VCH + 63) / 64;      // 1 KiB DMA blocks per tile
  constexpr int KBUF = KBLK * 512, VBUF = VBLK * 512;      // elements per ring slot (whole DMA blocks)
  constexpr int NDMA = (KBLK + VBLK + WAVES - 1) / WAVES;  // DMA instructions per wave and tile
  constexpr int QW = WAVES * 32;
  constexpr bool LTRICK = FAST && (HD % 32) != 0;
  constexpr int L_ROW = HD % 32, L_HI = (L_ROW >> 2) & 1, L_REG = (L_ROW & 3) + 4 * (L_ROW >> 3);
  constexpr float DEFER = FAST ? 8.f : 0.f;
  constexpr int NSTEP = 2 * NB;
  static_assert(NB == 2, "the slot plan is written for two key blocks per tile");
  static_assert(KSTR % 8 == 0 && VSTR % 8 == 0, "LDS rows must be whole 16-byte chunks");
  typedef typename Mfma32<T>::frag frag;
  typedef typename Mfma32<T>::half_frag hfrag;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T* Kring = reinterpret_cast<T*>(smem_raw);               // [3][KBUF]
  T* Vring = Kring + 3 * KBUF;                             // [2][VBUF]
  constexpr size_t TILE_BYTES = ((size_t)(3 * KBUF + 2 * VBUF) * sizeof(T) > (size_t)WAVES * 4096) ? (size_t)(3 * KBUF + 2 * VBUF) * sizeof(T) : (size_t)WAVES * 4096;
  unsigned int* bh_all = reinterpret_cast<unsigned int*>(smem_raw + TILE_BYTES);        // [kh + 1][QW] packed pairs; row kh = zeros

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                    // waves w and w + 4 share a SIMD
  const int li = lane & 31, hi = lane >> 5;
  const int kw = p.kw, kh = p.kh;

  int bh, qt;
  {
    const int id = blockIdx.x;
    if (p.swz) { bh = (id & 7) + 8 * ((id >> 3) / p.nqt); qt = (id >> 3) % p.nqt; }
    else { bh = id / p.nqt; qt = id % p.nqt; }
  }
  const int b = bh / p.H, h = bh % p.H;
  const long C = (long)p.H * HD;
  const T* Qg = reinterpret_cast<const T*>(p.qkv) + b * p.sb + (long)h * HD;
  const T* Kg = Qg + C;
  const T* Vg = Qg + 2 * C;
  T* Og = reinterpret_cast<T*>(p.out) + b * p.o_sb + (long)h * HD;

  const int qi = qt * QW + wave * 32 + li;
  const int qc = min(qi, p.N - 1);
  const int qy = qc / kw, qx = qc - qy * kw;

  frag qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag*>(Qg + (long)qc * p.st + 16 * ks + 8 * hi);

  // ---- decomposed rel-pos bias (as in vit_attn_kernel); bias_h is stored as packed 16-bit pairs ----
  f32x16 bw[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) bw[blk][r] = -INFINITY;
  {
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * (32 * 32);
    const T* TW = reinterpret_cast<const T*>(p.tab_w);
    const T* TH = reinterpret_cast<const T*>(p.tab_h);
    const int nbw = (2 * kw - 1 + 31) / 32, nbh = (2 * kh - 1 + 31) / 32;
    for (int jb = 0; jb < nbw; ++jb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const T* ap = TW + (long)min(32 * jb + li, 2 * kw - 2) * HD + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = Mfma32<T>::mma(*reinterpret_cast<const frag*>(ap + 16 * ks), qf[ks], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) stage[crow(r, hi) * 32 + li] = acc[r];
      __syncthreads();
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int slot = 32 * blk + crow(r, hi);
          const int j = qx + kw - 1 - slot - 32 * jb;
          if (slot < kw && j >= 0 && j < 32) bw[blk][r] = stage[j * 32 + li];
        }
      __syncthreads();
    }
    for (int jb = 0; jb < nbh; ++jb) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const T* ap = TH + (long)min(32 * jb + li, 2 * kh - 2) * HD + 8 * hi;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc = Mfma32<T>::mma(*reinterpret_cast<const frag*>(ap + 16 * ks), qf[ks], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ky = qy + kh - 1 - (32 * jb + crow(r, hi));
        if (ky >= 0 && ky < kh) bh_all[ky * QW + wave * 32 + li] = va_split2<T>(acc[r]);
      }
    }
    if (tid < QW) bh_all[kh * QW + tid] = 0u;        // the zeros row: what lane half 1 feeds into the offset k-step
    __syncthreads();                                  // stage (aliasing the rings) is dead from here on
  }
  // lane half 0 walks its query's column of the table, lane half 1 sits on the zeros row
  const unsigned int* bhp = bh_all + (hi ? kh * QW : 0) + wave * 32 + li;
  const int bhs = hi ? 0 : QW;

  // ---- tile streaming by LDS-DMA.  DMA block j of a tile: j < KBLK -> K block j, else V block j - KBLK; wave w issues blocks
  //      w, w + 8, w + 16.  Chunk c = 64 * block + lane of a padded tile image: row c / (K|V)CH, column c % (K|V)CH; columns
  //      >= CPR are padding and come from the 16-byte constants (K: zeros; V: the ones chunk then zeros). ----
  const char* dsrc[NDMA];          // this lane's source of the NEXT tile its DMA instruction r fetches
  long dstep[NDMA];                // its advance per tile (0 for pad chunks)
  const unsigned short* padc = __is_same(T, f16_t) ? g_va_pad_f16 : g_va_pad_bf16;
  const long tile_bytes = (long)kw * p.st * (long)sizeof(T);
#pragma unroll
  for (int r = 0; r < NDMA; ++r) {
    const int j = wave + WAVES * r;
    const bool isk = j < KBLK;
    const int c = 64 * (isk ? j : j - KBLK) + lane;
    const int nch = isk ? KCH : VCH;
    const int row = min(c / nch, KT - 1), col = c % nch;
    const T* base = isk ? Kg : Vg;
    if (col < CPR) {
      dsrc[r] = reinterpret_cast<const char*>(base + (long)min(row, kw - 1) * p.st + col * 8);     // padded slots: last key of the row
      dstep[r] = tile_bytes;
    } else {
      dsrc[r] = reinterpret_cast<const char*>(padc + ((!isk && col == CPR) ? 0 : 8));
      dstep[r] = 0;
    }
  }
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem_raw);
  // issue the DMA instructions of this wave for K tile tk -> ring slot ks_ and / or V tile tv -> ring slot vs_ (tile < 0: skip)
  // (r_only >= 0: only that one of the wave's instructions -- the loop spreads them over the tile, a burst blocks the wave's
  //  in-order issue for as long as the fill path is busy)
  auto dma_tiles = [&](const int tk, const int ks_, const int tv, const int vs_, const int r_only = -1) {
#pragma unroll
    for (int r = 0; r < NDMA; ++r) {
      const int j = wave + WAVES * r;
      if (r_only >= 0 && r != r_only) continue;
      if (j >= KBLK + VBLK) continue;
      const bool isk = j < KBLK;
      const int tt = isk ? tk : tv;
      if (tt < 0) continue;
      const unsigned int dst = lds0 + (isk ? (unsigned int)(ks_ * KBUF * sizeof(T)) + 1024u * j
                                             : (unsigned int)((3 * KBUF + vs_ * VBUF) * sizeof(T)) + 1024u * (j - KBLK));
      va_dma16(dsrc[r] + (long)tt * dstep[r], __builtin_amdgcn_readfirstlane(dst));
    }
  };

  // The offset k-step: K side = a ones fragment; Q side = [bh_hi, bh_lo, m_hi, m_lo, 0, 0, 0, 0] (lane half 0; zeros in half 1)
  frag kones;
#pragma unroll
  for (int j = 0; j < 8; ++j) kones[j] = (T)1.0f;
  unsigned int qa_bh = 0u, qa_m = 0u;
#define SP_QA() __builtin_bit_cast(frag, (u32x4){qa_bh, qa_m, 0u, 0u})

  f32x16 O[DB];
  float m_run = 0.f, l_run = 0.f;           // m_run: reference of the exponents already folded into S / O (on a 1/16 grid)
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;

  const int nt = kh;
  const int l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int vlane = (4 * hi + (l16 >> 2)) * VSTR + 16 * g1 + 4 * (l16 & 3);
  const int klane = li * KSTR + 8 * hi;

  // ---- prologue: K(0..2), V(0); S(0) un-overlapped ----
  dma_tiles(0, 0, 0, 0);
  dma_tiles(nt > 1 ? 1 : -1, 1, -1, 0);
  dma_tiles(nt > 2 ? 2 : -1, 2, -1, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this wave's DMA writes have landed
  __syncthreads();

  f32x16 SA[NB], SB[NB];
  {
    qa_bh = bhp[0];
    const frag qa = SP_QA();
    const T* kb = Kring + klane;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) SA[blk] = Mfma32<T>::mma(*reinterpret_cast<const frag*>(kb + 32 * blk * KSTR), qf[0], bw[blk]);
#pragma unroll
    for (int ks = 1; ks < KS; ++ks)
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
        SA[blk] = Mfma32<T>::mma(*reinterpret_cast<const frag*>(kb + 32 * blk * KSTR + 16 * ks), qf[ks], SA[blk]);
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) SA[blk] = Mfma32<T>::mma(kones, qa, SA[blk]);
    // first tile: the reference is the row maximum (rounded up to the 1/16 grid)
    float mx = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, SA[blk][r]), SA[blk][r + 1]);
    mx = va_xhalf_max(mx);
    m_run = __builtin_ceilf(mx * 16.f) * 0.0625f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) SA[blk][r] -= m_run;
    qa_bh = bhp[min(1, nt - 1) * bhs];
    qa_m = hi ? 0u : va_split2<T>(-m_run);
  }
  frag kfa[NB];                    // K fragments of the first k-step of the NEXT tile: fetched before the tile barrier
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) kfa[blk] = *reinterpret_cast<const frag*>(Kring + (nt > 1 ? 1 : 0) * KBUF + klane + 32 * blk * KSTR);

  long long dbg[4] = {0, 0, 0, 0};
  const long long cstart = (ABL >= 8) ? clock64() : 0;
  // the second-dispatched half of the workgroup loses the VALU arbitration against its SIMD partner on every tile (measured:
  // phase 1 takes it 1.6x as long); one static priority raise evens the two halves out
  if (p.prio && grp == 1) __builtin_amdgcn_s_setprio(1);

  // one tile: Sc = S(t) (reference m_run folded in), Sn receives S(t + 1).  kslot: ring slot of K(t + 1)
  auto tile = [&](f32x16 (&Sc)[NB], f32x16 (&Sn)[NB], const int t, const int kslot) {
    // The last iteration has no next tile: it still runs the QK^T MFMAs of phase 1 on a (stale but valid) K ring slot and
    // throws S_next away -- 1 / nt of wasted matrix work buys a loop body without a second, differently scheduled copy.
    const bool more = (t + 1 < nt);
    const long long c0 = (ABL >= 8) ? clock64() : 0;
    const T* Kt = Kring + kslot * KBUF + klane;                   // K(t + 1)
    const T* Vt = Vring + (t & 1) * VBUF + vlane;                 // V(t)
    const int kslot2 = (kslot == 2) ? 0 : kslot + 1;              // K(t + 2)
    const int kslot3 = (kslot2 == 2) ? 0 : kslot2 + 1;            // K(t + 3) replaces K(t)
    // K(t + 3) and V(t + 1): their ring slots were last read one iteration ago.  A DMA instruction costs its wave 100+ cycles
    // of issue time, so the two waves of a SIMD issue theirs in different phases (waves 0-3 here, waves 4-7 before phase 2)
    const int dtk = t + 3 < nt ? t + 3 : -1, dtv = more ? t + 1 : -1;
    if (ABL != 9) dma_tiles(dtk, kslot3, dtv, (t + 1) & 1);
    const frag qa = SP_QA();

    frag pf[NSTEP];
    hfrag vlo[DB], vhi[DB];

    // ================= phase 1: S_next = K(t+1) . Q'^T + bias_w + off   beside   P(t) steps 0 .. NSTEP-2 =================
    // unit u (0 .. 2*NSTEP-1) = 4 scores of P: step u/2, half u%2  ->  4 exp2 + 2 packs (+ 4 adds without the ones column)
#define SP_PUNIT(u_)                                                                                  \
    {                                                                                                 \
      constexpr int st_ = (u_) >> 1, hf_ = (u_) & 1;                                                  \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
        const float pv_ = __builtin_amdgcn_exp2f(Sc[st_ >> 1][8 * (st_ & 1) + 4 * hf_ + j]);          \
        if (!LTRICK) l_run += pv_;                                                                    \
        pf[st_][4 * hf_ + j] = (T)pv_;                                                                \
      }                                                                                               \
    }
    {
      frag kfb[NB];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        // K fragments of k-step ks + 1 are requested before the MFMAs of k-step ks
        if (ks + 1 < KS) {
#pragma unroll
          for (int blk = 0; blk < NB; ++blk) {
            if (ks & 1) kfa[blk] = *reinterpret_cast<const frag*>(Kt + 32 * blk * KSTR + 16 * (ks + 1));
            else kfb[blk] = *reinterpret_cast<const frag*>(Kt + 32 * blk * KSTR + 16 * (ks + 1));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          const frag kk = (ks & 1) ? kfb[blk] : kfa[blk];
          Sn[blk] = (ks == 0) ? Mfma32<T>::mma(kk, qf[0], bw[blk]) : Mfma32<T>::mma(kk, qf[ks], Sn[blk]);
          switch (ks * NB + blk) {       // the P units of PV steps 0 and 1, one behind every second MFMA
            case 1: SP_PUNIT(0); break;
            case 3: SP_PUNIT(1); break;
            case 5: SP_PUNIT(2); break;
            case 7: SP_PUNIT(3); break;
            default: break;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // V^T fragments of PV step 0, then the offset k-step
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        vlo[d] = Mfma32<T>::tr_read(Vt + 32 * d);
        vhi[d] = Mfma32<T>::tr_read(Vt + 32 * d + 8 * VSTR);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) Sn[blk] = Mfma32<T>::mma(kones, qa, Sn[blk]);
    }
    __builtin_amdgcn_sched_barrier(0);
    const long long c1 = (ABL >= 8) ? clock64() : 0;

    // ================= phase 2: O^T += V^T(t) . P(t)^T   beside   everything else of the tile ============================
    // slot plan of the NSTEP * DB PV MFMAs (12 for head dim 80, 8 for 64): P units 0-1, partial maxima 2-5, then the rest
    constexpr int SLOTS = NSTEP * DB;
    static_assert(SLOTS >= 8, "slot plan needs at least 8 PV MFMAs per tile");
    float mxp[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // four independent partial maxima of S_next (this lane's keys)
    bool grow_any = false;
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      frag vf[DB];
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[d][j] = vlo[d][j]; vf[d][4 + j] = vhi[d][j]; }
      if (step + 1 < NSTEP) {
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          vlo[d] = Mfma32<T>::tr_read(Vt + (16 * (step + 1)) * VSTR + 32 * d);
          vhi[d] = Mfma32<T>::tr_read(Vt + (16 * (step + 1)) * VSTR + 32 * d + 8 * VSTR);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        O[d] = Mfma32<T>::mma(vf[d], pf[step], O[d]);
        const int slot = step * DB + d;
        // P units of steps 2 and 3 just in time (step s starts at slot s * DB); maxima of S_next in the other slots
        constexpr int U4 = 0, U5 = 1, U6 = (SLOTS >= 12) ? 3 : 3, U7 = (SLOTS >= 12) ? 4 : 4;
        constexpr int M0 = 2, M1 = 5, M2 = (SLOTS >= 12) ? 6 : 5, M3 = (SLOTS >= 12) ? 7 : 6, FIN = (SLOTS >= 12) ? 8 : 6;
        if (slot == U4) { SP_PUNIT(4); }
        if (slot == U5) { SP_PUNIT(5); }
        if (slot == U6) { SP_PUNIT(6); }
        if (slot == U7) { SP_PUNIT(7); }
#define SP_MAXQ(q_)                                                                                   \
        {                                                                                             \
          constexpr int blk_ = (q_) >> 1, r0_ = 8 * ((q_) & 1);                                       \
          _Pragma("unroll") for (int r = r0_; r < r0_ + 8; r += 2)                                    \
            mxp[q_] = __builtin_fmaxf(__builtin_fmaxf(mxp[q_], Sn[blk_][r]), Sn[blk_][r + 1]);        \
          va_pin(mxp[q_]);                                                                            \
        }
        if (slot == M0) SP_MAXQ(0);
        if (slot == M1) SP_MAXQ(1);
        if (slot == M2) SP_MAXQ(2);
        if (slot == M3) SP_MAXQ(3);
#undef SP_MAXQ
        if (slot == FIN) {
          // a score above the reference window in ANY lane (each half holds half of a row's keys) -> cold path after the barrier
          mxp[0] = __builtin_fmaxf(__builtin_fmaxf(mxp[0], mxp[1]), __builtin_fmaxf(mxp[2], mxp[3]));
          grow_any = more && (__builtin_amdgcn_ballot_w64(mxp[0] > DEFER) != 0ull);
        }
        if (slot == SLOTS - 2) { qa_bh = bhp[min(t + 2, nt - 1) * bhs]; }
        if (slot == SLOTS - 1) {                     // first K fragments of the next tile (K(t + 2): complete since the last barrier)
#pragma unroll
          for (int blk = 0; blk < NB; ++blk) kfa[blk] = *reinterpret_cast<const frag*>(Kring + kslot2 * KBUF + klane + 32 * blk * KSTR);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef SP_PUNIT
    const long long c2 = (ABL >= 8) ? clock64() : 0;
    __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0): this wave's DMA writes of the iteration have landed
    __syncthreads();
    const long long c3 = (ABL >= 8) ? clock64() : 0;
    if (ABL >= 8) { dbg[0] += c1 - c0; dbg[1] += c2 - c1; dbg[2] += c3 - c2; dbg[3] += grow_any ? 1 : 0; }

    // ================= reference update (cold unless a score left the window) =================
    if (grow_any) {
      const float mx = va_xhalf_max(mxp[0]);                          // row maximum of S_next relative to m_run
      const float dlt = __builtin_ceilf(fmaxf(mx, 0.f) * 16.f) * 0.0625f;      // new reference = m_run + dlt, on the 1/16 grid
      const float alpha = __builtin_amdgcn_exp2f(-dlt);
      m_run += dlt;
      if (!LTRICK) l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) Sn[blk][r] -= dlt;
      qa_m = hi ? 0u : va_split2<T>(-m_run);
    }
  };

  {
    int kslot = (nt > 1) ? 1 : 0;
    for (int t = 0; t < nt; t += 2) {
      tile(SA, SB, t, kslot);
      kslot = (kslot == 2) ? 0 : kslot + 1;
      if (t + 1 < nt) {
        tile(SB, SA, t + 1, kslot);
        kslot = (kslot == 2) ? 0 : kslot + 1;
      }
    }
  }
#undef SP_QA

  if (ABL >= 8 && blockIdx.x == 8 && lane == 0 && (wave & 3) == 0) {
    const long long cend = clock64();
    unsigned long long* o = g_va_dbg + (wave >> 2) * 8;
    o[0] = dbg[0]; o[1] = dbg[1]; o[2] = dbg[2]; o[3] = dbg[3]; o[4] = cend - cstart; o[5] = nt;
  }
  {
    const float l_tot = LTRICK ? __shfl(O[DB - 1][L_REG], li + 32 * L_HI) : l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qi < p.N) {
      T* orow = Og + (long)qi * p.o_st;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int d0 = 32 * d + 8 * rr + 4 * hi;
          if (d0 < HD) {
            typedef T t4 __attribute__((ext_vector_type(4)));
            t4 o4;
#pragma unroll
            for (int e = 0; e < 4; ++e) o4[e] = (T)(O[d][4 * rr + e] * inv);
            *reinterpret_cast<t4*>(orow + d0) = o4;
          }
        }
    }
  }
}

template <typename T, int HD, int NB, bool FAST, int ABL>
static int launch_sp(VAParams& p, hipStream_t st) {
  constexpr int KT = 32 * NB, DB = (HD + 31) / 32, WAVES = 8;
  constexpr int KSTR = HD + 8, VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;
  constexpr int KBLK = (KT * (KSTR / 8) + 63) / 64, VBLK = (KT * (VSTR / 8) + 63) / 64;
  size_t lds = (size_t)(3 * KBLK + 2 * VBLK) * 1024;
  const size_t stage = (size_t)WAVES * 32 * 32 * sizeof(float);
  if (lds < stage) lds = stage;
  lds += (size_t)(p.kh + 1) * WAVES * 32 * sizeof(unsigned int);
  if (lds > 160 * 1024) return set_err(HIPIE_EINVAL, "vit_attn(sp): %zu bytes of LDS needed (grid %dx%d) > 160 KiB", lds, p.kh, p.kw);
  p.nqt = (p.N + WAVES * 32 - 1) / (WAVES * 32);
  p.swz = ((p.B * p.H) % 8 == 0) ? 1 : 0;
  const unsigned grid = (unsigned)(p.nqt * p.B * p.H);
  { static int prio = -1; if (prio < 0) { const char* e = study_env("HIPIE_VA_PRIO"); prio = e ? atoi(e) : 1; } p.prio = prio; }
  auto kern = vit_attn_sp_kernel<T, HD, NB, FAST, ABL>;
  if (lds > 64 * 1024) {
    static size_t lds_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || lds > lds_set[dev]) {
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (dev >= 0 && dev < 64) lds_set[dev] = lds;
    }
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, p);
  return check_launch("vit_attn(sp)");
}

template <typename T, int HD, bool FAST>
static int dispatch_va(VAParams& p, hipStream_t st) {
  if (p.kw == 14 && p.kh <= 96) return launch_va<T, HD, 1, 7, FAST, 2, 14, 0>(p, st);     // the 14x14 windows: 196 queries = 7 waves
  if (p.kw <= 32) return launch_va<T, HD, 1, 4, FAST, 1, 0, 0>(p, st);
  if (p.kw > 32 && p.kw <= 64 && p.kh <= 96 && p.N >= 1024) {
    static int mode = -1;       // 0: plain kernel, 2: software-pipelined 8-wave kernel (default)
    if (mode < 0) { const char* e = study_env("HIPIE_VA_MODE"); mode = e ? atoi(e) : 2; }
#ifdef HIPIE_VA_ABLATIONS
    if (mode == 2 && HD == 80 && FAST) {
      const char* e = study_env("HIPIE_VA_ABL");
      if (e && atoi(e) == 8) return launch_sp<T, HD, 2, FAST, 8>(p, st);
      if (e && atoi(e) == 9) return launch_sp<T, HD, 2, FAST, 9>(p, st);
    }
#endif
    if (mode == 2) return launch_sp<T, HD, 2, FAST, 0>(p, st);
  }
#ifdef HIPIE_VA_ABLATIONS
  if (p.kw == 64 && HD == 80 && FAST && sizeof(T) == 2) {
    const char* e = study_env("HIPIE_VA_ABL");
    switch (e ? atoi(e) : 0) {
      case 1: return launch_va<T, HD, 2, 4, FAST, 1, 0, 1>(p, st);
      case 2: return launch_va<T, HD, 2, 4, FAST, 1, 0, 2>(p, st);
      case 3: return launch_va<T, HD, 2, 4, FAST, 1, 0, 3>(p, st);
      case 4: return launch_va<T, HD, 2, 4, FAST, 1, 0, 4>(p, st);
      case 5: return launch_va<T, HD, 2, 4, FAST, 1, 0, 5>(p, st);
      case 6: return launch_va<T, HD, 2, 4, FAST, 1, 0, 6>(p, st);
      case 7: return launch_va<T, HD, 2, 4, FAST, 1, 0, 7>(p, st);
      case 8: return launch_va<T, HD, 2, 4, FAST, 1, 0, 8>(p, st);
      case 9: return launch_va<T, HD, 2, 4, FAST, 1, 0, 9>(p, st);
      default: break;
    }
  }
#endif
  if (p.kw <= 64) return launch_va<T, HD, 2, 4, FAST, 1, 0, 0>(p, st);
  if (p.kw <= 96) return launch_va<T, HD, 3, 8, FAST, 1, 0, 0>(p, st);
  return set_err(HIPIE_EINVAL, "vit_attn: token grids wider than 96 are not supported (got %dx%d)", p.kh, p.kw);
}

}  // namespace hipie

#ifdef HIPIE_VA_ABLATIONS
extern "C" int hipie_va_debug(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(hipie::g_va_dbg), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int hipie_vit_attn_rel(const void* qkv, const void* tab_h, const void* tab_w, void* out, int B, int gh, int gw,
                                  int heads, int hd, int dtype, int flags, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(qkv && out && tab_h && tab_w, "vit_attn_rel: null pointer");
  HIPIE_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)tab_h | (uintptr_t)tab_w) & 15) == 0, "vit_attn_rel: pointers must be 16-byte aligned");
  HIPIE_REQUIRE(B > 0 && gh > 0 && gw > 0 && heads > 0, "vit_attn_rel: bad shape B=%d grid %dx%d heads=%d", B, gh, gw, heads);
  HIPIE_REQUIRE(gh <= 160, "vit_attn_rel: more than 160 key rows (%d) are not supported", gh);
  const long N = (long)gh * gw, C = (long)heads * hd;
  VAParams p{};
  p.qkv = qkv; p.out = out; p.tab_h = tab_h; p.tab_w = tab_w;
  p.B = B; p.H = heads; p.N = (int)N; p.kh = gh; p.kw = gw;
  p.sb = N * 3 * C; p.st = 3 * C; p.o_sb = N * C; p.o_st = C;
  const bool fast = (flags & HIPIE_ATTN_FAST) != 0;
  hipStream_t st = (hipStream_t)stream;
#define VA_CASE(T_, HD_) return fast ? dispatch_va<T_, HD_, true>(p, st) : dispatch_va<T_, HD_, false>(p, st)
  if (dtype == HIPIE_F16 && hd == 80) { VA_CASE(f16_t, 80); }
  if (dtype == HIPIE_BF16 && hd == 80) { VA_CASE(bf16_t, 80); }
  if (dtype == HIPIE_F16 && hd == 64) { VA_CASE(f16_t, 64); }
  if (dtype == HIPIE_BF16 && hd == 64) { VA_CASE(bf16_t, 64); }
#undef VA_CASE
  return set_err(HIPIE_EINVAL, "vit_attn_rel: needs dtype f16/bf16 and head_dim 64/80 (got dtype %d, hd %d)", dtype, hd);
}

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
#include "mfma.h"

namespace hipie {

struct FAParams {
  const void* q; const void* k; const void* v; void* out;
  int B, H, Nq, Nk;
  long q_sb, q_st, q_sh, k_sb, k_st, k_sh, v_sb, v_st, v_sh, o_sb, o_st, o_sh;
  const float* bias_h; const float* bias_w; int kh, kw;
  const uint8_t* key_mask;
  float scale, clamp;
  int nqt, ntiles, swz;
};

constexpr int FA_WAVES = 4;
constexpr int FA_QT = FA_WAVES * 32;
constexpr float kLog2e = 1.4426950408889634f;

template <typename T, int HD, int NB, bool BIAS, bool USE_TR>
__global__ __launch_bounds__(FA_WAVES * 64, (HD <= 80) ? 2 : 1) void flash_attn_kernel(const FAParams p) {
  constexpr int KT = 32 * NB;                  // keys per tile
  constexpr int KS = HD / 16;                  // k16 steps of QK^T
  constexpr int DB = (HD + 31) / 32;           // 32-row d blocks of O^T
  constexpr int KSTR = HD + 8;                 // K tile row stride (elements): 16-B aligned, conflict-free b128 reads
  constexpr int VSTR = DB * 32 + 8;            // V tile row stride (covers the padded d blocks)
  constexpr int CPR = HD / 8;                  // 16-byte chunks per row
  constexpr int NCH = KT * CPR;                // chunks per tile
  constexpr int NT = FA_WAVES * 64;
  constexpr int CPT = (NCH + NT - 1) / NT;     // chunks per thread
  constexpr int BUF = KT * (KSTR + VSTR);      // elements per LDS buffer
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

  const int qi = qt * FA_QT + wave * 32 + li;          // this lane's query
  const int qc = min(qi, p.Nq - 1);

  // zero LDS once (pad columns of the V tile feed the padded d rows of O^T, which are discarded but must stay finite-free of traps)
  for (int i = tid; i < 2 * BUF / 8; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  // ---- Q fragments (B operand): lane (q = li, half hi) holds Q[q][16 ks + 8 hi + j] ----
  frag qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    qf[ks] = *reinterpret_cast<const frag*>(Qg + (long)qc * p.q_st + 16 * ks + 8 * hi);

  // ---- bias_w in S^T register order ----
  float bw[NB][16];
  const float* bhp = nullptr;
  if (BIAS) {
    const float* bwp = p.bias_w + ((long)bh * p.Nq + qc) * p.kw;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) bw[blk][r] = bwp[min(32 * blk + crow(r, hi), p.kw - 1)];
    bhp = p.bias_h + ((long)bh * p.Nq + qc) * p.kh;
  }

  // ---- tile streaming helpers ----
  uint4 kr[CPT], vr[CPT];
  auto tile_geom = [&](int t, int& key0, int& nkeys) {
    if (BIAS) { key0 = t * p.kw; nkeys = p.kw; }
    else { key0 = t * KT; nkeys = min(KT, p.Nk - key0); }
  };
  auto load_regs = [&](int t) {
    int key0, nkeys;
    tile_geom(t, key0, nkeys);
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int idx = c * NT + tid;
      if (idx < NCH) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const long key = key0 + min(row, nkeys - 1);      // padded rows re-read the last valid key (finite data, P = 0)
        kr[c] = *reinterpret_cast<const uint4*>(Kg + key * p.k_st + ch * 8);
        vr[c] = *reinterpret_cast<const uint4*>(Vg + key * p.v_st + ch * 8);
      }
    }
  };
  auto store_lds = [&](int buf) {
    T* Ks = smem + buf * BUF;
    T* Vs = Ks + KT * KSTR;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int idx = c * NT + tid;
      if (idx < NCH) {
        const int row = idx / CPR, ch = idx - row * CPR;
        *reinterpret_cast<uint4*>(Ks + row * KSTR + ch * 8) = kr[c];
        *reinterpret_cast<uint4*>(Vs + row * VSTR + ch * 8) = vr[c];
      }
    }
  };

  // ---- online softmax state ----
  f32x16 O[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int nt = p.ntiles;
  __syncthreads();                 // LDS zero fill done
  load_regs(0);
  store_lds(0);
  __syncthreads();
  if (nt > 1) load_regs(1);
  float bh_next = BIAS ? bhp[0] : 0.f;

  for (int t = 0; t < nt; ++t) {
    const T* Ks = smem + (t & 1) * BUF;
    const T* Vs = Ks + KT * KSTR;
    int key0, nkeys;
    tile_geom(t, key0, nkeys);
    const float bh_t = bh_next;
    if (BIAS && t + 1 < nt) bh_next = bhp[t + 1];

    // validity words: bit i of word w <=> key (64 w + i) of this tile is attended to
    unsigned long long vw[(KT + 63) / 64];
#pragma unroll
    for (int w = 0; w < (KT + 63) / 64; ++w) {
      const int kk = 64 * w + lane;
      bool ok = kk < nkeys;
      if (Mg != nullptr && ok) ok = Mg[key0 + kk] != 0;
      vw[w] = __ballot(ok);
    }

    // ---- S^T = K . Q^T ----
    f32x16 S[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[blk][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        const frag kf = *reinterpret_cast<const frag*>(Ks + (32 * blk + li) * KSTR + 16 * ks + 8 * hi);
        S[blk] = Mfma32<T>::mma(kf, qf[ks], S[blk]);
      }
    }

    // ---- scale, clamp, bias, mask; tile max ----
    float mx = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = 32 * blk + crow(r, hi);
        float s = S[blk][r] * p.scale;
        if (p.clamp > 0.f) s = fminf(fmaxf(s, -p.clamp), p.clamp);
        if (BIAS) s += bh_t + bw[blk][r];
        const bool ok = (vw[kk >> 6] >> (kk & 63)) & 1ull;
        s = ok ? s : -INFINITY;
        S[blk][r] = s;
        mx = fmaxf(mx, s);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;      // all keys so far masked: keep exp() finite
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_use) * kLog2e);   // m_run = -inf -> 0
    m_run = m_new;
    float lsum = 0.f;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = __builtin_amdgcn_exp2f((S[blk][r] - m_use) * kLog2e);
        S[blk][r] = pv;
        lsum += pv;
      }
    l_run = l_run * alpha + lsum;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] *= alpha;

    // ---- O^T += V^T . P^T ----
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        frag pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (T)S[blk][8 * s + j];
        // keys of this lane-half for k slots j = 0..3 and 4..7
        const int krow0 = 32 * blk + 16 * s + 4 * hi;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          frag vf;
          if (USE_TR) {
            const int l16 = lane & 15, g1 = (lane >> 4) & 1;
            const T* a0 = Vs + (krow0 + (l16 >> 2)) * VSTR + 32 * d + 16 * g1 + 4 * (l16 & 3);
            const hfrag lo = Mfma32<T>::tr_read(a0);
            const hfrag hi4 = Mfma32<T>::tr_read(a0 + 8 * VSTR);
#pragma unroll
            for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi4[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) vf[j] = Vs[(krow0 + (j & 3) + 8 * (j >> 2)) * VSTR + 32 * d + li];
          }
          O[d] = Mfma32<T>::mma(vf, pf, O[d]);
        }
      }
    }

    if (t + 1 < nt) store_lds((t + 1) & 1);
    __syncthreads();
    if (t + 2 < nt) load_regs(t + 2);
  }

  // ---- epilogue ----
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  if (qi < p.Nq) {
    T* orow = Og + (long)qi * p.o_st;
#pragma unroll
    for (int d = 0; d < DB; ++d) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int d0 = 32 * d + 8 * rr + 4 * hi;           // rows crow(4 rr + (0..3), hi) of block d
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

template <typename T, int HD, int NB, bool BIAS>
static int launch_fa(const FAParams& p, hipStream_t st, bool use_tr) {
  constexpr int KT = 32 * NB, DB = (HD + 31) / 32;
  constexpr size_t lds = (size_t)2 * KT * ((HD + 8) + (DB * 32 + 8)) * sizeof(T);
  const unsigned grid = (unsigned)(p.nqt * p.B * p.H);
  if (use_tr) {
    auto kern = flash_attn_kernel<T, HD, NB, BIAS, true>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FA_WAVES * 64), lds, st, p);
  } else {
    auto kern = flash_attn_kernel<T, HD, NB, BIAS, false>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(FA_WAVES * 64), lds, st, p);
  }
  return check_launch("flash_attn");
}

template <typename T, int HD>
static int dispatch_nb(FAParams& p, hipStream_t st, bool use_tr) {
  if (p.bias_h != nullptr) {
    const int nb = (p.kw + 31) / 32;
    p.ntiles = p.kh;
    switch (nb) {
      case 1: return launch_fa<T, HD, 1, true>(p, st, use_tr);
      case 2: return launch_fa<T, HD, 2, true>(p, st, use_tr);
      case 3: return launch_fa<T, HD, 3, true>(p, st, use_tr);
      default: return set_err(HIPIE_EINVAL, "flash_attn: kw=%d > 96 unsupported with rel-pos bias", p.kw);
    }
  }
  p.ntiles = (p.Nk + 63) / 64;
  return launch_fa<T, HD, 2, false>(p, st, use_tr);
}

template <typename T>
static int dispatch_hd(FAParams& p, int hd, hipStream_t st, bool use_tr) {
  switch (hd) {
    case 32: return dispatch_nb<T, 32>(p, st, use_tr);
    case 64: return dispatch_nb<T, 64>(p, st, use_tr);
    case 80: return dispatch_nb<T, 80>(p, st, use_tr);
    case 256: return dispatch_nb<T, 256>(p, st, use_tr);
    default: return set_err(HIPIE_EINVAL, "flash_attn: head_dim %d unsupported (32, 64, 80, 256)", hd);
  }
}

static int flash_attn_impl(FAParams p, int hd, int dtype, void* stream) {
  HIPIE_REQUIRE(p.q && p.k && p.v && p.out, "flash_attn: null pointer");
  HIPIE_REQUIRE(p.B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "flash_attn: bad shape B=%d H=%d Nq=%d Nk=%d", p.B, p.H, p.Nq, p.Nk);
  HIPIE_REQUIRE((p.bias_h == nullptr) == (p.bias_w == nullptr), "flash_attn: bias_h and bias_w must be given together");
  if (p.bias_h) HIPIE_REQUIRE(p.kh > 0 && p.kw > 0 && (long)p.kh * p.kw == p.Nk, "flash_attn: kh*kw=%d*%d != Nk=%d", p.kh, p.kw, p.Nk);
  const long strides[] = {p.q_sb, p.q_st, p.q_sh, p.k_sb, p.k_st, p.k_sh, p.v_sb, p.v_st, p.v_sh, p.o_sb, p.o_st, p.o_sh};
  for (long s : strides) HIPIE_REQUIRE(s % 8 == 0, "flash_attn: strides must be multiples of 8 elements (16 bytes), got %ld", s);
  HIPIE_REQUIRE((((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v | (uintptr_t)p.out) & 15) == 0, "flash_attn: pointers must be 16-byte aligned");
  p.nqt = (p.Nq + FA_QT - 1) / FA_QT;
  p.swz = ((p.B * p.H) % 8 == 0) ? 1 : 0;
  const char* e = getenv("HIPIE_FA_NO_TR");
  const bool use_tr = !(e && e[0] == '1');
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HIPIE_F16: return dispatch_hd<f16_t>(p, hd, st, use_tr);
    case HIPIE_BF16: return dispatch_hd<bf16_t>(p, hd, st, use_tr);
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

extern "C" int hipie_bi_xattn(const void* q, const void* k, const void* vv, const void* vl, const uint8_t* text_mask,
                              void* out_v, void* out_l, int B, int H, int Nv, int L, int hd, float clamp, int dtype,
                              void* stream) {
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
  int rc = flash_attn_impl(a, hd, dtype, stream);
  if (rc != HIPIE_OK) return rc;
  // text rows attend over ALL image tokens: softmax over Nv of the transposed scores, no mask (fuse_helper.py:86-95)
  FAParams t{};
  t.q = k; t.k = q; t.v = vv; t.out = out_l;
  t.B = B; t.H = H; t.Nq = L; t.Nk = Nv;
  t.q_sb = (long)L * E; t.q_st = E; t.q_sh = hd;
  t.k_sb = t.v_sb = (long)Nv * E; t.k_st = t.v_st = E; t.k_sh = t.v_sh = hd;
  t.o_sb = (long)L * E; t.o_st = E; t.o_sh = hd;
  t.key_mask = nullptr; t.scale = 1.f; t.clamp = clamp;
  return flash_attn_impl(t, hd, dtype, stream);
}

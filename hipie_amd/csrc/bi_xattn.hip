// bi_xattn.hip -- the image -> text direction of the VL fusion attention (BiMultiHeadAttention, fuse_helper.py:69-121) for the
// shape the path runs it in: 21760 image tokens per image attend over ONE short text (L <= 224 tokens), head dim 256.
//
// The generic flash kernel (flash_attn.hip) walks the text in 32-key tiles with an online softmax; with head dim 256 that is 32
// MFMAs per 16 scores and lane, and its per-tile bookkeeping (running max, rescale of a 256-wide accumulator, mask / clamp per
// element, tile streaming) costs 10.7 VALU instructions per MFMA: 13 % matrix-pipe utilisation (profiles/r02_pmc.md).
// Here the whole text is ONE softmax window:
//   phase A  S^T = K . Q^T for all NKB 32-key blocks (K resident in LDS for the lifetime of the workgroup, Q fragments straight
//            from global memory as the B operand), 16 * NKB accumulators per lane;
//   phase B  one max / exp2 / sum over the lane's 16 * NKB scores (+ one cross-half exchange): no running state, no rescale;
//            clamp is one v_med3, the text mask one add of a per-row 0 / -inf table;
//   phase C  O^T = VL^T . P^T with the text values streamed through a two-slot LDS ring in 32-key tiles (the ring repeats the
//            same NKB tiles for every query tile), P used in-register as the B operand, VL^T fetched by ds_read_b64_tr_b16.
// A workgroup (8 waves x 32 queries) walks QT consecutive query tiles of one (image, head), so K is loaded once per 1280 queries.
// Bound: matrix pipe / LDS bandwidth (32 flop per LDS byte with 32 queries per wave = the CU's MFMA : LDS ratio).
#include <algorithm>
#include <cstdlib>

#include "mfma.h"

namespace hipie {

constexpr int XA_HD = 256, XA_KS = 16, XA_DB = 8;
constexpr int XA_KSTR = XA_HD + 8;       // K row stride (elements): conflict-free b128 reads (flash_attn.hip)
constexpr int XA_VSTR = XA_HD + 32;      // V row stride: conflict-free transposing reads
constexpr int XA_WAVES = 8;
constexpr int XA_QT = 5;                 // query tiles (of 256) per workgroup

struct XAParams {
  const void *q, *k, *vl;
  const uint8_t* mask;                   // (B, L) or null
  void* out;
  int B, H, Nv, L;
  long q_sb, q_st, kv_sb, kv_st, o_sb, o_st;     // element strides: batch, token (head stride = 256)
  float clamp_l2;                        // clamp * log2(e), 0 = no clamp
};

template <typename T, int NKB>
__global__ __launch_bounds__(512) void xattn_i2t_kernel(XAParams p) {
  typedef typename Mfma32<T>::frag frag;
  typedef typename Mfma32<T>::half_frag hfrag;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char xa_smem[];
  T* Ks = reinterpret_cast<T*>(xa_smem);                          // [NKB*32][KSTR]
  T* Vr = Ks + NKB * 32 * XA_KSTR;                                // [2][32][VSTR]
  float* mb = reinterpret_cast<float*>(Vr + 2 * 32 * XA_VSTR);    // [NKB][2][16]: 0 / -inf per S^T row in register order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5, l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const T* Qg = reinterpret_cast<const T*>(p.q) + (long)b * p.q_sb + (long)h * XA_HD;
  const T* Kg = reinterpret_cast<const T*>(p.k) + (long)b * p.kv_sb + (long)h * XA_HD;
  const T* Vg = reinterpret_cast<const T*>(p.vl) + (long)b * p.kv_sb + (long)h * XA_HD;
  T* Og = reinterpret_cast<T*>(p.out) + (long)b * p.o_sb + (long)h * XA_HD;
  const uint8_t* Mg = p.mask ? p.mask + (long)b * p.L : nullptr;

  // ---- K resident (rows >= L zero), mask table ----
  for (int idx = tid; idx < NKB * 32 * 32; idx += 512) {
    const int row = idx >> 5, ch = idx & 31;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < p.L) v = *reinterpret_cast<const u32x4*>(Kg + (long)row * p.kv_st + ch * 8);
    *reinterpret_cast<u32x4*>(Ks + row * XA_KSTR + ch * 8) = v;
  }
  for (int i = tid; i < NKB * 32; i += 512) {
    const int blk = i >> 5, hh = (i >> 4) & 1, r = i & 15;
    const int key = 32 * blk + crow(r, hh);
    const bool ok = key < p.L && (Mg == nullptr || Mg[key] != 0);
    mb[i] = ok ? 0.f : -INFINITY;
  }
  // VL tile streaming: thread -> two 16-byte chunks of a 32-row tile
  const int vrow0 = tid >> 5, vch = tid & 31;          // rows vrow0 and vrow0 + 16
  u32x4 vreg[2];
  auto vload = [&](int tile) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int row = 32 * tile + vrow0 + 16 * c;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (row < p.L) v = *reinterpret_cast<const u32x4*>(Vg + (long)row * p.kv_st + vch * 8);
      vreg[c] = v;
    }
  };
  auto vstore = [&](int slot) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      *reinterpret_cast<u32x4*>(Vr + (slot * 32 + vrow0 + 16 * c) * XA_VSTR + vch * 8) = vreg[c];
  };
  // zero the pad columns of the ring once (never rewritten; they only feed nothing, but keep LDS NaN-free)
  for (int i = tid; i < 2 * 32 * (XA_VSTR - XA_HD) / 8; i += 512) {
    const int row = i / ((XA_VSTR - XA_HD) / 8), c = i % ((XA_VSTR - XA_HD) / 8);
    *reinterpret_cast<u32x4*>(Vr + row * XA_VSTR + XA_HD + c * 8) = u32x4{0u, 0u, 0u, 0u};
  }
  vload(0);
  vstore(0);
  vload(1 % NKB);
  int step = 0;                                          // global tile step: slot = step & 1, tile = step % NKB

  const int nqt = (p.Nv + 255) >> 8;
  const int qt0 = blockIdx.x * XA_QT, qt1 = min(qt0 + XA_QT, nqt);
  const float kL2 = 1.4426950408889634f;
  for (int qt = qt0; qt < qt1; ++qt) {
    const int q = qt * 256 + wave * 32 + li;
    const int qc = min(q, p.Nv - 1);
    // ---- phase A ----
    frag qf[XA_KS];
#pragma unroll
    // k-slot labelling (free as long as A and B agree, mfma.h): lane half hi covers head-dim [128 hi, 128 hi + 128), k-step ks its
    // 8 elements at 8 ks -- every lane reads 256 CONTIGUOUS bytes of its query row over the 16 loads (two full 128-byte lines, each
    // touched by one lane only) instead of 16 scattered 16-byte pieces whose lines are shared by four instructions
    for (int ks = 0; ks < XA_KS; ++ks) qf[ks] = *reinterpret_cast<const frag*>(Qg + (long)qc * p.q_st + 128 * hi + 8 * ks);
    if (qt == qt0) __syncthreads();                      // K, mask table, ring slot 0 visible
    f32x16 S[NKB];
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[blk][r] = 0.f;
      const T* kb = Ks + (32 * blk + li) * XA_KSTR + 128 * hi;
#pragma unroll
      for (int ks = 0; ks < XA_KS; ++ks) S[blk] = Mfma32<T>::mma(*reinterpret_cast<const frag*>(kb + 8 * ks), qf[ks], S[blk]);
    }
    // ---- phase B: s = clamp(log2e * qk) + mask; one softmax over all 32 * NKB keys ----
    float mx = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(mb + (blk * 2 + hi) * 16 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = S[blk][4 * g + e] * kL2;
          if (p.clamp_l2 > 0.f) s = __builtin_amdgcn_fmed3f(s, -p.clamp_l2, p.clamp_l2);
          s += m4[e];
          S[blk][4 * g + e] = s;
          mx = fmaxf(mx, s);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mref = (mx == -INFINITY) ? 0.f : mx;
    float lsum = 0.f;
    frag pf[NKB][2];
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float pv = __builtin_amdgcn_exp2f(S[blk][8 * s2 + j] - mref);
          lsum += pv;
          pf[blk][s2][j] = (T)pv;
        }
    lsum += __shfl_xor(lsum, 32);
    const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
    // ---- phase C ----
    f32x16 O[XA_DB];
#pragma unroll
    for (int d = 0; d < XA_DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
#pragma unroll
    for (int blk = 0; blk < NKB; ++blk) {
      if (!(qt == qt0 && blk == 0)) __syncthreads();     // tile `step` is in its slot; the other slot is free
      vstore((step + 1) & 1);                            // tile step + 1 (prefetched) -> the free slot
      vload((step + 2) % NKB);
      const T* Vs = Vr + (step & 1) * 32 * XA_VSTR;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int krow0 = 16 * s2 + 4 * hi;
#pragma unroll
        for (int d = 0; d < XA_DB; ++d) {
          const T* a0 = Vs + (krow0 + (l16 >> 2)) * XA_VSTR + 32 * d + 16 * g1 + 4 * (l16 & 3);
          const hfrag lo = Mfma32<T>::tr_read(a0);
          const hfrag hi4 = Mfma32<T>::tr_read(a0 + 8 * XA_VSTR);
          frag vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi4[j]; }
          O[d] = Mfma32<T>::mma(vf, pf[blk][s2], O[d]);
        }
      }
      ++step;
    }
    // ---- epilogue: O^T rows = head-dim index, lane = query ----
    // the C layout gives a lane 4 consecutive head-dim values per (d, i) and its partner in the other lane half the next 4: the
    // halves exchange every second group (v_permlane32_swap) so that each lane stores 16 contiguous bytes and the two lanes of a
    // query complete a 32-byte sector per instruction (half the store instructions, no partial sectors)
    {
      T* orow = Og + (long)min(q, p.Nv - 1) * p.o_st;
      const bool live = q < p.Nv;
#pragma unroll
      for (int d = 0; d < XA_DB; ++d)
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {                   // register groups i = 2 ip (kept by half 0) and 2 ip + 1 (kept by half 1)
          typedef T t4 __attribute__((ext_vector_type(4)));
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          t4 a4, b4;
#pragma unroll
          for (int e = 0; e < 4; ++e) { a4[e] = (T)(O[d][8 * ip + e] * inv); b4[e] = (T)(O[d][8 * ip + 4 + e] * inv); }
          u32x2 a = __builtin_bit_cast(u32x2, a4), b = __builtin_bit_cast(u32x2, b4);
          // half 0 sends b (group 2 ip + 1), half 1 sends a (group 2 ip): after the exchange `mine`/`theirs` are the two 8-byte
          // pieces of one 16-byte run: half 0 -> head-dim 32 d + 16 ip + [0, 8), half 1 -> 32 d + 16 ip + [8, 16)
          const u32x2 send = hi ? a : b, keep = hi ? b : a;
          u32x2 recv;
          recv[0] = __shfl_xor((int)send[0], 32);
          recv[1] = __shfl_xor((int)send[1], 32);
          u32x4 o;
          if (hi == 0) { o[0] = keep[0]; o[1] = keep[1]; o[2] = recv[0]; o[3] = recv[1]; }
          else { o[0] = recv[0]; o[1] = recv[1]; o[2] = keep[0]; o[3] = keep[1]; }
          if (live) *reinterpret_cast<u32x4*>(orow + 32 * d + 16 * ip + 8 * hi) = o;
        }
    }
  }
}

template <typename T, int NKB>
static int launch_i2t(XAParams& p, hipStream_t st) {
  const size_t lds = (size_t)(NKB * 32 * XA_KSTR + 2 * 32 * XA_VSTR) * sizeof(T) + (size_t)NKB * 32 * sizeof(float);
  auto kern = xattn_i2t_kernel<T, NKB>;
  static size_t lds_set[64] = {0};
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || lds > lds_set[dev])) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = lds;
  }
  const int nqt = (p.Nv + 255) / 256;
  dim3 grid((nqt + XA_QT - 1) / XA_QT, p.B * p.H);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
  return check_launch("bi_xattn_i2t");
}

// image -> text direction for hd 256 and 64 < L <= 224; returns HIPIE_OK after launching, or 1 if the shape is not covered
int xattn_i2t_try(const void* q, const void* k, const void* vl, const uint8_t* mask, void* out, int B, int H, int Nv, int L, int hd,
                  long E, float clamp, int dtype, hipStream_t st) {
  if (hd != XA_HD || L <= 64 || L > 224 || Nv <= 0) return 1;
  XAParams p{};
  p.q = q; p.k = k; p.vl = vl; p.mask = mask; p.out = out;
  p.B = B; p.H = H; p.Nv = Nv; p.L = L;
  p.q_sb = (long)Nv * E; p.q_st = E; p.kv_sb = (long)L * E; p.kv_st = E; p.o_sb = (long)Nv * E; p.o_st = E;
  p.clamp_l2 = clamp * 1.4426950408889634f;
  if (dtype == HIPIE_F16) return L <= 128 ? launch_i2t<f16_t, 4>(p, st) : launch_i2t<f16_t, 7>(p, st);
  if (dtype == HIPIE_BF16) return L <= 128 ? launch_i2t<bf16_t, 4>(p, st) : launch_i2t<bf16_t, 7>(p, st);
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// text -> image direction: every text token attends over ALL image tokens of its image (softmax over 21760 keys, no mask;
// fuse_helper.py:86-95).  Only L <= 224 queries per (image, head): the generic kernel has 2 workgroups of work per head and
// walks 680 key tiles serially (1.06 ms).  Here the roles are swapped so that the SHORT side sits in registers:
//   S = Qimg . Ktxt^T with the image tile as the MFMA rows (A operand from LDS) and one 32-token text block per wave as the
//   columns, its K fragments held in registers for the whole kernel (64 VGPRs);  the softmax index (image token) is then the
//   register index of the C layout and the text token the LANE: running max / sum are per-lane scalars, no cross-lane
//   reduction beyond one exchange between the lane halves;  P is used in place as the B operand of
//   out_l^T (head-dim x text) += vv^T . P,  vv^T fetched from the LDS tile by ds_read_b64_tr_b16.
// The image range of a head is split over SP workgroups (flash-decoding); each writes (m, l, unnormalised acc) to a workspace
// and a small kernel combines the splits.  7 waves compute (text blocks), all 8 stream the image tiles through a 3-slot ring.
struct XTParams {
  const void *q, *k, *vv;                // q: image queries (B, Nv, E); k: text keys (B, L, E); vv: image values (B, Nv, E)
  void* out;                             // (B, L, E)
  float* ws;                             // [BH][SP][ m[TB] | l[TB] | acc[TB][256] ],  TB = 32 * NKB
  int B, H, Nv, L, SP, tiles_per_split;
  long q_sb, q_st, kv_sb, kv_st, o_sb, o_st;
  float clamp_l2;
};

template <typename T, int NKB>
__global__ __launch_bounds__(512) void xattn_t2i_kernel(XTParams p) {
  typedef typename Mfma32<T>::frag frag;
  typedef typename Mfma32<T>::half_frag hfrag;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) unsigned char xa_smem[];
  constexpr int QTILE = 32 * XA_KSTR, VTILE = 32 * XA_VSTR, STAGE = QTILE + VTILE;
  T* ring = reinterpret_cast<T*>(xa_smem);                        // [3][ Q tile | vv tile ]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5, l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H, sp = blockIdx.x;
  const T* Qg = reinterpret_cast<const T*>(p.q) + (long)b * p.q_sb + (long)h * XA_HD;
  const T* Vg = reinterpret_cast<const T*>(p.vv) + (long)b * p.q_sb + (long)h * XA_HD;
  const T* Kg = reinterpret_cast<const T*>(p.k) + (long)b * p.kv_sb + (long)h * XA_HD;
  const int ntiles = (p.Nv + 31) >> 5;
  const int t0 = sp * p.tiles_per_split, t1 = min(t0 + p.tiles_per_split, ntiles);
  const bool compute = wave < NKB;

  // this wave's text block as the B operand, in registers (rows >= L: clamped duplicates, never stored)
  frag kf[XA_KS];
  {
    const int trow = min(32 * min(wave, NKB - 1) + li, p.L - 1);
#pragma unroll
    for (int ks = 0; ks < XA_KS; ++ks) kf[ks] = *reinterpret_cast<const frag*>(Kg + (long)trow * p.kv_st + 128 * hi + 8 * ks);
  }
  // tile streaming: 512 threads x (2 Q chunks + 2 vv chunks) of 16 bytes
  const int crow0 = tid >> 5, cch = tid & 31;
  u32x4 qreg[2], vreg[2];
  auto tload = [&](int tile) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int row = min(32 * tile + crow0 + 16 * c, p.Nv - 1);          // tail rows: finite duplicates, masked below
      qreg[c] = *reinterpret_cast<const u32x4*>(Qg + (long)row * p.q_st + cch * 8);
      vreg[c] = *reinterpret_cast<const u32x4*>(Vg + (long)row * p.q_st + cch * 8);
    }
  };
  auto tstore = [&](int slot) {
    T* Qs = ring + slot * STAGE;
    T* Vs = Qs + QTILE;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      *reinterpret_cast<u32x4*>(Qs + (crow0 + 16 * c) * XA_KSTR + cch * 8) = qreg[c];
      *reinterpret_cast<u32x4*>(Vs + (crow0 + 16 * c) * XA_VSTR + cch * 8) = vreg[c];
    }
  };
  f32x16 O[XA_DB];
#pragma unroll
  for (int d = 0; d < XA_DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float kL2 = 1.4426950408889634f;

  if (t0 < t1) {
    tload(t0);
    tstore(0);
    if (t0 + 1 < t1) tload(t0 + 1);
  }
  for (int t = t0; t < t1; ++t) {
    const int it = t - t0;
    __syncthreads();                                   // tile t is in slot it % 3; slot (it + 1) % 3 was last read in iteration it - 2
    if (t + 1 < t1) tstore((it + 1) % 3);
    if (t + 2 < t1) tload(t + 2);
    if (!compute) continue;
    const T* Qs = ring + (it % 3) * STAGE;
    const T* Vs = Qs + QTILE;
    // ---- S = Qimg . Ktxt^T : rows = image tokens of the tile, columns = this wave's text tokens ----
    f32x16 S;
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = 0.f;
    const T* qa = Qs + li * XA_KSTR + 128 * hi;
#pragma unroll
    for (int ks = 0; ks < XA_KS; ++ks) S = Mfma32<T>::mma(*reinterpret_cast<const frag*>(qa + 8 * ks), kf[ks], S);
    float mx = -INFINITY;
    const bool tail = (32 * t + 32 > p.Nv);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = S[r] * kL2;
      if (p.clamp_l2 > 0.f) s = __builtin_amdgcn_fmed3f(s, -p.clamp_l2, p.clamp_l2);
      if (tail && 32 * t + crow(r, hi) >= p.Nv) s = -INFINITY;
      S[r] = s;
      mx = fmaxf(mx, s);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    if (__builtin_amdgcn_ballot_w64(mx > m_run) != 0ull) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);       // m_run = -inf -> 0 (m_new is finite: a tile has >= 1 valid row)
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < XA_DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
    }
    frag pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = __builtin_amdgcn_exp2f(S[8 * s2 + j] - m_run);
        l_run += pv;
        pf[s2][j] = (T)pv;
      }
    // ---- out_l^T (head-dim x text) += vv^T . P ----
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int krow0 = 16 * s2 + 4 * hi;
#pragma unroll
      for (int d = 0; d < XA_DB; ++d) {
        const T* a0 = Vs + (krow0 + (l16 >> 2)) * XA_VSTR + 32 * d + 16 * g1 + 4 * (l16 & 3);
        const hfrag lo = Mfma32<T>::tr_read(a0);
        const hfrag hi4 = Mfma32<T>::tr_read(a0 + 8 * XA_VSTR);
        frag vf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi4[j]; }
        O[d] = Mfma32<T>::mma(vf, pf[s2], O[d]);
      }
    }
  }
  if (!compute) return;
  // ---- partial result of this split: m, l per text token, unnormalised accumulator (text, head-dim) ----
  l_run += __shfl_xor(l_run, 32);
  constexpr int TB = 32 * NKB;
  float* wsp = p.ws + ((long)bh * p.SP + sp) * (2 * TB + (long)TB * XA_HD);
  const int tcol = 32 * wave + li;
  if (hi == 0) { wsp[tcol] = m_run; wsp[TB + tcol] = l_run; }
  float* acc = wsp + 2 * TB + (long)tcol * XA_HD;
#pragma unroll
  for (int d = 0; d < XA_DB; ++d)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<f32x4*>(acc + 32 * d + 8 * i + 4 * hi) = f32x4{O[d][4 * i], O[d][4 * i + 1], O[d][4 * i + 2], O[d][4 * i + 3]};
}

// out_l[b, t, h*256 + c] = sum_sp acc_sp * 2^(m_sp - M) / sum_sp l_sp * 2^(m_sp - M)
template <typename T>
__global__ __launch_bounds__(256) void xattn_t2i_combine_kernel(const float* __restrict__ ws, T* __restrict__ out, int H, int L, int SP,
                                                                int TB, long o_sb, long o_st) {
  const int t = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H, c = threadIdx.x;
  const long stride = 2 * TB + (long)TB * XA_HD;
  const float* base = ws + (long)bh * SP * stride;
  float M = -INFINITY;
  for (int s = 0; s < SP; ++s) M = fmaxf(M, base[s * stride + t]);
  float num = 0.f, den = 0.f;
  for (int s = 0; s < SP; ++s) {
    const float m = base[s * stride + t];
    const float w = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - M);        // a split without tiles contributes nothing
    den += base[s * stride + TB + t] * w;
    num += base[s * stride + 2 * TB + (long)t * XA_HD + c] * w;
  }
  out[(long)b * o_sb + (long)t * o_st + (long)h * XA_HD + c] = (T)(den > 0.f ? num / den : 0.f);
}

template <typename T, int NKB>
static int launch_t2i(XTParams& p, hipStream_t st) {
  const size_t lds = (size_t)3 * (32 * XA_KSTR + 32 * XA_VSTR) * sizeof(T);
  auto kern = xattn_t2i_kernel<T, NKB>;
  static size_t lds_set[64] = {0};
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && (dev < 0 || dev >= 64 || lds > lds_set[dev])) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = lds;
  }
  hipLaunchKernelGGL(kern, dim3(p.SP, p.B * p.H), dim3(512), lds, st, p);
  int rc = check_launch("bi_xattn_t2i");
  if (rc != HIPIE_OK) return rc;
  hipLaunchKernelGGL((xattn_t2i_combine_kernel<T>), dim3(p.L, p.B * p.H), dim3(256), 0, st, p.ws, (T*)p.out, p.H, p.L, p.SP, 32 * NKB,
                     p.o_sb, p.o_st);
  return check_launch("bi_xattn_t2i_combine");
}

static int t2i_splits(int B, int H, int Nv) {
  const int ntiles = (Nv + 31) / 32;
  static const int forced = study_env("HIPIE_XT_SP") ? atoi(study_env("HIPIE_XT_SP")) : 0;      // tuning experiments
  int sp = forced > 0 ? forced : (256 + B * H - 1) / (B * H);      // one workgroup per CU (measured at B*H = 64: SP 4 1.22 ms both directions, 8 1.26, 16 1.32)
  sp = std::max(1, std::min(sp, std::min(16, ntiles / 8)));   // at least 8 tiles per split
  return std::max(sp, 1);
}

size_t xattn_t2i_workspace(int B, int H, int Nv, int L, int hd) {
  if (hd != XA_HD || L <= 64 || L > 224 || Nv <= 0) return 0;
  const int TB = (L <= 128) ? 128 : 224;
  return (size_t)B * H * t2i_splits(B, H, Nv) * (2 * TB + (size_t)TB * XA_HD) * sizeof(float);
}

// text -> image direction; 1 = shape not covered / no workspace
int xattn_t2i_try(const void* q, const void* k, const void* vv, void* out, float* ws, size_t ws_bytes, int B, int H, int Nv, int L,
                  int hd, long E, float clamp, int dtype, hipStream_t st) {
  const size_t need = xattn_t2i_workspace(B, H, Nv, L, hd);
  if (need == 0 || ws == nullptr || ws_bytes < need) return 1;
  XTParams p{};
  p.q = q; p.k = k; p.vv = vv; p.out = out; p.ws = ws;
  p.B = B; p.H = H; p.Nv = Nv; p.L = L;
  p.SP = t2i_splits(B, H, Nv);
  const int ntiles = (Nv + 31) / 32;
  p.tiles_per_split = (ntiles + p.SP - 1) / p.SP;
  p.q_sb = (long)Nv * E; p.q_st = E; p.kv_sb = (long)L * E; p.kv_st = E; p.o_sb = (long)L * E; p.o_st = E;
  p.clamp_l2 = clamp * 1.4426950408889634f;
  if (dtype == HIPIE_F16) return L <= 128 ? launch_t2i<f16_t, 4>(p, st) : launch_t2i<f16_t, 7>(p, st);
  if (dtype == HIPIE_BF16) return L <= 128 ? launch_t2i<bf16_t, 4>(p, st) : launch_t2i<bf16_t, 7>(p, st);
  return 1;
}

}  // namespace hipie

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
  int abl;                               // timing ablations (tools/bench_xattn.py, HIPIE_XA_ABL); 0 in production
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
      for (int ks = 0; ks < XA_KS; ++ks) {
        if ((p.abl & 16) && ks >= XA_KS / 2) break;
        S[blk] = Mfma32<T>::mma(*reinterpret_cast<const frag*>(kb + 8 * ks), qf[ks], S[blk]);
      }
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
          const float pv = (p.abl & 2) ? S[blk][8 * s2 + j] : __builtin_amdgcn_exp2f(S[blk][8 * s2 + j] - mref);
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
      if (!(p.abl & 4)) {
        if (!(qt == qt0 && blk == 0)) __syncthreads();   // tile `step` is in its slot; the other slot is free
        vstore((step + 1) & 1);                          // tile step + 1 (prefetched) -> the free slot
        vload((step + 2) % NKB);
      }
      const T* Vs = Vr + (step & 1) * 32 * XA_VSTR;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        if ((p.abl & 8) && s2 == 1) break;
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
      const bool live = q < p.Nv && (!(p.abl & 1) || inv == 12345.f);
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
  { const char* e = getenv("HIPIE_XA_ABL"); p.abl = e ? atoi(e) : 0; }
  if (dtype == HIPIE_F16) return L <= 128 ? launch_i2t<f16_t, 4>(p, st) : launch_i2t<f16_t, 7>(p, st);
  if (dtype == HIPIE_BF16) return L <= 128 ? launch_i2t<bf16_t, 4>(p, st) : launch_i2t<bf16_t, 7>(p, st);
  return 1;
}

}  // namespace hipie

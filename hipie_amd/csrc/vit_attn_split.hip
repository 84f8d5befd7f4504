// vit_attn_split.hip -- ViT attention (global and 14x14-windowed blocks, decomposed relative-position bias computed in the kernel)
// on SPLIT fp16 operands: the fp32-class form of hipie_vit_attn_rel for the policy that has to match the reference's fp32
// attention (hipie/backbone/vit.py:67-83 + hipie/backbone/utils.py:96-125) to 1e-3 through 32 blocks.
//
//     out[i,:] = softmax_j( q'_i.k_j + q'_i.Rh'[yi - yj + kh - 1] + q'_i.Rw'[xi - xj + kw - 1] ) . v_j        (exp2 domain)
//
// Operands (HIPIE_HL8: groups of 8 values as 8 fp16 hi + 8 fp16 lo, include/hipie_mi355.h): the packed qkv rows (B, N, 3, heads,
// hd) with q' = scale * log2(e) * q, and the two tables R' = R / scale.  Every product that feeds a LOGIT is formed from both
// halves of both operands,  a.b = a_lo.b_hi + a_hi.b_lo + a_hi.b_hi  (fp32 accumulation; fp16 x fp16 products are exact in
// fp32), so the scores and both bias terms are fp32-class: a score error enters the probabilities multiplied by exp().  The
// probabilities are an fp16 PAIR as well (P_hi + P_lo) and V keeps both halves:  O += V_hi^T.(P_hi + P_lo) + V_lo^T.P_hi -- three
// products.  (Round 3 ran P as ONE fp16 at first: fine at the fixture depths, but at the headline configuration the six decoder layers
// amplify the 2e-5 that leaves in the backbone features into 2e-3 on the CondInst mask logits; tools/dec_err_full.py.)  Lazy running maximum (moves on growth > 2^6), fp32 row sums of the UNROUNDED probabilities.
// tests/study/prec_sim.py: with single-fp16 q / k the a22 outputs of the full-depth fixture move by 1.4e-3, with single-fp16 V by 6e-4.
//
// Structure = vit_attn_kernel (vit_attn.hip): swapped products S^T = K.Q'^T (C operand = bias_w, -inf on padded key slots) and
// O^T = V^T.P^T with P used in place as the B operand and V^T fetched by ds_read_b64_tr_b16; one key tile = R key rows of the token
// grid.  Differences: the K / V tiles go L2 -> LDS by LDS-DMA with the hi / lo halves DE-INTERLEAVED into separate planes (the DMA
// writes LDS lane-linearly but reads a per-lane source address), so every LDS read pattern is the conflict-free one of the
// single-fp16 kernel; two tile buffers, one barrier per tile; the output is written as HL8 = the A operand of the projection GEMM.
#include <stdlib.h>

#include "common.h"
#include "mfma.h"

#ifndef HIPIE_VS_TAIL
#define HIPIE_VS_TAIL 1        // 0: hd 80 as three 32-row blocks with a ones column (the round-4 form), for A/B timing builds only
#endif
#ifndef HIPIE_VS_TAIL_MAXNB
#define HIPIE_VS_TAIL_MAXNB 3  // the tail form is used by instances of up to this many 32-key blocks per tile (2: the 96-slot instance keeps three padded blocks)
#endif

namespace hipie {

constexpr float VS_LAZY = 6.f;       // exp2(6) = 64: P <= 64, far inside fp16; a tile's P.V partial sums stay far inside fp32

struct VSParams {
  const f16_t* qkv; f16_t* out; const f16_t* tab_h; const f16_t* tab_w;
  int B, H, N, kh, kw;
  long sb, st;                  // qkv strides in fp16 elements: batch, token (= 2 * 3C); q at +0, k at +2C, v at +4C (C = H * HD)
  long o_sb, o_st;              // out strides in fp16 elements (token rows of 2C)
  int nqt, swz;
  // TRANSPOSED walk (grids wider than 96 tokens whose height is <= 96, e.g. 64 x 128 = a 1024 x 2048 image): the kernel's "key row" is
  // a COLUMN of the token grid -- kh / kw / tab_h / tab_w arrive swapped, logical token i = (ly, lx) lives at memory token lx * kwm + ly.
  // The logits are symmetric in the two axes, every per-token access is per-lane already, so only three address maps change.
  int tr, kwm;
  long krs, kts;                // fp16 elements between consecutive key slots of a tile / between consecutive tiles
};

__device__ __forceinline__ float vs_max3(float a, float b, float c) {
  float d;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

__device__ __forceinline__ float vs_xhalf_max(float x) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const unsigned int u = __builtin_bit_cast(unsigned int, x);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)r[0]), __builtin_bit_cast(float, (unsigned int)r[1]));
}

// two probabilities -> one VGPR of fp16 hi halves and one of lo halves, lo = fp16(p - hi) from ONE instruction each: v_fma_mix{lo,hi}_f16
// forms p - hi exactly (the fp16 hi enters as an fp16 source operand) and rounds once.  The C++ form `(T)(pv - (float)(T)pv)` costs a
// v_cvt_f16_f32, a v_cvt_f32_f16 and a v_sub_f32 per VALUE on top of the two packs: 4 instead of 1.5 VALU per probability, 80 of the
// ~230 VALU instructions a wave issues per 64-key tile of the global-attention instance.
__device__ __forceinline__ void vs_split2(const float a, const float b, unsigned int& H, unsigned int& L) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(HIPIE_NO_FMA_MIX)
  typedef _Float16 h2v __attribute__((ext_vector_type(2)));
  h2v h, l;
  h[0] = (f16_t)a; h[1] = (f16_t)b;
  l[0] = (f16_t)(a - (float)h[0]); l[1] = (f16_t)(b - (float)h[1]);
  H = __builtin_bit_cast(unsigned int, h);
  L = __builtin_bit_cast(unsigned int, l);
#elif defined(__HIP_DEVICE_COMPILE__)
  unsigned int h, l;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(a));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(b));
  H = h;
  L = l;
#endif
}

// hipcc's hazard recogniser does not look inside inline asm: a VGPR written by the statements above and read by the NEXT instruction as an
// MFMA operand or by v_permlane*_swap is read too early (gfx950 needs 2 wait states there; round 5: the a22 error of the full-depth
// fixture went from 5e-5 to 1e-3 -- isolated stale fragments -- until this was added; tools/split_form_check.py shows the split itself
// is bit-identical to the C++ form).  One s_nop behind a block of splits, tied to every register the block wrote.
__device__ __forceinline__ void vs_settle(unsigned int (&h)[4], unsigned int (&l)[4]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 1" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(l[0]), "+v"(l[1]), "+v"(l[2]), "+v"(l[3]));
#endif
}

// The other direction: the probabilities come out of v_exp_f32, and on gfx950 a transcendental result needs one wait state before a
// non-transcendental VALU instruction reads it.  hipcc inserts it for instructions it can see -- not for the asm statements of vs_split2.
// Where the scheduler happened to put something between the two nothing showed; in the 96-slot instance with the 16-row tail it did not,
// and the splits read stale registers (garbage outputs, round 5).  One s_nop behind the block of exps, tied to all of them.
__device__ __forceinline__ void vs_exp_settle(float (&pv)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_nop 0" : "+v"(pv[0]), "+v"(pv[1]), "+v"(pv[2]), "+v"(pv[3]), "+v"(pv[4]), "+v"(pv[5]), "+v"(pv[6]), "+v"(pv[7]));
#endif
}

// LDS-DMA of 16 bytes per lane: LDS[lds_dst + 16 * lane] <- *(sbase + voff) for the lanes of `mask` (see gemm.hip / vit_attn.hip for the
// inline-asm form).  The lane mask (a tile's row-padding chunks are neither fetched nor written) is applied inside the statement --
// s_and_saveexec / s_mov exec around the load, no branch -- and M0 is simply overwritten (nothing else in this kernel uses it): 5 scalar
// instructions per DMA where the compiler's own predication + an M0 save / restore took 12.
__device__ __forceinline__ void vs_dma16(const char* sbase, unsigned int voff, unsigned int lds_dst, unsigned long long mask) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned long long save;
  asm volatile("s_and_saveexec_b64 %0, %4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
               : "=&s"(save) : "v"(voff), "s"(sbase), "s"(lds_dst), "s"(mask) : "memory", "m0");
#endif
}

// HD: head dim (80 | 64);  NB: 32-slot key blocks per tile;  R: key rows of the token grid per tile;  KW: compile-time grid width when R > 1
// BHC: 0 = the bias_h of every key row is computed up front ([kh][queries] floats of LDS); > 0 = it is recomputed per chunk of BHC
// key rows (<= 31: a wave's 32 queries span at most two query rows, so 32 consecutive table rows cover a chunk) -- the form for grids
// wider than 64 (the 84x84 grid of the 1344-pixel configuration), whose 96-slot double-buffered tiles leave no room for kh rows.
template <int HD, int NB, int WAVES, int R, int KW, int BHC>
__global__ __launch_bounds__(WAVES * 64, (NB >= 3 ? 1 : 2)) void vit_attn_split_kernel(const VSParams p) {
  typedef f16_t T;
  constexpr int KT = 32 * NB;                  // key slots per tile
  constexpr int KS = HD / 16;                  // k16 steps of QK^T
  constexpr int DB = (HD + 31) / 32;           // 32-row d blocks of O^T
  constexpr int KSTR = HD + 8;                 // K plane row stride (elements): an odd number of 16-byte chunks -> conflict-free b128 reads
  constexpr int VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;     // V plane row stride (vit_attn.hip)
  constexpr int CPR = HD / 8;                  // data chunks per plane row
  constexpr int KCH = KSTR / 8, VCH = VSTR / 8;
  constexpr int KBLK = (KT * KCH + 63) / 64, VBLK = (KT * VCH + 63) / 64;      // 1 KiB DMA blocks per plane
  constexpr int KPL = KBLK * 512, VPL = VBLK * 512;                            // elements per plane
  constexpr int BUF = 2 * KPL + 2 * VPL;                                       // elements per tile buffer: K_hi | K_lo | V_hi | V_lo
  constexpr int NBLK = 2 * KBLK + 2 * VBLK;
  constexpr int NDMA = (NBLK + WAVES - 1) / WAVES;
  constexpr int QW = WAVES * 32;
  constexpr int NT = WAVES * 64;
  // hd = 80 leaves d rows 80 .. 95 of the last O^T block unused: column 80 of the V_hi plane holds 1.0 (written once; the DMA skips the
  // padding chunks), so O^T row 80 accumulates the row sums of the (fp16) probabilities on the matrix pipe -- no VALU adds for them
  // hd = 80 = 2 x 32 + 16: the d rows 64 .. 79 of O^T are a 16-row TAIL computed with v_mfma_f32_16x16x32_f16 (16 d rows x 16 queries x 32
  // keys) instead of a third 32-row block of which half would be padding -- 60 instead of 66 MFMA-equivalents per 64-key tile (the matrix
  // pipe is the power-limited resource of this kernel: time follows the MFMA work issued, not the instruction count).  The tail's B operand
  // needs a lane's 16 queries x 4 k-groups where the S^T layout has 32 queries x 2 halves: two v_permlane16_swap per VGPR re-deal the two
  // 16-key steps of a 32-key block into the fragments of queries 0-15 (X) and 16-31 (Y); see the PV loop.
  // (all instances; the 96-slot one of the 84 x 84 grid gave garbage with it until vs_exp_settle: its splits read v_exp_f32 results early)
  constexpr bool TAIL = (HD % 32 == 16) && (HIPIE_VS_TAIL != 0) && NB <= HIPIE_VS_TAIL_MAXNB;
  constexpr int DBB = TAIL ? HD / 32 : DB;     // full 32-row d blocks that run on the 32x32x16 MFMA
  constexpr bool ONES = (DB * 32 > HD) && !TAIL;
  static_assert(R == 1 || (KW > 0 && R * KW <= KT), "R key rows of KW keys must fit the tile");
  static_assert(BHC == 0 || (R == 1 && BHC <= 31), "chunked bias_h: one key row per tile, at most 31 rows per chunk");
  static_assert(KSTR % 8 == 0 && VSTR % 8 == 0, "plane rows are whole 16-byte chunks");
  typedef Mfma32<T>::frag frag;
  typedef Mfma32<T>::half_frag hfrag;

  extern __shared__ __attribute__((aligned(1024))) char smem_raw[];
  T* smem = reinterpret_cast<T*>(smem_raw);
  constexpr size_t TILE_BYTES = ((size_t)2 * BUF * sizeof(T) > (size_t)WAVES * 4096) ? (size_t)2 * BUF * sizeof(T) : (size_t)WAVES * 4096;
  float* bh_all = reinterpret_cast<float*>(smem_raw + TILE_BYTES);            // [kh | BHC][QW], log2 domain

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int kw = (R > 1) ? KW : p.kw, kh = p.kh;

  int bh, qt;
  {
    const int id = blockIdx.x;
    if (p.swz) { bh = (id & 7) + 8 * ((id >> 3) / p.nqt); qt = (id >> 3) % p.nqt; }
    else { bh = id / p.nqt; qt = id % p.nqt; }
  }
  const int b = bh / p.H, h = bh % p.H;
  const long C2 = 2L * p.H * HD;                               // fp16 elements of one of q / k / v per token
  const T* Qg = p.qkv + b * p.sb + (long)h * (2 * HD);
  const T* Kg = Qg + C2;
  T* Og = p.out + b * p.o_sb + (long)h * (2 * HD);

  const int qi = qt * QW + wave * 32 + li;
  const int qc = min(qi, p.N - 1);
  const int qy = qc / kw, qx = qc - qy * kw;
  const long qm = p.tr ? (long)qx * p.kwm + qy : (long)qc;       // memory token of this lane's query
  const int nkt = R * kw;

  // ---- Q fragments (B operand), both halves: lane (q = li, half hi) holds group 2 ks + hi of its query row ----
  frag qh[KS], ql[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const T* s = Qg + qm * p.st + 16 * (2 * ks + hi);
    qh[ks] = *reinterpret_cast<const frag*>(s);
    ql[ks] = *reinterpret_cast<const frag*>(s + 8);
  }

  // ---- decomposed rel-pos bias from the split tables (three products each), exactly as vit_attn_kernel lays it out ----
  f32x16 bw[NB];
#pragma unroll
  for (int blk = 0; blk < NB; ++blk)
#pragma unroll
    for (int r = 0; r < 16; ++r) bw[blk][r] = -INFINITY;
  {
    float* stage = reinterpret_cast<float*>(smem_raw) + wave * (32 * 32);      // aliases the (still unused) tile buffers
    const int nbw = (2 * kw - 1 + 31) / 32, nbh = (2 * kh - 1 + 31) / 32;
    auto table_block = [&](const T* tab, const int rows, const int jb) -> f32x16 {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const T* ap = tab + (long)min(32 * jb + li, rows - 1) * (2 * HD) + 16 * hi;        // (vs_table_rows below: the same at any start row)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag th = *reinterpret_cast<const frag*>(ap + 32 * ks), tl = *reinterpret_cast<const frag*>(ap + 32 * ks + 8);
        acc = Mfma32<T>::mma(tl, qh[ks], acc);
        acc = Mfma32<T>::mma(th, ql[ks], acc);
        acc = Mfma32<T>::mma(th, qh[ks], acc);
      }
      return acc;
    };
    for (int jb = 0; jb < nbw; ++jb) {
      const f32x16 acc = table_block(p.tab_w, 2 * kw - 1, jb);
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
    if (BHC == 0) {
      for (int jb = 0; jb < nbh; ++jb) {
        const f32x16 acc = table_block(p.tab_h, 2 * kh - 1, jb);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ky = qy + kh - 1 - (32 * jb + crow(r, hi));
          if (ky >= 0 && ky < kh) bh_all[ky * QW + wave * 32 + li] = acc[r];
        }
      }
    }
    __syncthreads();             // every wave is done with its stage before the tile buffers are filled
  }
  // chunked bias_h: rows [t0, t0 + BHC) of this wave's 32 queries from the 32 table rows j0 .. j0 + 31, j0 = (first query row of the wave)
  // + kh - 1 - (t0 + BHC - 1); a wave writes and reads only its own 32 columns of bh_all, LDS operations of a wave execute in order
  const int qy0 = __builtin_amdgcn_readfirstlane(qy);
  auto bias_h_chunk = [&](const int t0) {
    const int j0 = qy0 + kh - 1 - (t0 + BHC - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const T* ap = p.tab_h + (long)min(max(j0 + li, 0), 2 * kh - 2) * (2 * HD) + 16 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const frag th = *reinterpret_cast<const frag*>(ap + 32 * ks), tl = *reinterpret_cast<const frag*>(ap + 32 * ks + 8);
      acc = Mfma32<T>::mma(tl, qh[ks], acc);
      acc = Mfma32<T>::mma(th, ql[ks], acc);
      acc = Mfma32<T>::mma(th, qh[ks], acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ky = qy + kh - 1 - (j0 + crow(r, hi));
      if (ky >= t0 && ky < t0 + BHC && ky < kh) bh_all[(ky - t0) * QW + wave * 32 + li] = acc[r];
    }
  };

  // ---- tile streaming by LDS-DMA.  Block j of a tile image: plane = K_hi, K_lo, V_hi, V_lo; chunk c = 64 * (block in plane) + lane ->
  //      plane row c / (K|V)CH, column c % (K|V)CH; columns >= CPR are padding (never read for K; they feed the discarded d rows of
  //      O^T for V) and fetch the row's chunk 0.  Source = hl8 chunk 2 * column + (lo plane) of the key's k / v row. ----
  const int nt = (kh + R - 1) / R;
  const bool ragged = (R > 1) && (nt * nkt > p.N);           // odd number of key rows: the last tile has fewer
  auto dma_voff = [&](const int r, const int maxrow) -> unsigned int {
    const int j = wave + WAVES * r;
    const int jj = min(j, NBLK - 1);
    const bool isk = jj < 2 * KBLK;
    const int pl = isk ? jj / KBLK : (jj - 2 * KBLK) / VBLK;                  // 0 hi, 1 lo
    const int c = 64 * (isk ? jj - pl * KBLK : jj - 2 * KBLK - pl * VBLK) + lane;
    const int nch = isk ? KCH : VCH;
    const int row = min(min(c / nch, KT - 1), maxrow);
    int col = c % nch;
    if (col >= CPR) col = 0;          // padding chunk: this lane is masked out of the DMA (dskip)
    return (unsigned int)(((long)row * p.krs + (isk ? 0 : C2) + 16 * col + 8 * pl) * (long)sizeof(T));
  };
  unsigned int dvoff[NDMA];
  unsigned long long dmask[NDMA];     // lanes of DMA instruction r whose chunk is DATA (row-padding chunks are not fetched, not written)
#pragma unroll
  for (int r = 0; r < NDMA; ++r) {
    dvoff[r] = dma_voff(r, nkt - 1);
    const int jj = min(wave + WAVES * r, NBLK - 1);
    const bool isk = jj < 2 * KBLK;
    const int pl = isk ? jj / KBLK : (jj - 2 * KBLK) / VBLK;
    const int c = 64 * (isk ? jj - pl * KBLK : jj - 2 * KBLK - pl * VBLK) + lane;
    dmask[r] = __builtin_amdgcn_ballot_w64(c % (isk ? KCH : VCH) < CPR);
  }
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem_raw);
  const char* kbase0 = reinterpret_cast<const char*>(Kg);
  const long tile_bytes = (R > 1 ? (long)nkt * p.st : p.kts) * (long)sizeof(T);
  // the NDMA * WAVES slots of a round may exceed the NBLK blocks of a tile image: a surplus slot re-fetches the LAST block (same bytes to
  // the same place as the wave that owns it) instead of branching around the instruction
  auto dma_tile = [&](const int t, const int buf, const int r) {
    const int j = (WAVES * r + WAVES - 1 < NBLK) ? wave + WAVES * r : min(wave + WAVES * r, NBLK - 1);
    const bool last_ragged = ragged && (t == nt - 1);
    const unsigned int vo = last_ragged ? dma_voff(r, p.N - t * nkt - 1) : dvoff[r];
    const unsigned int ldst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(buf * BUF * (int)sizeof(T) + 1024 * j));
    vs_dma16(kbase0 + (long)t * tile_bytes, vo, ldst, dmask[r]);
  };

  // row padding of the V planes of both buffers (the prologue's staging area overlapped them): zeros, and 1.0 in column HD of V_hi
  {
    constexpr int PADC = VCH - CPR;
    for (int idx = tid; idx < 2 * 2 * KT * PADC; idx += NT) {
      const int pc = idx % PADC, row = (idx / PADC) % KT, pl = (idx / (PADC * KT)) & 1, buf = idx / (2 * PADC * KT);
      f16x8 z;
#pragma unroll
      for (int e = 0; e < 8; ++e) z[e] = (T)0.f;
      if (ONES && pl == 0 && pc == 0) z[0] = (T)1.f;
      *reinterpret_cast<f16x8*>(smem + buf * BUF + 2 * KPL + pl * VPL + row * VSTR + (CPR + pc) * 8) = z;
    }
  }

  f32x16 O[DBB];
  f32x4 OT[2];                  // TAIL: d rows 32 DBB + 4 (lane >> 4) + r of queries (lane & 15) [0] and 16 + (lane & 15) [1] of this wave
  float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
  for (int d = 0; d < DBB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) { OT[0][r] = 0.f; OT[1][r] = 0.f; }

#pragma unroll
  for (int r = 0; r < NDMA; ++r) dma_tile(0, 0, r);
  __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
  __syncthreads();

  const int l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int vlane = (4 * hi + (l16 >> 2)) * VSTR + 16 * g1 + 4 * (l16 & 3);   // this lane's V^T read offset inside a (step, d) block
  const int klane = li * KSTR + 8 * hi;

  for (int t = 0; t < nt; ++t) {
    const T* Kh = smem + (t & 1) * BUF;
    const T* Kl = Kh + KPL;
    const T* Vh = Kh + 2 * KPL;
    const T* Vl = Vh + VPL;
    // the DMA of the NEXT tile is issued unconditionally: behind the last tile it re-fetches that tile into the idle buffer (1 / nt more
    // L2 traffic) -- no `more` branches around the six DMA statements of the loop body
    const int tn = min(t + 1, nt - 1);
    if (BHC > 0 && t % (BHC > 0 ? BHC : 1) == 0) bias_h_chunk(t);
    float bh0 = bh_all[(BHC > 0 ? t % (BHC > 0 ? BHC : 1) : R * t) * QW + wave * 32 + li], bh1 = 0.f;
    if (R > 1) bh1 = (R * t + 1 < kh) ? bh_all[(R * t + 1) * QW + wave * 32 + li] : -INFINITY;

    // ---- S^T = K . Q'^T + bias_w (three products per k-step); the fragments of k-step ks + 1 are requested before the MFMAs of ks;
    //      the DMA instructions of tile t + 1 are spread over the k-steps ----
    f32x16 S[NB];
    {
      frag kfh[2][NB], kfl[2][NB];
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        kfh[0][blk] = *reinterpret_cast<const frag*>(Kh + klane + 32 * blk * KSTR);
        kfl[0][blk] = *reinterpret_cast<const frag*>(Kl + klane + 32 * blk * KSTR);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
          for (int blk = 0; blk < NB; ++blk) {
            kfh[(ks + 1) & 1][blk] = *reinterpret_cast<const frag*>(Kh + klane + 32 * blk * KSTR + 16 * (ks + 1));
            kfl[(ks + 1) & 1][blk] = *reinterpret_cast<const frag*>(Kl + klane + 32 * blk * KSTR + 16 * (ks + 1));
          }
        }
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
          S[blk] = Mfma32<T>::mma(kfl[ks & 1][blk], qh[ks], (ks == 0) ? bw[blk] : S[blk]);
          S[blk] = Mfma32<T>::mma(kfh[ks & 1][blk], ql[ks], S[blk]);
          S[blk] = Mfma32<T>::mma(kfh[ks & 1][blk], qh[ks], S[blk]);
        }
        if (ks < NDMA) dma_tile(tn, (t + 1) & 1, ks);
      }
#pragma unroll
      for (int r = KS; r < NDMA; ++r) dma_tile(tn, (t + 1) & 1, r);
    }
    // V^T fragments of the first PV product (step 0, hi plane): in flight while the softmax statistics are computed
    hfrag va[2][DBB], vb[2][DBB];               // [parity][d]: rows 0-3 / 8-11 of the 16-key step for this lane half
#pragma unroll
    for (int d = 0; d < DBB; ++d) {
      va[0][d] = Mfma32<T>::tr_read(Vh + vlane + 32 * d);
      vb[0][d] = Mfma32<T>::tr_read(Vh + vlane + 32 * d + 8 * VSTR);
    }

    // ---- row maximum; R > 1: add the key row's bias_h first ----
    if (R > 1) {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) S[blk][r] += (32 * blk + crow(r, hi) < kw) ? bh0 : bh1;
    }
    // four independent v_max3 chains: a single chain made hipcc put a wait state behind every link (inline-asm VALU feeding the next one).
    // hipcc places the MFMA -> VALU wait states only for reads it can see, not for the v_max3 asm statements -- which it is free to schedule
    // right behind the MFMA that writes their operand (tools/isa_hazard_lint.py found chains that started from -inf doing exactly that, 0
    // wait states after the MFMA; harmless in practice only because a stale maximum merely moves the softmax reference point).  So every
    // chain starts from `t0`, a C++ maximum over one element of EVERY score block: the compiler waits for each block's last MFMA before
    // it, and every asm link depends on it.
    float t0 = __builtin_fmaxf(S[0][0], S[0][1]);           // a real VALU instruction also when NB == 1 (a plain copy would be folded into the asm operand)
#pragma unroll
    for (int blk = 1; blk < NB; ++blk) t0 = __builtin_fmaxf(t0, S[blk][0]);
    float mx = t0, mxb = t0, mxc = t0, mxd = t0;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; r += 8) {
        mx = vs_max3(mx, S[blk][r], S[blk][r + 1]);
        mxb = vs_max3(mxb, S[blk][r + 2], S[blk][r + 3]);
        mxc = vs_max3(mxc, S[blk][r + 4], S[blk][r + 5]);
        mxd = vs_max3(mxd, S[blk][r + 6], S[blk][r + 7]);
      }
    mx = vs_max3(mx, mxb, mxc);
    mx = vs_max3(mx, mxd, mxd);
    mx = vs_xhalf_max(mx);
    mx += (R > 1 ? 0.f : bh0);
    // LAZY running maximum: the reference point only moves when some row's maximum grew by more than 2^VS_LAZY (log2 domain), so
    // probabilities may reach 2^VS_LAZY instead of 1 -- fp16 holds them at the same relative precision, the sums are fp32 -- and the
    // rescale of the 48 accumulator registers (taken on most tiles with the classic rule: one of 32 rows nearly always grows a little)
    // becomes the rare path it is meant to be
    if (__builtin_amdgcn_ballot_w64(mx > m_run + VS_LAZY) != 0ull) {
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);          // m_run = -inf -> 0 (every tile has a valid key)
      m_run = m_new;
      if (!ONES) l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DBB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
      if (TAIL) {               // the tail accumulators hold query (lane & 15) / 16 + (lane & 15): their factors live in those lanes
        const float ax = __shfl(alpha, lane & 15), ay = __shfl(alpha, 16 + (lane & 15));
#pragma unroll
        for (int r = 0; r < 4; ++r) { OT[0][r] *= ax; OT[1][r] *= ay; }
      }
    }
    const float off = (R > 1 ? 0.f : bh0) - m_run;

    // ---- P = exp2(S + off) as an fp16 PAIR;  O^T += V_hi^T . (P_hi + P_lo)^T + V_lo^T . P_hi^T.  Product u = (step, plane): the V^T reads of product
    //      u + 1 and (before a new step) the exps / packs of the next P fragment are issued ahead of the MFMAs of product u ----
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (TAIL) {
      // p[b][s]: the probabilities of 16-key step s of 32-key block (parity b) as packed fp16 pairs, hi and lo halves -- in the S^T layout
      // (lane = query li, the lane half's 8 keys of the step), which IS the B operand of the 32x32x16 products
      u32x4 p_h[2][2], p_l[2][2];
      auto make_p = [&](const int blk, const int st, u32x4& Hh, u32x4& Ll) {
        unsigned int hh[4], ll[4];
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(S[blk][8 * st + j] + off);
        vs_exp_settle(pv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          l_run += pv[2 * j] + pv[2 * j + 1];  // fp32 row sums of the unrounded probabilities (no spare O^T row for a ones column here)
          vs_split2(pv[2 * j], pv[2 * j + 1], hh[j], ll[j]);
        }
        vs_settle(hh, ll);
#pragma unroll
        for (int j = 0; j < 4; ++j) { Hh[j] = hh[j]; Ll[j] = ll[j]; }
      };
      // tail A operand (V^T rows 32 DBB .. + 15 over the 32 keys of a block): lane group g = lane >> 4 holds the k-group the swapped P
      // fragments give it -- keys {0-3, 8-11}, {16-19, 24-27}, {4-7, 12-15}, {20-23, 28-31} of the block for g = 0 .. 3 -- as two transposing reads
      const int tg = lane >> 4, tj = lane & 15;
      const int tlane = ((tg & 1) * 16 + (tg >> 1) * 4 + (tj >> 2)) * VSTR + 32 * DBB + 4 * (tj & 3);
      hfrag ta_h[2], ta_l[2];
      make_p(0, 0, p_h[0][0], p_l[0][0]);
#pragma unroll
      for (int u = 0; u < 4 * NB; ++u) {
        const int step = u >> 1, pl = u & 1, blk = step >> 1, st = step & 1, bp = blk & 1;
        frag vf[DBB];
#pragma unroll
        for (int d = 0; d < DBB; ++d)
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[d][j] = va[u & 1][d][j]; vf[d][4 + j] = vb[u & 1][d][j]; }
        if (u + 1 < 4 * NB) {
          const int nstep = (u + 1) >> 1, npl = (u + 1) & 1;
          const T* vbp = (npl ? Vl : Vh) + (16 * nstep) * VSTR + vlane;
#pragma unroll
          for (int d = 0; d < DBB; ++d) {
            va[(u + 1) & 1][d] = Mfma32<T>::tr_read(vbp + 32 * d);
            vb[(u + 1) & 1][d] = Mfma32<T>::tr_read(vbp + 32 * d + 8 * VSTR);
          }
          if (npl == 0) make_p(nstep >> 1, nstep & 1, p_h[(nstep >> 1) & 1][nstep & 1], p_l[(nstep >> 1) & 1][nstep & 1]);
        }
        if (st == 1 && pl == 0) {           // the tail's V^T fragments of this block, requested one product ahead of their use
          ta_h[0] = Mfma32<T>::tr_read(Vh + (32 * blk) * VSTR + tlane);
          ta_h[1] = Mfma32<T>::tr_read(Vh + (32 * blk + 8) * VSTR + tlane);
          ta_l[0] = Mfma32<T>::tr_read(Vl + (32 * blk) * VSTR + tlane);
          ta_l[1] = Mfma32<T>::tr_read(Vl + (32 * blk + 8) * VSTR + tlane);
        }
        // plane 0 (V_hi): both halves of P;  plane 1 (V_lo): P_hi only (lo x lo is below fp32 rounding)
        const frag pf = __builtin_bit_cast(frag, p_h[bp][st]);
#pragma unroll
        for (int d = 0; d < DBB; ++d) O[d] = Mfma32<T>::mma(vf[d], pf, O[d]);
        if (pl == 0) {
          const frag pfl = __builtin_bit_cast(frag, p_l[bp][st]);
#pragma unroll
          for (int d = 0; d < DBB; ++d) O[d] = Mfma32<T>::mma(vf[d], pfl, O[d]);
        }
        if (st == 1 && pl == 1) {
          // ---- the 16-row tail of this 32-key block.  v_permlane16_swap exchanges lanes 16-31 / 48-63 of its first operand with lanes
          //      0-15 / 32-47 of the second: (step 0, step 1) -> X = the four k-groups of queries 0-15, Y = those of queries 16-31 ----
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          u32x4 xh, yh, xl, yl;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const u32x2 sh = __builtin_amdgcn_permlane16_swap(p_h[bp][0][j], p_h[bp][1][j], false, false);
            const u32x2 sl = __builtin_amdgcn_permlane16_swap(p_l[bp][0][j], p_l[bp][1][j], false, false);
            xh[j] = sh[0]; yh[j] = sh[1]; xl[j] = sl[0]; yl[j] = sl[1];
          }
          frag ah, al;
#pragma unroll
          for (int j = 0; j < 4; ++j) { ah[j] = ta_h[0][j]; ah[4 + j] = ta_h[1][j]; al[j] = ta_l[0][j]; al[4 + j] = ta_l[1][j]; }
          const frag fxh = __builtin_bit_cast(frag, xh), fyh = __builtin_bit_cast(frag, yh);
          OT[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, fxh, OT[0], 0, 0, 0);
          OT[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, fyh, OT[1], 0, 0, 0);
          OT[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(frag, xl), OT[0], 0, 0, 0);
          OT[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(frag, yl), OT[1], 0, 0, 0);
          OT[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, fxh, OT[0], 0, 0, 0);
          OT[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, fyh, OT[1], 0, 0, 0);
        }
      }
    } else {
    frag pf, pfl;                // the probabilities of a 16-key step as an fp16 pair: hi, and lo = fp16(p - hi)
    {
      unsigned int hh[4], ll[4];
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(S[0][j] + off);
      vs_exp_settle(pv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (!ONES) l_run += pv[2 * j] + pv[2 * j + 1];
        vs_split2(pv[2 * j], pv[2 * j + 1], hh[j], ll[j]);
      }
      vs_settle(hh, ll);
      pf = __builtin_bit_cast(frag, (u32x4){hh[0], hh[1], hh[2], hh[3]});
      pfl = __builtin_bit_cast(frag, (u32x4){ll[0], ll[1], ll[2], ll[3]});
    }
#pragma unroll
    for (int u = 0; u < 4 * NB; ++u) {
      const int step = u >> 1, pl = u & 1;
      frag vf[DB];
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[d][j] = va[u & 1][d][j]; vf[d][4 + j] = vb[u & 1][d][j]; }
      frag pn = pf, pnl = pfl;
      if (u + 1 < 4 * NB) {
        const int nstep = (u + 1) >> 1, npl = (u + 1) & 1;
        const T* vbp = (npl ? Vl : Vh) + (16 * nstep) * VSTR + vlane;
#pragma unroll
        for (int d = 0; d < DB; ++d) {
          va[(u + 1) & 1][d] = Mfma32<T>::tr_read(vbp + 32 * d);
          vb[(u + 1) & 1][d] = Mfma32<T>::tr_read(vbp + 32 * d + 8 * VSTR);
        }
        if (npl == 0) {           // a new step follows: its probabilities
          const int nb_ = nstep >> 1, ns_ = nstep & 1;
          unsigned int hh[4], ll[4];
          float pv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(S[nb_][8 * ns_ + j] + off);
          vs_exp_settle(pv);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (!ONES) l_run += pv[2 * j] + pv[2 * j + 1];
            vs_split2(pv[2 * j], pv[2 * j + 1], hh[j], ll[j]);
          }
          vs_settle(hh, ll);
          pn = __builtin_bit_cast(frag, (u32x4){hh[0], hh[1], hh[2], hh[3]});
          pnl = __builtin_bit_cast(frag, (u32x4){ll[0], ll[1], ll[2], ll[3]});
        }
      }
      // plane 0 (V_hi): both halves of P;  plane 1 (V_lo): P_hi only (lo x lo is below fp32 rounding)
#pragma unroll
      for (int d = 0; d < DB; ++d) O[d] = Mfma32<T>::mma(vf[d], pf, O[d]);
      if (pl == 0) {
#pragma unroll
        for (int d = 0; d < DB; ++d) O[d] = Mfma32<T>::mma(vf[d], pfl, O[d]);
      }
      pf = pn;
      pfl = pnl;
      (void)step;
    }

    }

    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMA writes of tile t + 1 have landed
    __syncthreads();                          // ... and everybody's; all reads of tile t are done
  }

  // ---- epilogue: O^T / l as HL8: a lane owns 4 consecutive d = one half of a group of 8 ----
  {
    // ONES: d row HD = row HD % 32 of the last block = register 8 (crow(8, 0) = 16 = 80 - 64) of the lower lane half
    static_assert(!ONES || (HD % 32 == 16), "the ones column is read back from O^T row 16 of the last block");
    const float l_tot = ONES ? __shfl(O[DBB - 1][8], li) : l_run + __shfl_xor(l_run, 32);
    const float inv = 1.f / l_tot;
    if (qi < p.N) {
      T* orow = Og + qm * p.o_st;
#pragma unroll
      for (int d = 0; d < DBB; ++d)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int d0 = 32 * d + 8 * rr + 4 * hi;
          if (d0 < HD) {
            f16x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              T hh, ll;
              hl_split(O[d][4 * rr + e] * inv, hh, ll);
              oh[e] = hh;
              ol[e] = ll;
            }
            T* dst = orow + 16 * (d0 >> 3) + 4 * hi;
            *reinterpret_cast<f16x4*>(dst) = oh;
            *reinterpret_cast<f16x4*>(dst + 8) = ol;
          }
        }
    }
    if (TAIL) {
      // tail rows: lane holds d = 32 DBB + 4 (lane >> 4) + e of query (lane & 15) [OT[0]] and 16 + (lane & 15) [OT[1]] of this wave
      const int d0 = 32 * DBB + 4 * (lane >> 4);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int ql = 16 * h2 + (lane & 15);                 // query inside the wave's 32
        const float iv = __shfl(inv, ql);
        const int qi2 = qt * QW + wave * 32 + ql;
        const int qc2 = min(qi2, p.N - 1);
        const int qy2 = qc2 / kw, qx2 = qc2 - qy2 * kw;
        const long qm2 = p.tr ? (long)qx2 * p.kwm + qy2 : (long)qc2;
        if (qi2 < p.N) {
          f16x4 oh, ol;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            T hh, ll;
            hl_split(OT[h2][e] * iv, hh, ll);
            oh[e] = hh;
            ol[e] = ll;
          }
          T* dst = Og + qm2 * p.o_st + 16 * (d0 >> 3) + (d0 & 4);
          *reinterpret_cast<f16x4*>(dst) = oh;
          *reinterpret_cast<f16x4*>(dst + 8) = ol;
        }
      }
    }
  }
}

template <int HD, int NB, int WAVES, int R, int KW, int BHC = 0>
static int launch_vs(VSParams& p, hipStream_t st) {
  constexpr int KT = 32 * NB, DB = (HD + 31) / 32;
  constexpr int KSTR = HD + 8, VSTR = (DB * 32 == 96 || DB * 32 == 32) ? DB * 32 : DB * 32 + 32;
  constexpr int KBLK = (KT * (KSTR / 8) + 63) / 64, VBLK = (KT * (VSTR / 8) + 63) / 64;
  size_t lds = (size_t)2 * (2 * KBLK + 2 * VBLK) * 1024;
  const size_t stage = (size_t)WAVES * 32 * 32 * sizeof(float);
  if (lds < stage) lds = stage;
  lds += (size_t)(BHC > 0 ? BHC : p.kh) * WAVES * 32 * sizeof(float);
  if (lds > 160 * 1024) return set_err(HIPIE_EINVAL, "vit_attn(split): %zu bytes of LDS needed (grid %dx%d) > 160 KiB", lds, p.kh, p.kw);
  p.nqt = (p.N + WAVES * 32 - 1) / (WAVES * 32);
  p.swz = ((p.B * p.H) % 8 == 0) ? 1 : 0;
  const unsigned grid = (unsigned)(p.nqt * p.B * p.H);
  auto kern = vit_attn_split_kernel<HD, NB, WAVES, R, KW, BHC>;
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
  return check_launch("vit_attn(split)");
}

template <int HD>
static int dispatch_vs(VSParams& p, hipStream_t st) {
  if (p.kw == 14 && p.kh <= 96) return launch_vs<HD, 1, 7, 2, 14>(p, st);     // the 14x14 windows: 196 queries = 7 waves
  if (p.kw <= 32) return launch_vs<HD, 1, 4, 1, 0>(p, st);
  if (p.kw <= 64 && p.kh <= 64) return launch_vs<HD, 2, 8, 1, 0>(p, st);
  if (p.kw <= 64) return launch_vs<HD, 2, 4, 1, 0>(p, st);
  if (p.kw <= 96) return launch_vs<HD, 3, 8, 1, 0, 16>(p, st);                // 84x84: the 1344-pixel configuration
  return set_err(HIPIE_EINVAL, "vit_attn(split): token grids wider than 96 in BOTH directions are not supported (got %dx%d)", p.kh, p.kw);
}

}  // namespace hipie

extern "C" int hipie_vit_attn_split(const void* qkv, const void* tab_h, const void* tab_w, void* out, int B, int gh, int gw,
                                    int heads, int hd, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(qkv && out && tab_h && tab_w, "vit_attn_split: null pointer");
  HIPIE_REQUIRE((((uintptr_t)qkv | (uintptr_t)out | (uintptr_t)tab_h | (uintptr_t)tab_w) & 15) == 0, "vit_attn_split: pointers must be 16-byte aligned");
  HIPIE_REQUIRE(B > 0 && gh > 0 && gw > 0 && heads > 0, "vit_attn_split: bad shape B=%d grid %dx%d heads=%d", B, gh, gw, heads);
  HIPIE_REQUIRE(gh <= 160, "vit_attn_split: more than 160 key rows (%d) are not supported", gh);
  const long N = (long)gh * gw, C = (long)heads * hd;
  HIPIE_REQUIRE(N * 6 * C * 2 < (1L << 31), "vit_attn_split: an image's qkv block must stay below 2 GiB");
  VSParams p{};
  p.qkv = (const f16_t*)qkv; p.out = (f16_t*)out; p.tab_h = (const f16_t*)tab_h; p.tab_w = (const f16_t*)tab_w;
  p.B = B; p.H = heads; p.N = (int)N; p.kh = gh; p.kw = gw;
  p.sb = N * 6 * C; p.st = 6 * C; p.o_sb = N * 2 * C; p.o_st = 2 * C;
  p.tr = 0; p.kwm = gw; p.krs = p.st; p.kts = (long)gw * p.st;
  {
    // grids wider than 96 tokens (eval yamls: MAX_SIZE_TEST 2048 -> up to 64 x 128): walk the grid column by column
    static int force_tr = -1;
    if (force_tr < 0) { const char* e = study_env("HIPIE_VA_TRANSPOSE"); force_tr = e ? atoi(e) : 0; }      // 1: always (tests)
    if ((gw > 96 && gh <= 96) || (force_tr == 1 && gw != 14)) {
      p.tr = 1; p.kh = gw; p.kw = gh; p.tab_h = (const f16_t*)tab_w; p.tab_w = (const f16_t*)tab_h;
      p.krs = (long)gw * p.st; p.kts = p.st;
    }
  }
  hipStream_t st = (hipStream_t)stream;
  if (hd == 80) return dispatch_vs<80>(p, st);
  if (hd == 64) return dispatch_vs<64>(p, st);
  return set_err(HIPIE_EINVAL, "vit_attn_split: head_dim 64 / 80 only (got %d)", hd);
}

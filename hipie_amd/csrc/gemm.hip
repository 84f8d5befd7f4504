// gemm.hip -- the linears of the path (ViT qkv / proj / fc1 / fc2, BERT, encoder / decoder FFNs and projections) as one
// hand-written MFMA GEMM for gfx950:      out = epilogue( alpha * A (M x K) . W^T (N x K) + bias )
//
// Both operands are K-contiguous (torch.nn.Linear keeps W as (N, K)), so both MFMA fragments are 16-byte row pieces: no
// transposes anywhere.  Two operand formats:
//   HIPIE_F16   one fp16 per element                                                       1 MFMA per tile and k-step
//   HIPIE_HL8   SPLIT fp16: every group of 8 k-elements is stored as 8 fp16 "hi" then 8 fp16 "lo" with hi + lo == x to 2^-22
//               (32 bytes per group = the bytes of fp32).  The product is formed as  W_lo.X_hi + W_hi.X_lo + W_hi.X_hi  with
//               fp32 accumulation: every fp16 x fp16 product is exact in fp32, the dropped lo x lo term is 2^-22 relative, so
//               the result is fp32-class (the reference runs these linears in fp32, hipie/backbone/vit.py:67-83,212-230;
//               deformable_transformer_dino.py:378-394) at 3 MFMAs per k-step on the 16-bit matrix pipe -- gfx950's fp32 MFMA
//               runs at 1/16 of the fp16 rate.  A row's 32-element k slice is one 128-byte line.
//
// Design (MI355X_MICROARCH.md, cdna_hip_programming.md section 5):
//   * workgroup tile 256 (M) x BN (N), BN = 320 | 256, 8 waves as 4 (M) x 2 (N): a wave owns 64 tokens x BN/2 features =
//     2 x (BN/64) MFMA tiles of 32x32 (160 / 128 accumulator registers).  BN = 320 tiles N = 1280 / 3840 / 5120 of ViT-H
//     exactly and makes 512 / 1536 / 2048 workgroups at 32768 tokens = whole waves of the 256 CUs;
//   * the MFMA computes out^T = W . X^T (features are the 32 MFMA rows, tokens the 32 columns), so a lane owns ONE token and
//     4 consecutive features per accumulator quad: bias / activation / residual / split are per-lane vector work and the stores
//     are 16 bytes;
//   * k tile = 128 bytes per row in both formats (64 fp16 elements, or 32 split elements); operand tiles go L2 -> LDS by
//     LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write), 2 LDS stages of (256 + BN) x 128 B, one barrier
//     per stage, the DMA instructions of stage t+1 spread between the MFMAs of stage t;
//   * LDS rows are 128 B, which would put a ds_read_b128 lane group on two 16-byte slots (8-way conflict); the 16-byte chunk c
//     of row r is therefore stored at chunk position c ^ ((r >> 1) & 7).  The DMA writes LDS lane-linearly, so the swizzle is
//     applied to the per-lane global SOURCE address; the reader applies the same XOR.  Conflict-free for every b128 lane group;
//   * block -> tile map: the blocks of one XCD (blockIdx % 8) walk a contiguous range of tiles -- N index fastest for up to 4 column
//     tiles, else in groups of 8 row panels with the row panel fastest (GemmParams::group_m) -- so the 32 workgroups resident on an
//     XCD share 8 A row panels and 4 W panels through that XCD's L2.
// Epilogue (runtime switches, once per tile): * alpha, + bias, exact-erf GELU | ReLU, + fp32 residual, then the output as
// fp32, fp16 or HL8 (optionally scaled) -- the HL8 form is directly the A operand of the next GEMM.
#include <stdlib.h>

#include "common.h"
#include "mfma.h"

namespace hipie {

struct GemmParams {
  const char* A; const char* W; const float* bias; const float* resid; char* out;
  const int32_t* out_row;     // optional: row m of the product goes to output / residual row out_row[m] (< 0: dropped) -- window un-partition
  const int32_t* a_row;       // optional: row m of the product READS operand row a_row[m] (gather; hipie_gemm_gather) -- the real tokens of a padded window layout
  long lda_b, ldw_b;          // row strides of A / W in BYTES
  long ldr, ldo;              // row strides of resid (fp32 elements) / out (elements of the output format: fp32 | fp16; HL8: fp16 elements)
  int M, N, K;
  int nkt;                    // 128-byte k tiles
  int tiles_m, tiles_n;
  int group_m;                // block -> tile order: 0 / 1 = N fastest; g > 1 = groups of g M-panels, M fastest inside a group
  int out_fmt, act;
  float alpha, oscale;
  // batched form (hipie_gemm_batched): blockIdx.y = outer * nbi + inner; operand / output base offsets in BYTES per outer / inner index
  int nbi;
  long a_bo, a_bi, w_bo, w_bi, o_bo, o_bi;
  // 3 x 3 convolution as an implicit GEMM on a zero-PADDED pixel grid (hipie_conv3x3_split): K = 9 taps x C channels, the A rows of k tile kt
  // come from the pixel row shifted by tap (dy, dx): byte offset ((dy - 1) * conv_wp + (dx - 1)) * lda_b + (kt % conv_kpt) * 128 -- the same
  // for every row, so only the scalar source base of the A tile changes.  conv_kpt = k tiles per tap (0: plain GEMM).
  int conv_kpt, conv_wp;
  // row softmax in the epilogue (hipie_gemm_batched_softmax: the logits GEMM of the image -> text fusion attention, N = text length <= 256
  // = ONE column tile): out = HL8 of softmax_j( clamp(acc) masked ) per row; sm_mask (n_outer, sm_L) uint8 or null, column j valid iff
  // j < sm_L && mask[outer][j].  0 = off.
  int softmax, sm_L;
  float sm_clamp;
  const unsigned char* sm_mask;
  // residual add + LayerNorm in the epilogue (hipie_gemm_ln: N = 256 = ONE column tile, so a workgroup holds whole rows):
  // y = LN(alpha * acc + bias + resid) * ln_g + ln_b; out = y as fp32, out2 (optional) = y as HL8 rows (row stride ldo2 fp16 elements)
  const float* sm_bias = nullptr;            // softmax epilogue (VAR 8): per-column logit bias, (n_outer * n_inner, N) fp32, added before the clamp
  long r_bo = 0, r_bi = 0;                   // batched form: offsets of `resid` per outer / inner index in fp32 ELEMENTS (bias is shared)
  const float* ln_g = nullptr; const float* ln_b = nullptr; float ln_eps = 0.f;
  char* out2 = nullptr; long ldo2 = 0;
  int variant;                // timing experiments (HIPIE_GEMM_VARIANTS builds only)
  int prio_mode;              // gemm2: 0 none, 1 blocks 256..511 at low priority (phase offset), 2 by dispatch-round parity
};

// LDS-DMA, 16 bytes per lane: LDS[m0 + 16 * lane] <- *(sbase + voff).  Inline asm (see vit_attn.hip: the builtin makes hipcc
// drain vmcnt(0) before every later ds_read); completion is counted by hand -- vmcnt(0) before the stage barrier.
__device__ __forceinline__ void gm_dma16(const char* sbase, unsigned int voff, unsigned int lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}

// exact-erf GELU (nn.GELU() default, timm Mlp: hipie/backbone/vit.py:193-197), branch-free:  gelu(x) = x * Phi(x),
//   Phi(x) = 1 - h(|x|) for x >= 0,  h(|x|) for x < 0,   h(a) = P(t) * exp(-a^2 / 2),  t = 1 / (1 + p a)
// -- the erfc form of Abramowitz & Stegun 7.1.26 with one more term, the seven constants re-fitted for this code (minimax over [0, 6 sqrt 2]:
// |erf error| 9.2e-9 against 1.4e-7 for the handbook's five-term constants; tools/fit_gelu_erf.py).  Evaluated in fp32: max |error| 3.8e-7
// over |x| <= 12, relative error <= 2.4e-7 |x| -- the figures of 0.5 x (1 + erff(x / sqrt 2)) with a correctly rounded erff, and better for
// x < -4, where 1 + erf cancels.  16 VALU instructions, two of them transcendental, no branch: ocml's erff is two polynomial branches
// (both executed by a wavefront) around an exp -- the fc1 epilogue (160 values per lane and tile) was 0.145 ms per launch behind fc2's.
__device__ __forceinline__ float gm_gelu(float x) {
#ifdef HIPIE_GELU_ERFF
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
#endif
  const float a = __builtin_fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.27601078152656555f, a, 1.f));
  float q = -0.113462433218956f;
  q = __builtin_fmaf(q, t, 0.4407985508441925f);
  q = __builtin_fmaf(q, t, -0.31384575366973877f);
  q = __builtin_fmaf(q, t, 0.32216209173202515f);
  q = __builtin_fmaf(q, t, 0.046716462820768356f);
  q = __builtin_fmaf(q, t, 0.11763110756874084f);
  const float e = __builtin_amdgcn_exp2f(-0.7213475108146667f * (a * a));
  const float h = (q * t) * e;
  return x * (x >= 0.f ? 1.f - h : h);
}

__device__ __forceinline__ unsigned int gm_pack2(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v;
  v[0] = (f16_t)a;
  v[1] = (f16_t)b;
  return __builtin_bit_cast(unsigned int, v);
}

__device__ __forceinline__ unsigned int gm_pack2h(f16_t a, f16_t b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 v;
  v[0] = a;
  v[1] = b;
  return __builtin_bit_cast(unsigned int, v);
}


// ---- tile epilogue of one 32-feature x 32-token MFMA block (both kernels): lane = token, 16 accumulator values = 4 quads of 4
// consecutive features.  The activation / residual / scale switches are taken once per block (not per value: the per-value form cost
// ~1400 scalar branches per wave), and the HL8 split works on PAIRS: v_cvt_pk_f16_f32 for the hi and the lo halves.
typedef _Float16 gm_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gm_split2(float x0, float x1, unsigned int& H, unsigned int& L) {
  x0 = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
  x1 = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPIE_NO_FMA_MIX)
  // hi = fp16(x) for the pair in one v_cvt_pk_f16_f32; lo = fp16(x - hi) in ONE v_fma_mix{lo,hi}_f16 each (the fp16 hi is an fp16 source
  // operand of the fma, x - hi is exact, one rounding): the same bits as `(f16)(x - (float)(f16)x)`, which costs a convert, a convert back
  // and a subtract per value.  The asm operands are the register values themselves: nothing for the compiler to re-fold (see hl_split).
  unsigned int h, l;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(x0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
  H = h;
  L = l;
#else
  gm_h2 h, l;
  h[0] = (f16_t)x0; h[1] = (f16_t)x1;
  l[0] = (f16_t)(x0 - (float)h[0]); l[1] = (f16_t)(x1 - (float)h[1]);
  H = __builtin_bit_cast(unsigned int, h);
  L = __builtin_bit_cast(unsigned int, l);
#endif
}

// quads [G0, G0 + NG) of the block; rq = the residual quads (zeros when there is no residual); sb = this block's 32 bias values in LDS
// x = the raw accumulator values of quads [G0, G0 + NG) of the block (G0 may be a runtime value: 0 | 2 for half blocks)
template <int NG>
__device__ __forceinline__ void gm_epi_vals(const float (&x)[NG][4], const int G0, const float4* rq, const float* sb, const long m, const bool mok,
                                            const int nb, const int hi, const GemmParams& p, const bool has_res) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const float alpha = p.alpha, osc = p.oscale;
  const int act = p.act, ofmt = p.out_fmt;
  float v[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const float4 b4 = *reinterpret_cast<const float4*>(sb + 8 * (G0 + g) + 4 * hi);
    v[g][0] = x[g][0] * alpha + b4.x;
    v[g][1] = x[g][1] * alpha + b4.y;
    v[g][2] = x[g][2] * alpha + b4.z;
    v[g][3] = x[g][3] * alpha + b4.w;
  }
  if (act == 1) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = gm_gelu(v[g][e]);
  } else if (act == 2) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = fmaxf(v[g][e], 0.f);
  } else if (act == 3) {                       // QuickGELU of the OpenAI CLIP weights: y * sigmoid(1.702 y) (open_clip's QuickGELU; hipie/open_vocab/clip.py towers)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] = v[g][e] / (1.f + expf(-1.702f * v[g][e]));
  }
  if (has_res) {
#pragma unroll
    for (int g = 0; g < NG; ++g) { v[g][0] += rq[g].x; v[g][1] += rq[g].y; v[g][2] += rq[g].z; v[g][3] += rq[g].w; }
  }
  if (osc != 1.f) {
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g][e] *= osc;
  }
  if (ofmt == HIPIE_F16) {
    // quads g and g + 1 of the two lane halves are exchanged so that the lower half stores features 8g .. 8g+7 and the upper
    // half 8(g+1) .. 8(g+1)+7 as ONE 16-byte piece each
#pragma unroll
    for (int g = 0; g < NG; g += 2) {
      const u32x2 s0 = __builtin_amdgcn_permlane32_swap(gm_pack2(v[g][0], v[g][1]), gm_pack2(v[g + 1][0], v[g + 1][1]), false, false);
      const u32x2 s1 = __builtin_amdgcn_permlane32_swap(gm_pack2(v[g][2], v[g][3]), gm_pack2(v[g + 1][2], v[g + 1][3]), false, false);
      const int n = nb + 8 * (G0 + g + hi);
      if (mok && n < p.N) *reinterpret_cast<u32x4*>(reinterpret_cast<f16_t*>(p.out) + m * p.ldo + n) = (u32x4){s0[0], s1[0], s0[1], s1[1]};
    }
  } else {
    // fp32 and HL8: the block's 32 features are a 128-byte span of the output row, of which this lane holds the four 16-byte
    // pieces at byte 32 g + 16 hi (fp32: features 8g+4hi ..+3; HL8: the lower lane half ends up with the 8 hi values of group g,
    // the upper half with its 8 lo values).  One store per piece: 32 rows x 32 bytes per instruction.
    u32x4 piece[NG];
    if (ofmt == HIPIE_F32) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
        piece[g] = (u32x4){__builtin_bit_cast(unsigned int, v[g][0]), __builtin_bit_cast(unsigned int, v[g][1]),
                           __builtin_bit_cast(unsigned int, v[g][2]), __builtin_bit_cast(unsigned int, v[g][3])};
    } else {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        unsigned int H0, L0, H1, L1;
        gm_split2(v[g][0], v[g][1], H0, L0);
        gm_split2(v[g][2], v[g][3], H1, L1);
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPIE_NO_FMA_MIX)
        // the split above is inline asm, which hipcc's hazard recogniser does not look into: v_permlane32_swap must not read a VGPR in the
        // two wait states behind the VALU instruction that wrote it (see vs_settle in vit_attn_split.hip)
        asm volatile("s_nop 1" : "+v"(H0), "+v"(L0), "+v"(H1), "+v"(L1));
#endif
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(H0, L0, false, false);    // lower: (H0 own, H0 of upper); upper: (L0 of lower, L0 own)
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(H1, L1, false, false);
        piece[g] = (u32x4){s0[0], s1[0], s0[1], s1[1]};
      }
    }
    const long rowb = (m * p.ldo) * (ofmt == HIPIE_F32 ? 4 : 2) + (long)nb * 4 + 16 * hi;
#pragma unroll
    for (int g = 0; g < NG; ++g)
      if (mok && nb + 8 * (G0 + g) < p.N) *reinterpret_cast<u32x4*>(p.out + rowb + 32 * (G0 + g)) = piece[g];
  }
}

template <int G0, int NG>
__device__ __forceinline__ void gm_epi_quads(const f32x16& a, const float4* rq, const float* sb, const long m, const bool mok, const int nb,
                                             const int hi, const GemmParams& p, const bool has_res) {
  float x[NG][4];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int e = 0; e < 4; ++e) x[g][e] = a[4 * (G0 + g) + e];
  gm_epi_vals<NG>(x, G0, rq, sb, m, mok, nb, hi, p, has_res);
}

template <int BN, bool SPLIT, int VAR>
__global__ __launch_bounds__(512, 2) void gemm_kernel(const GemmParams pin) {
  GemmParams p = pin;
  int vtile = -1;
  if (VAR == 8) {
    // batched with the INNER index fastest (grid: tiles * n_inner x n_outer): the n_inner problems of one row tile share their A operand (the
    // heads of the folded fusion attention all read the visual stream), so they run back to back inside ONE XCD's contiguous range of ids
    // and the 256 KB A tile is fetched once into that XCD's L2 instead of once per head (PMC: 1.57 GB fetched per launch for 0.18 GB of A)
    const int nblk = p.tiles_m * p.tiles_n * p.nbi;
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    const int bo = blockIdx.y, bi = v % p.nbi;
    vtile = v / p.nbi;
    p.A += bo * p.a_bo + bi * p.a_bi;
    p.W += bo * p.w_bo + bi * p.w_bi;
    p.out += bo * p.o_bo + bi * p.o_bi;
    if (p.sm_mask != nullptr) p.sm_mask += (long)bo * p.sm_L;
    if (p.sm_bias != nullptr) p.sm_bias += ((long)bo * p.nbi + bi) * p.N;
  } else if (gridDim.y > 1) {                   // batched: one (outer, inner) problem per blockIdx.y
    const int bo = blockIdx.y / p.nbi, bi = blockIdx.y - bo * p.nbi;
    p.A += bo * p.a_bo + bi * p.a_bi;
    p.W += bo * p.w_bo + bi * p.w_bi;
    p.out += bo * p.o_bo + bi * p.o_bi;
    if (p.sm_mask != nullptr) p.sm_mask += (long)bo * p.sm_L;
    if (p.sm_bias != nullptr) p.sm_bias += (long)blockIdx.y * p.N;
    if (p.resid != nullptr) p.resid += bo * p.r_bo + bi * p.r_bi;
  }
  constexpr int BM = 256;
  constexpr int ROWS = BM + BN;                // rows of one LDS stage: the A tile then the W tile
  constexpr int STAGE = ROWS * 128;            // bytes
  constexpr int NI = ROWS / 64;                // DMA instructions per wave and stage (one covers 8 rows x 128 B)
  constexpr int NJ = BN / 64;                  // 32-feature blocks per wave
  constexpr int KS = SPLIT ? 2 : 4;            // k16 steps per stage
  constexpr int SUB = KS * NJ;                 // (k-step, feature block) sub-steps per stage
  constexpr bool AF32 = VAR == 2 || VAR == 6;  // A rows are plain fp32, split in registers
  constexpr bool LNE = VAR == 4 || VAR == 6;   // residual add + LayerNorm epilogue (separate instances: the plain kernels' code is unchanged)
  typedef Mfma32<f16_t>::frag frag;

  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int li = lane & 31, hi = lane >> 5;

  // ---- block -> tile (bijective XCD-aware order: XCD x owns a contiguous range of tile ids) ----
  int tm, tn;
  {
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int v = (VAR == 8) ? vtile : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    if (p.group_m > 1) {
      const int gsz = p.group_m * p.tiles_n;
      const int g = v / gsz, w = v - g * gsz;
      const int rows = min(p.group_m, p.tiles_m - g * p.group_m);      // the last group may be shorter
      tn = w / rows;
      tm = g * p.group_m + (w - tn * rows);
    } else {
      tm = v / p.tiles_n;
      tn = v - tm * p.tiles_n;
    }
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- DMA plan: instruction i of this wave fills rows 8 * (8 i + wave) .. + 7 of the stage image; lane -> (row, chunk position) ----
  unsigned int dvoff[NI];
  {
    const int rl = lane >> 3, cp = lane & 7;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = 8 * (8 * i + wave) + rl;              // stage row
      const int c = cp ^ ((r >> 1) & 7);                  // logical chunk stored at this position
      if (r < BM) {
        const int mr = min(r, p.M - 1 - m0);
        // gather: the offset is taken from the START of A (all of A within 4 GB: checked on the host)
        dvoff[i] = p.a_row != nullptr ? (unsigned int)((long)p.a_row[m0 + mr] * p.lda_b + 16 * c) : (unsigned int)((long)mr * p.lda_b + 16 * c);
      } else dvoff[i] = (unsigned int)((long)min(r - BM, p.N - 1 - n0) * p.ldw_b + 16 * c);
    }
  }
  const char* abase = p.a_row != nullptr ? p.A : p.A + (long)m0 * p.lda_b;
  const char* wbase = p.W + (long)n0 * p.ldw_b;
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);

  auto dma = [&](const int i, const int kt, const int stage) {
    const bool isa = (8 * (8 * i + wave)) < BM;           // wave-uniform: an instruction is all-A or all-W (BM % 64 == 0)
    long ko = (long)kt * 128;
    if (isa && p.conv_kpt) {                              // implicit 3 x 3 convolution: tap of this k tile -> row shift on the padded grid
      const int tap = kt / p.conv_kpt, r = kt - tap * p.conv_kpt, dy = tap / 3;
      ko = ((long)(dy - 1) * p.conv_wp + (tap - 3 * dy - 1)) * p.lda_b + (long)r * 128;
    }
    const char* sb = (isa ? abase : wbase) + ko;
    gm_dma16(sb, dvoff[i], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(stage * STAGE + 1024 * (8 * i + wave))));
  };

  // ---- fragment addresses: row = tile base (multiple of 32) + li, so the swizzle term is ((li >> 1) & 7) for every tile ----
  const int swz = (li >> 1) & 7;
  const char* xrow = smem + (wm * 64 + li) * 128;                    // + t * 32 * 128
  const char* wrow = smem + (BM + wn * (BN / 2) + li) * 128;         // + j * 32 * 128
  // logical chunk of (k-step ks, lane half, lo): plain 2 ks + hi; split 2 (2 ks + hi) + lo
  auto choff = [&](const int ks, const int lo) -> int { return 16 * ((SPLIT ? (2 * (2 * ks + hi) + lo) : (2 * ks + hi)) ^ swz); };

  f32x16 acc[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  // ---- prologue: stage 0 ----
#ifdef HIPIE_GEMM_VARIANTS
  const int nkt = (VAR == 3) ? 0 : p.nkt;     // timing experiment: the epilogue alone
  if (VAR != 3)
#else
  const int nkt = p.nkt;
#endif
  {
#pragma unroll
    for (int i = 0; i < NI; ++i) dma(i, 0, 0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0)
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    const bool more = kt + 1 < nkt;
    const char* xs = xrow + st * STAGE;
    const char* ws = wrow + st * STAGE;
    // software pipeline inside the stage: the fragments of sub-step s + 1 are requested before the MFMAs of sub-step s
    frag xa[2][2][2];            // [k-step parity][hi | lo][token tile]
    frag wa[2][2];               // [sub-step parity][hi | lo]
    auto load_x = [&](const int ks) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xa[ks & 1][0][t] = *reinterpret_cast<const frag*>(xs + t * 4096 + choff(ks, 0));
        if (SPLIT) xa[ks & 1][1][t] = *reinterpret_cast<const frag*>(xs + t * 4096 + choff(ks, 1));
      }
    };
    auto load_w = [&](const int s) {
      const int ks = s / NJ, j = s % NJ;
      wa[s & 1][0] = *reinterpret_cast<const frag*>(ws + j * 4096 + choff(ks, 0));
      if (SPLIT) wa[s & 1][1] = *reinterpret_cast<const frag*>(ws + j * 4096 + choff(ks, 1));
    };
    load_x(0);
    load_w(0);
    // DMA plan: the instructions of stage t + 1 are issued in the FIRST sub-steps of stage t, PER sub-step as many as it takes to be
    // done by sub-step DMA_BY: a fill needs 1-2 us from issue to landing and the stage ends with vmcnt(0), so a late issue stalls
    // every wave at the barrier (measured: one DMA per sub-step over the whole stage cost ~15 % of the split kernel's rate)
    constexpr int DMA_BY = SPLIT ? 4 : 6;                        // sub-steps that carry DMA instructions
    constexpr int PER = (NI + DMA_BY - 1) / DMA_BY;
    frag cx[2][2];               // AF32 (fp32 A rows): the k-step's A fragments split in registers, [hi | lo][token tile]
#pragma unroll
    for (int s = 0; s < SUB; ++s) {
      const int ks = s / NJ, j = s % NJ;
      if (s + 1 < SUB) {
        if ((s + 1) % NJ == 0) load_x(ks + 1);
        load_w(s + 1);
      }
      if (AF32 && j == 0) {
        // the two 16-byte chunks of a group hold x0..x3 / x4..x7 as fp32 (the same 32 bytes an HL8 group takes): split them here
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 a = __builtin_bit_cast(f32x4, xa[ks & 1][0][t]), b = __builtin_bit_cast(f32x4, xa[ks & 1][1][t]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f16_t hh, ll;
            hl_split(a[e], hh, ll);
            cx[0][t][e] = hh; cx[1][t][e] = ll;
            hl_split(b[e], hh, ll);
            cx[0][t][4 + e] = hh; cx[1][t][4 + e] = ll;
          }
        }
      }
      const frag wh = wa[s & 1][0];
      const frag xh0 = AF32 ? cx[0][0] : xa[ks & 1][0][0], xh1 = AF32 ? cx[0][1] : xa[ks & 1][0][1];
      if (SPLIT) {
        const frag wl = wa[s & 1][1];
        const frag xl0 = AF32 ? cx[1][0] : xa[ks & 1][1][0], xl1 = AF32 ? cx[1][1] : xa[ks & 1][1][1];
        acc[j][0] = Mfma32<f16_t>::mma(wl, xh0, acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wl, xh1, acc[j][1]);
        acc[j][0] = Mfma32<f16_t>::mma(wh, xl0, acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wh, xl1, acc[j][1]);
      }
      acc[j][0] = Mfma32<f16_t>::mma(wh, xh0, acc[j][0]);
      acc[j][1] = Mfma32<f16_t>::mma(wh, xh1, acc[j][1]);
      if (more) {
#pragma unroll
        for (int i = s * PER; i < (s + 1) * PER && i < NI; ++i) dma(i, kt + 1, st ^ 1);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMA writes of stage t+1 have landed
    __syncthreads();                          // ... and everybody's; all reads of stage t are done
  }

  // ---- epilogue: lane = token (column of the MFMA tile), registers = features ----
  bool has_res = p.resid != nullptr;
#ifdef HIPIE_GEMM_VARIANTS
  if (VAR == 1 && p.alpha != 12345.f) return;        // timing experiment: no epilogue at all (tools/bench_gemm2.py variants)
#endif
  const int act = p.act, ofmt = p.out_fmt;
  const float alpha = p.alpha, osc = p.oscale;
  // the tile's bias values go through LDS once (the stage buffers are free after the last barrier): the per-quad bias reads are then
  // LDS reads the compiler can schedule freely between the global stores (a global read behind every store serialised the epilogue)
  float* sbias = reinterpret_cast<float*>(smem);
  if (tid < BN) sbias[tid] = (p.bias != nullptr && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
  if (SPLIT && LNE && BN == 256) {
    // ---- residual add + LayerNorm over the tile's 256 columns (the whole row: one column tile), the statistics of hipie_add_layernorm_dec
    //      (two passes, fp32): a lane owns 64 of its token's 256 columns per token tile; lane-local sums, one exchange with the other lane
    //      half (xor 32), one with the partner wave (wn ^ 1) through LDS.  Every residual value of a row is read before any store of that
    //      row (rows belong to ONE workgroup), so `out` may alias `resid`. ----
    float* lg = sbias + 256;                    // [256] gamma
    float* lb = lg + 256;                       // [256] beta
    float* red = lb + 256;                      // [2 wn][256 tokens] partial sums, then partial squared deviations
    if (tid < 256) { lg[tid] = p.ln_g[tid]; lb[tid] = p.ln_b[tid]; }
    __syncthreads();
    float su[2] = {0.f, 0.f};
    // residual quads of block (t, j + 1) are requested before block (t, j) is summed (two buffers: hoisting all 32 loads would spill)
    f32x4 rr[2][4];
    auto ln_res = [&](const int blk, f32x4 (&dst)[4]) {
      const int t = blk / NJ, j = blk % NJ;
      const int m = min(m0 + wm * 64 + t * 32 + li, p.M - 1);       // rows beyond M re-read the last row (never stored): no branch
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = wn * (BN / 2) + j * 32 + 8 * g + 4 * hi;
        dst[g] = *reinterpret_cast<const f32x4*>(p.resid + (long)m * p.ldr + n);
      }
    };
    ln_res(0, rr[0]);
#pragma unroll
    for (int blk = 0; blk < 2 * NJ; ++blk) {
      const int t = blk / NJ, j = blk % NJ;
      if (blk + 1 < 2 * NJ) ln_res(blk + 1, rr[(blk + 1) & 1]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = wn * (BN / 2) + j * 32 + 8 * g + 4 * hi;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = acc[j][t][4 * g + e] * p.alpha + b4[e];
          x += rr[blk & 1][g][e];
          acc[j][t][4 * g + e] = x;
          su[t] += x;
        }
      }
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      su[t] += __shfl_xor(su[t], 32);
      if (hi == 0) red[wn * 256 + wm * 64 + t * 32 + li] = su[t];
    }
    __syncthreads();
    float mean[2], sq[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      mean[t] = (su[t] + red[(wn ^ 1) * 256 + wm * 64 + t * 32 + li]) / 256.f;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc[j][t][r] - mean[t]; q += d * d; }
      sq[t] = q + __shfl_xor(q, 32);
    }
    __syncthreads();                             // everybody has read the partial sums
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (hi == 0) red[wn * 256 + wm * 64 + t * 32 + li] = sq[t];
    __syncthreads();
    float rstd[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) rstd[t] = rsqrtf((sq[t] + red[(wn ^ 1) * 256 + wm * 64 + t * 32 + li]) / 256.f + p.ln_eps);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      // t outside e: a (t = 0, t = 1) pair of `x - mean[t]` would be SLP-packed into v_pk_add_f32 with op_sel [0,1] -- the form of the
      // gfx950 packed-fp32 erratum (tests/test_isa_hazards.py refuses it)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = wn * (BN / 2) + j * 32 + 8 * g + 4 * hi;
          const f32x4 g4 = *reinterpret_cast<const f32x4*>(lg + n), b4 = *reinterpret_cast<const f32x4*>(lb + n);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][t][4 * g + e] = (acc[j][t][4 * g + e] - mean[t]) * rstd[t] * g4[e] + b4[e];
        }
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_sched_barrier(0);         // one block's gamma / beta reads at a time (hoisted together they spill)
#endif
    }
    __syncthreads();                             // the bias table has been read by everybody
    if (tid < BN) sbias[tid] = 0.f;              // the values are final: the store path below adds a zero bias, no residual, alpha 1
    p.alpha = 1.f; p.act = 0; p.oscale = 1.f; p.out_fmt = HIPIE_F32;   // compile-time facts of this instance from here on
    has_res = false;
  }
  if (SPLIT && (VAR == 0 || VAR == 8) && BN == 256 && p.softmax) {
    // ---- row softmax over the tile's columns (the whole row: one column tile).  A lane owns 64 of its token's 256 columns per token
    //      tile (its lane half's 4 of every 8, this wave's 128-column half): lane-local reduction, one exchange with the other lane half
    //      (xor 32), one with the partner wave (wn ^ 1) through LDS.  Column validity enters as a 0 / -inf table. ----
    float* kb = sbias + 256;                    // [256] 0 | -inf per column
    float* red = kb + 256;                      // [2 wn][256 tokens] partial max, then partial sums
    float* cb = red + 512;                      // [256] VAR 8: the logit bias of the column (q-side bias folded into the keys: bq . k_j)
    if (tid < 256) kb[tid] = (tid < p.sm_L && (p.sm_mask == nullptr || p.sm_mask[tid] != 0)) ? 0.f : -INFINITY;
    if (VAR == 8 && tid < 256) cb[tid] = (p.sm_bias != nullptr && tid < p.N) ? p.sm_bias[tid] : 0.f;
    __syncthreads();
    const float cl = p.sm_clamp;
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 k4 = *reinterpret_cast<const f32x4*>(kb + wn * (BN / 2) + j * 32 + 8 * g + 4 * hi);
          f32x4 c4 = {0.f, 0.f, 0.f, 0.f};
          if (VAR == 8) c4 = *reinterpret_cast<const f32x4*>(cb + wn * (BN / 2) + j * 32 + 8 * g + 4 * hi);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[j][t][4 * g + e] * p.alpha;
            if (VAR == 8) x += c4[e];
            if (cl > 0.f) x = __builtin_amdgcn_fmed3f(x, -cl, cl);
            x += k4[e];
            acc[j][t][4 * g + e] = x;
            mx[t] = fmaxf(mx[t], x);
          }
        }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32));
      if (hi == 0) red[wn * 256 + wm * 64 + t * 32 + li] = mx[t];
    }
    __syncthreads();
    float sm[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float m = fmaxf(mx[t], red[(wn ^ 1) * 256 + wm * 64 + t * 32 + li]);
      const float m2 = (m == -INFINITY) ? 0.f : m * 1.4426950408889634f;
      float su = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(acc[j][t][r] * 1.4426950408889634f - m2);     // exp2(-inf) = 0 on masked columns
          acc[j][t][r] = pv;
          su += pv;
        }
      sm[t] = su + __shfl_xor(su, 32);
    }
    __syncthreads();                             // everybody has read the partial maxima
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (hi == 0) red[wn * 256 + wm * 64 + t * 32 + li] = sm[t];
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float tot = sm[t] + red[(wn ^ 1) * 256 + wm * 64 + t * 32 + li];
      const float inv = tot > 0.f ? 1.f / tot : 0.f;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][t][r] *= inv;
    }
    p.alpha = 1.f;                               // the values are final: the store path below adds the (zero) bias and writes HL8
  }
  __syncthreads();
  // residual rows: the four quads of block (t, j + 1) are requested before block (t, j) is processed
  float4 rq[2][4];
  // output row of this lane's two tokens (identity, or the caller's row map: -1 drops the row)
  long orow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = m0 + wm * 64 + t * 32 + li;
    orow[t] = (m < p.M) ? (p.out_row != nullptr ? (long)p.out_row[m] : (long)m) : -1;
  }
  auto load_res = [&](const int t, const int j, float4 (&dst)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + wn * (BN / 2) + j * 32 + 8 * g + 4 * hi;
      dst[g] = (has_res && orow[t] >= 0 && n < p.N) ? *reinterpret_cast<const float4*>(p.resid + orow[t] * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // LayerNorm epilogue: the normalised rows leave twice -- fp32 (the stream), then HL8 (the operand of the GEMM that follows)
  const int passes = (LNE && p.out2 != nullptr) ? 2 : 1;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    if (pass >= passes) break;
    if (pass == 1) { p.out = p.out2; p.ldo = p.ldo2; p.out_fmt = HIPIE_HL8; }
    load_res(0, 0, rq[0]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const long m = orow[t];
      const bool mok = m >= 0;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int blk = t * NJ + j;
        if (blk + 1 < 2 * NJ) load_res((blk + 1) / NJ, (blk + 1) % NJ, rq[(blk + 1) & 1]);
        const int nb = n0 + wn * (BN / 2) + j * 32;             // first feature of the 32-row MFMA block
        gm_epi_quads<0, 4>(acc[j][t], rq[blk & 1], sbias + (nb - n0), m, mok, nb, hi, p, has_res);
#if defined(__HIP_DEVICE_COMPILE__)
        if (LNE) __builtin_amdgcn_sched_barrier(0);            // its store blocks are straight-line code: scheduled together they spill
#endif
      }
    }
  }
}



// ------------------------------------------------------------------------------------------------------------------------------
// gemm_small_kernel: the split product for SMALL problems (round 4) -- the decoder / BERT / head linears (M = 1.5k .. 8k rows) fill
// 7 .. 80 of the 256 x 256 tiles above, i.e. a fraction of the 256 CUs, each walking the whole K range alone: 14 ms of the step were
// ~250 launches of 0.03 .. 0.19 ms that are pure latency.  Here the tile is 64 tokens x 128 features on 4 waves (wave w owns feature
// block w: one 32-row MFMA block x 2 token tiles = 32 accumulator registers), 3 LDS slots of a k32 step (192 rows x 128 B = 24 KB:
// two workgroups per CU), one barrier per step: M = 2400, N = 256 becomes 76 workgroups of 8 short steps instead of 10 of them, and
// M = 1552, N = 768, K = 3072 (BERT's output dense) 150 instead of 21.  Same operand formats, swizzle and epilogue as gemm_kernel.
template <int VAR>
__global__ __launch_bounds__(256, 2) void gemm_small_kernel(const GemmParams p) {
  constexpr int BM = 64, BN = 128, ROWS = BM + BN, STAGE = ROWS * 128, NI = ROWS / 32;      // NI: DMA instructions per wave and stage
  typedef Mfma32<f16_t>::frag frag;
  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int tm = blockIdx.x / p.tiles_n, tn = blockIdx.x - tm * p.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  unsigned int dvoff[NI];
  {
    const int rl = lane >> 3, cp = lane & 7;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int r = 8 * (4 * i + wave) + rl;              // stage row
      const int c = cp ^ ((r >> 1) & 7);
      if (r < BM) dvoff[i] = (unsigned int)((long)min(r, p.M - 1 - m0) * p.lda_b + 16 * c);
      else dvoff[i] = (unsigned int)((long)min(r - BM, p.N - 1 - n0) * p.ldw_b + 16 * c);
    }
  }
  const char* abase = p.A + (long)m0 * p.lda_b;
  const char* wbase = p.W + (long)n0 * p.ldw_b;
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  auto dma_stage = [&](const int kt, const int slot) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const bool isa = (8 * (4 * i + wave)) < BM;         // wave-uniform (BM % 8 == 0)
      gm_dma16((isa ? abase : wbase) + (long)kt * 128, dvoff[i],
               __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(slot * STAGE + 1024 * (4 * i + wave))));
    }
  };

  const int swz = (li >> 1) & 7;
  const char* xrow = smem + li * 128;                               // + t * 32 * 128
  const char* wrow = smem + (BM + wave * 32 + li) * 128;
  auto choff = [&](const int ks, const int lo) -> int { return 16 * ((2 * (2 * ks + hi) + lo) ^ swz); };

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nkt = p.nkt;
  dma_stage(0, 0);
  if (nkt > 1) dma_stage(1, 1);
  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) __builtin_amdgcn_s_waitcnt(0x0F70 | NI);      // stage kt landed; stage kt + 1 may still be in flight
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();                                                // ... for every wave; all reads of stage kt - 1 are done
    if (kt + 2 < nkt) dma_stage(kt + 2, slot == 0 ? 2 : slot - 1);
    const char* xs = xrow + slot * STAGE;
    const char* ws = wrow + slot * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      frag xh[2], xl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xh[t] = *reinterpret_cast<const frag*>(xs + t * 4096 + choff(ks, 0));
        xl[t] = *reinterpret_cast<const frag*>(xs + t * 4096 + choff(ks, 1));
      }
      const frag wh = *reinterpret_cast<const frag*>(ws + choff(ks, 0));
      const frag wl = *reinterpret_cast<const frag*>(ws + choff(ks, 1));
      if (VAR == 2) {
        // fp32 A rows: the two 16-byte pieces hold x0..x3 / x4..x7 of the lane's k group (gemm_kernel VAR 2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 a = __builtin_bit_cast(f32x4, xh[t]), b = __builtin_bit_cast(f32x4, xl[t]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f16_t hh, ll;
            hl_split(a[e], hh, ll);
            xh[t][e] = hh; xl[t][e] = ll;
            hl_split(b[e], hh, ll);
            xh[t][4 + e] = hh; xl[t][4 + e] = ll;
          }
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = Mfma32<f16_t>::mma(wl, xh[t], acc[t]);
        acc[t] = Mfma32<f16_t>::mma(wh, xl[t], acc[t]);
        acc[t] = Mfma32<f16_t>::mma(wh, xh[t], acc[t]);
      }
    }
    slot = slot == 2 ? 0 : slot + 1;
  }

  // ---- epilogue ----
  const bool has_res = p.resid != nullptr;
  __syncthreads();                                                  // the last stage's reads are done: its slot holds the bias values now
  float* sbias = reinterpret_cast<float*>(smem);
  if (tid < BN) sbias[tid] = (p.bias != nullptr && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
  __syncthreads();
  const int nb = n0 + wave * 32;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int mm = m0 + t * 32 + li;
    const long m = (mm < p.M) ? (p.out_row != nullptr ? (long)p.out_row[mm] : (long)mm) : -1;
    float4 rq[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = nb + 8 * g + 4 * hi;
      rq[g] = (has_res && m >= 0 && n < p.N) ? *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    gm_epi_quads<0, 4>(acc[t], rq, sbias + wave * 32, m, m >= 0, nb, hi, p, has_res);
  }
}

template <int VAR>
static int launch_gemm_small(GemmParams& p, hipStream_t st) {
  constexpr size_t lds = (size_t)3 * (64 + 128) * 128;
  p.tiles_m = (p.M + 63) / 64;
  p.tiles_n = (p.N + 127) / 128;
  auto kern = gemm_small_kernel<VAR>;
  static bool lds_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), lds, st, p);
  return check_launch("gemm_small");
}

#ifdef HIPIE_GEMM_VARIANTS
#include "../../tools/ubench/gemm_overlap_study.h"     // round-4 timing study: not part of the product, lives with the micro-benchmarks
#endif

template <int BN, bool SPLIT, int VAR = 0>
static int launch_gemm(GemmParams& p, hipStream_t st, int batches = 1) {
  constexpr size_t lds = (size_t)2 * (256 + BN) * 128;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + BN - 1) / BN;
  // wide outputs (qkv: 12 column tiles, fc1: 16): the blocks of an XCD walk groups of 8 row panels with the row panel fastest, so the 32
  // workgroups resident on an XCD hold 8 A panels x 4 W panels instead of 2 x 16 -- 40 % fewer operand rows through that XCD's L2.
  // Same-box A/B (profiles/r06_gemm_tile_order.txt): qkv 0.894 -> 0.874 ms, fc1 1.099 -> 1.076 ms; up to 4 column tiles the plain order already is 8 x 4.
  p.group_m = p.tiles_n > 4 ? 8 : 0;
  auto kern = gemm_kernel<BN, SPLIT, VAR>;
  static bool lds_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = true;
  }
  if (VAR == 8)       // inner index fastest inside blockIdx.x (see the kernel): grid = tiles * n_inner x n_outer
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n * p.nbi), (unsigned)(batches / p.nbi)), dim3(512), lds, st, p);
  else
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n), (unsigned)batches), dim3(512), lds, st, p);
  return check_launch("gemm");
}

// gemm_ln.hip includes this file for the kernel template and its launcher only (its LayerNorm-epilogue instances are built without the
// SLP vectoriser, see the Makefile): everything below belongs to gemm.o alone.
#ifndef HIPIE_GEMM_LN_TU
// fp32 / fp16 rows -> HL8 (optionally scaled): the generic producer of split operands (weights are split once on the host)
template <typename T>
__global__ __launch_bounds__(256) void to_hl8_kernel(const T* __restrict__ x, f16_t* __restrict__ out, long rows, int K, long ldx, long ldo,
                                                     float scale) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;       // one group of 8 elements per thread
  const int gpr = K / 8;
  if (gid >= rows * gpr) return;
  const long r = gid / gpr;
  const int g = (int)(gid - r * gpr);
  const T* src = x + r * ldx + 8 * g;
  f16x8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    f16_t hh, ll;
    hl_split((float)src[e] * scale, hh, ll);
    h[e] = hh;
    l[e] = ll;
  }
  f16_t* dst = out + r * ldo + 16 * g;
  *reinterpret_cast<f16x8*>(dst) = h;
  *reinterpret_cast<f16x8*>(dst + 8) = l;
}


// x (rows x C fp32, row stride ldx) -> out (C x 2 rows_p) HL8: the TRANSPOSE as a split operand, for products that contract over the rows
// (the weight gradients of the training step: dW = dy^T . x needs dy^T and x^T with the token dimension as K).  One pass instead of a
// strided transpose copy followed by hipie_to_hl8: a 128-row x 64-column tile goes through LDS; columns beyond `rows` (up to rows_p, a
// multiple of 8) are written as zeros.
constexpr int kTrRows = 128, kTrCols = 64;
__global__ __launch_bounds__(256) void to_hl8_t_kernel(const float* __restrict__ x, f16_t* __restrict__ out, long rows, int C, long ldx,
                                                       long ldo, long rows_p, float scale) {
  __shared__ float tile[kTrRows][kTrCols + 1];
  const long r0 = (long)blockIdx.x * kTrRows;
  const int c0 = blockIdx.y * kTrCols;
  const int tid = threadIdx.x;
  for (int k = tid; k < kTrRows * (kTrCols / 4); k += 256) {           // coalesced along the columns
    const int i = k / (kTrCols / 4), j = (k % (kTrCols / 4)) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (r0 + i < rows) {
      const float* src = x + (r0 + i) * ldx + c0 + j;
      if (c0 + j + 3 < C && (((uintptr_t)src) & 15) == 0) {
        const float4 q = *reinterpret_cast<const float4*>(src);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
        for (int e = 0; e < 4; ++e) v[e] = c0 + j + e < C ? src[e] : 0.f;
      }
    }
    for (int e = 0; e < 4; ++e) tile[i][j + e] = v[e];
  }
  __syncthreads();
  for (int u = tid; u < kTrCols * (kTrRows / 8); u += 256) {           // unit = (output row c, group of 8 source rows)
    const int g = u % (kTrRows / 8), c = u / (kTrRows / 8);      // lanes along the 16 row groups: 512 contiguous bytes of one output row
    const long m = r0 + 8 * g;
    if (c0 + c >= C || m >= rows_p) continue;
    f16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f16_t hh, ll;
      hl_split(tile[8 * g + e][c] * scale, hh, ll);
      h[e] = hh;
      l[e] = ll;
    }
    f16_t* dst = out + (long)(c0 + c) * ldo + 2 * m;
    *reinterpret_cast<f16x8*>(dst) = h;
    *reinterpret_cast<f16x8*>(dst + 8) = l;
  }
}

}  // namespace hipie

using namespace hipie;

namespace hipie {       // gemm_k256.hip
int launch_gemm_k256(const void* X, long ldx_b, int x_f32, const void* W, long ldw_b, const float* bias, float* out, long ldo, int M, int N,
                     hipStream_t st);
}

static int gemm_impl(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                     void* out, int64_t ldo, const int32_t* out_row, const int32_t* a_row, int64_t a_rows, int M, int N, int K, int in_fmt,
                     int out_fmt, int act, float alpha, float oscale, void* stream) {
  HIPIE_REQUIRE(A && W && out, "gemm: null pointer");
  HIPIE_REQUIRE(in_fmt == HIPIE_F16 || in_fmt == HIPIE_HL8 || in_fmt == HIPIE_F32, "gemm: operand format %d (HIPIE_F16 | HIPIE_HL8 | HIPIE_F32)",
                in_fmt);
  HIPIE_REQUIRE(out_fmt == HIPIE_F32 || out_fmt == HIPIE_F16 || out_fmt == HIPIE_HL8, "gemm: output format %d", out_fmt);
  HIPIE_REQUIRE(act >= 0 && act <= 3, "gemm: activation %d (0 none, 1 gelu, 2 relu, 3 quick-gelu)", act);
  HIPIE_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0, "gemm: M=%d N=%d K=%d (N must be a multiple of 8)", M, N, K);
  const bool a_f32 = in_fmt == HIPIE_F32;       // A rows are plain fp32 (lda in fp32 elements), split in the kernel; W is HL8
  const bool split = in_fmt == HIPIE_HL8 || a_f32;
  if (a_f32) lda *= 2;                          // from here on in fp16 units like the HL8 form: the same bytes per row
  const int kq = split ? 32 : 64;               // elements per 128-byte k tile
  HIPIE_REQUIRE(K % kq == 0, "gemm: K=%d must be a multiple of %d", K, kq);
  const int epr = split ? 2 * K : K;            // fp16 elements per operand row
  HIPIE_REQUIRE(lda >= epr && ldw >= epr && lda % 8 == 0 && ldw % 8 == 0, "gemm: operand row strides %ld / %ld (>= %d, multiples of 8)",
                (long)lda, (long)ldw, epr);
  HIPIE_REQUIRE((long)256 * lda * 2 < (1L << 31) && (long)320 * ldw * 2 < (1L << 31), "gemm: row stride too large");
  const int opr = out_fmt == HIPIE_HL8 ? 2 * N : N;
  HIPIE_REQUIRE(ldo >= opr && ldo % 4 == 0, "gemm: output row stride %ld (>= %d)", (long)ldo, opr);
  HIPIE_REQUIRE(resid == nullptr || (ldr >= N && ldr % 4 == 0), "gemm: residual row stride %ld", (long)ldr);
  HIPIE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0 &&
                ((uintptr_t)bias % 16) == 0 && ((uintptr_t)resid % 16) == 0, "gemm: pointers must be 16-byte aligned");
  GemmParams p;
  p.A = (const char*)A; p.W = (const char*)W; p.bias = bias; p.resid = resid; p.out = (char*)out; p.out_row = out_row; p.a_row = a_row;
  p.lda_b = lda * 2; p.ldw_b = ldw * 2; p.ldr = ldr; p.ldo = ldo;
  p.M = M; p.N = N; p.K = K; p.nkt = K / kq;
  p.out_fmt = out_fmt; p.act = act; p.alpha = alpha; p.oscale = oscale;
  p.nbi = 1; p.a_bo = p.a_bi = p.w_bo = p.w_bi = p.o_bo = p.o_bi = 0;
  p.conv_kpt = 0; p.conv_wp = 0; p.softmax = 0; p.sm_L = 0; p.sm_clamp = 0.f; p.sm_mask = nullptr;
  p.ln_g = p.ln_b = nullptr; p.ln_eps = 0.f; p.out2 = nullptr; p.ldo2 = 0;
  hipStream_t st = (hipStream_t)stream;
  // K = 256 linears over many rows with a plain fp32 result: the thin-K kernel (gemm_k256.hip: X rows live in registers, the weight
  // streams through LDS in 32-feature chunks) where it is faster than the 256-column tiles -- N >= 384 (tools/bench_gemm_k256.py: -20 % at
  // N = 384, -7 % at 1024, -9 % at 2304; level at N = 256).  HIPIE_GEMM_K256=0: never, =1: every eligible shape (A/B timing)
  // (diagnostic switches are read from the environment ONCE per process, not per launch)
  static const int k256_mode = [] { const char* e = study_env("HIPIE_GEMM_K256"); return e ? atoi(e) : 2; }();
  const bool k256_on = k256_mode == 1 || (k256_mode == 2 && N >= 384);
  // the thin-K kernel addresses X rows with 32-bit offsets from the base of the whole matrix: M rows must stay below 4 GiB
  if (k256_on && split && K == 256 && out_fmt == HIPIE_F32 && act == 0 && resid == nullptr && out_row == nullptr && a_row == nullptr &&
      alpha == 1.f && oscale == 1.f && N % 32 == 0 && M >= 8192 && ldw * 2 == 1024 && (long)M * p.lda_b < (1L << 32))
    return launch_gemm_k256(A, p.lda_b, a_f32 ? 1 : 0, W, p.ldw_b, bias, (float*)out, ldo, M, N, st);
  const bool wide = (N % 320 == 0);
  // problems that fill less than 3/8 of the CUs with 256-row tiles go to the 64 x 128 tile kernel (HIPIE_GEMM_SMALL=0: never; A/B timing)
  static const int small_on = [] { const char* e = study_env("HIPIE_GEMM_SMALL"); return e ? atoi(e) : 1; }();
  static const long small_tiles = [] { const char* e = study_env("HIPIE_GEMM_SMALL_MAXTILES"); return e ? atol(e) : 96L; }();      // A/B runs: tools/bench_gemm_small.py big
  const bool small_ok = small_on && (long)((M + 255) / 256) * ((N + (wide ? 319 : 255)) / (wide ? 320 : 256)) < small_tiles;
#ifdef HIPIE_GEMM_VARIANTS
  { const char* e = study_env("HIPIE_GEMM_VARIANT"); const int v = e ? atoi(e) : 0;
    if (split && wide && v == 1) return launch_gemm<320, true, 1>(p, st);
    if (split && wide && v == 3) return launch_gemm<320, true, 3>(p, st); }
#endif
  p.prio_mode = 0;
  p.variant = 0;
#ifdef HIPIE_GEMM_VARIANTS
  { const char* e = study_env("HIPIE_GEMM_VARIANT"); p.variant = e ? atoi(e) : 0; }
  p.prio_mode = gemm2_prio();
  if (split && gemm2_mode() == 4) {
    const bool w160 = (N % 160 == 0);
    if (a_f32) return w160 ? launch_gemm4<5, 2>(p, st) : launch_gemm4<4, 2>(p, st);
    return w160 ? launch_gemm4<5, 0>(p, st) : launch_gemm4<4, 0>(p, st);
  }
  if (split && gemm2_mode() == 2) {
    const bool w160 = (N % 160 == 0);
    if (a_f32) return w160 ? launch_gemm3<5, 2>(p, st) : launch_gemm3<4, 2>(p, st);
    return w160 ? launch_gemm3<5, 0>(p, st) : launch_gemm3<4, 0>(p, st);
  }
  if (split && gemm2_mode() == 1) {
    const bool w160 = (N % 160 == 0);
    if (a_f32) return w160 ? launch_gemm2<5, 2>(p, st) : launch_gemm2<4, 2>(p, st);
    return w160 ? launch_gemm2<5, 0>(p, st) : launch_gemm2<4, 0>(p, st);
  }
#endif
  if (split && small_ok && a_row == nullptr) return a_f32 ? launch_gemm_small<2>(p, st) : launch_gemm_small<0>(p, st);
  if (a_f32) return wide ? launch_gemm<320, true, 2>(p, st) : launch_gemm<256, true, 2>(p, st);
  if (split) return wide ? launch_gemm<320, true>(p, st) : launch_gemm<256, true>(p, st);
  return wide ? launch_gemm<320, false>(p, st) : launch_gemm<256, false>(p, st);
}

extern "C" int hipie_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* resid, int64_t ldr,
                          void* out, int64_t ldo, const int32_t* out_row, int M, int N, int K, int in_fmt, int out_fmt, int act, float alpha,
                          float oscale, void* stream) {
  return gemm_impl(A, lda, W, ldw, bias, resid, ldr, out, ldo, out_row, nullptr, 0, M, N, K, in_fmt, out_fmt, act, alpha, oscale, stream);
}

extern "C" int hipie_gemm_gather(const void* A, int64_t lda, int64_t a_rows, const int32_t* a_row, const void* W, int64_t ldw, const float* bias,
                                 const float* resid, int64_t ldr, void* out, int64_t ldo, const int32_t* out_row, int M, int N, int K, int in_fmt,
                                 int out_fmt, int act, float alpha, float oscale, void* stream) {
  HIPIE_REQUIRE(a_row != nullptr && a_rows > 0, "gemm_gather: a_row map / operand row count missing");
  HIPIE_REQUIRE(in_fmt == HIPIE_HL8 || in_fmt == HIPIE_F32, "gemm_gather: split operands only (HIPIE_HL8 | HIPIE_F32 rows)");
  HIPIE_REQUIRE((long)a_rows * lda * (in_fmt == HIPIE_F32 ? 4 : 2) < (1L << 32), "gemm_gather: the gathered operand must stay below 4 GiB");
  return gemm_impl(A, lda, W, ldw, bias, resid, ldr, out, ldo, out_row, a_row, a_rows, M, N, K, in_fmt, out_fmt, act, alpha, oscale, stream);
}

extern "C" int hipie_gemm_batched(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                                  int64_t w_inner, void* out, int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner, int M,
                                  int N, int K, int out_fmt, float alpha, void* stream) {
  HIPIE_REQUIRE(A && W && out, "gemm_batched: null pointer");
  HIPIE_REQUIRE(out_fmt == HIPIE_F32 || out_fmt == HIPIE_HL8, "gemm_batched: output format %d (HIPIE_F32 | HIPIE_HL8)", out_fmt);
  HIPIE_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 32 == 0, "gemm_batched: M=%d N=%d K=%d (N %% 8, K %% 32)", M, N, K);
  HIPIE_REQUIRE(n_outer > 0 && n_inner > 0 && (long)n_outer * n_inner <= 65535, "gemm_batched: %d x %d problems", n_outer, n_inner);
  HIPIE_REQUIRE(lda >= 2 * K && ldw >= 2 * K && lda % 8 == 0 && ldw % 8 == 0, "gemm_batched: operand row strides %ld / %ld", (long)lda, (long)ldw);
  HIPIE_REQUIRE((long)256 * lda * 2 < (1L << 31) && (long)320 * ldw * 2 < (1L << 31), "gemm_batched: row stride too large");
  const int opr = out_fmt == HIPIE_HL8 ? 2 * N : N;
  HIPIE_REQUIRE(ldo >= opr && ldo % 4 == 0, "gemm_batched: output row stride %ld (>= %d)", (long)ldo, opr);
  HIPIE_REQUIRE(((a_outer | a_inner | w_outer | w_inner) % 8) == 0 && ((o_outer | o_inner) % 4) == 0, "gemm_batched: batch offsets must keep 16-byte alignment");
  HIPIE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_batched: pointers must be 16-byte aligned");
  GemmParams p;
  p.A = (const char*)A; p.W = (const char*)W; p.bias = nullptr; p.resid = nullptr; p.out = (char*)out; p.out_row = nullptr; p.a_row = nullptr;
  p.lda_b = lda * 2; p.ldw_b = ldw * 2; p.ldr = 0; p.ldo = ldo;
  p.M = M; p.N = N; p.K = K; p.nkt = K / 32;
  p.out_fmt = out_fmt; p.act = 0; p.alpha = alpha; p.oscale = 1.f;
  p.conv_kpt = 0; p.conv_wp = 0; p.softmax = 0; p.sm_L = 0; p.sm_clamp = 0.f; p.sm_mask = nullptr;
  const long osz = out_fmt == HIPIE_F32 ? 4 : 2;
  p.nbi = n_inner;
  p.a_bo = a_outer * 2; p.a_bi = a_inner * 2; p.w_bo = w_outer * 2; p.w_bi = w_inner * 2; p.o_bo = o_outer * osz; p.o_bi = o_inner * osz;
  hipStream_t st = (hipStream_t)stream;
  const int batches = n_outer * n_inner;
  p.prio_mode = 0;
  p.variant = 0;
#ifdef HIPIE_GEMM_VARIANTS
  if (gemm2_mode() == 1) return (N % 160 == 0) ? launch_gemm2<5, 0>(p, st, batches) : launch_gemm2<4, 0>(p, st, batches);
#endif
  return (N % 320 == 0) ? launch_gemm<320, true>(p, st, batches) : launch_gemm<256, true>(p, st, batches);
}

extern "C" int hipie_gemm_batched_resid(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                                        int64_t w_inner, const float* bias, const float* resid, int64_t ldr, int64_t r_outer, int64_t r_inner,
                                        float* out, int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner, int M, int N, int K,
                                        float alpha, void* stream) {
  HIPIE_REQUIRE(A && W && out, "gemm_batched_resid: null pointer");
  HIPIE_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 32 == 0, "gemm_batched_resid: M=%d N=%d K=%d (N %% 8, K %% 32)", M, N, K);
  HIPIE_REQUIRE(n_outer > 0 && n_inner > 0 && (long)n_outer * n_inner <= 65535, "gemm_batched_resid: %d x %d problems", n_outer, n_inner);
  HIPIE_REQUIRE(lda >= 2 * K && ldw >= 2 * K && lda % 8 == 0 && ldw % 8 == 0, "gemm_batched_resid: operand row strides %ld / %ld", (long)lda, (long)ldw);
  HIPIE_REQUIRE((long)256 * lda * 2 < (1L << 31) && (long)320 * ldw * 2 < (1L << 31), "gemm_batched_resid: row stride too large");
  HIPIE_REQUIRE(ldo >= N && ldo % 4 == 0, "gemm_batched_resid: output row stride %ld (>= %d)", (long)ldo, N);
  HIPIE_REQUIRE(resid == nullptr || (ldr >= N && ldr % 4 == 0 && ((r_outer | r_inner) % 4) == 0), "gemm_batched_resid: residual strides %ld / %ld / %ld",
                (long)ldr, (long)r_outer, (long)r_inner);
  HIPIE_REQUIRE(((a_outer | a_inner | w_outer | w_inner) % 8) == 0 && ((o_outer | o_inner) % 4) == 0, "gemm_batched_resid: batch offsets must keep 16-byte alignment");
  HIPIE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
                ((uintptr_t)resid % 16) == 0, "gemm_batched_resid: pointers must be 16-byte aligned");
  GemmParams p;
  p.A = (const char*)A; p.W = (const char*)W; p.bias = bias; p.resid = resid; p.out = (char*)out; p.out_row = nullptr; p.a_row = nullptr;
  p.lda_b = lda * 2; p.ldw_b = ldw * 2; p.ldr = ldr; p.ldo = ldo;
  p.M = M; p.N = N; p.K = K; p.nkt = K / 32;
  p.out_fmt = HIPIE_F32; p.act = 0; p.alpha = alpha; p.oscale = 1.f;
  p.conv_kpt = 0; p.conv_wp = 0; p.softmax = 0; p.sm_L = 0; p.sm_clamp = 0.f; p.sm_mask = nullptr;
  p.nbi = n_inner;
  p.a_bo = a_outer * 2; p.a_bi = a_inner * 2; p.w_bo = w_outer * 2; p.w_bi = w_inner * 2; p.o_bo = o_outer * 4; p.o_bi = o_inner * 4;
  p.r_bo = r_outer; p.r_bi = r_inner;
  p.prio_mode = 0; p.variant = 0;
  return (N % 320 == 0) ? launch_gemm<320, true>(p, (hipStream_t)stream, n_outer * n_inner) : launch_gemm<256, true>(p, (hipStream_t)stream, n_outer * n_inner);
}

extern "C" int hipie_gemm_batched_softmax(const void* A, int64_t lda, int64_t a_outer, int64_t a_inner, const void* W, int64_t ldw, int64_t w_outer,
                                          int64_t w_inner, void* out, int64_t ldo, int64_t o_outer, int64_t o_inner, int n_outer, int n_inner,
                                          int M, int N, int K, const unsigned char* mask, int L, float clamp, float alpha, void* stream) {
  HIPIE_REQUIRE(A && W && out, "gemm_batched_softmax: null pointer");
  HIPIE_REQUIRE(M > 0 && N > 0 && N <= 256 && N % 8 == 0 && K > 0 && K % 32 == 0 && L > 0 && L <= N,
                "gemm_batched_softmax: M=%d N=%d K=%d L=%d (N <= 256: the row must fit one column tile)", M, N, K, L);
  HIPIE_REQUIRE(n_outer > 0 && n_inner > 0 && (long)n_outer * n_inner <= 65535, "gemm_batched_softmax: %d x %d problems", n_outer, n_inner);
  HIPIE_REQUIRE(lda >= 2 * K && ldw >= 2 * K && lda % 8 == 0 && ldw % 8 == 0, "gemm_batched_softmax: operand row strides %ld / %ld", (long)lda, (long)ldw);
  HIPIE_REQUIRE((long)256 * lda * 2 < (1L << 31) && (long)320 * ldw * 2 < (1L << 31), "gemm_batched_softmax: row stride too large");
  HIPIE_REQUIRE(ldo >= 2 * N && ldo % 4 == 0, "gemm_batched_softmax: output row stride %ld (HL8: >= %d)", (long)ldo, 2 * N);
  HIPIE_REQUIRE(((a_outer | a_inner | w_outer | w_inner) % 8) == 0 && ((o_outer | o_inner) % 4) == 0, "gemm_batched_softmax: batch offsets must keep 16-byte alignment");
  HIPIE_REQUIRE(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0, "gemm_batched_softmax: pointers must be 16-byte aligned");
  GemmParams p;
  p.A = (const char*)A; p.W = (const char*)W; p.bias = nullptr; p.resid = nullptr; p.out = (char*)out; p.out_row = nullptr; p.a_row = nullptr;
  p.lda_b = lda * 2; p.ldw_b = ldw * 2; p.ldr = 0; p.ldo = ldo;
  p.M = M; p.N = N; p.K = K; p.nkt = K / 32;
  p.out_fmt = HIPIE_HL8; p.act = 0; p.alpha = alpha; p.oscale = 1.f;
  p.nbi = n_inner;
  p.a_bo = a_outer * 2; p.a_bi = a_inner * 2; p.w_bo = w_outer * 2; p.w_bi = w_inner * 2; p.o_bo = o_outer * 2; p.o_bi = o_inner * 2;
  p.conv_kpt = 0; p.conv_wp = 0; p.prio_mode = 0; p.variant = 0;
  p.softmax = 1; p.sm_L = L; p.sm_clamp = clamp; p.sm_mask = mask;
  return launch_gemm<256, true>(p, (hipStream_t)stream, n_outer * n_inner);
}

extern "C" int hipie_conv3x3_split(const void* x, int64_t ldx, const void* w, const float* bias, void* out, int64_t ldo, int64_t rows, int Wp,
                                   int C, int N, int in_fmt, int out_fmt, int act, void* stream) {
  HIPIE_REQUIRE(x && w && out, "conv3x3_split: null pointer");
  HIPIE_REQUIRE(in_fmt == HIPIE_HL8 || in_fmt == HIPIE_F32, "conv3x3_split: input format %d (HIPIE_HL8 | HIPIE_F32 rows)", in_fmt);
  HIPIE_REQUIRE(out_fmt == HIPIE_F32 || out_fmt == HIPIE_HL8, "conv3x3_split: output format %d (HIPIE_F32 | HIPIE_HL8)", out_fmt);
  HIPIE_REQUIRE(act >= 0 && act <= 2, "conv3x3_split: activation %d", act);
  HIPIE_REQUIRE(rows > 0 && rows < (1L << 31) && Wp >= 3 && C > 0 && C % 32 == 0 && N > 0 && N % 8 == 0, "conv3x3_split: rows=%ld Wp=%d C=%d N=%d",
                (long)rows, Wp, C, N);
  if (in_fmt == HIPIE_F32) ldx *= 2;                 // from here on in fp16 units (the same bytes per row as HL8)
  HIPIE_REQUIRE(ldx >= 2 * C && ldx % 8 == 0, "conv3x3_split: input row stride %ld", (long)ldx);
  HIPIE_REQUIRE((long)(256 + Wp + 2) * ldx * 2 < (1L << 31), "conv3x3_split: row stride too large");
  const int opr = out_fmt == HIPIE_HL8 ? 2 * N : N;
  HIPIE_REQUIRE(ldo >= opr && ldo % 4 == 0, "conv3x3_split: output row stride %ld", (long)ldo);
  HIPIE_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)out % 16) == 0 && ((uintptr_t)bias % 16) == 0,
                "conv3x3_split: pointers must be 16-byte aligned");
  GemmParams p;
  p.A = (const char*)x; p.W = (const char*)w; p.bias = bias; p.resid = nullptr; p.out = (char*)out; p.out_row = nullptr; p.a_row = nullptr;
  p.lda_b = ldx * 2; p.ldw_b = (long)2 * 9 * C * 2; p.ldr = 0; p.ldo = ldo;
  p.M = (int)rows; p.N = N; p.K = 9 * C; p.nkt = 9 * C / 32;
  p.out_fmt = out_fmt; p.act = act; p.alpha = 1.f; p.oscale = 1.f;
  p.nbi = 1; p.a_bo = p.a_bi = p.w_bo = p.w_bi = p.o_bo = p.o_bi = 0;
  p.conv_kpt = C / 32; p.conv_wp = Wp; p.softmax = 0; p.sm_L = 0; p.sm_clamp = 0.f; p.sm_mask = nullptr;
  p.prio_mode = 0; p.variant = 0;
  hipStream_t st = (hipStream_t)stream;
  const bool wide = (N % 320 == 0);
  if (in_fmt == HIPIE_F32) return wide ? launch_gemm<320, true, 2>(p, st) : launch_gemm<256, true, 2>(p, st);
  return wide ? launch_gemm<320, true>(p, st) : launch_gemm<256, true>(p, st);
}

extern "C" int hipie_to_hl8(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int K, int x_dtype, float scale, void* stream) {
  HIPIE_REQUIRE(x && out && rows > 0 && K > 0 && K % 8 == 0, "to_hl8: rows=%ld K=%d (K must be a multiple of 8)", (long)rows, K);
  HIPIE_REQUIRE(ldx >= K && ldo >= 2 * K && ldo % 8 == 0, "to_hl8: row strides %ld / %ld", (long)ldx, (long)ldo);
  const long n = rows * (K / 8);
  const dim3 grid((unsigned)((n + 255) / 256));
  hipStream_t st = (hipStream_t)stream;
  switch (x_dtype) {
    case HIPIE_F32: hipLaunchKernelGGL(to_hl8_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (f16_t*)out, rows, K, ldx, ldo, scale); break;
    case HIPIE_F16: hipLaunchKernelGGL(to_hl8_kernel<f16_t>, grid, dim3(256), 0, st, (const f16_t*)x, (f16_t*)out, rows, K, ldx, ldo, scale); break;
    default: return set_err(HIPIE_EINVAL, "to_hl8: dtype %d", x_dtype);
  }
  return check_launch("to_hl8");
}

extern "C" int hipie_to_hl8_t(const void* x, int64_t ldx, void* out, int64_t ldo, int64_t rows, int C, int64_t rows_p, float scale, void* stream) {
  HIPIE_REQUIRE(x && out && rows > 0 && C > 0, "to_hl8_t: rows=%ld C=%d", (long)rows, C);
  HIPIE_REQUIRE(rows_p >= rows && rows_p % 8 == 0 && ldx >= C && ldo >= 2 * rows_p && ldo % 8 == 0, "to_hl8_t: rows_p=%ld (>= rows, multiple of 8), strides %ld / %ld",
                (long)rows_p, (long)ldx, (long)ldo);
  HIPIE_REQUIRE(((uintptr_t)out % 16) == 0, "to_hl8_t: output must be 16-byte aligned");
  const long tiles_r = (rows_p + kTrRows - 1) / kTrRows, tiles_c = (C + kTrCols - 1) / kTrCols;
  HIPIE_REQUIRE(tiles_c <= 65535, "to_hl8_t: C=%d too wide", C);
  hipLaunchKernelGGL(to_hl8_t_kernel, dim3((unsigned)tiles_r, (unsigned)tiles_c), dim3(256), 0, (hipStream_t)stream, (const float*)x, (f16_t*)out,
                     (long)rows, C, (long)ldx, (long)ldo, (long)rows_p, scale);
  return check_launch("to_hl8_t");
}
#else
}  // namespace hipie (the part gemm_ln.hip uses ends inside it)
#endif  // HIPIE_GEMM_LN_TU

// mask_einsum.hip -- mask-logit contraction out[b,q,p] = sum_c embed[b,q,c] * feats[b,c,p]   (SURVEY row a21).
//
// Shape at 1024^2: Q = 300, C = 256, P = H*W = 65536 per image; 10.07 GFLOP but 67 MB of features in and 79 MB (f32)
// of logits out -> AI = 69 flop/B (f32 out), far below the MI355X ridge: HBM-bound, the output write dominates.
//
// Decomposition: one workgroup (8 waves, two per SIMD) owns 256 consecutive pixels p and ALL queries; wave w owns one
// 32-wide MFMA column block and keeps its whole (320 x 32) f32 accumulator in registers (10 x 16 = 160 AGPRs).  So every feature element is read from HBM exactly once
// and every logit written exactly once; `embed` (307 KB) is re-read per workgroup from L2 and staged through LDS in
// K-chunks of 16 channels.
//   B operand (features): feats is (C, P) with p fastest, lanes run along p -> each lane's 8 k-values are 8 different
//     rows, each row load is a 128-byte coalesced segment per half-wave; no LDS, no transpose needed because no other
//     wave ever uses these elements.
//   A operand (embed): LDS tile [320 q][16 c], row stride padded to keep ds_read_b128/b32 conflict-free.
// Precision modes:
//   0  v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain) -- MFMA-f32-bound at 157 TF (about 65 us / image);
//   1  bf16x3: a = a_hi + a_lo, b = b_hi + b_lo in bf16, out += a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (f32 accumulate):
//      ~2^-16 relative error at 3/16 of the f32-MFMA time -> back under the HBM roof;
//   2  single bf16 MFMA (2^-8 relative inputs).
#include <stdlib.h>

#include "common.h"
#include "mfma.h"

namespace hipie {

constexpr int ME_QB = 10;        // 32-row query blocks per pass (Q <= 320 per pass)
constexpr int ME_KC = 16;        // channels per K chunk (one k16 MFMA step / eight k2 steps)
constexpr int ME_PB = 1;         // 32-pixel MFMA column blocks per wave
constexpr int ME_WAVES = 8;      // waves per workgroup (two per SIMD, 256 registers each)
constexpr int ME_TP = ME_WAVES * ME_PB * 32;   // pixels per workgroup
constexpr int ME_F32_STRIDE = 17;   // floats per LDS row (mode 0): bank = (17 q + c) % 32 distinct over q
constexpr int ME_B16_STRIDE = 24;   // bf16 per LDS row (modes 1,2): 48 B rows -> 12 q % 64 banks, conflict-free b128

template <int PREC, typename OutT>
__global__ __launch_bounds__(ME_WAVES * 64) void mask_einsum_kernel(const float* __restrict__ embed,
                                                             const float* __restrict__ feats, OutT* __restrict__ out,
                                                             int Q, int C, int P) {
  __shared__ __attribute__((aligned(16))) char smem[(PREC == 0) ? ME_QB * 32 * ME_F32_STRIDE * 4
                                                                 : ME_QB * 32 * ME_B16_STRIDE * 2 * 2];
  float* e32 = reinterpret_cast<float*>(smem);
  bf16_t* ehi = reinterpret_cast<bf16_t*>(smem);
  bf16_t* elo = ehi + ME_QB * 32 * ME_B16_STRIDE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const long p0 = (long)blockIdx.x * ME_TP + wave * (ME_PB * 32);
  const float* E = embed + (long)b * Q * C;
  const float* F = feats + (long)b * C * P;
  OutT* O = out + (long)b * Q * P;
  // pixel columns of this lane; the tail workgroup clamps its loads and predicates its stores
  long pc[ME_PB];
  bool pv[ME_PB];
#pragma unroll
  for (int hb = 0; hb < ME_PB; ++hb) {
    pc[hb] = min(p0 + 32 * hb + li, (long)P - 1);
    pv[hb] = p0 + 32 * hb + li < P;
  }

  for (int q0 = 0; q0 < Q; q0 += ME_QB * 32) {
    const int nq = min(Q - q0, ME_QB * 32);
    const int nqb = (nq + 31) / 32;
    f32x16 acc[ME_QB][ME_PB];
#pragma unroll
    for (int i = 0; i < ME_QB; ++i)
#pragma unroll
      for (int j = 0; j < ME_PB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kc = 0; kc < C; kc += ME_KC) {
      // ---- B operands for the whole chunk straight from HBM (issued first: they overlap the LDS staging) ----
      float bv[ME_PB][ME_KC];   // [pixel block][k index inside chunk as seen by this lane]
      if (PREC == 0) {
        // step kk (0..15): lane holds F[kc + 2*kk + hi][p]
#pragma unroll
        for (int kk = 0; kk < ME_KC / 2; ++kk)
#pragma unroll
          for (int hb = 0; hb < ME_PB; ++hb) bv[hb][kk] = F[(long)(kc + 2 * kk + hi) * P + pc[hb]];
      } else {
        // step s (0..1), j (0..7): lane holds F[kc + 16*s + 8*hi + j][p]
#pragma unroll
        for (int s = 0; s < ME_KC / 16; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int hb = 0; hb < ME_PB; ++hb)
              bv[hb][8 * s + j] = F[(long)(kc + 16 * s + 8 * hi + j) * P + pc[hb]];
      }
      // ---- stage embed[q0 .. q0+320, kc .. kc+32) into LDS ----
      __syncthreads();   // previous chunk's readers are done
      for (int i = tid; i < ME_QB * 32 * (ME_KC / 4); i += ME_WAVES * 64) {
        const int q = i / (ME_KC / 4), c4 = (i % (ME_KC / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) v = *reinterpret_cast<const float4*>(E + (long)(q0 + q) * C + kc + c4);
        if (PREC == 0) {
          float* d = e32 + q * ME_F32_STRIDE + c4;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        } else {
          const float f[4] = {v.x, v.y, v.z, v.w};
          bf16x4 h, l;
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            h[t] = (bf16_t)f[t];
            l[t] = (bf16_t)(f[t] - (float)h[t]);
          }
          *reinterpret_cast<bf16x4*>(ehi + q * ME_B16_STRIDE + c4) = h;
          if (PREC == 1) *reinterpret_cast<bf16x4*>(elo + q * ME_B16_STRIDE + c4) = l;
        }
      }
      __syncthreads();
      // ---- MFMA ----
      if (PREC == 0) {
#pragma unroll
        for (int kk = 0; kk < ME_KC / 2; ++kk) {
#pragma unroll
          for (int qb = 0; qb < ME_QB; ++qb) {
            if (qb < nqb) {
              const float a = e32[(qb * 32 + li) * ME_F32_STRIDE + 2 * kk + hi];
#pragma unroll
              for (int hb = 0; hb < ME_PB; ++hb)
                acc[qb][hb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[hb][kk], acc[qb][hb], 0, 0, 0);
            }
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < ME_KC / 16; ++s) {
          bf16x8 bh[ME_PB], bl[ME_PB];
#pragma unroll
          for (int hb = 0; hb < ME_PB; ++hb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float f = bv[hb][8 * s + j];
              bh[hb][j] = (bf16_t)f;
              bl[hb][j] = (bf16_t)(f - (float)bh[hb][j]);
            }
#pragma unroll
          for (int qb = 0; qb < ME_QB; ++qb) {
            if (qb < nqb) {
              const int off = (qb * 32 + li) * ME_B16_STRIDE + 16 * s + 8 * hi;
              const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ehi + off);
              if (PREC == 1) {
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(elo + off);
#pragma unroll
                for (int hb = 0; hb < ME_PB; ++hb) {
                  acc[qb][hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[hb], acc[qb][hb], 0, 0, 0);
                  acc[qb][hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[hb], acc[qb][hb], 0, 0, 0);
                  acc[qb][hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[hb], acc[qb][hb], 0, 0, 0);
                }
              } else {
#pragma unroll
                for (int hb = 0; hb < ME_PB; ++hb)
                  acc[qb][hb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[hb], acc[qb][hb], 0, 0, 0);
              }
            }
          }
        }
      }
    }
    // ---- epilogue: each register r of a (qb, hb) block is one row q, 32 consecutive pixels per half-wave ----
#pragma unroll
    for (int qb = 0; qb < ME_QB; ++qb) {
      if (qb < nqb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qb * 32 + crow(r, hi);
          if (q < nq) {
#pragma unroll
            for (int hb = 0; hb < ME_PB; ++hb)
              if (pv[hb]) O[(long)(q0 + q) * P + pc[hb]] = elem<OutT>::from_f32(acc[qb][hb][r]);
          }
        }
      }
    }
  }
}

template <int PREC, typename OutT>
static int launch_me(const float* e, const float* f, void* out, int B, int Q, int C, int P, hipStream_t st) {
  hipLaunchKernelGGL((mask_einsum_kernel<PREC, OutT>), dim3((P + ME_TP - 1) / ME_TP, B), dim3(ME_WAVES * 64), 0, st, e, f, (OutT*)out, Q, C, P);
  return check_launch("mask_einsum");
}

template <int PREC>
static int dispatch_me(const float* e, const float* f, void* out, int B, int Q, int C, int P, int odt, hipStream_t st) {
  switch (odt) {
    case HIPIE_F32: return launch_me<PREC, float>(e, f, out, B, Q, C, P, st);
    case HIPIE_F16: return launch_me<PREC, f16_t>(e, f, out, B, Q, C, P, st);
    case HIPIE_BF16: return launch_me<PREC, bf16_t>(e, f, out, B, Q, C, P, st);
    default: return set_err(HIPIE_EINVAL, "mask_einsum: bad out_dtype %d", odt);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// 16-bit form (the fast policy): features arrive in the activation dtype (f16 / bf16), logits leave in 16 bit (or f32).
//   algorithmic bytes per image at 1024^2: 33.5 MB of features in + 39.3 MB of logits out = 73 MB (AI 138 flop/B) -- HBM bound.
// The product is computed TRANSPOSED, out^T = feats^T . embed^T: pixels are the MFMA rows, queries the columns.  Then
//   * the A operand of a lane is (pixel, 8 channels): with TWO interleaved pixel blocks per wave (block hb = pixels 2 i + hb)
//     one 4-byte load per channel row fetches a lane's pixel pair -- 128 contiguous bytes per half-wave and channel;
//   * in the accumulator a lane owns ONE query row and registers 4g .. 4g+3 of the two blocks are 8 CONSECUTIVE pixels of it:
//     the epilogue is one 16-byte store per lane (16-bit output) instead of sixteen 2-byte ones;
//   * the B operand (embed, pre-split by the host into 16-bit hi + lo parts: two MFMAs per product keep the query embedding
//     at ~2^-22; lo == NULL: single product) is a plain 16-byte load per lane from the L2-resident (Q, C) matrix.
// Workgroup = 8 waves = 256 pixels x 320 queries: wave w owns pixel group w >> 1 (64 pixels) and query blocks 5 (w & 1) ..
template <typename T, typename OutT, bool LO, int ABL>
__global__ __launch_bounds__(512, 2) void mask_einsum16_kernel(const T* __restrict__ ehi, const T* __restrict__ elo,
                                                               const T* __restrict__ feats, const float* __restrict__ row_bias,
                                                               OutT* __restrict__ out, int Q, int C, int P) {
  typedef typename Mfma32<T>::frag frag;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // embed k-chunk (16 channels of all 320 query rows, hi and lo parts) staged in LDS once per workgroup and k-step, double
  // buffered: every wave reads its 5 (x2) B fragments from there -- fetched per wave from L2 they were 5x the feature bytes
  constexpr int ESTR = 24;                               // elements per LDS row: 48 B -> conflict-free ds_read_b128
  constexpr int EBUF = 320 * ESTR;                       // one part (hi or lo) of one buffer
  __shared__ __attribute__((aligned(16))) T esm[2 * (LO ? 2 : 1) * EBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const int qh = wave & 1;
  const long p0 = (long)blockIdx.x * 256 + (wave >> 1) * 64;
  const T* F = feats + (long)b * C * P;
  const T* EH = ehi + (long)b * Q * C;
  const T* EL = LO ? elo + (long)b * Q * C : nullptr;
  OutT* O = out + (long)b * Q * P;
  const long pl = min(p0 + 2 * li, (long)P - 2);          // this lane's pixel pair (clamped: the tail predicates its stores)

  f32x16 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging role of this thread: 640 16-byte chunks per part and k-step (row q = chunk / 2, half = chunk % 2)
  constexpr int NCH = 640, PARTS = LO ? 2 : 1, CPT = (NCH * PARTS + 511) / 512;
  const T* esrc[CPT];
  int edst[CPT];
  bool eval_[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int idx = c * 512 + tid;
    eval_[c] = idx < NCH * PARTS;
    const int part = min(idx / NCH, PARTS - 1), ch = idx % NCH, q = min(ch >> 1, Q - 1), hf = ch & 1;
    esrc[c] = (part ? EL : EH) + (long)q * C + 8 * hf;
    edst[c] = part * EBUF + (ch >> 1) * ESTR + 8 * hf;
  }
  // Operand pipeline, two k-steps deep (one workgroup per CU: nothing else hides the L2 / HBM latency): at the top of step k
  // the loads of step k + 2 are issued; the embed chunk of step k + 1 (loaded one step ago) goes to LDS at the end of step k.
  u32x4 e1[CPT], e2[CPT];           // embed chunks of steps k + 1, k + 2 (registers)
  unsigned int fw[8], f1[8], f2[8]; // feature words of steps k, k + 1, k + 2
  const int nk = C / 16;
#pragma unroll
  for (int c = 0; c < CPT; ++c) e1[c] = *reinterpret_cast<const u32x4*>(esrc[c]);
#pragma unroll
  for (int j = 0; j < 8; ++j) fw[j] = *reinterpret_cast<const unsigned int*>(F + (long)(8 * hi + j) * P + pl);
#pragma unroll
  for (int c = 0; c < CPT; ++c)
    if (eval_[c]) *reinterpret_cast<u32x4*>(esm + edst[c]) = e1[c];
  if (nk > 1) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) e1[c] = *reinterpret_cast<const u32x4*>(esrc[c] + 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) f1[j] = *reinterpret_cast<const unsigned int*>(F + (long)(16 + 8 * hi + j) * P + pl);
  }
  __syncthreads();

  for (int k = 0; k < nk; ++k) {
    const T* ecur = esm + (k & 1) * PARTS * EBUF;
    if (k + 2 < nk) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) e2[c] = *reinterpret_cast<const u32x4*>(esrc[c] + 16 * (k + 2));
#pragma unroll
      for (int j = 0; j < 8; ++j) f2[j] = *reinterpret_cast<const unsigned int*>(F + (long)(16 * (k + 2) + 8 * hi + j) * P + pl);
    }
    u32x4 a0, a1;                                        // even-pixel / odd-pixel A fragments: 8 channels x 16 bit
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0[j] = __builtin_amdgcn_perm(fw[2 * j + 1], fw[2 * j], 0x05040100u);      // lo halves of (fw[2j], fw[2j+1])
      a1[j] = __builtin_amdgcn_perm(fw[2 * j + 1], fw[2 * j], 0x07060302u);      // hi halves
    }
    const frag af[2] = {__builtin_bit_cast(frag, a0), __builtin_bit_cast(frag, a1)};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = 32 * (5 * qh + i) + li;
      const frag bh = *reinterpret_cast<const frag*>(ecur + row * ESTR + 8 * hi);
      if (LO) {
        const frag bl = *reinterpret_cast<const frag*>(ecur + EBUF + row * ESTR + 8 * hi);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) acc[i][hb] = Mfma32<T>::mma(af[hb], bl, acc[i][hb]);
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        if (ABL == 1) { acc[i][hb][0] += (float)af[hb][0] + (float)bh[0]; }
        else acc[i][hb] = Mfma32<T>::mma(af[hb], bh, acc[i][hb]);
      }
    }
    if (k + 1 < nk) {
      T* enext = esm + ((k + 1) & 1) * PARTS * EBUF;     // last read in step k - 1, before the previous barrier
#pragma unroll
      for (int c = 0; c < CPT; ++c)
        if (eval_[c]) *reinterpret_cast<u32x4*>(enext + edst[c]) = e1[c];
#pragma unroll
      for (int c = 0; c < CPT; ++c) e1[c] = e2[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) { fw[j] = f1[j]; f1[j] = f2[j]; }
    }
    __syncthreads();
  }
  // ---- epilogue: lane = query row; registers 4g .. 4g+3 of the two blocks = pixels p0 + 16 g + 8 hi + (0 .. 7) ----
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int q = 32 * (5 * qh + i) + li;
    if (q >= Q) continue;
    if (ABL == 2 && acc[i][0][0] != 12345.f) continue;        // timing ablation: no stores
    OutT* orow = O + (long)q * P + p0;
    const float rb = row_bias ? row_bias[(long)b * Q + q] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long px = p0 + 16 * g + 8 * hi;
      if (px + 8 <= P) {
        typedef OutT o8 __attribute__((ext_vector_type(8)));
        o8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = elem<OutT>::from_f32(acc[i][0][4 * g + e] + rb);
          v[2 * e + 1] = elem<OutT>::from_f32(acc[i][1][4 * g + e] + rb);
        }
        *reinterpret_cast<o8*>(orow + 16 * g + 8 * hi) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (px + 2 * e < P) orow[16 * g + 8 * hi + 2 * e] = elem<OutT>::from_f32(acc[i][0][4 * g + e] + rb);
          if (px + 2 * e + 1 < P) orow[16 * g + 8 * hi + 2 * e + 1] = elem<OutT>::from_f32(acc[i][1][4 * g + e] + rb);
        }
      }
    }
  }
}

template <typename T, typename OutT>
static int launch_me16(const void* eh, const void* el, const void* f, const float* rb, void* out, int B, int Q, int C, int P, hipStream_t st) {
  dim3 grid((P + 255) / 256, B);
#ifdef HIPIE_VA_ABLATIONS
  { const char* e = getenv("HIPIE_ME_ABL"); const int a = e ? atoi(e) : 0;
    if (a == 1) { hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, false, 1>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P); return check_launch("me16"); }
    if (a == 2) { hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, false, 2>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P); return check_launch("me16"); } }
#endif
  if (el != nullptr)
    hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, true, 0>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P);
  else
    hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, false, 0>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P);
  return check_launch("mask_einsum16");
}

}  // namespace hipie

extern "C" int hipie_mask_einsum(const float* embed, const float* feats, void* out, int B, int Q, int C, int HW,
                                 int precision, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(embed && feats && out, "mask_einsum: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q > 0 && C > 0 && HW > 0, "mask_einsum: bad shape");
  HIPIE_REQUIRE(C % ME_KC == 0, "mask_einsum: C=%d must be a multiple of %d", C, ME_KC);
  if (B == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (precision) {
    case 0: return dispatch_me<0>(embed, feats, out, B, Q, C, HW, out_dtype, st);
    case 1: return dispatch_me<1>(embed, feats, out, B, Q, C, HW, out_dtype, st);
    case 2: return dispatch_me<2>(embed, feats, out, B, Q, C, HW, out_dtype, st);
    default: return set_err(HIPIE_EINVAL, "mask_einsum: bad precision %d", precision);
  }
}

extern "C" int hipie_mask_einsum16(const void* embed_hi, const void* embed_lo, const void* feats, const float* row_bias, void* out,
                                   int B, int Q, int C, int HW, int dtype, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(embed_hi && feats && out, "mask_einsum16: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q > 0 && Q <= 320 && C > 0 && HW > 0, "mask_einsum16: bad shape (Q=%d must be in 1..320)", Q);
  HIPIE_REQUIRE(C % 16 == 0 && HW % 2 == 0, "mask_einsum16: C=%d must be a multiple of 16 and HW=%d even", C, HW);
  HIPIE_REQUIRE((((uintptr_t)embed_hi | (uintptr_t)embed_lo | (uintptr_t)feats | (uintptr_t)out) & 15) == 0, "mask_einsum16: pointers must be 16-byte aligned");
  HIPIE_REQUIRE(out_dtype == dtype || out_dtype == HIPIE_F32, "mask_einsum16: out_dtype must be the feature dtype or f32");
  if (B == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == HIPIE_F16) {
    if (out_dtype == HIPIE_F16) { HIPIE_REQUIRE(HW % 8 == 0, "mask_einsum16: HW %% 8"); return launch_me16<f16_t, f16_t>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st); }
    return launch_me16<f16_t, float>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st);
  }
  if (dtype == HIPIE_BF16) {
    if (out_dtype == HIPIE_BF16) { HIPIE_REQUIRE(HW % 8 == 0, "mask_einsum16: HW %% 8"); return launch_me16<bf16_t, bf16_t>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st); }
    return launch_me16<bf16_t, float>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st);
  }
  return set_err(HIPIE_EINVAL, "mask_einsum16: dtype must be f16 or bf16 (got %d)", dtype);
}

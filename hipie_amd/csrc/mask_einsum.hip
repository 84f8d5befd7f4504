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

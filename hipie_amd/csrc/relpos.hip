// relpos.hip -- decomposed relative-position bias tables for the ViT attention.
//
// Reference: add_decomposed_rel_pos (hipie/backbone/utils.py:96-125):
//     rel_h[q, kh] = q . Rh[hq, kh],  rel_w[q, kw] = q . Rw[wq, kw],   Rh[hq, kh] = tab_h[hq - kh + (H - 1)]  (get_rel_pos, :63-93)
// i.e. both are Toeplitz gathers of ONE small GEMM  T = Q . [tab_h ; tab_w]^T  ((N x hd) x (hd x (2H-1 + 2W-1))).  In eager
// PyTorch this is a cast of q to fp32, two einsums (each a permuted copy in, an fp32 batched GEMM and a permuted copy out):
// ~20 ms per ViT-H forward at batch 8.  Here one launch per block reads q in place from the packed 16-bit qkv tensor, runs T on
// MFMA (the Q fragments are the same registers the attention kernel uses), transposes / skews through LDS and writes the two
// tables exactly as hipie_vit_attn consumes them: rel_h (B*heads, H, N) key-row major, rel_w (B*heads, N, W).
// HBM-write bound: 4 B x N x (H + W) per (batch, head) = 268 MB per ViT-H global block at batch 8.
#include "common.h"
#include "mfma.h"

namespace hipie {

constexpr int RP_WAVES = 4;
constexpr int RP_MAXG = 96;         // token grid side limit (2*96 - 1 = 191 table rows -> 6 column blocks)

template <typename T, int HD>
__global__ __launch_bounds__(RP_WAVES * 64) void relpos_kernel(const T* __restrict__ qkv, const T* __restrict__ tab_h,
                                                               const T* __restrict__ tab_w, float* __restrict__ rel_h,
                                                               float* __restrict__ rel_w, int heads, int gh, int gw) {
  constexpr int KS = HD / 16;
  typedef typename Mfma32<T>::frag frag;
  extern __shared__ __attribute__((aligned(16))) float rp_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int N = gh * gw;
  const int bh = blockIdx.y, b = bh / heads, h = bh % heads;
  const int q0 = blockIdx.x * (RP_WAVES * 32) + wave * 32;
  const long C3 = 3L * heads * HD;
  const T* Qg = qkv + (long)b * N * C3 + (long)h * HD;
  // per-wave LDS tiles: hbuf[gh][33] (kh-major, 32 queries + pad), wbuf[32][gw + 1]
  float* hbuf = rp_smem + wave * (gh * 33 + 32 * (gw + 1));
  float* wbuf = hbuf + gh * 33;

  // A operand: this lane's query row (same fragment layout as the attention kernel's Q operand)
  const int qa = min(q0 + li, N - 1);
  frag qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const frag*>(Qg + (long)qa * C3 + 16 * ks + 8 * hi);

  for (int part = 0; part < 2; ++part) {
    const T* tab = part == 0 ? tab_h : tab_w;
    const int g = part == 0 ? gh : gw;
    const int nrows = 2 * g - 1;
    for (int j0 = 0; j0 < nrows; j0 += 32) {
      const int jr = min(j0 + li, nrows - 1);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag tf = *reinterpret_cast<const frag*>(tab + (long)jr * HD + 16 * ks + 8 * hi);
        acc = Mfma32<T>::mma(qf[ks], tf, acc);          // D[q][j] = sum_c Q[q][c] tab[j][c]
      }
      const int j = j0 + li;
      if (j < nrows) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ql = crow(r, hi);                     // local query 0..31
          const int q = min(q0 + ql, N - 1);
          const int pos = part == 0 ? q / gw : q % gw;    // hq or wq
          const int kk = pos + (g - 1) - j;               // key row / column this table entry belongs to
          if (kk >= 0 && kk < g) {
            if (part == 0) hbuf[kk * 33 + ql] = acc[r];
            else wbuf[ql * (gw + 1) + kk] = acc[r];
          }
        }
      }
    }
  }
  __syncthreads();
  // coalesced write-out: rel_h rows of 32 consecutive queries, rel_w rows of gw consecutive key columns
  const int nq = min(32, N - q0);
  if (nq > 0) {
    float* oh = rel_h + (long)bh * gh * N + q0;
    for (int i = lane; i < gh * 32; i += 64) {
      const int kh = i >> 5, ql = i & 31;
      if (ql < nq) oh[(long)kh * N + ql] = hbuf[kh * 33 + ql];
    }
    float* ow = rel_w + ((long)bh * N + q0) * gw;
    for (int i = lane; i < nq * gw; i += 64) {
      const int ql = i / gw, kw = i - ql * gw;
      ow[i] = wbuf[ql * (gw + 1) + kw];
    }
  }
}

template <typename T>
static int launch_rp(const void* qkv, const void* th, const void* tw, float* rh, float* rw, int B, int gh, int gw, int heads,
                     int hd, hipStream_t st) {
  const int N = gh * gw;
  dim3 grid((N + RP_WAVES * 32 - 1) / (RP_WAVES * 32), B * heads);
  const size_t lds = (size_t)RP_WAVES * (gh * 33 + 32 * (gw + 1)) * sizeof(float);
  switch (hd) {
    case 64: hipLaunchKernelGGL((relpos_kernel<T, 64>), grid, dim3(RP_WAVES * 64), lds, st, (const T*)qkv, (const T*)th, (const T*)tw, rh, rw, heads, gh, gw); break;
    case 80: hipLaunchKernelGGL((relpos_kernel<T, 80>), grid, dim3(RP_WAVES * 64), lds, st, (const T*)qkv, (const T*)th, (const T*)tw, rh, rw, heads, gh, gw); break;
    default: return set_err(HIPIE_EINVAL, "vit_relpos: head_dim %d unsupported (64, 80)", hd);
  }
  return check_launch("vit_relpos");
}

}  // namespace hipie

extern "C" int hipie_vit_relpos(const void* qkv, const void* tab_h, const void* tab_w, float* rel_h, float* rel_w, int B,
                                int gh, int gw, int heads, int hd, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(qkv && tab_h && tab_w && rel_h && rel_w, "vit_relpos: null pointer");
  HIPIE_REQUIRE(B > 0 && gh > 0 && gw > 0 && gh <= RP_MAXG && gw <= RP_MAXG && heads > 0, "vit_relpos: bad shape");
  HIPIE_REQUIRE(B * heads < 65536, "vit_relpos: B*heads too large for one launch");
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HIPIE_F16: return launch_rp<f16_t>(qkv, tab_h, tab_w, rel_h, rel_w, B, gh, gw, heads, hd, st);
    case HIPIE_BF16: return launch_rp<bf16_t>(qkv, tab_h, tab_w, rel_h, rel_w, B, gh, gw, heads, hd, st);
    default: return set_err(HIPIE_EINVAL, "vit_relpos: dtype must be HIPIE_F16 or HIPIE_BF16");
  }
}

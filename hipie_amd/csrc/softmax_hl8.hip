// softmax_hl8.hip -- row softmax of fp32 logits with a per-batch key mask, written as the split-fp16 (HL8) A operand of the GEMM that
// follows: the middle step of the image -> text direction of the vision-language fusion in the split policy,
//     S = Q . K^T (hipie_gemm_batched, three products)  ->  P = softmax(clamp(S) masked)  ->  out = P . V_text (hipie_gemm_batched),
// the fp32-class form of  attn_weights_v = softmax(clamp(q k^T) + mask) ; bmm(attn_probs_v, value_l_states)
// (models/deformable_detr/fuse_helper.py:77-121).  Single-fp16 q / k in this attention seed 2e-5 of error into the 21760-token
// memory, which the six decoder layers amplify to 2e-3 on the mask logits at the headline configuration (tools/dec_err_full.py).
// One wavefront per row; a lane owns whole groups of 8 columns (32 bytes in, 32 bytes out); masked and padding columns become 0.
#include "common.h"

namespace hipie {

template <int MAXG>      // groups of 8 columns per lane: columns <= 512 * MAXG
__global__ __launch_bounds__(256) void softmax_hl8_kernel(const float* __restrict__ S, long lds_, f16_t* __restrict__ P, long ldp, long rows,
                                                          int L, int Lp, const unsigned char* __restrict__ mask, long rows_per_batch,
                                                          float clamp) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* s = S + row * lds_;
  const unsigned char* mk = mask ? mask + (row / rows_per_batch) * L : nullptr;
  const int ngroups = Lp / 8;
  float v[MAXG][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = lane + 64 * i;
    if (g < ngroups) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(s + 8 * g), b = *reinterpret_cast<const f32x4*>(s + 8 * g + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = 8 * g + e;
        float x = e < 4 ? a[e] : b[e - 4];
        if (clamp > 0.f) x = fminf(fmaxf(x, -clamp), clamp);
        const bool keep = c < L && (mk == nullptr || mk[c] != 0);
        v[i][e] = keep ? x : -INFINITY;
        mx = fmaxf(mx, v[i][e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = -INFINITY;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  const float m2 = (mx == -INFINITY) ? 0.f : mx * 1.4426950408889634f;
#pragma unroll
  for (int i = 0; i < MAXG; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v[i][e] = __builtin_amdgcn_exp2f(v[i][e] * 1.4426950408889634f - m2);      // exp2(-inf) = 0 on masked / padding columns
      sum += v[i][e];
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
  f16_t* pr = P + row * ldp;
#pragma unroll
  for (int i = 0; i < MAXG; ++i) {
    const int g = lane + 64 * i;
    if (g < ngroups) {
      f16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        f16_t hh, ll;
        hl_split(v[i][e] * inv, hh, ll);
        h[e] = hh;
        l[e] = ll;
      }
      *reinterpret_cast<f16x8*>(pr + 16 * g) = h;
      *reinterpret_cast<f16x8*>(pr + 16 * g + 8) = l;
    }
  }
}

}  // namespace hipie

extern "C" int hipie_softmax_hl8(const float* S, int64_t lds, void* P, int64_t ldp, int64_t rows, int L, int Lp, const unsigned char* mask,
                                 int64_t rows_per_batch, float clamp, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(S && P && rows > 0, "softmax_hl8: null pointer / no rows");
  HIPIE_REQUIRE(L > 0 && Lp >= L && Lp % 8 == 0 && Lp <= 4096, "softmax_hl8: L=%d Lp=%d (Lp a multiple of 8, <= 4096)", L, Lp);
  HIPIE_REQUIRE(lds >= Lp && lds % 4 == 0 && ldp >= 2 * Lp && ldp % 8 == 0, "softmax_hl8: row strides %ld / %ld", (long)lds, (long)ldp);
  HIPIE_REQUIRE(((uintptr_t)S % 16) == 0 && ((uintptr_t)P % 16) == 0, "softmax_hl8: pointers must be 16-byte aligned");
  HIPIE_REQUIRE(mask == nullptr || rows_per_batch > 0, "softmax_hl8: rows_per_batch");
  const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (Lp <= 512) hipLaunchKernelGGL(softmax_hl8_kernel<1>, grid, block, 0, st, S, (long)lds, (f16_t*)P, (long)ldp, (long)rows, L, Lp, mask, (long)rows_per_batch, clamp);
  else if (Lp <= 1024) hipLaunchKernelGGL(softmax_hl8_kernel<2>, grid, block, 0, st, S, (long)lds, (f16_t*)P, (long)ldp, (long)rows, L, Lp, mask, (long)rows_per_batch, clamp);
  else hipLaunchKernelGGL(softmax_hl8_kernel<8>, grid, block, 0, st, S, (long)lds, (f16_t*)P, (long)ldp, (long)rows, L, Lp, mask, (long)rows_per_batch, clamp);
  return check_launch("softmax_hl8");
}

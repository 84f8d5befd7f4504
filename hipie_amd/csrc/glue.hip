// glue.hip -- small fused kernels of the two DINO decoders: each replaces a chain of 8-25 tiny eager
// launches per decoder layer (15 layers per forward) with one launch.  Latency-bound by nature; exact fp32 arithmetic in
// the reference's operation order.
#include "common.h"

namespace hipie {

// get_sine_pos_embed (deformable_transformer_dino.py:636-670) == gen_sineembed_for_position (maskdino/utils/utils.py:74-100):
//   out[n, c' * F + 2k]     = sin(x_c * 2pi / dim_t[2k])
//   out[n, c' * F + 2k + 1] = cos(x_c * 2pi / dim_t[2k + 1])         dim_t[i] = T^(2 * floor(i/2) / F)  (host table, same formula)
// with the output coordinate order (y, x, w, h) for input (x, y, w, h) (exchange_xy).  One thread per output element.
template <typename OutT>
__global__ __launch_bounds__(256) void sine_embed_kernel(const float* __restrict__ ref, const float* __restrict__ dim_t,
                                                         OutT* __restrict__ out, long n, int nc, int F, int ref_stride, float scale) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = n * nc * F;
  if (i >= total) return;
  const int f = (int)(i % F);
  const int co = (int)((i / F) % nc);
  const long row = i / ((long)F * nc);
  const int ci = (co == 0) ? 1 : (co == 1) ? 0 : co;              // exchange_xy: output block 0 <- y, 1 <- x
  const float s = ref[row * ref_stride + ci] * scale / dim_t[f];
  out[i] = elem<OutT>::from_f32((f & 1) ? cosf(s) : sinf(s));
}

// iterative box refinement (deformable_transformer_dino.py:502-520, dino_decoder.py:150-160):
//   new = sigmoid(delta + inverse_sigmoid(ref)),  inverse_sigmoid(x) = log(clamp(clamp(x,0,1), eps) / clamp(1 - clamp(x,0,1), eps))
template <typename Td>
__global__ __launch_bounds__(256) void box_refine_kernel(const Td* __restrict__ delta, const float* __restrict__ ref,
                                                         float* __restrict__ out, long n, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float x = fminf(fmaxf(ref[i], 0.f), 1.f);
  const float x1 = fmaxf(x, eps), x2 = fmaxf(1.f - x, eps);
  const float v = elem<Td>::to_f32(delta[i]) + logf(x1 / x2);
  out[i] = 1.f / (1.f + expf(-v));
}

}  // namespace hipie

extern "C" int hipie_sine_embed(const float* ref, const float* dim_t, void* out, int64_t n, int n_coord, int num_pos_feats,
                                int ref_stride, float scale, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(ref && dim_t && out, "sine_embed: null pointer");
  HIPIE_REQUIRE(n >= 0 && (n_coord == 2 || n_coord == 4) && num_pos_feats > 0 && ref_stride >= n_coord, "sine_embed: bad shape");
  if (n == 0) return HIPIE_OK;
  const long total = n * n_coord * num_pos_feats;
  const unsigned grid = (unsigned)((total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case HIPIE_F32: hipLaunchKernelGGL((sine_embed_kernel<float>), dim3(grid), dim3(256), 0, st, ref, dim_t, (float*)out, (long)n, n_coord, num_pos_feats, ref_stride, scale); break;
    case HIPIE_F16: hipLaunchKernelGGL((sine_embed_kernel<f16_t>), dim3(grid), dim3(256), 0, st, ref, dim_t, (f16_t*)out, (long)n, n_coord, num_pos_feats, ref_stride, scale); break;
    case HIPIE_BF16: hipLaunchKernelGGL((sine_embed_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, ref, dim_t, (bf16_t*)out, (long)n, n_coord, num_pos_feats, ref_stride, scale); break;
    default: return set_err(HIPIE_EINVAL, "sine_embed: bad out_dtype %d", out_dtype);
  }
  return check_launch("sine_embed");
}

extern "C" int hipie_box_refine(const void* delta, const float* ref, float* out, int64_t n, float eps, int delta_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(delta && ref && out, "box_refine: null pointer");
  HIPIE_REQUIRE(n >= 0, "box_refine: bad size");
  if (n == 0) return HIPIE_OK;
  const unsigned grid = (unsigned)((n + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  switch (delta_dtype) {
    case HIPIE_F32: hipLaunchKernelGGL((box_refine_kernel<float>), dim3(grid), dim3(256), 0, st, (const float*)delta, ref, out, (long)n, eps); break;
    case HIPIE_F16: hipLaunchKernelGGL((box_refine_kernel<f16_t>), dim3(grid), dim3(256), 0, st, (const f16_t*)delta, ref, out, (long)n, eps); break;
    case HIPIE_BF16: hipLaunchKernelGGL((box_refine_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, (const bf16_t*)delta, ref, out, (long)n, eps); break;
    default: return set_err(HIPIE_EINVAL, "box_refine: bad delta_dtype %d", delta_dtype);
  }
  return check_launch("box_refine");
}

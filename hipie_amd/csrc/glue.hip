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

// dst row rows[i] <- the ONE source row (row_bytes a multiple of 16): e.g. the HL8 bias row into the padding rows of a window-layout qkv
// buffer.  One workgroup per destination row, 16-byte stores.
__global__ __launch_bounds__(256) void fill_rows_kernel(char* __restrict__ dst, long ld_bytes, const int* __restrict__ rows,
                                                        const char* __restrict__ src, int row_bytes) {
  const long r = rows[blockIdx.x];
  if (r < 0) return;
  char* d = dst + r * ld_bytes;
  for (int o = threadIdx.x * 16; o < row_bytes; o += 256 * 16)
    *reinterpret_cast<uint4*>(d + o) = *reinterpret_cast<const uint4*>(src + o);
}

}  // namespace hipie

extern "C" int hipie_fill_rows(void* dst, int64_t ld_bytes, const int32_t* rows, int64_t n_rows, const void* src_row, int64_t row_bytes,
                               void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(n_rows >= 0 && n_rows < (1L << 31), "fill_rows: n_rows=%ld", (long)n_rows);
  if (n_rows == 0) return HIPIE_OK;
  HIPIE_REQUIRE(dst && rows && src_row, "fill_rows: null pointer");
  HIPIE_REQUIRE(row_bytes > 0 && row_bytes % 16 == 0 && row_bytes < (1L << 30) && ld_bytes >= row_bytes && ld_bytes % 16 == 0 &&
                ((uintptr_t)dst % 16) == 0 && ((uintptr_t)src_row % 16) == 0, "fill_rows: rows of %ld bytes, stride %ld (16-byte units)",
                (long)row_bytes, (long)ld_bytes);
  hipLaunchKernelGGL(fill_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, (char*)dst, (long)ld_bytes, rows,
                     (const char*)src_row, (int)row_bytes);
  return check_launch("fill_rows");
}

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

// ---------------------------------------------------------------------------------------------------------------------------
// The two small MLP heads that run once per decoder layer on a few thousand query rows (deformable_transformer_dino.py:484-520):
//   ref_point_head:  query_pos = W2 . relu(W1 . sine(ref) + b1) + b2          (512 -> 256 -> 256; was sine kernel + 2 GEMMs)
//   bbox_embed:      new_ref   = sigmoid(W3 . relu(W2 . relu(W1 . x + b1) + b2) + b3 + inverse_sigmoid(ref))
//                                                                              (256 -> 256 -> 256 -> 4; was 3 GEMMs + refine)
// Latency-bound (0.3 - 1 GF): one launch each.  A workgroup owns 8 rows; the input / hidden rows sit in LDS (k-major, so a
// thread fetches its 8 row values with two 16-byte broadcast reads), thread j owns output column j and streams column j of the
// TRANSPOSED weight (in, out) -- coalesced across the workgroup, L2-resident across workgroups.  fp32 FMA accumulation; 16-bit
// weights / activations are rounded exactly where the library path rounded them (GEMM inputs and outputs).
namespace hipie {

constexpr int MLP_R = 8;

template <typename W>
__device__ __forceinline__ void mlp_layer(const float* __restrict__ xs /*[IN][8]*/, const W* __restrict__ wt /*[IN][OUT]*/,
                                          const W* __restrict__ bias, int IN, int OUT, int j, float (&acc)[MLP_R]) {
  const float b = (j < OUT) ? elem<W>::to_f32(bias[j]) : 0.f;
#pragma unroll
  for (int r = 0; r < MLP_R; ++r) acc[r] = b;
  if (j >= OUT) return;
#pragma unroll 4
  for (int k = 0; k < IN; ++k) {
    const float w = elem<W>::to_f32(wt[(long)k * OUT + j]);
    const f32x4 a = *reinterpret_cast<const f32x4*>(xs + k * MLP_R);
    const f32x4 c = *reinterpret_cast<const f32x4*>(xs + k * MLP_R + 4);
    acc[0] = fmaf(w, a[0], acc[0]); acc[1] = fmaf(w, a[1], acc[1]); acc[2] = fmaf(w, a[2], acc[2]); acc[3] = fmaf(w, a[3], acc[3]);
    acc[4] = fmaf(w, c[0], acc[4]); acc[5] = fmaf(w, c[1], acc[5]); acc[6] = fmaf(w, c[2], acc[6]); acc[7] = fmaf(w, c[3], acc[7]);
  }
}

// query_pos of one decoder layer.  ref (n, ref_stride >= 4) f32; w1t (512, 256), w2t (256, 256) transposed weights, b1, b2 in W;
// out (n, 256) in W.  The sine features and the hidden activations are rounded to W (what the 16-bit GEMM path feeds / emits).
template <typename W>
__global__ __launch_bounds__(256) void ref_point_mlp_kernel(const float* __restrict__ ref, const float* __restrict__ dim_t,
                                                            const W* __restrict__ w1t, const W* __restrict__ b1,
                                                            const W* __restrict__ w2t, const W* __restrict__ b2,
                                                            W* __restrict__ out, long n, int ref_stride, float scale) {
  __shared__ __attribute__((aligned(16))) float xs[512 * MLP_R];
  __shared__ __attribute__((aligned(16))) float hs[256 * MLP_R];
  const long row0 = (long)blockIdx.x * MLP_R;
  const int j = threadIdx.x;
  // sine embedding, output coordinate order (y, x, w, h), 128 features each (sin on even, cos on odd feature index)
  for (int i = j; i < 512 * MLP_R; i += 256) {
    const int r = i & (MLP_R - 1), f = i >> 3;                  // feature f = co * 128 + ff
    const int co = f >> 7, ff = f & 127;
    const int ci = (co == 0) ? 1 : (co == 1) ? 0 : co;
    const long row = min(row0 + r, n - 1);
    const float s = ref[row * ref_stride + ci] * scale / dim_t[ff];
    xs[f * MLP_R + r] = elem<W>::to_f32(elem<W>::from_f32((ff & 1) ? cosf(s) : sinf(s)));
  }
  __syncthreads();
  float acc[MLP_R];
  mlp_layer<W>(xs, w1t, b1, 512, 256, j, acc);
#pragma unroll
  for (int r = 0; r < MLP_R; ++r) hs[j * MLP_R + r] = elem<W>::to_f32(elem<W>::from_f32(fmaxf(acc[r], 0.f)));
  __syncthreads();
  mlp_layer<W>(hs, w2t, b2, 256, 256, j, acc);
#pragma unroll
  for (int r = 0; r < MLP_R; ++r)
    if (row0 + r < n) out[(row0 + r) * 256 + j] = elem<W>::from_f32(acc[r]);
}

// box head + iterative refinement of one decoder layer, all fp32: x (n, 256), ref (n, 4) -> out (n, 4)
__global__ __launch_bounds__(256) void box_head_kernel(const float* __restrict__ x, const float* __restrict__ ref,
                                                       const float* __restrict__ w1t, const float* __restrict__ b1,
                                                       const float* __restrict__ w2t, const float* __restrict__ b2,
                                                       const float* __restrict__ w3t, const float* __restrict__ b3,
                                                       float* __restrict__ out, long n, float eps) {
  __shared__ __attribute__((aligned(16))) float xs[256 * MLP_R];
  __shared__ __attribute__((aligned(16))) float hs[256 * MLP_R];
  const long row0 = (long)blockIdx.x * MLP_R;
  const int j = threadIdx.x;
#pragma unroll
  for (int r = 0; r < MLP_R; ++r) xs[j * MLP_R + r] = x[min(row0 + r, n - 1) * 256 + j];
  __syncthreads();
  float acc[MLP_R];
  mlp_layer<float>(xs, w1t, b1, 256, 256, j, acc);
#pragma unroll
  for (int r = 0; r < MLP_R; ++r) hs[j * MLP_R + r] = fmaxf(acc[r], 0.f);
  __syncthreads();
  mlp_layer<float>(hs, w2t, b2, 256, 256, j, acc);
  __syncthreads();                                       // xs is reused for the second hidden layer
#pragma unroll
  for (int r = 0; r < MLP_R; ++r) xs[j * MLP_R + r] = fmaxf(acc[r], 0.f);
  __syncthreads();
  if (j < 4 * MLP_R) {                                   // 8 rows x 4 outputs: one (row, coordinate) per thread
    const int r = j >> 2, c = j & 3;
    float a = b3[c];
    for (int k = 0; k < 256; ++k) a = fmaf(w3t[k * 4 + c], xs[k * MLP_R + r], a);
    if (row0 + r < n) {
      const float rv = fminf(fmaxf(ref[(row0 + r) * 4 + c], 0.f), 1.f);
      const float v = a + logf(fmaxf(rv, eps) / fmaxf(1.f - rv, eps));
      out[(row0 + r) * 4 + c] = 1.f / (1.f + expf(-v));
    }
  }
}

}  // namespace hipie

extern "C" int hipie_ref_point_mlp(const float* ref, const float* dim_t, const void* w1t, const void* b1, const void* w2t,
                                   const void* b2, void* out, int64_t n, int ref_stride, float scale, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(ref && dim_t && w1t && b1 && w2t && b2 && out, "ref_point_mlp: null pointer");
  HIPIE_REQUIRE(n >= 0 && ref_stride >= 4, "ref_point_mlp: bad shape");
  if (n == 0) return HIPIE_OK;
  const unsigned grid = (unsigned)((n + MLP_R - 1) / MLP_R);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HIPIE_F32: hipLaunchKernelGGL((ref_point_mlp_kernel<float>), dim3(grid), dim3(256), 0, st, ref, dim_t, (const float*)w1t, (const float*)b1, (const float*)w2t, (const float*)b2, (float*)out, (long)n, ref_stride, scale); break;
    case HIPIE_F16: hipLaunchKernelGGL((ref_point_mlp_kernel<f16_t>), dim3(grid), dim3(256), 0, st, ref, dim_t, (const f16_t*)w1t, (const f16_t*)b1, (const f16_t*)w2t, (const f16_t*)b2, (f16_t*)out, (long)n, ref_stride, scale); break;
    case HIPIE_BF16: hipLaunchKernelGGL((ref_point_mlp_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, ref, dim_t, (const bf16_t*)w1t, (const bf16_t*)b1, (const bf16_t*)w2t, (const bf16_t*)b2, (bf16_t*)out, (long)n, ref_stride, scale); break;
    default: return set_err(HIPIE_EINVAL, "ref_point_mlp: bad dtype %d", dtype);
  }
  return check_launch("ref_point_mlp");
}

extern "C" int hipie_box_head(const float* x, const float* ref, const float* w1t, const float* b1, const float* w2t, const float* b2,
                              const float* w3t, const float* b3, float* out, int64_t n, float eps, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(x && ref && w1t && b1 && w2t && b2 && w3t && b3 && out, "box_head: null pointer");
  HIPIE_REQUIRE(n >= 0, "box_head: bad shape");
  if (n == 0) return HIPIE_OK;
  hipLaunchKernelGGL(box_head_kernel, dim3((unsigned)((n + MLP_R - 1) / MLP_R)), dim3(256), 0, (hipStream_t)stream, x, ref, w1t, b1, w2t,
                     b2, w3t, b3, out, (long)n, eps);
  return check_launch("box_head");
}

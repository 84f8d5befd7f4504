// msda_bwd.hip -- backward of multi-scale deformable attention: the second entry point of the reference's operator plugin
// (ms_deform_attn_backward, ops/src/vision.cpp:13-16 -> ms_deform_attn_cuda_backward, ops/src/cuda/ms_deform_attn_cuda.cu:83-153 ->
// ms_deformable_col2im_gpu_kernel_*, ops/src/cuda/ms_deform_im2col_cuda.cuh:301-1320) -- SURVEY row f-4's first item.  Inference never
// calls it; it completes the drop-in boundary b2 (the reference's own ops/test.py gradient-checks the op in double for
// D = 30, 32, 64, 71, 1025, 2048, 3096).
//
//   out[b,q,m,c]        = sum_{l,p} A[b,q,m,l,p] * bilinear(value_l[b,:,m,c], loc[b,q,m,l,p])           (forward, msda.hip)
//   grad_value[corner]  += w_corner * A * g            g = grad_out[b,q,m,c]
//   grad_A[b,q,m,l,p]    = sum_c g * bilinear
//   grad_loc[...,x]      = W_l * A * sum_c g * d bilinear / d w_im,      [...,y] = H_l * A * sum_c g * d bilinear / d h_im
// (zero padding outside the map, align_corners = False: w_im = x * W - 0.5 -- ms_deform_attn_col2im_bilinear, cuh:76-160).
//
// Decomposition for gfx950: an ITEM is one (b, q, head) and is owned by LPI = 8 | 16 | 32 | 64 adjacent lanes of a wavefront (the
// smallest power of two >= D, so the production D = 32 packs two items per wave).  A lane walks channels sub, sub + LPI, ...: the four
// corner reads and the four grad_value atomics of a point are contiguous along D (value is (B, S, M, D)) -- one 128-byte segment per
// corner at D = 32.  The channel sums of grad_A / grad_loc are xor-shuffle reductions inside the item's lanes: every (b,q,m,l,p) has
// exactly one owner, so those two outputs are plain stores (the reference needs shared-memory trees or atomics depending on D) and
// only grad_value -- where different queries meet on a pixel -- uses fp32 / fp64 global atomics (no return value: L2 atomics).
// Consequently grad_A and grad_loc are deterministic; grad_value sums in arrival order, as the reference's does.
#include "common.h"

namespace hipie {

template <typename T>
__device__ __forceinline__ T bw_shfl_xor(T v, int mask) { return __shfl_xor(v, mask); }

template <typename T, int LPI>
__global__ __launch_bounds__(256) void msda_bwd_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart, const T* __restrict__ loc,
                                                       const T* __restrict__ attn, const T* __restrict__ gout, T* __restrict__ gvalue,
                                                       T* __restrict__ gloc, T* __restrict__ gattn, int S, int M, int D, int L, int Lq,
                                                       int P, long items) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int sub = (int)(gid % LPI);
  long item = gid / LPI;
  const bool live = item < items;                 // dead items keep their lanes in the shuffles, read item 0, write nothing
  if (!live) item = 0;
  const int m = (int)(item % M);
  const long bq = item / M;
  const int b = (int)(bq / Lq);
  const long row = (long)M * D;
  const T* vb = value + (long)b * S * row + (long)m * D;
  T* gvb = gvalue + (long)b * S * row + (long)m * D;
  const T* go = gout + item * D;
  const T* lp = loc + item * (long)L * P * 2;
  const T* wp = attn + item * (long)L * P;
  T* glp = gloc + item * (long)L * P * 2;
  T* gwp = gattn + item * (long)L * P;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long lbase = (long)lstart[l] * row;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      const T a = wp[i];
      const T h_im = lp[2 * i + 1] * (T)H - (T)0.5, w_im = lp[2 * i] * (T)W - (T)0.5;
      T sa = 0, sx = 0, sy = 0;
      if (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)H && w_im < (T)W) {
        const int h0 = (int)floor(h_im), w0 = (int)floor(w_im), h1 = h0 + 1, w1 = w0 + 1;
        const T lh = h_im - (T)h0, lw = w_im - (T)w0, hh = (T)1 - lh, hw = (T)1 - lw;
        const bool ok1 = h0 >= 0 && w0 >= 0, ok2 = h0 >= 0 && w1 <= W - 1, ok3 = h1 <= H - 1 && w0 >= 0, ok4 = h1 <= H - 1 && w1 <= W - 1;
        const long o1 = lbase + ((long)h0 * W + w0) * row, o2 = lbase + ((long)h0 * W + w1) * row;
        const long o3 = lbase + ((long)h1 * W + w0) * row, o4 = lbase + ((long)h1 * W + w1) * row;
        const T c1 = hh * hw, c2 = hh * lw, c3 = lh * hw, c4 = lh * lw;
        for (int c = sub; c < D; c += LPI) {
          const T g = go[c];
          const T tg = g * a;
          const T v1 = ok1 ? vb[o1 + c] : (T)0, v2 = ok2 ? vb[o2 + c] : (T)0, v3 = ok3 ? vb[o3 + c] : (T)0, v4 = ok4 ? vb[o4 + c] : (T)0;
          if (live) {
            if (ok1) unsafeAtomicAdd(gvb + o1 + c, c1 * tg);         // hardware L2 atomic (global_atomic_add_f32 / _f64), not a CAS loop
            if (ok2) unsafeAtomicAdd(gvb + o2 + c, c2 * tg);
            if (ok3) unsafeAtomicAdd(gvb + o3 + c, c3 * tg);
            if (ok4) unsafeAtomicAdd(gvb + o4 + c, c4 * tg);
          }
          sa += g * (c1 * v1 + c2 * v2 + c3 * v3 + c4 * v4);
          sx += tg * (hh * (v2 - v1) + lh * (v4 - v3));          // d/d w_im
          sy += tg * (hw * (v3 - v1) + lw * (v4 - v2));          // d/d h_im
        }
      }
#pragma unroll
      for (int s = LPI / 2; s > 0; s >>= 1) {
        sa += bw_shfl_xor(sa, s);
        sx += bw_shfl_xor(sx, s);
        sy += bw_shfl_xor(sy, s);
      }
      if (live && sub == 0) {
        gwp[i] = sa;
        glp[2 * i] = (T)W * sx;
        glp[2 * i + 1] = (T)H * sy;
      }
    }
  }
}

template <typename T>
static int launch_msda_bwd(const void* value, const int64_t* shapes, const int64_t* lstart, const void* loc, const void* attn,
                           const void* gout, void* gvalue, void* gloc, void* gattn, int B, int S, int M, int D, int L, int Lq, int P,
                           hipStream_t st) {
  const long items = (long)B * Lq * M;
  if (hipMemsetAsync(gvalue, 0, (size_t)B * S * M * D * sizeof(T), st) != hipSuccess) return set_err(HIPIE_ELAUNCH, "msda_backward: memset failed");
  if (items == 0) return HIPIE_OK;
  const int lpi = D <= 8 ? 8 : D <= 16 ? 16 : D <= 32 ? 32 : 64;
  const long threads = items * lpi;
  const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
#define HIPIE_BWD(LPI_)                                                                                                                 \
  hipLaunchKernelGGL((msda_bwd_kernel<T, LPI_>), grid, block, 0, st, (const T*)value, shapes, lstart, (const T*)loc, (const T*)attn,    \
                     (const T*)gout, (T*)gvalue, (T*)gloc, (T*)gattn, S, M, D, L, Lq, P, items)
  switch (lpi) {
    case 8: HIPIE_BWD(8); break;
    case 16: HIPIE_BWD(16); break;
    case 32: HIPIE_BWD(32); break;
    default: HIPIE_BWD(64); break;
  }
#undef HIPIE_BWD
  return check_launch("msda_backward");
}

}  // namespace hipie

extern "C" int hipie_msda_backward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const void* sampling_loc,
                                   const void* attn_weight, const void* grad_output, void* grad_value, void* grad_sampling_loc,
                                   void* grad_attn_weight, int B, int S, int M, int D, int L, int Lq, int P, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(B >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda_backward: bad shape");
  HIPIE_REQUIRE(dtype == HIPIE_F32 || dtype == HIPIE_F64, "msda_backward: dtype %d (HIPIE_F32 | HIPIE_F64, as the reference dispatches)", dtype);
  if (B == 0) return HIPIE_OK;
  HIPIE_REQUIRE(value && spatial_shapes && level_start && grad_value, "msda_backward: null pointer");
  HIPIE_REQUIRE(Lq == 0 || (sampling_loc && attn_weight && grad_output && grad_sampling_loc && grad_attn_weight), "msda_backward: null pointer");
  HIPIE_REQUIRE((long)B * Lq * M * 64 < (1L << 40), "msda_backward: too many (batch, query, head) items");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == HIPIE_F64)
    return launch_msda_bwd<double>(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                                   grad_attn_weight, B, S, M, D, L, Lq, P, st);
  return launch_msda_bwd<float>(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                                grad_attn_weight, B, S, M, D, L, Lq, P, st);
}

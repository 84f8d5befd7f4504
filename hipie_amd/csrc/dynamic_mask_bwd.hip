// dynamic_mask_bwd.hip -- backward of the fused CondInst dynamic mask head (forward: dynamic_mask.hip, row a19).
//
// Reference: autograd through DDETRSegmUniDN.dynamic_mask_with_coords (models/ddetrs_dn.py:1411-1502: relative coordinates + the grouped
// 1x1 convs of mask_heads_forward :1390-1408) and aligned_bilinear (:1832-1854) -- in training the masks of the matched instances are
// produced by this head and the dice / focal losses differentiate through it into the mask features, the controller's 169 parameters
// per instance and the reference points.
//
//   per instance n of image b and low-resolution pixel (r, c):
//     x  = [rx - (stride c + stride/2), ry - (stride r + stride/2), feats[b, 0..7, r, c]]
//     h1 = relu(W1 x + b1), h2 = relu(W2 h1 + b2), y = W3 h2 + b3, out = aligned_bilinear(y, up)
//   gy  = aligned_bilinear^T(grad_out)         (factor 2: out[0] = in[0], out[2m+1] = in[m], out[2m] = (in[m-1] + in[m]) / 2 per axis,
//                                               so gy[m] = g[2m+1] + (m ? 1/2 : 1) g[2m] + [m + 1 < H] 1/2 g[2m+2], separably)
//   gW3 = gy h2, gb3 = gy;   gh2 = gy W3 [h2 > 0];   gW2 = gh2 h1^T, gb2 = gh2;   gh1 = W2^T gh2 [h1 > 0];   gW1 = gh1 x^T, gb1 = gh1;
//   gx  = W1^T gh1:  grad_refs[n] = sum_pixels gx[0..1],  grad_feats[b, :, r, c] = sum_n gx[2..9]
//
// Decomposition: a workgroup owns 256 low-resolution pixels of one image and a contiguous chunk of its instances; a thread owns ONE
// pixel for the whole chunk.  The pixel's 8 features are read once; an instance's 169 parameters and reference point are wave-uniform
// (scalar loads), as in the forward.  Per instance a thread's 171 per-pixel terms (169 parameter gradients + the two reference-point
// terms) are reduced over the wave by a halving exchange -- at every step a lane keeps one half of its values and sends the other to its
// partner, 96 + 48 + 24 + 12 + 6 + 3 = 189 ds_bpermute instead of 171 x 6 -- after which lane L holds the sums of parameters
// 3 bitrev-weighted indices apart and adds them to grad_params / grad_refs with global atomics (HW / 64 addends per element).
// grad_feats needs no cross-lane traffic: 8 running sums per thread over the chunk, one atomic per (channel, pixel, chunk) at the end.
#include "common.h"

namespace hipie {

constexpr int DMB_SLOTS = 192;          // 171 values padded to 3 x 64

__device__ __forceinline__ float dmb_gy(const float* __restrict__ g, int r, int c, int H, int W, int up) {
  if (up == 1) return g[(long)r * W + c];
  const int OW = 2 * W;
  // taps of the transposed x2 upsampling along one axis: (index, weight) x 3
  const int Y0 = 2 * r, X0 = 2 * c;
  const float wy0 = r ? 0.5f : 1.f, wx0 = c ? 0.5f : 1.f;
  const float wy2 = (r + 1 < H) ? 0.5f : 0.f, wx2 = (c + 1 < W) ? 0.5f : 0.f;
  const int Y2 = (r + 1 < H) ? Y0 + 2 : Y0, X2 = (c + 1 < W) ? X0 + 2 : X0;       // clamped: weight 0 there
  const float* r0 = g + (long)Y0 * OW;
  const float* r1 = r0 + OW;
  const float* r2 = g + (long)Y2 * OW;
  const float a0 = wx0 * r0[X0] + r0[X0 + 1] + wx2 * r0[X2];
  const float a1 = wx0 * r1[X0] + r1[X0 + 1] + wx2 * r1[X2];
  const float a2 = wx0 * r2[X0] + r2[X0 + 1] + wx2 * r2[X2];
  return wy0 * a0 + a1 + wy2 * a2;
}

__global__ __launch_bounds__(256) void dynamic_mask_bwd_kernel(const float* __restrict__ feats, const float* __restrict__ refs,
                                                               const float* __restrict__ params, const float* __restrict__ gout,
                                                               float* __restrict__ gfeats, float* __restrict__ grefs,
                                                               float* __restrict__ gparams, int Q, int H, int W, int stride, int up,
                                                               int nchunk) {
  const int HW = H * W;
  const int b = blockIdx.y / nchunk, ch = blockIdx.y - b * nchunk;
  const int per = (Q + nchunk - 1) / nchunk;
  const int qa = ch * per, qb = min(Q, qa + per);
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const bool live = pix < HW;
  const int pc = live ? pix : HW - 1;
  const int r = pc / W, c = pc - r * W;
  const int lane = threadIdx.x & 63;
  const float half = (float)(stride / 2);
  const float px = (float)(stride * c) + half, py = (float)(stride * r) + half;
  float f[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) f[k] = feats[((long)b * 8 + k) * HW + pc];
  float gf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long osz = (long)(up * up) * HW;
  for (int q = qa; q < qb; ++q) {
    const long n = (long)b * Q + q;
    const float* __restrict__ p = params + n * 169;
    float x[10];
    x[0] = refs[2 * n] - px;
    x[1] = refs[2 * n + 1] - py;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[2 + k] = f[k];
    // ---- forward, recomputed ----
    float h1[8], h2[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float a = p[152 + o];
#pragma unroll
      for (int i = 0; i < 10; ++i) a = fmaf(p[o * 10 + i], x[i], a);
      h1[o] = fmaxf(a, 0.f);
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      float a = p[160 + o];
#pragma unroll
      for (int i = 0; i < 8; ++i) a = fmaf(p[80 + o * 8 + i], h1[i], a);
      h2[o] = fmaxf(a, 0.f);
    }
    // ---- backward through the three layers ----
    const float gy = live ? dmb_gy(gout + n * osz, r, c, H, W, up) : 0.f;
    float v[DMB_SLOTS];
    float gh2[8], gh1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[144 + i] = gy * h2[i];
      gh2[i] = h2[i] > 0.f ? gy * p[144 + i] : 0.f;
      v[160 + i] = gh2[i];
    }
    v[168] = gy;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float a = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        v[80 + o * 8 + i] = gh2[o] * h1[i];
        a = fmaf(p[80 + o * 8 + i], gh2[o], a);
      }
      gh1[i] = h1[i] > 0.f ? a : 0.f;
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      v[152 + o] = gh1[o];
#pragma unroll
      for (int i = 0; i < 10; ++i) v[o * 10 + i] = gh1[o] * x[i];
    }
    float gx[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      float a = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) a = fmaf(p[o * 10 + i], gh1[o], a);
      gx[i] = a;
    }
    v[169] = gx[0];
    v[170] = gx[1];
#pragma unroll
    for (int i = 171; i < DMB_SLOTS; ++i) v[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) gf[k] += gx[2 + k];
    // ---- wave reduction by halving exchange: after the step with mask m a lane holds cnt / 2 partial sums ----
    int base = 0;
#pragma unroll
    for (int step = 0; step < 6; ++step) {
      const int m = 32 >> step;
      const int cnt = DMB_SLOTS >> step;            // values held before this step
      const bool up_half = (lane & m) != 0;
#pragma unroll
      for (int i = 0; i < cnt / 2; ++i) {
        const float lo = v[i], hi = v[i + cnt / 2];
        const float send = up_half ? lo : hi;
        const float keep = up_half ? hi : lo;
        v[i] = keep + __shfl_xor(send, m);
      }
      if (up_half) base += cnt / 2;
    }
    // lane now owns slots base .. base + 2
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int j = base + i;
      if (j < 169) unsafeAtomicAdd(gparams + n * 169 + j, v[i]);
      else if (j < 171) unsafeAtomicAdd(grefs + 2 * n + (j - 169), v[i]);
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (nchunk == 1) gfeats[((long)b * 8 + k) * HW + pix] = gf[k];
      else unsafeAtomicAdd(gfeats + ((long)b * 8 + k) * HW + pix, gf[k]);
    }
  }
}

}  // namespace hipie

extern "C" int hipie_dynamic_mask_backward(const float* feats, const float* refs, const float* params, const float* grad_out,
                                           float* grad_feats, float* grad_refs, float* grad_params, int B, int Q, int H, int W,
                                           int stride, int up, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(B >= 0 && Q >= 0 && H > 0 && W > 0 && stride > 0, "dynamic_mask_backward: bad shape");
  HIPIE_REQUIRE(up == 1 || up == 2, "dynamic_mask_backward: up = %d (1 | 2)", up);
  HIPIE_REQUIRE(grad_feats && grad_refs && grad_params, "dynamic_mask_backward: null output pointer");
  hipStream_t st = (hipStream_t)stream;
  const long HW = (long)H * W;
  if (B == 0) return HIPIE_OK;
  if (hipMemsetAsync(grad_feats, 0, (size_t)B * 8 * HW * sizeof(float), st) != hipSuccess) return set_err(HIPIE_ELAUNCH, "dynamic_mask_backward: memset failed");
  if (Q == 0) return HIPIE_OK;
  HIPIE_REQUIRE(feats && refs && params && grad_out, "dynamic_mask_backward: null pointer");
  if (hipMemsetAsync(grad_refs, 0, (size_t)B * Q * 2 * sizeof(float), st) != hipSuccess ||
      hipMemsetAsync(grad_params, 0, (size_t)B * Q * 169 * sizeof(float), st) != hipSuccess)
    return set_err(HIPIE_ELAUNCH, "dynamic_mask_backward: memset failed");
  const int tiles = (int)((HW + 255) / 256);
  // instance chunks: enough workgroups for four per CU, never more chunks than instances
  int nchunk = (int)((1024 + (long)B * tiles - 1) / ((long)B * tiles));
  nchunk = std::max(1, std::min(nchunk, Q));
  HIPIE_REQUIRE((long)B * nchunk <= 65535, "dynamic_mask_backward: %d images x %d instance chunks", B, nchunk);
  hipLaunchKernelGGL(dynamic_mask_bwd_kernel, dim3((unsigned)tiles, (unsigned)(B * nchunk)), dim3(256), 0, st, feats, refs, params,
                     grad_out, grad_feats, grad_refs, grad_params, Q, H, W, stride, up, nchunk);
  return check_launch("dynamic_mask_backward");
}

// dynamic_mask.hip -- fused CondInst dynamic mask head.
//
// Reference (models/ddetrs_dn.py:1411-1502): builds a (1, N*10, H, W) tensor of [rel_x, rel_y, 8 feature channels] per
// instance (596 MB / image at N = 910, H = W = 128), runs three grouped 1x1 convs with groups = N, then
// aligned_bilinear x2.  Here one workgroup owns (instance n, 16 low-resolution rows):
//   phase 1  each thread evaluates the 10 -> 8 -> 8 -> 1 MLP for 4 pixels at a time.  The instance's 169 parameters and its
//            reference point are wave-uniform (indexed by blockIdx), so the compiler keeps them in SGPRs / the scalar
//            cache and every FMA takes its weight as a scalar operand; the 8 feature channels of a pixel are read
//            coalesced along W from the (B,8,H,W) map, which stays L2-resident across the 910 instances of an image.
//            Logits of the 16 rows + 1 halo row above go to LDS.
//   phase 2  aligned_bilinear(x, 2) (ddetrs_dn.py:1832-1854) reduces, for factor 2, to
//                out[0] = in[0];  out[2m+1] = in[m];  out[2m] = (in[m-1] + in[m]) / 2   (m >= 1)
//            per axis (replicate pad + align_corners=True bilinear to 2h+1 + front pad 1 + crop).  Each thread emits 4
//            consecutive output pixels (one 16-byte store) from the LDS tile -> the 238 MB/image f32 output is written
//            once, fully coalesced.  Nothing else touches HBM.
// Roofline: HBM-write bound (out = N * 4*H*W * 4 B) with 4.5 GFLOP / image of f32 VALU work alongside.
#include "common.h"

namespace hipie {

constexpr int DM_TH = 16;       // low-res rows per workgroup
constexpr int DM_PB = 4;        // pixels per thread per iteration
constexpr int DM_MAXW = 512;    // LDS tile width bound

template <typename OutT, int UP>
__global__ __launch_bounds__(256) void dynamic_mask_kernel(const float* __restrict__ feats, const float* __restrict__ refs,
                                                           const float* __restrict__ params, OutT* __restrict__ out,
                                                           int Q, int H, int W, int stride) {
  extern __shared__ __attribute__((aligned(16))) float ylo[];   // [(DM_TH + 1)][W]
  const int n = blockIdx.y;
  const int b = n / Q;
  const int r0 = blockIdx.x * DM_TH;             // first low-res row of this tile
  const int rows = min(DM_TH, H - r0);
  const float* __restrict__ p = params + (long)n * 169;
  const float rx = refs[2 * (long)n], ry = refs[2 * (long)n + 1];
  const float* fb = feats + (long)b * 8 * H * W;
  const int HW = H * W;
  const float half = (float)(stride / 2);

  // ---- phase 1: low-res logits for rows r0-1 .. r0+rows-1 (row r0-1 clamped to 0) ----
  const int npx = (rows + 1) * W;
  for (int base = 0; base < npx; base += 256 * DM_PB) {
    float x[DM_PB][10];
    int pix[DM_PB];
#pragma unroll
    for (int k = 0; k < DM_PB; ++k) {
      const int i = min(base + k * 256 + (int)threadIdx.x, npx - 1);
      pix[k] = i;
      const int lr = i / W, col = i - lr * W;
      const int row = max(r0 - 1 + lr, 0);
      x[k][0] = rx - ((float)(stride * col) + half);
      x[k][1] = ry - ((float)(stride * row) + half);
#pragma unroll
      for (int c = 0; c < 8; ++c) x[k][2 + c] = fb[(long)c * HW + row * W + col];
    }
    float h1[DM_PB][8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
      for (int k = 0; k < DM_PB; ++k) {
        float a = p[152 + o];
#pragma unroll
        for (int i = 0; i < 10; ++i) a = fmaf(p[o * 10 + i], x[k][i], a);
        h1[k][o] = fmaxf(a, 0.f);
      }
    }
    float h2[DM_PB][8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
#pragma unroll
      for (int k = 0; k < DM_PB; ++k) {
        float a = p[160 + o];
#pragma unroll
        for (int i = 0; i < 8; ++i) a = fmaf(p[80 + o * 8 + i], h1[k][i], a);
        h2[k][o] = fmaxf(a, 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < DM_PB; ++k) {
      float a = p[168];
#pragma unroll
      for (int i = 0; i < 8; ++i) a = fmaf(p[144 + i], h2[k][i], a);
      if (base + k * 256 + (int)threadIdx.x < npx) ylo[pix[k]] = a;
    }
  }
  __syncthreads();

  // ---- phase 2: upsample and store ----
  if (UP == 1) {
    OutT* ob = out + (long)n * HW + (long)r0 * W;
    for (int i = threadIdx.x; i < rows * W; i += 256) ob[i] = elem<OutT>::from_f32(ylo[W + i]);
    return;
  }
  const int OW = 2 * W;
  OutT* ob = out + (long)n * 4 * HW;
  const int nout = 2 * rows * (OW / 4);           // groups of 4 output pixels
  for (int i = threadIdx.x; i < nout; i += 256) {
    const int oy = i / (OW / 4), x4 = (i - oy * (OW / 4)) * 4;
    const int Y = 2 * r0 + oy;
    // vertical taps: rows (ra, rb) with weight 0.5 each, or a single row
    int ma, mb;
    if (Y & 1) { ma = mb = (Y - 1) >> 1; }
    else if (Y == 0) { ma = mb = 0; }
    else { ma = (Y >> 1) - 1; mb = Y >> 1; }
    const float* la = ylo + (ma - (r0 - 1)) * W;   // LDS row index = low-res row - (r0 - 1); row r0-1 of tile 0 holds row 0
    const float* lb = ylo + (mb - (r0 - 1)) * W;
    float o[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int X = x4 + t;
      int ca, cb;
      if (X & 1) { ca = cb = (X - 1) >> 1; }
      else if (X == 0) { ca = cb = 0; }
      else { ca = (X >> 1) - 1; cb = X >> 1; }
      const float top = 0.5f * (la[ca] + la[cb]);
      const float bot = 0.5f * (lb[ca] + lb[cb]);
      o[t] = 0.5f * (top + bot);
    }
    OutT* dst = ob + (long)Y * OW + x4;
    if (sizeof(OutT) == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) dst[t] = elem<OutT>::from_f32(o[t]);
    }
  }
}

template <typename OutT>
static int launch_dm(const float* feats, const float* refs, const float* params, void* out, int B, int Q, int H, int W,
                     int stride, int up, hipStream_t st) {
  dim3 grid((H + DM_TH - 1) / DM_TH, B * Q);
  const size_t lds = (size_t)(DM_TH + 1) * W * sizeof(float);
  if (up == 1)
    hipLaunchKernelGGL((dynamic_mask_kernel<OutT, 1>), grid, dim3(256), lds, st, feats, refs, params, (OutT*)out, Q, H, W, stride);
  else
    hipLaunchKernelGGL((dynamic_mask_kernel<OutT, 2>), grid, dim3(256), lds, st, feats, refs, params, (OutT*)out, Q, H, W, stride);
  return check_launch("dynamic_mask");
}

}  // namespace hipie

extern "C" int hipie_dynamic_mask(const float* feats, const float* refs, const float* params, void* out, int B, int Q,
                                  int H, int W, int stride, int up, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(feats && refs && params && out, "dynamic_mask: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q >= 0 && H > 0 && W > 0 && stride > 0, "dynamic_mask: bad shape");
  HIPIE_REQUIRE(up == 1 || up == 2, "dynamic_mask: up=%d unsupported (mask feature stride 8, MASK_STRIDE 8 or 4)", up);
  HIPIE_REQUIRE(W <= DM_MAXW && W % 2 == 0, "dynamic_mask: W=%d must be even and <= %d", W, DM_MAXW);
  HIPIE_REQUIRE((long)B * Q < 65536, "dynamic_mask: too many instances for one launch");
  if (B * Q == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case HIPIE_F32: return launch_dm<float>(feats, refs, params, out, B, Q, H, W, stride, up, st);
    case HIPIE_F16: return launch_dm<f16_t>(feats, refs, params, out, B, Q, H, W, stride, up, st);
    case HIPIE_BF16: return launch_dm<bf16_t>(feats, refs, params, out, B, Q, H, W, stride, up, st);
    default: return set_err(HIPIE_EINVAL, "dynamic_mask: bad out_dtype %d", out_dtype);
  }
}

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
#include <cstdlib>

#include "mfma.h"

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


// ---------------------------------------------------------------------------------------------------------------------------
// The same head with the three 1x1 layers on the matrix pipe (16-bit operands, fp32 accumulate) -- the 16-bit policies.
//
// The fp32 kernel above is VALU-bound (185 FMAs per low-resolution pixel and instance: 1.5 ms for 8 x 910 instances, 8 % of the
// HBM write roof).  Here FOUR instances of one image share a 32-row MFMA tile (4 x 8 channels) over 32 pixels:
//   layer 1  h1 = A1 . X + c1       X (16 x 32 px) = [-px, -px, -py, -py, f0 .. f7, 0 ..] is the SAME for every instance:
//            rel = ref - pixel, so W . rel = (W . ref) - W . pixel and the per-instance part moves into the fp32 accumulator
//            init c1 = b1 + wx * rx + wy * ry.  wx, wy are split hi + lo (two k slots each) because they multiply pixel
//            coordinates of ~1000; -px, -py are exact in 16 bit (multiples of 4 up to 8192).  k = 12 of 16 used.
//   layer 2  h2 = A2 . relu(h1) + c2   A2 = block-diagonal 4 x (8 x 8); the C layout of h1 (lane = pixel, 16 rows) IS a valid
//            B operand for two k steps after relu + pack, with the k order (row permutation) folded into A2's columns.
//   layer 3  y = A3 . relu(h2)         rows 0..3 = the four instances' 1 x 8 weights, again block-structured.
// 5 MFMAs per (4 instances x 32 pixels); no cross-lane traffic between the layers.
// One WAVE owns (4 instances, R low-res rows): it walks the rows, keeps the logits of the last two rows in a private LDS
// ring and emits the two output rows of aligned_bilinear(x2) per step with 16-byte stores.  No workgroup barrier.

// SPLIT (round 4; T = fp16): the fp32-class form for the split policy.  Every weight and every activation is an fp16 PAIR x = hi + lo and
// each product is the three-term sum  W_hi.x_hi + W_lo.x_hi + W_hi.x_lo  (fp32 accumulation; the coordinate columns are exact pairs
// already): 15 MFMAs per tile instead of 5, ReLU in fp32 before the split.  Retires the fp32 VALU kernel (1.57 ms per step) in that policy.
template <typename T, typename OutT, int R, int TU, bool SPLIT>
__global__ __launch_bounds__(256) void dynamic_mask_mfma_kernel(const float* __restrict__ feats, const float* __restrict__ refs,
                                                                const float* __restrict__ params, OutT* __restrict__ out,
                                                                int Q, int H, int W, int stride) {
  typedef Mfma32<T> M;
  typedef typename M::frag frag;
  extern __shared__ __attribute__((aligned(16))) float dm_ring[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, g = lane >> 5;
  const int GQ = (Q + 3) >> 2;
  const int b = blockIdx.y / GQ, q0 = (blockIdx.y - b * GQ) * 4;
  const int r0 = (blockIdx.x * 4 + wave) * R;
  if (r0 >= H) return;                                   // whole wave; there is no barrier below
  const int rows = min(R, H - r0);
  const int WP = W + 8;                                  // LDS row pitch (floats): index c + 1 holds column c, index 0 = column 0 again
  float* ring = dm_ring + wave * (2 * 4 * WP);           // [slot][instance][WP]
  const long ibase = (long)b * Q;
  const int HW = H * W;
  const float half = (float)(stride / 2);

  // ---- operands that depend on the instances only ----
  const int myinst = n >> 3, o = n & 7;
  const float* pm = params + (ibase + min(q0 + myinst, Q - 1)) * 169;
  const float* p3 = params + (ibase + min(q0 + (n & 3), Q - 1)) * 169;
  frag A1, A2[2], A3[2];
  frag A1l, A2l[2], A3l[2];                              // SPLIT: the lo halves of the weights (coordinate columns: none)
  auto lo_of = [](const float x) -> T { return (T)(x - (float)(T)x); };
  {
    float v[8];
    if (g == 0) {
      const float wx = pm[o * 10], wy = pm[o * 10 + 1];
      const T xh = (T)wx, yh = (T)wy;
      v[0] = (float)xh; v[1] = wx - (float)xh; v[2] = (float)yh; v[3] = wy - (float)yh;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[4 + c] = pm[o * 10 + 2 + c];
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) { v[c] = 0.f; v[4 + c] = pm[o * 10 + 6 + c]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      A1[j] = (T)v[j];
      A1l[j] = (SPLIT && j >= 4) ? lo_of(v[j]) : (T)0.f;
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int inst2 = 2 * s + (j >> 2), ch = 4 * g + (j & 3);
      const float w2 = inst2 == myinst ? pm[80 + o * 8 + ch] : 0.f;
      const float w3 = (n < 4 && inst2 == n) ? p3[144 + ch] : 0.f;
      A2[s][j] = (T)w2;
      A3[s][j] = (T)w3;
      A2l[s][j] = SPLIT ? lo_of(w2) : (T)0.f;
      A3l[s][j] = SPLIT ? lo_of(w3) : (T)0.f;
    }
  }
  f32x16 c1, c2, c3;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int inst = r >> 2, ch = 4 * g + (r & 3);
    const long iq = ibase + min(q0 + inst, Q - 1);
    const float* pi = params + iq * 169;
    const float rx = refs[2 * iq], ry = refs[2 * iq + 1];
    c1[r] = fmaf(pi[ch * 10 + 1], ry, fmaf(pi[ch * 10], rx, pi[152 + ch]));
    c2[r] = pi[160 + ch];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)                          // rows 0..3 of the last product = the four instances: their output bias
    c3[r] = (r < 4 && g == 0) ? params[(ibase + min(q0 + r, Q - 1)) * 169 + 168] : 0.f;

  const float* fb = feats + (long)b * 8 * HW + (long)(4 * g) * HW;      // this lane half's four feature planes
  const int NT = (W + 31) >> 5;
  const int CG = W >> 2;
  const int OW = 2 * W;

  // output addressing of the upsampling step: lane -> (instance g, g + 2; column group n, n + 32, ..)
  OutT* obase = out + ((ibase + q0 + g) * (long)(2 * H)) * OW + 8 * n;
  const long ostep = 2L * (2 * H) * OW;                   // two instances further
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  const s16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  for (int rr = -1; rr < rows; ++rr) {
    const int row = max(r0 + rr, 0);                     // rr = -1: the halo row above the strip (row 0 for the first strip)
    float* cur = ring + ((rr + 1) & 1) * 4 * WP;
    const float npy = g == 0 ? -((float)(stride * row) + half) : 0.f;
    // ---- low-resolution logits of this row, TU 32-pixel tiles in flight ----
    for (int t = 0; t < NT; t += TU) {
      frag X[TU], Xl[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int col = min((t + u) * 32 + n, W - 1);
        const float* fp = fb + row * W + col;
        const float f0 = fp[0], f1 = fp[HW], f2 = fp[2 * HW], f3 = fp[3 * HW];
        const float npx = g == 0 ? -((float)(stride * col) + half) : 0.f;
        X[u][0] = (T)npx; X[u][1] = (T)npx; X[u][2] = (T)npy; X[u][3] = (T)npy;      // zero in the upper lane half
        X[u][4] = (T)f0; X[u][5] = (T)f1; X[u][6] = (T)f2; X[u][7] = (T)f3;
        if (SPLIT) {
          Xl[u][0] = (T)0.f; Xl[u][1] = (T)0.f; Xl[u][2] = (T)0.f; Xl[u][3] = (T)0.f;
          Xl[u][4] = lo_of(f0); Xl[u][5] = lo_of(f1); Xl[u][6] = lo_of(f2); Xl[u][7] = lo_of(f3);
        }
      }
      f32x16 h[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        h[u] = M::mma(A1, X[u], c1);
        if (SPLIT) h[u] = M::mma(A1, Xl[u], M::mma(A1l, X[u], h[u]));
      }
      // ReLU after the rounding (the same value: rounding is monotonic and keeps 0): one packed integer max per two
      // activations -- a negative 16-bit float is a negative int16
      frag Hb[TU][2], Hl[TU][2];
      auto activate = [&]() {
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            f32x8 w8;
#pragma unroll
            for (int j = 0; j < 8; ++j) w8[j] = SPLIT ? fmaxf(h[u][8 * s + j], 0.f) : h[u][8 * s + j];
            Hb[u][s] = __builtin_convertvector(w8, frag);          // packed conversions (v_cvt_pk_*), round to nearest even
            if (SPLIT) {
              f32x8 r8;
#pragma unroll
              for (int j = 0; j < 8; ++j) r8[j] = w8[j] - (float)Hb[u][s][j];
              Hl[u][s] = __builtin_convertvector(r8, frag);
            } else {
              Hb[u][s] = __builtin_bit_cast(frag, __builtin_elementwise_max(__builtin_bit_cast(s16x8, Hb[u][s]), zero8));
            }
          }
      };
      activate();
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        h[u] = M::mma(A2[1], Hb[u][1], M::mma(A2[0], Hb[u][0], c2));
        if (SPLIT) h[u] = M::mma(A2[1], Hl[u][1], M::mma(A2[0], Hl[u][0], M::mma(A2l[1], Hb[u][1], M::mma(A2l[0], Hb[u][0], h[u]))));
      }
      activate();
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        h[u] = M::mma(A3[1], Hb[u][1], M::mma(A3[0], Hb[u][0], c3));
        if (SPLIT) h[u] = M::mma(A3[1], Hl[u][1], M::mma(A3[0], Hl[u][0], M::mma(A3l[1], Hb[u][1], M::mma(A3l[0], Hb[u][0], h[u]))));
      }
#pragma unroll
      for (int u = 0; u < TU; ++u) {
        const int col = (t + u) * 32 + n;
        if (g == 0 && col < W) {                         // rows 0..3 of the product = the four instances, pixel = lane
#pragma unroll
          for (int r = 0; r < 4; ++r) cur[r * WP + 1 + col] = h[u][r];
        }
      }
    }
    if (rr < 0) continue;
    // ---- aligned_bilinear x2: output rows 2m (mean of rows m-1, m) and 2m+1 (row m); 8 output columns per lane ----
    const float* prev = ring + (rr & 1) * 4 * WP;
    OutT* orow = obase + (long)(2 * (r0 + rr)) * OW;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      const int inst = g + 2 * i2;
      if (q0 + inst >= Q) continue;
      for (int cg = n; cg < CG; cg += 32) {
        const float* ca = cur + inst * WP + 4 * cg;
        const float* pa = prev + inst * WP + 4 * cg;
        const float4 a4 = *reinterpret_cast<const float4*>(ca);
        const float4 p4 = *reinterpret_cast<const float4*>(pa);
        // columns 4cg-1 .. 4cg+3; column -1 = column 0 (LDS index 0 is never written)
        const float a[5] = {cg == 0 ? a4.y : a4.x, a4.y, a4.z, a4.w, ca[4]};
        const float p[5] = {cg == 0 ? p4.y : p4.x, p4.y, p4.z, p4.w, pa[4]};
        float v[5], top[8], bot[8];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = 0.5f * (p[k] + a[k]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          top[2 * c] = 0.5f * (v[c] + v[c + 1]); top[2 * c + 1] = v[c + 1];
          bot[2 * c] = 0.5f * (a[c] + a[c + 1]); bot[2 * c + 1] = a[c + 1];
        }
        OutT* dst = orow + i2 * ostep + 8 * (cg - n);
        if (sizeof(OutT) == 4) {
          float4* d0 = reinterpret_cast<float4*>(dst);
          float4* d1 = reinterpret_cast<float4*>(dst + OW);
          d0[0] = make_float4(top[0], top[1], top[2], top[3]); d0[1] = make_float4(top[4], top[5], top[6], top[7]);
          d1[0] = make_float4(bot[0], bot[1], bot[2], bot[3]); d1[1] = make_float4(bot[4], bot[5], bot[6], bot[7]);
        } else {
          typedef OutT o8 __attribute__((ext_vector_type(8)));
          o8 t8, b8;
#pragma unroll
          for (int k = 0; k < 8; ++k) { t8[k] = elem<OutT>::from_f32(top[k]); b8[k] = elem<OutT>::from_f32(bot[k]); }
          *reinterpret_cast<o8*>(dst) = t8;
          *reinterpret_cast<o8*>(dst + OW) = b8;
        }
      }
    }
  }
}

template <typename T, typename OutT, int R, int TU, bool SPLIT = false>
static int launch_dm_mfma_v(const float* feats, const float* refs, const float* params, void* out, int B, int Q, int H, int W,
                            int stride, hipStream_t st) {
  const int strips = (H + R - 1) / R;
  dim3 grid((strips + 3) / 4, B * ((Q + 3) / 4));
  const size_t lds = (size_t)4 * 2 * 4 * (W + 8) * sizeof(float);
  hipLaunchKernelGGL((dynamic_mask_mfma_kernel<T, OutT, R, TU, SPLIT>), grid, dim3(256), lds, st, feats, refs, params, (OutT*)out, Q, H, W, stride);
  return check_launch("dynamic_mask16");
}

template <typename T, typename OutT, bool SPLIT = false>
static int launch_dm_mfma(const float* feats, const float* refs, const float* params, void* out, int B, int Q, int H, int W,
                          int stride, hipStream_t st) {
  // measured at B = 8, Q = 910, 128 x 128 (f16 out): 16 rows / 4 tiles in flight 0.321 ms; 16 / 2 0.356; 32 / 4 0.347; 32 / 2 0.373
  if (SPLIT) return launch_dm_mfma_v<T, OutT, 16, 2, SPLIT>(feats, refs, params, out, B, Q, H, W, stride, st);
  if (W > 64) return launch_dm_mfma_v<T, OutT, 16, 4>(feats, refs, params, out, B, Q, H, W, stride, st);
  return launch_dm_mfma_v<T, OutT, 16, 2>(feats, refs, params, out, B, Q, H, W, stride, st);
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

extern "C" int hipie_dynamic_mask16(const float* feats, const float* refs, const float* params, void* out, int B, int Q,
                                    int H, int W, int stride, int dtype, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(feats && refs && params && out, "dynamic_mask16: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q >= 0 && H > 0 && W > 0 && stride > 0, "dynamic_mask16: bad shape");
  HIPIE_REQUIRE(dtype == HIPIE_F16 || dtype == HIPIE_BF16 || dtype == HIPIE_HL8, "dynamic_mask16: operand dtype must be f16, bf16 or HL8 (split fp16 pairs)");
  HIPIE_REQUIRE(W % 4 == 0 && W <= DM_MAXW, "dynamic_mask16: W=%d must be a multiple of 4 and <= %d", W, DM_MAXW);
  HIPIE_REQUIRE(stride % 4 == 0 && (long)stride * W <= 8192 && (long)stride * H <= 8192,
                "dynamic_mask16: pixel coordinates must be exact in 16 bit (stride %% 4 == 0, stride * size <= 8192)");
  HIPIE_REQUIRE((long)B * ((Q + 3) / 4) < 65536, "dynamic_mask16: too many instances for one launch");
  if (B * Q == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
#define HIPIE_DM16(T)                                                                                                     \
  switch (out_dtype) {                                                                                                    \
    case HIPIE_F32: return launch_dm_mfma<T, float>(feats, refs, params, out, B, Q, H, W, stride, st);                    \
    case HIPIE_F16: return launch_dm_mfma<T, f16_t>(feats, refs, params, out, B, Q, H, W, stride, st);                    \
    case HIPIE_BF16: return launch_dm_mfma<T, bf16_t>(feats, refs, params, out, B, Q, H, W, stride, st);                  \
    default: return set_err(HIPIE_EINVAL, "dynamic_mask16: bad out_dtype %d", out_dtype);                                 \
  }
  if (dtype == HIPIE_HL8) {            // fp32-class: weights and activations as fp16 pairs, three products each
    switch (out_dtype) {
      case HIPIE_F32: return launch_dm_mfma<f16_t, float, true>(feats, refs, params, out, B, Q, H, W, stride, st);
      case HIPIE_F16: return launch_dm_mfma<f16_t, f16_t, true>(feats, refs, params, out, B, Q, H, W, stride, st);
      default: return set_err(HIPIE_EINVAL, "dynamic_mask16 (split): out_dtype %d (fp32 | fp16)", out_dtype);
    }
  }
  if (dtype == HIPIE_F16) { HIPIE_DM16(f16_t) } else { HIPIE_DM16(bf16_t) }
#undef HIPIE_DM16
}

// sem_pan.hip -- fused semantic + panoptic map kernel of the post-processing row.
//
// Replaces, for one image, the tensor part of HIPIE_IMG.inference's detection tail (projects/HIPIE/hipie/hipie_img.py:716-748)
//     up   = F.interpolate(masks, x4, bilinear)[:, :, :h, :w];  up = F.interpolate(up, size=(oh, ow), bilinear)
//     sig  = up.sigmoid()
//     sem  = einsum("qc,qhw->chw", cls, sig)                                (semantic_inference, :870-878)
//     ids  = (scores[keep, None, None] * sig[keep]).argmax(0);  own = sig[ids] >= 0.5;  area[q] = (sig[q] >= 0.5).sum()
//                                                                            (panoptic_inference, :473-505)
// The reference materialises `up` and `sig` (N x H x W fp32 each: 4.8 GB per 1024^2 image at N = 1200) and reads them three
// more times; here every sigmoid value is produced in registers from the stride-4 logits -- directly in the MFMA B-operand
// layout of the (classes x queries) . (queries x pixels) contraction -- and is consumed on the spot by the contraction, the
// running arg-max and the area ballots.  HBM traffic per image: the low-resolution logits (L2-resident, 315 MB) + the
// outputs (C x H x W fp32 semantic map, 600 MB at C = 150; 5 bytes per pixel of panoptic data).
//
// Work split: a workgroup = 4 waves = 256 consecutive pixels of one output row; a wave owns 64 of them (two 32-pixel MFMA
// column tiles) and all classes (MT row tiles of 32).  Queries are walked in k-steps of 16: lane (li, hi) evaluates the
// sigmoid of queries q0 + 8 hi + j (j = 0..7) at its pixel li of both tiles -- exactly the 8 k-slots its B fragment holds.
// Precision: PREC 0 splits both operands into bf16 hi + lo and issues 3 MFMAs per product (error ~2^-16, like
// hipie_mask_einsum's default); PREC 1 is plain bf16.
#include "common.h"
#include "mfma.h"

namespace hipie {

struct SemPanParams {
  const float* masks;        // (N, hm, wm) stride-`up` logits
  const bf16_t* cls_hi;      // (MT*32, Npad) class probabilities, transposed, bf16 high part (zero padded)
  const bf16_t* cls_lo;      // low part (PREC 0)
  const float* pscore;       // (Npad) panoptic weight: score if the query is kept, <= 0 otherwise
  float* sem;                // (C, oh, ow)
  int32_t* pan_idx;          // (oh, ow): arg-max query, -1 if no query is kept
  uint8_t* pan_own;          // (oh, ow): sigmoid of the winner >= 0.5
  int32_t* area;             // (Npad): += number of pixels with sigmoid >= 0.5
  int N, Npad, C, hm, wm, up, crop_h, crop_w, oh, ow;
};

// one axis of the two-stage resize: output index -> up to TAPS (low-res index, weight) pairs.
//   stage 2 (only if TAPS == 4): bilinear align_corners=False from the cropped up-grid (size crop) to the output (size out)
//   stage 1: bilinear align_corners=False from the low-res grid (size lo) to the up-grid (lo * up)
template <int TAPS>
__device__ __forceinline__ void axis_taps(int o, int out, int crop, int lo, int up, int (&idx)[TAPS], float (&w)[TAPS]) {
  int u[2];
  float wu[2];
  if (TAPS == 4) {
    const float sc = (float)crop / (float)out;
    const float s = fmaxf(((float)o + 0.5f) * sc - 0.5f, 0.f);
    u[0] = min((int)s, crop - 1);
    u[1] = min(u[0] + 1, crop - 1);
    wu[1] = s - (float)u[0];
    wu[0] = 1.f - wu[1];
  } else {
    u[0] = o; u[1] = o; wu[0] = 1.f; wu[1] = 0.f;
  }
  const float inv = 1.f / (float)up;
#pragma unroll
  for (int a = 0; a < TAPS / 2; ++a) {
    const float s = fmaxf(((float)u[a] + 0.5f) * inv - 0.5f, 0.f);
    const int i0 = (int)s;
    const int i1 = i0 + (i0 < lo - 1 ? 1 : 0);
    const float l1 = s - (float)i0;
    idx[2 * a] = i0; idx[2 * a + 1] = i1;
    w[2 * a] = wu[a] * (1.f - l1); w[2 * a + 1] = wu[a] * l1;
  }
}

template <int MT, int PREC, int TAPS>
__global__ __launch_bounds__(256) void sem_pan_kernel(const SemPanParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s_ps = reinterpret_cast<float*>(smem_raw);                  // [Npad]
  int* s_area = reinterpret_cast<int*>(s_ps + p.Npad);               // [Npad]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int oy = blockIdx.y;
  const int px0 = blockIdx.x * 256 + wave * 64;                     // first pixel (column) of this wave
  for (int i = tid; i < p.Npad; i += 256) { s_ps[i] = p.pscore[i]; s_area[i] = 0; }

  // ---- per-lane sampling geometry of its two pixels (fixed over the query loop) ----
  int yi[TAPS];
  float yw[TAPS];
  axis_taps<TAPS>(min(oy, p.oh - 1), p.oh, p.crop_h, p.hm, p.up, yi, yw);
  int xo[2][TAPS];
  float xw[2][TAPS];
  bool pvalid[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const int ox = px0 + 32 * n + li;
    pvalid[n] = ox < p.ow;
    axis_taps<TAPS>(min(ox, p.ow - 1), p.ow, p.crop_w, p.wm, p.up, xo[n], xw[n]);
  }
  int yoff[TAPS];
#pragma unroll
  for (int a = 0; a < TAPS; ++a) yoff[a] = yi[a] * p.wm;

  f32x16 acc[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][n][r] = 0.f;
  float best_v[2] = {-1.f, -1.f}, best_s[2] = {0.f, 0.f};
  int best_q[2] = {-1, -1};
  __syncthreads();

  const long plane = (long)p.hm * p.wm;
  for (int q0 = 0; q0 < p.Npad; q0 += 16) {
    // ---- sigmoid values of queries q0 + 8 hi + j at this lane's two pixels ----
    float sg[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = min(q0 + 8 * hi + j, p.N - 1);                  // padded k-slots re-read the last query (their cls column is 0)
      const float* mq = p.masks + (long)q * plane;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float v = 0.f;
#pragma unroll
        for (int a = 0; a < TAPS; ++a) {
          float rowv = 0.f;
#pragma unroll
          for (int c = 0; c < TAPS; ++c) rowv = fmaf(xw[n][c], mq[yoff[a] + xo[n][c]], rowv);
          v = fmaf(yw[a], rowv, v);
        }
        sg[n][j] = 1.f / (1.f + __expf(-v));
      }
      if (TAPS == 4) __builtin_amdgcn_sched_barrier(0);      // 16 taps per value: keep one query's loads in flight at a time
    }
    // ---- panoptic arg-max and areas ----
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = q0 + 8 * hi + j;
      const float ps = s_ps[q];
      const bool real = q < p.N;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const float v = ps * sg[n][j];
        if (ps > 0.f && v > best_v[n]) { best_v[n] = v; best_q[n] = q; best_s[n] = sg[n][j]; }
        const unsigned long long b = __ballot(real && pvalid[n] && sg[n][j] >= 0.5f);
        if (b != 0ull && li == 0) {                                   // lanes 0 and 32: one per half
          const int cnt = __popc((unsigned)(hi ? (b >> 32) : (b & 0xffffffffull)));
          if (cnt) atomicAdd(&s_area[q], cnt);
        }
      }
    }
    // ---- contraction: sem[class, pixel] += cls[q, class] * sigmoid[q, pixel] ----
    bf16x8 bh[2], bl[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bf16_t h = (bf16_t)sg[n][j];
        bh[n][j] = h;
        if (PREC == 0) bl[n][j] = (bf16_t)(sg[n][j] - (float)h);
      }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const long aoff = (long)(32 * mt + li) * p.Npad + q0 + 8 * hi;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(p.cls_hi + aoff);
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[mt][n] = Mfma32<bf16_t>::mma(ah, bh[n], acc[mt][n]);
      if (PREC == 0) {
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(p.cls_lo + aoff);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[mt][n] = Mfma32<bf16_t>::mma(ah, bl[n], acc[mt][n]);
          acc[mt][n] = Mfma32<bf16_t>::mma(al, bh[n], acc[mt][n]);
        }
      }
    }
  }

  // ---- outputs ----
  const long opix = (long)p.oh * p.ow;
  if (oy < p.oh) {
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      // merge the two lane halves (they saw disjoint queries of the same pixel): larger value, then smaller index
      const float ov = __shfl_xor(best_v[n], 32), os = __shfl_xor(best_s[n], 32);
      const int oq = __shfl_xor(best_q[n], 32);
      const bool take = (oq >= 0) && (best_q[n] < 0 || ov > best_v[n] || (ov == best_v[n] && oq < best_q[n]));
      const int wq = take ? oq : best_q[n];
      const float ws = take ? os : best_s[n];
      const int ox = px0 + 32 * n + li;
      if (pvalid[n]) {
        if (hi == 0) {
          p.pan_idx[(long)oy * p.ow + ox] = wq;
          p.pan_own[(long)oy * p.ow + ox] = (wq >= 0 && ws >= 0.5f) ? 1 : 0;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * mt + crow(r, hi);
            if (c < p.C) p.sem[(long)c * opix + (long)oy * p.ow + ox] = acc[mt][n][r];
          }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < p.N; i += 256)
    if (s_area[i]) atomicAdd(&p.area[i], s_area[i]);
}

template <int MT, int PREC>
static int launch_sem_pan(const SemPanParams& p, bool ident, hipStream_t st) {
  dim3 grid(ceil_div(p.ow, 256), p.oh);
  const size_t lds = (size_t)p.Npad * 8;
  if (ident) hipLaunchKernelGGL((sem_pan_kernel<MT, PREC, 2>), grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL((sem_pan_kernel<MT, PREC, 4>), grid, dim3(256), lds, st, p);
  return check_launch("sem_pan");
}

}  // namespace hipie

extern "C" int hipie_sem_pan(const float* masks, const void* cls_hi, const void* cls_lo, const float* pscore, float* sem,
                             int32_t* pan_idx, uint8_t* pan_own, int32_t* area, int N, int Npad, int C, int hm, int wm,
                             int up, int crop_h, int crop_w, int out_h, int out_w, int precision, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(N > 0 && Npad >= N && Npad % 16 == 0 && Npad <= 8192, "sem_pan: bad N=%d Npad=%d", N, Npad);
  HIPIE_REQUIRE(C > 0 && C <= 160, "sem_pan: C=%d classes unsupported (1..160)", C);
  HIPIE_REQUIRE(hm > 0 && wm > 0 && up > 0 && crop_h > 0 && crop_w > 0 && crop_h <= hm * up && crop_w <= wm * up, "sem_pan: bad mask geometry");
  HIPIE_REQUIRE(out_h > 0 && out_w > 0 && out_h <= 65535, "sem_pan: bad output size");
  HIPIE_REQUIRE(precision == 0 || precision == 1, "sem_pan: precision must be 0 (bf16x3) or 1 (bf16)");
  HIPIE_REQUIRE(masks && cls_hi && pscore && sem && pan_idx && pan_own && area && (precision == 1 || cls_lo), "sem_pan: null pointer");
  SemPanParams p{};
  p.masks = masks; p.cls_hi = (const bf16_t*)cls_hi; p.cls_lo = (const bf16_t*)cls_lo; p.pscore = pscore;
  p.sem = sem; p.pan_idx = pan_idx; p.pan_own = pan_own; p.area = area;
  p.N = N; p.Npad = Npad; p.C = C; p.hm = hm; p.wm = wm; p.up = up; p.crop_h = crop_h; p.crop_w = crop_w;
  p.oh = out_h; p.ow = out_w;
  const bool ident = (out_h == crop_h && out_w == crop_w);
  hipStream_t st = (hipStream_t)stream;
  const int mt = (C + 31) / 32;
#define HIPIE_SP(MT_)                                                                          \
  return precision == 0 ? launch_sem_pan<MT_, 0>(p, ident, st) : launch_sem_pan<MT_, 1>(p, ident, st)
  if (mt <= 1) { HIPIE_SP(1); }
  if (mt <= 3) { HIPIE_SP(3); }
  HIPIE_SP(5);
#undef HIPIE_SP
}

// msda.hip -- multi-scale deformable attention sampling (forward) for gfx950.
//
// What it computes: SURVEY.md row a10 / include/hipie_mi355.h hipie_msda_forward.  For every (batch b, query q,
// head m) the weighted sum over L levels x P points of a bilinear sample of value[b, level pixels, m, 0:D].
//
// Data layout in HBM: value is (B, S, M, D) with D fastest, i.e. the M*D = 256 channels of one pixel are one
// contiguous 1 KiB (f32) / 512 B (16-bit) line and one head's slice is 128 / 64 contiguous bytes.
//
// Mapping (D == 32 fast path): 8 lanes own one (b,q,m); each lane owns 4 consecutive channels, so a corner fetch of one
// head is ONE 128-byte coalesced segment (8 lanes x 16 B) and a wave covers 8 heads = the full 1 KiB pixel line when
// M == 8.  Two phases per (query, head): the location / bilinear / softmax arithmetic of the L*P points is split over the 8
// lanes and published through LDS as (4 corner offsets, 4 weights) records; then all 8 lanes run a pure gather + FMA loop
// over the records (it used to be repeated by every lane, which made the kernel VALU-bound: 0.9 -> 0.47 ms per bs-8
// encoder call).  Control flow is branch-free: out-of-range samples / corners are clamped to a legal address and get a zero
// weight, exactly reproducing the zero padding of ms_deform_attn_im2col_bilinear (ms_deform_im2col_cuda.cuh:33-84).
// Tried and dropped: staging per-tile sampling windows of all levels in LDS (bit-identical results, 1.03 ms vs 0.47 ms) --
// the gather is served well by L1/L2; occupancy and the per-point arithmetic are what matter.
//
// The FUSED variant additionally computes the sampling locations and the softmax over the L*P logits in registers
// (ops/modules/ms_deform_attn.py:99-114) so the (B,Lq,M,L,P,2) location tensor never exists in HBM.
//
// Roofline: HBM/L2 bound.  Algorithmic bytes per call (f32): S*M*D*4 (value) + Lq*M*L*P*(2+1)*4 (loc, weights) +
// Lq*M*D*4 (out) per image = 78 MB at Nv = 21760.
#include <stdlib.h>

#include "common.h"

namespace hipie {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  typedef float4 raw;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
    for (int i = 0; i < 4; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    bf16x4 r;
    for (int i = 0; i < 4; ++i) r[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = r;
  }
};
template <> struct Vec4<f16_t> {
  static __device__ __forceinline__ void load(const f16_t* p, float (&v)[4]) {
    f16x4 r = *reinterpret_cast<const f16x4*>(p);
    for (int i = 0; i < 4; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void store(f16_t* p, const float (&v)[4]) {
    f16x4 r;
    for (int i = 0; i < 4; ++i) r[i] = (f16_t)v[i];
    *reinterpret_cast<f16x4*>(p) = r;
  }
};

constexpr int kMaxLP = 32;  // L*P supported by the fused softmax (reference geometry: 4*4 = 16)

// ---- shared by the D == 32 kernels: the per-point record (4 corner offsets + 4 weights) --------------------------------
// The location / bilinear / softmax arithmetic of one sampling point is ~70 VALU instructions; the 8 lanes that share a
// (query, head) used to repeat all of it (the kernel was VALU-bound at ~1800 instructions per lane, not gather-bound).  Now
// each lane does it for LP/8 of the points (phase 1), publishes a record of 4 clamped corner offsets (elements, relative to
// the head's slice of the image) and 4 weights (bilinear x attention weight, zero for padded corners) through LDS, and all
// 8 lanes then run the pure gather + FMA loop over the records (phase 2).

struct PointRec {
  int o[4];
  float w[4];
};

// sampling point at normalised (x, y) of an H x W level whose first pixel is `lbase` pixels into the image; `row` elements
// per pixel; aw = attention weight.  Zero padding exactly as ms_deform_attn_im2col_bilinear (ms_deform_im2col_cuda.cuh:33-84).
__device__ __forceinline__ PointRec point_record(float x, float y, int H, int W, long lbase, long row, float aw) {
  const float h_im = y * (float)H - 0.5f;
  const float w_im = x * (float)W - 0.5f;
  const bool inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const float lh = h_im - hf, lw = w_im - wf, hh = 1.f - lh, hw = 1.f - lw;
  const int h0 = inside ? (int)hf : 0, w0 = inside ? (int)wf : 0;
  const int h1 = h0 + 1, w1 = w0 + 1;
  const bool okh0 = inside && h0 >= 0, okh1 = inside && h1 <= H - 1;
  const bool okw0 = w0 >= 0, okw1 = w1 <= W - 1;
  const int ch0 = max(h0, 0), ch1 = min(h1, H - 1), cw0 = max(w0, 0), cw1 = min(w1, W - 1);
  PointRec r;
  r.o[0] = (int)((lbase + (long)ch0 * W + cw0) * row);
  r.o[1] = (int)((lbase + (long)ch0 * W + cw1) * row);
  r.o[2] = (int)((lbase + (long)ch1 * W + cw0) * row);
  r.o[3] = (int)((lbase + (long)ch1 * W + cw1) * row);
  r.w[0] = (okh0 && okw0) ? hh * hw * aw : 0.f;
  r.w[1] = (okh0 && okw1) ? hh * lw * aw : 0.f;
  r.w[2] = (okh1 && okw0) ? lh * hw * aw : 0.f;
  r.w[3] = (okh1 && okw1) ? lh * lw * aw : 0.f;
  return r;
}

// max / sum over the 8 lanes of a (query, head) group
__device__ __forceinline__ float group_max8(float v) {
  v = fmaxf(v, __shfl_xor(v, 1)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 4));
  return v;
}
__device__ __forceinline__ float group_sum8(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
  return v;
}

// phase 1 for one (batch*query row bq, head m): lane `sub` handles points sub, sub + 8, ...; records go to rec[point * 8 ..]
template <typename A, bool FUSED>
__device__ __forceinline__ void publish_records(float* rec, int sub, const A* lp, const A* wp, const float* refrow,
                                                int ref_dim, const int64_t* shapes, const int64_t* lstart, int L, int P,
                                                long row) {
  const int LP = L * P;
  float wmax = 0.f, winv = 1.f;
  if (FUSED) {
    float mx = -INFINITY;
    for (int i = sub; i < LP; i += 8) mx = fmaxf(mx, elem<A>::to_f32(wp[i]));
    wmax = group_max8(mx);
    float sm = 0.f;
    for (int i = sub; i < LP; i += 8) sm += expf(elem<A>::to_f32(wp[i]) - wmax);
    winv = 1.f / group_sum8(sm);
  }
  for (int i = sub; i < LP; i += 8) {
    const int l = i / P;
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    float x = elem<A>::to_f32(lp[2 * i]), y = elem<A>::to_f32(lp[2 * i + 1]), aw;
    if (FUSED) {
      const float* r = refrow + l * ref_dim;
      if (ref_dim == 2) {                              // ref + off / (W_l, H_l)
        x = r[0] + x / (float)W;
        y = r[1] + y / (float)H;
      } else {                                         // ref_xy + off / P * ref_wh * 0.5
        x = r[0] + x / (float)P * r[2] * 0.5f;
        y = r[1] + y / (float)P * r[3] * 0.5f;
      }
      aw = expf(elem<A>::to_f32(wp[i]) - wmax) * winv;
    } else {
      aw = elem<A>::to_f32(wp[i]);
    }
    const PointRec pr = point_record(x, y, H, W, lstart[l], row, aw);
    *reinterpret_cast<int4*>(rec + i * 8) = make_int4(pr.o[0], pr.o[1], pr.o[2], pr.o[3]);
    *reinterpret_cast<float4*>(rec + i * 8 + 4) = make_float4(pr.w[0], pr.w[1], pr.w[2], pr.w[3]);
  }
}

// D == 32: 8 lanes per (b,q,m), 4 channels per lane.
template <typename T, typename A, bool FUSED, int U>
__global__ __launch_bounds__(256) void msda_d32_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart,
                                                       const A* __restrict__ loc_or_off, const A* __restrict__ w_or_logit,
                                                       const float* __restrict__ ref, T* __restrict__ out, int S, int M,
                                                       int L, int Lq, int P, int ref_dim, long total_groups,
                                                       long off_stride, long w_stride, long vrow, int map) {
  extern __shared__ __attribute__((aligned(16))) float recs[];   // 32 groups x (LP * 8 + 4) words
  // XCD-aware block order: workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md); give every XCD one CONTIGUOUS range of
  // (image, query) groups so that neighbouring queries -- which sample neighbouring value pixels -- share that XCD's L2
  // instead of all eight L2s streaming the whole value tensor.  Placement only affects speed.
  // bijective for any grid size: XCD x owns q+1 blocks if x < r else q  (nblk = 8 q + r)
  const long nblk = gridDim.x, qn = nblk >> 3, rn = nblk & 7;
  const long xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const long blk = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  // Which 32 (query, head) groups this workgroup owns (placement only: every group computes the same thing wherever it runs).
  //   map 0  heads fastest: a wave = the 8 heads of ONE query.  Its 8 lane groups fetch 8 unrelated 128-byte lines per corner
  //          and nothing is reused inside a wave; the L1 hit rate is what co-resident neighbour queries happen to share.
  //   map 1  queries fastest: a workgroup = 32 CONSECUTIVE queries of ONE head, a wave = 8 of them.  Neighbouring queries of a
  //          pyramid sample neighbouring pixels of that head, so the x+1 corner of one lane group is the x corner of the next
  //          and a corner instruction of a coarse level touches 2-5 distinct lines instead of 8.
  //   map 2  (queries are the pyramid's own pixels, Lq == S, and every level is a multiple of 4 rows x 8 columns) a workgroup =
  //          an 8-wide x 4-high TILE of one level, wave w = row w of the tile: the y+1 corners of a wave are the y corners of
  //          the next wave.  Falls back to map 1 when a level does not divide.
  long g;
  bool live;
  if (map == 0) {
    g = blk * 32 + (threadIdx.x >> 3);
    live = g < total_groups;
  } else {
    const long chunk = blk / M;
    const int mh = (int)(blk - chunk * M);
    long bqn = chunk * 32 + (threadIdx.x >> 3);
    if (map == 2) {
      bool tiles = true;
      for (int l = 0; l < L; ++l) tiles = tiles && (shapes[2 * l] % 4 == 0) && (shapes[2 * l + 1] % 8 == 0);
      if (tiles) {
        const long cpi = Lq / 32;                        // tiles per image (every level divides, so does their sum)
        const long bi = chunk / cpi;
        long ci = chunk - bi * cpi;
        int l = 0;
        for (; l < L - 1; ++l) {
          const long nt = (shapes[2 * l] / 4) * (shapes[2 * l + 1] / 8);
          if (ci < nt) break;
          ci -= nt;
        }
        const long Wl = shapes[2 * l + 1], tw = Wl / 8;
        const long ty = ci / tw, tx = ci - ty * tw;
        bqn = bi * Lq + lstart[l] + (ty * 4 + (threadIdx.x >> 6)) * Wl + tx * 8 + ((threadIdx.x >> 3) & 7);
      }
    }
    live = bqn < total_groups / M;
    g = (live ? bqn : total_groups / M - 1) * M + mh;
  }
  const long gc = live ? g : total_groups - 1;         // dead groups recompute the last one (the shuffles need every lane)
  const int sub = threadIdx.x & 7;
  const int m = (int)(gc % M);
  const long bq = gc / M;
  const int b = (int)(bq / Lq);
  const int LP = L * P;
  const long row = vrow;                               // elements from one pixel's channels to the next pixel's (>= M * 32)
  float* rec = recs + (threadIdx.x >> 3) * (LP * 8 + 4);      // +4 words: the 8 groups of a wave land on disjoint banks
  // row = one (batch, query); offsets / logits of head m inside the row (dense rows when the strides are M*L*P*2 / M*L*P)
  publish_records<A, FUSED>(rec, sub, loc_or_off + bq * off_stride + (long)m * (LP * 2), w_or_logit + bq * w_stride + (long)m * LP,
                            FUSED ? ref + bq * L * ref_dim : nullptr, ref_dim, shapes, lstart, L, P, row);
  // the 8 lanes of a group live in one wave: no block barrier is needed; the records are waited for (lgkmcnt(0)) before they are read back
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_wave_barrier();

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const T* vb = value + (long)b * S * row + m * 32 + sub * 4;
#pragma unroll U
  for (int i = 0; i < LP; ++i) {
    const int4 o = *reinterpret_cast<const int4*>(rec + i * 8);
    const float4 w = *reinterpret_cast<const float4*>(rec + i * 8 + 4);
    float v1[4], v2[4], v3[4], v4[4];
    Vec4<T>::load(vb + o.x, v1);
    Vec4<T>::load(vb + o.y, v2);
    Vec4<T>::load(vb + o.z, v3);
    Vec4<T>::load(vb + o.w, v4);
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = fmaf(w.w, v4[c], fmaf(w.z, v3[c], fmaf(w.y, v2[c], fmaf(w.x, v1[c], acc[c]))));
  }
  if (live) Vec4<T>::store(out + g * 32 + sub * 4, acc);
}

// any D: one thread per output element (b,q,m,c), the reference's own decomposition.
template <typename T, typename A, bool FUSED>
__global__ __launch_bounds__(256) void msda_generic_kernel(const T* __restrict__ value, const int64_t* __restrict__ shapes,
                                                           const int64_t* __restrict__ lstart,
                                                           const A* __restrict__ loc_or_off,
                                                           const A* __restrict__ w_or_logit,
                                                           const float* __restrict__ ref, T* __restrict__ out, int S,
                                                           int M, int D, int L, int Lq, int P, int ref_dim, long n,
                                                           long off_stride, long w_stride, long vrow) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % D);
  const long g = idx / D;
  const int m = (int)(g % M);
  const long bq = g / M;
  const int b = (int)(bq / Lq);
  const int LP = L * P;
  const A* lp = loc_or_off + bq * off_stride + (long)m * (LP * 2);
  const A* wp = w_or_logit + bq * w_stride + (long)m * LP;
  float wmax = 0.f, winv = 1.f;
  if (FUSED) {
    wmax = -INFINITY;
    for (int i = 0; i < LP; ++i) wmax = fmaxf(wmax, elem<A>::to_f32(wp[i]));
    float s = 0.f;
    for (int i = 0; i < LP; ++i) s += expf(elem<A>::to_f32(wp[i]) - wmax);
    winv = 1.f / s;
  }
  const long row = vrow;
  const T* vb = value + (long)b * S * row + m * D + c;
  float col = 0.f;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const T* vl = vb + (long)lstart[l] * row;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      float x = elem<A>::to_f32(lp[2 * i]), y = elem<A>::to_f32(lp[2 * i + 1]), w;
      if (FUSED) {
        const float* r = ref + (bq * L + l) * ref_dim;
        if (ref_dim == 2) {
          x = r[0] + x / (float)W;
          y = r[1] + y / (float)H;
        } else {
          x = r[0] + x / (float)P * r[2] * 0.5f;
          y = r[1] + y / (float)P * r[3] * 0.5f;
        }
        w = expf(elem<A>::to_f32(wp[i]) - wmax) * winv;
      } else {
        w = elem<A>::to_f32(wp[i]);
      }
      const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
      if (h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W) {
        const int h0 = (int)floorf(h_im), w0 = (int)floorf(w_im), h1 = h0 + 1, w1 = w0 + 1;
        const float lh = h_im - h0, lw = w_im - w0, hh = 1.f - lh, hw = 1.f - lw;
        float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
        if (h0 >= 0 && w0 >= 0) v1 = elem<T>::to_f32(vl[((long)h0 * W + w0) * row]);
        if (h0 >= 0 && w1 <= W - 1) v2 = elem<T>::to_f32(vl[((long)h0 * W + w1) * row]);
        if (h1 <= H - 1 && w0 >= 0) v3 = elem<T>::to_f32(vl[((long)h1 * W + w0) * row]);
        if (h1 <= H - 1 && w1 <= W - 1) v4 = elem<T>::to_f32(vl[((long)h1 * W + w1) * row]);
        col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * w;
      }
    }
  }
  out[idx] = elem<T>::from_f32(col);
}

// value row stride of the call being dispatched (host-side plumbing through the dtype switch; 0 = dense M * D)
static thread_local long g_value_row = 0;


// double instantiation of the reference op (AT_DISPATCH_FLOATING_TYPES, ms_deform_attn_cuda.cu:56): one thread per output element,
// every quantity in f64 in the reference's operation order (ms_deform_attn_im2col_bilinear, ms_deform_im2col_cuda.cuh:21-73, 237-299).
// Not on the product path -- it completes the plugin boundary (ops/test.py checks the op in double).
__global__ __launch_bounds__(256) void msda_f64_kernel(const double* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ lstart, const double* __restrict__ loc,
                                                       const double* __restrict__ attn, double* __restrict__ out, int S, int M, int D,
                                                       int L, int Lq, int P, long n) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int c = (int)(idx % D);
  const long g = idx / D;
  const int m = (int)(g % M);
  const long bq = g / M;
  const int b = (int)(bq / Lq);
  const long row = (long)M * D;
  const double* vb = value + (long)b * S * row + m * D + c;
  const double* lp = loc + g * (long)L * P * 2;
  const double* wp = attn + g * (long)L * P;
  double col = 0.0;
  for (int l = 0; l < L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const double* vl = vb + (long)lstart[l] * row;
    for (int p = 0; p < P; ++p) {
      const int i = l * P + p;
      const double h_im = lp[2 * i + 1] * H - 0.5, w_im = lp[2 * i] * W - 0.5;
      if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) {
        const int h0 = (int)floor(h_im), w0 = (int)floor(w_im), h1 = h0 + 1, w1 = w0 + 1;
        const double lh = h_im - h0, lw = w_im - w0, hh = 1 - lh, hw = 1 - lw;
        double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
        if (h0 >= 0 && w0 >= 0) v1 = vl[((long)h0 * W + w0) * row];
        if (h0 >= 0 && w1 <= W - 1) v2 = vl[((long)h0 * W + w1) * row];
        if (h1 <= H - 1 && w0 >= 0) v3 = vl[((long)h1 * W + w0) * row];
        if (h1 <= H - 1 && w1 <= W - 1) v4 = vl[((long)h1 * W + w1) * row];
        col += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * wp[i];
      }
    }
  }
  out[idx] = col;
}

template <typename T, typename A, bool FUSED>
static int launch_msda(const void* value, const int64_t* shapes, const int64_t* lstart, const void* a, const void* w,
                       const float* ref, void* out, int B, int S, int M, int D, int L, int Lq, int P, int ref_dim,
                       long off_stride, long w_stride, hipStream_t st) {
  const long groups = (long)B * Lq * M;
  if (groups == 0) return HIPIE_OK;
  const long vrow = g_value_row > 0 ? g_value_row : (long)M * D;
  if (D == 32) {
    // group -> workgroup map (see the kernel): tiles of the pyramid when the queries are its pixels, else runs of 32 queries of
    // one head; HIPIE_MSDA_MAP=0 restores the heads-fastest order (A/B timing: tools/bench_msda.py)
    static const int map_env = [] { const char* me = study_env("HIPIE_MSDA_MAP"); return me ? atoi(me) : -1; }();      // once per process
    const int map = map_env >= 0 ? (map_env == 2 && Lq != S ? 1 : map_env) : (Lq == S ? 2 : 1);
    const long blocks = map == 0 ? (groups + 31) / 32 : (((long)B * Lq + 31) / 32) * M;
    const size_t lds = (size_t)32 * (L * P * 8 + 4) * sizeof(float);
    // unroll 2 of the record loop: 0.467 ms vs 0.480 (1, 4, 8) on the bs-8 encoder geometry (tools/bench_msda.py)
    hipLaunchKernelGGL((msda_d32_kernel<T, A, FUSED, 2>), dim3((unsigned)blocks), dim3(256), lds, st, (const T*)value, shapes,
                       lstart, (const A*)a, (const A*)w, ref, (T*)out, S, M, L, Lq, P, ref_dim, groups, off_stride, w_stride, vrow, map);
  } else {
    const long n = groups * D;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL((msda_generic_kernel<T, A, FUSED>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)value,
                       shapes, lstart, (const A*)a, (const A*)w, ref, (T*)out, S, M, D, L, Lq, P, ref_dim, n, off_stride, w_stride, vrow);
  }
  return check_launch("msda");
}

template <typename T, bool FUSED>
static int dispatch_aux(int aux_dtype, const void* value, const int64_t* shapes, const int64_t* lstart, const void* a,
                        const void* w, const float* ref, void* out, int B, int S, int M, int D, int L, int Lq, int P,
                        int ref_dim, long off_stride, long w_stride, hipStream_t st) {
  switch (aux_dtype) {
    case HIPIE_F32: return launch_msda<T, float, FUSED>(value, shapes, lstart, a, w, ref, out, B, S, M, D, L, Lq, P, ref_dim, off_stride, w_stride, st);
    case HIPIE_F16: return launch_msda<T, f16_t, FUSED>(value, shapes, lstart, a, w, ref, out, B, S, M, D, L, Lq, P, ref_dim, off_stride, w_stride, st);
    case HIPIE_BF16: return launch_msda<T, bf16_t, FUSED>(value, shapes, lstart, a, w, ref, out, B, S, M, D, L, Lq, P, ref_dim, off_stride, w_stride, st);
    default: return set_err(HIPIE_EINVAL, "msda: unsupported aux dtype %d", aux_dtype);
  }
}


template <bool FUSED>
static int dispatch_msda(const void* value, const int64_t* shapes, const int64_t* lstart, const void* a, const void* w,
                         const float* ref, void* out, int B, int S, int M, int D, int L, int Lq, int P, int ref_dim,
                         int dtype, int aux_dtype, long off_stride, long w_stride, void* stream) {
  HIPIE_REQUIRE(B >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda: bad shape B=%d S=%d M=%d D=%d L=%d Lq=%d P=%d", B, S, M, D, L, Lq, P);
  if (B == 0 || Lq == 0) return HIPIE_OK;              // no queries: nothing to write (pointers of empty tensors may be null)
  HIPIE_REQUIRE(value && shapes && lstart && a && w && out, "msda: null pointer");
  HIPIE_REQUIRE((long)B * S * M * D < (1L << 40), "msda: tensor too large");
  HIPIE_REQUIRE(off_stride >= (long)M * L * P * 2 && w_stride >= (long)M * L * P, "msda: row strides too small");
  HIPIE_REQUIRE((long)S * (g_value_row > 0 ? g_value_row : (long)M * D) < (1L << 31), "msda: one image's value block exceeds the 32-bit sample offsets");
  HIPIE_REQUIRE(g_value_row == 0 || (g_value_row >= (long)M * D && g_value_row % 8 == 0), "msda: value row stride %ld < M*D or not a multiple of 8", g_value_row);
  if (FUSED) {
    HIPIE_REQUIRE(ref != nullptr && (ref_dim == 2 || ref_dim == 4), "msda_fused: ref_dim must be 2 or 4 (got %d)", ref_dim);
    HIPIE_REQUIRE(L * P <= kMaxLP, "msda_fused: L*P=%d > %d", L * P, kMaxLP);
  }
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HIPIE_F32: return dispatch_aux<float, FUSED>(aux_dtype, value, shapes, lstart, a, w, ref, out, B, S, M, D, L, Lq, P, ref_dim, off_stride, w_stride, st);
    case HIPIE_F16: return dispatch_aux<f16_t, FUSED>(aux_dtype, value, shapes, lstart, a, w, ref, out, B, S, M, D, L, Lq, P, ref_dim, off_stride, w_stride, st);
    case HIPIE_BF16: return dispatch_aux<bf16_t, FUSED>(aux_dtype, value, shapes, lstart, a, w, ref, out, B, S, M, D, L, Lq, P, ref_dim, off_stride, w_stride, st);
    default: return set_err(HIPIE_EINVAL, "msda: unsupported dtype %d", dtype);
  }
}

}  // namespace hipie

static int msda_forward_f64(const void* value, const int64_t* shapes, const int64_t* lstart, const void* loc, const void* attn, void* out,
                            int B, int S, int M, int D, int L, int Lq, int P, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(B >= 0 && S > 0 && M > 0 && D > 0 && L > 0 && Lq >= 0 && P > 0, "msda: bad shape");
  if (B == 0 || Lq == 0) return HIPIE_OK;
  HIPIE_REQUIRE(value && shapes && lstart && loc && attn && out, "msda: null pointer");
  const long n = (long)B * Lq * M * D;
  hipLaunchKernelGGL(msda_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const double*)value, shapes,
                     lstart, (const double*)loc, (const double*)attn, (double*)out, S, M, D, L, Lq, P, n);
  return check_launch("msda_f64");
}

extern "C" int hipie_msda_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                  const void* sampling_loc, const void* attn_weight, void* out, int B, int S, int M,
                                  int D, int L, int Lq, int P, int value_dtype, void* stream) {
  if (value_dtype == HIPIE_F64)
    return msda_forward_f64(value, spatial_shapes, level_start, sampling_loc, attn_weight, out, B, S, M, D, L, Lq, P, stream);
  return hipie::dispatch_msda<false>(value, spatial_shapes, level_start, sampling_loc, attn_weight, nullptr, out, B, S,
                                     M, D, L, Lq, P, 2, value_dtype, HIPIE_F32, (long)M * L * P * 2, (long)M * L * P, stream);
}

extern "C" int hipie_msda_fused_forward(const void* value, const int64_t* spatial_shapes, const int64_t* level_start,
                                        const float* ref, const void* offsets, const void* logits, void* out, int B,
                                        int S, int M, int D, int L, int Lq, int P, int ref_dim, int value_dtype,
                                        int aux_dtype, int64_t off_row_stride, int64_t logit_row_stride, void* stream) {
  return hipie::dispatch_msda<true>(value, spatial_shapes, level_start, offsets, logits, ref, out, B, S, M, D, L, Lq, P,
                                    ref_dim, value_dtype, aux_dtype, off_row_stride, logit_row_stride, stream);
}

extern "C" int hipie_msda_fused_forward_strided(const void* value, int64_t value_row_stride, const int64_t* spatial_shapes,
                                                const int64_t* level_start, const float* ref, const void* offsets,
                                                const void* logits, void* out, int B, int S, int M, int D, int L, int Lq, int P,
                                                int ref_dim, int value_dtype, int aux_dtype, int64_t off_row_stride,
                                                int64_t logit_row_stride, void* stream) {
  hipie::g_value_row = value_row_stride;
  const int rc = hipie::dispatch_msda<true>(value, spatial_shapes, level_start, offsets, logits, ref, out, B, S, M, D, L, Lq, P,
                                            ref_dim, value_dtype, aux_dtype, off_row_stride, logit_row_stride, stream);
  hipie::g_value_row = 0;
  return rc;
}

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
#include <algorithm>

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


// ---- the gather form (D == 32, fp32, caller-provided workspace): no floating-point atomics ------------------------------------------
// The atomic form above is bound by the rate of the fp32 atomic units: 713 M adds per encoder call (B = 2) at ~460 G adds/s = 1.55 of its
// 2.1 ms, the same for clustered and for uniformly spread sampling points and for any placement of the heads on the XCDs (measured,
// tools/bench_msda.py bwd) -- the count of adds is what costs.  Here grad_value is computed from the DESTINATION side instead:
//   1. bin    count the valid corners per (pixel, head) destination in LDS counters (one workgroup per (image, head) plane and query slice)
//   2. scan   exclusive prefix over the (destination, slice) counts (three small launches)
//   3. fill   the same workgroups again: every corner takes a slot of its destination and leaves a record (source row b*Lq+q, weight)
//   4. gather 8 lanes x float4 per destination walk its records and sum weight x grad_out[row, head, :] in registers: one plain
//             128-byte store per destination, every destination written (no memset), coarse levels first (their lists are the longest)
//   5. coef   grad_attn_weight / grad_sampling_loc: 8 lanes x float4 per (b, q, head) item, corner reads of 128 bytes, three xor
//             shuffles per sum instead of the five of the 32-lane form
// Summation order inside a destination follows the slot order (arrival order of step 3): as with the atomics, and as in the reference,
// grad_value is not bit-reproducible between runs; the other two outputs are.

struct BwdRec {
  int row;      // b * Lq + q: the grad_out row (and image) the sample came from
  float w;      // attention weight x bilinear weight of this corner
};

struct BwdCorners {
  bool inside;
  bool ok[4];
  long pix[4];          // pixel row inside the image's S rows (level_start + h * W + w)
  float c[4];           // bilinear weights
  float hh, hw, lh, lw;
  int H, W;
};

// geometry of one sampling point exactly as ms_deform_attn_col2im_bilinear (ms_deform_im2col_cuda.cuh:76-160) and its caller (:301-345)
__device__ __forceinline__ BwdCorners bwd_corners(float x, float y, int H, int W, long lstart) {
  BwdCorners r;
  r.H = H; r.W = W;
  const float h_im = y * (float)H - 0.5f, w_im = x * (float)W - 0.5f;
  r.inside = h_im > -1.f && w_im > -1.f && h_im < (float)H && w_im < (float)W;
  const int h0 = r.inside ? (int)floorf(h_im) : 0, w0 = r.inside ? (int)floorf(w_im) : 0, h1 = h0 + 1, w1 = w0 + 1;
  r.lh = h_im - (float)h0; r.lw = w_im - (float)w0; r.hh = 1.f - r.lh; r.hw = 1.f - r.lw;
  r.ok[0] = r.inside && h0 >= 0 && w0 >= 0;
  r.ok[1] = r.inside && h0 >= 0 && w1 <= W - 1;
  r.ok[2] = r.inside && h1 <= H - 1 && w0 >= 0;
  r.ok[3] = r.inside && h1 <= H - 1 && w1 <= W - 1;
  const int ch0 = max(h0, 0), cw0 = max(w0, 0), ch1 = min(h1, H - 1), cw1 = min(w1, W - 1);      // a legal address for masked corners
  r.pix[0] = lstart + (long)ch0 * W + cw0;
  r.pix[1] = lstart + (long)ch0 * W + cw1;
  r.pix[2] = lstart + (long)ch1 * W + cw0;
  r.pix[3] = lstart + (long)ch1 * W + cw1;
  r.c[0] = r.hh * r.hw; r.c[1] = r.hh * r.lw; r.c[2] = r.lh * r.hw; r.c[3] = r.lh * r.lw;
  return r;
}

// steps 1 and 3.  Scattered global integer atomics are no cheaper than the float ones (measured: 22 M of them = 1.2 ms; the atomic units
// take ~15 G REQUESTS per second, a 128-byte row of 32 floats being one request), so the bins live in LDS: workgroup (plane = b*M + m,
// slice) handles the queries [slice * per, (slice + 1) * per) of image b for head m with one LDS counter per pixel row of the image
// (S ints <= 160 KB).  Counting pass: ds_add_u32, then the slice's counts are stored to bins[plane][slice][0:S].  After the scan the
// same array holds the first slot of every (destination, slice) segment; the fill pass loads it back into LDS and every corner takes its
// slot with a returning LDS add.
template <bool FILL>
__global__ __launch_bounds__(1024) void msda_bwd_bin_kernel(const int64_t* __restrict__ shapes, const int64_t* __restrict__ lstart,
                                                            const float* __restrict__ loc, const float* __restrict__ attn,
                                                            int* __restrict__ bins, BwdRec* __restrict__ recs, int S, int M, int L, int Lq,
                                                            int P, int K) {
  extern __shared__ int bwd_lds[];
  const int tid = threadIdx.x;
  const int slice = (int)(blockIdx.x % K), plane = (int)(blockIdx.x / K);
  const int b = plane / M, m = plane % M;
  int* mine = bins + ((long)plane * K + slice) * S;
  for (int s = tid; s < S; s += 1024) bwd_lds[s] = FILL ? mine[s] : 0;
  __syncthreads();
  const int per = (Lq + K - 1) / K;
  const int q0 = slice * per, q1 = min(Lq, q0 + per);
  const int LP = L * P;
  const long n = (long)max(q1 - q0, 0) * LP;
  for (long t = tid; t < n; t += 1024) {
    const int i = (int)(t % LP), l = i / P;
    const long row = (long)b * Lq + q0 + t / LP;
    const long sidx = (row * M + m) * LP + i;
    const float2 xy = *reinterpret_cast<const float2*>(loc + 2 * sidx);
    const BwdCorners g = bwd_corners(xy.x, xy.y, (int)shapes[2 * l], (int)shapes[2 * l + 1], (long)lstart[l]);
    if (!g.inside) continue;
    const float a = FILL ? attn[sidx] : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!g.ok[k]) continue;
      if (FILL) {
        const int slot = atomicAdd(&bwd_lds[g.pix[k]], 1);
        BwdRec r;
        r.row = (int)row;
        r.w = g.c[k] * a;
        recs[slot] = r;
      } else {
        atomicAdd(&bwd_lds[g.pix[k]], 1);
      }
    }
  }
  if (!FILL) {
    __syncthreads();
    for (int s = tid; s < S; s += 1024) mine[s] = bwd_lds[s];
  }
}

// step 2: exclusive scan over the bins in the order (plane, pixel row, slice) -- the segments of a destination end up adjacent, slice by
// slice -- 4096 entries per workgroup: (a) workgroup totals, (b) scan of the totals by one workgroup, (c) local scan + total prefix,
// written back in place; the slot where a destination's first segment starts also goes to starts[plane * S + s], the grand total to
// starts[planes * S].  Entry j lives at bins[(plane * K + slice) * S + s]: with K = 16 = the entries per thread, the lanes of a wave read
// consecutive pixel rows of one slice.
constexpr int kScanPer = 16;                               // entries per thread, 256 threads per workgroup

__device__ __forceinline__ long bin_addr(long j, int S, int K) {
  const long per_plane = (long)S * K;
  const long plane = j / per_plane, rem = j % per_plane;
  return (plane * K + rem % K) * S + rem / K;
}

__device__ __forceinline__ int block_exclusive_256(int v, int* sh, int& total) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int u = tid >= off ? sh[tid - off] : 0;
    __syncthreads();
    sh[tid] += u;
    __syncthreads();
  }
  total = sh[255];
  return sh[tid] - v;
}

__global__ __launch_bounds__(256) void msda_bwd_scan_totals_kernel(const int* __restrict__ bins, int* __restrict__ totals, long n, int S, int K) {
  __shared__ int sh[256];
  const long base = ((long)blockIdx.x * 256 + threadIdx.x) * kScanPer;
  int s = 0;
  for (int k = 0; k < kScanPer; ++k) s += base + k < n ? bins[bin_addr(base + k, S, K)] : 0;
  int total;
  block_exclusive_256(s, sh, total);
  if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void msda_bwd_scan_top_kernel(int* __restrict__ totals, int nblocks) {
  __shared__ int sh[256];
  int carry = 0;
  for (int base = 0; base < nblocks; base += 256) {       // chunks of 256 totals, in sequence
    const int idx = base + threadIdx.x;
    const int v = idx < nblocks ? totals[idx] : 0;
    int total;
    const int ex = block_exclusive_256(v, sh, total);
    if (idx < nblocks) totals[idx] = carry + ex;
    carry += total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void msda_bwd_scan_apply_kernel(int* __restrict__ bins, int* __restrict__ starts,
                                                                  const int* __restrict__ totals, long n, int S, int K) {
  __shared__ int sh[256];
  const long base = ((long)blockIdx.x * 256 + threadIdx.x) * kScanPer;
  int c[kScanPer], s = 0;
  for (int k = 0; k < kScanPer; ++k) {
    c[k] = base + k < n ? bins[bin_addr(base + k, S, K)] : 0;
    s += c[k];
  }
  int total;
  int run = block_exclusive_256(s, sh, total) + totals[blockIdx.x];
  for (int k = 0; k < kScanPer; ++k) {
    const long j = base + k;
    if (j < n) {
      bins[bin_addr(j, S, K)] = run;
      if (j % K == 0) starts[j / K] = run;
      if (j == n - 1) starts[n / K] = run + c[k];
    }
    run += c[k];
  }
}

// 8-lane groups of steps 4 and 5 -> (row t of T, head m).  With M a multiple of 8, workgroup w (which the dispatcher places on XCD w % 8)
// takes head w % 8 (+ 8 j) and 32 consecutive rows: an XCD then reads only its own heads' eighth of grad_out / value (5.6 of 44.5 MB at
// B = 2: close to its 4 MB L2) instead of streaming all of it from the Infinity Cache.  A placement hint: results do not depend on it.
__device__ __forceinline__ bool bwd_group(long T, int M, long& t, int& m) {
  const int gi = threadIdx.x >> 3;
  if ((M & 7) == 0) {
    const long w = blockIdx.x, idx = w >> 3;
    const int mg = M >> 3;
    m = (int)(w & 7) + 8 * (int)(idx % mg);
    t = (idx / mg) * 32 + gi;
  } else {
    const long g = (long)blockIdx.x * 32 + gi;
    m = (int)(g % M);
    t = g / M;
  }
  const bool live = t < T;
  if (!live) { t = 0; m = 0; }
  return live;
}
static unsigned bwd_group_grid(long T, int M) {
  return (M & 7) == 0 ? (unsigned)(M * ((T + 31) / 32)) : (unsigned)((T * M + 31) / 32);
}

// step 4: a group of 8 lanes owns destination (b, s, m); rows t = (S - 1 - s) * B + b: the LAST pixel rows first (the coarse levels, whose
// lists are the longest, start first).
__global__ __launch_bounds__(256) void msda_bwd_gather_kernel(const float* __restrict__ gout, const int* __restrict__ starts,
                                                              const BwdRec* __restrict__ recs, float* __restrict__ gvalue, int B, int S,
                                                              int M) {
  const int sub = threadIdx.x & 7;
  long t;
  int m;
  const bool live = bwd_group((long)B * S, M, t, m);
  const int b = (int)(t % B);
  const long s = (long)S - 1 - t / B;
  const long d = ((long)b * S + s) * M + m;            // row of grad_value
  const long dp = ((long)b * M + m) * S + s;           // the destination in the scan's (plane, pixel row) order
  const int start = starts[dp];
  const int n = live ? starts[dp + 1] - start : 0;
  const float* gsrc = gout + (long)m * 32 + sub * 4;
  const long grow = (long)M * 32;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < n; r += 8) {
    BwdRec rc;
    rc.row = 0;                                       // padding records: weight 0, a row that exists
    rc.w = 0.f;
    if (r + sub < n) rc = recs[start + r + sub];      // 8 records = one 64-byte piece per group
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = __shfl(rc.row, j, 8);
      const float w = __shfl(rc.w, j, 8);
      const float4 gv = *reinterpret_cast<const float4*>(gsrc + (long)row * grow);
      acc.x = fmaf(w, gv.x, acc.x);
      acc.y = fmaf(w, gv.y, acc.y);
      acc.z = fmaf(w, gv.z, acc.z);
      acc.w = fmaf(w, gv.w, acc.w);
    }
  }
  if (live) *reinterpret_cast<float4*>(gvalue + d * 32 + sub * 4) = acc;
}

// step 5: item = (b, q, head) on 8 lanes, lane `sub` holds channels 4 sub .. 4 sub + 3
__global__ __launch_bounds__(256) void msda_bwd_coef_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                            const int64_t* __restrict__ lstart, const float* __restrict__ loc,
                                                            const float* __restrict__ attn, const float* __restrict__ gout,
                                                            float* __restrict__ gloc, float* __restrict__ gattn, int S, int M, int L,
                                                            int Lq, int P, long items) {
  const int sub = threadIdx.x & 7;
  long bq;
  int m;
  const bool live = bwd_group(items / M, M, bq, m);
  const long item = bq * M + m;
  const int b = (int)(bq / Lq);
  const int LP = L * P;
  const long row = (long)M * 32;
  const float* vb = value + (long)b * S * row + (long)m * 32 + sub * 4;
  const float4 g4 = *reinterpret_cast<const float4*>(gout + item * 32 + sub * 4);
  const float* lp = loc + item * (long)LP * 2;
  const float* wp = attn + item * (long)LP;
  const float gc[4] = {g4.x, g4.y, g4.z, g4.w};
  // branch-free body (masked corners and outside points read a legal address and are weighted 0), four points at a time so that their 16
  // corner reads are in flight together
  constexpr int U = 4;                                // 136 VGPRs; 1 / 2 / 8 points at a time: 0.39 / 0.34 / 0.33 ms against 0.28
  for (int i0 = 0; i0 < LP; i0 += U) {
    BwdCorners g[U];
    float a[U];
    float4 v[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = min(i0 + u, LP - 1), l = i / P;
      a[u] = wp[i];
      g[u] = bwd_corners(lp[2 * i], lp[2 * i + 1], (int)shapes[2 * l], (int)shapes[2 * l + 1], (long)lstart[l]);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[u][k] = *reinterpret_cast<const float4*>(vb + g[u].pix[k] * row);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float m0 = g[u].ok[0] ? 1.f : 0.f, m1 = g[u].ok[1] ? 1.f : 0.f, m2 = g[u].ok[2] ? 1.f : 0.f, m3 = g[u].ok[3] ? 1.f : 0.f;
      const float v0[4] = {v[u][0].x * m0, v[u][0].y * m0, v[u][0].z * m0, v[u][0].w * m0};
      const float v1[4] = {v[u][1].x * m1, v[u][1].y * m1, v[u][1].z * m1, v[u][1].w * m1};
      const float v2[4] = {v[u][2].x * m2, v[u][2].y * m2, v[u][2].z * m2, v[u][2].w * m2};
      const float v3[4] = {v[u][3].x * m3, v[u][3].y * m3, v[u][3].z * m3, v[u][3].w * m3};
      float sa = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float tg = gc[c] * a[u];
        sa += gc[c] * (g[u].c[0] * v0[c] + g[u].c[1] * v1[c] + g[u].c[2] * v2[c] + g[u].c[3] * v3[c]);
        sx += tg * (g[u].hh * (v1[c] - v0[c]) + g[u].lh * (v3[c] - v2[c]));        // d / d w_im
        sy += tg * (g[u].hw * (v2[c] - v0[c]) + g[u].lw * (v3[c] - v1[c]));        // d / d h_im
      }
#pragma unroll
      for (int sft = 4; sft > 0; sft >>= 1) {
        sa += __shfl_xor(sa, sft);
        sx += __shfl_xor(sx, sft);
        sy += __shfl_xor(sy, sft);
      }
      const int i = i0 + u;
      if (live && sub == 0 && i < LP) {
        gattn[item * LP + i] = sa;
        gloc[(item * LP + i) * 2] = (float)g[u].W * sx;
        gloc[(item * LP + i) * 2 + 1] = (float)g[u].H * sy;
      }
    }
  }
}

constexpr long kBwdMaxRows = 160 * 1024 / 4;           // pixel rows per image whose counters fit the LDS of a CU

struct BwdWorkspace {
  size_t bins, starts, totals, recs, bytes;
  long ndest, nbins, nblocks, samples;
  int K;                                                 // query slices per (image, head) plane
};

static BwdWorkspace bwd_workspace(long B, long S, long M, long L, long Lq, long P) {
  BwdWorkspace w;
  const long planes = B * M;
  w.K = (int)std::min<long>(16, std::max<long>(1, 256 / planes));      // one workgroup per CU holds the S counters of its plane
  w.ndest = planes * S;
  w.nbins = w.ndest * w.K;
  w.nblocks = (w.nbins + 256 * kScanPer - 1) / (256 * kScanPer);
  w.samples = B * Lq * M * L * P;
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  w.bins = 0;
  w.starts = up(w.bins + (size_t)w.nbins * 4);
  w.totals = up(w.starts + (size_t)(w.ndest + 1) * 4);
  w.recs = up(w.totals + (size_t)w.nblocks * 4);
  w.bytes = up(w.recs + (size_t)w.samples * 4 * sizeof(BwdRec));
  return w;
}

static int launch_msda_bwd_gather(const void* value, const int64_t* shapes, const int64_t* lstart, const void* loc, const void* attn,
                                  const void* gout, void* gvalue, void* gloc, void* gattn, int B, int S, int M, int L, int Lq, int P,
                                  void* workspace, hipStream_t st) {
  const BwdWorkspace w = bwd_workspace(B, S, M, L, Lq, P);
  char* ws = (char*)workspace;
  int* bins = (int*)(ws + w.bins);
  int* starts = (int*)(ws + w.starts);
  int* totals = (int*)(ws + w.totals);
  BwdRec* recs = (BwdRec*)(ws + w.recs);
  const dim3 block(256), bin_grid((unsigned)(B * M * w.K)), bin_block(1024);
  const size_t lds = (size_t)S * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)msda_bwd_bin_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)msda_bwd_bin_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((msda_bwd_bin_kernel<false>), bin_grid, bin_block, lds, st, shapes, lstart, (const float*)loc, (const float*)attn, bins,
                     recs, S, M, L, Lq, P, w.K);
  hipLaunchKernelGGL(msda_bwd_scan_totals_kernel, dim3((unsigned)w.nblocks), block, 0, st, bins, totals, w.nbins, S, w.K);
  hipLaunchKernelGGL(msda_bwd_scan_top_kernel, dim3(1), block, 0, st, totals, (int)w.nblocks);
  hipLaunchKernelGGL(msda_bwd_scan_apply_kernel, dim3((unsigned)w.nblocks), block, 0, st, bins, starts, totals, w.nbins, S, w.K);
  hipLaunchKernelGGL((msda_bwd_bin_kernel<true>), bin_grid, bin_block, lds, st, shapes, lstart, (const float*)loc, (const float*)attn, bins,
                     recs, S, M, L, Lq, P, w.K);
  hipLaunchKernelGGL(msda_bwd_gather_kernel, dim3(bwd_group_grid((long)B * S, M)), block, 0, st, (const float*)gout, starts, recs,
                     (float*)gvalue, B, S, M);
  const long items = (long)B * Lq * M;
  hipLaunchKernelGGL(msda_bwd_coef_kernel, dim3(bwd_group_grid((long)B * Lq, M)), block, 0, st, (const float*)value, shapes, lstart,
                     (const float*)loc, (const float*)attn, (const float*)gout, (float*)gloc, (float*)gattn, S, M, L, Lq, P, items);
  return check_launch("msda_backward_ws");
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

extern "C" int64_t hipie_msda_backward_workspace(int B, int S, int M, int L, int Lq, int P) {
  if (B <= 0 || S <= 0 || M <= 0 || L <= 0 || Lq < 0 || P <= 0) return 0;
  return (int64_t)hipie::bwd_workspace(B, S, M, L, Lq, P).bytes;
}

extern "C" int hipie_msda_backward_ws(const void* value, const int64_t* spatial_shapes, const int64_t* level_start, const void* sampling_loc,
                                      const void* attn_weight, const void* grad_output, void* grad_value, void* grad_sampling_loc,
                                      void* grad_attn_weight, int B, int S, int M, int D, int L, int Lq, int P, int dtype, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  using namespace hipie;
  if (dtype != HIPIE_F32 || D != 32 || B <= 0 || Lq <= 0 || S > hipie::kBwdMaxRows)   // not covered by the gather form: the atomic kernel, same results
    return hipie_msda_backward(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                               grad_attn_weight, B, S, M, D, L, Lq, P, dtype, stream);
  HIPIE_REQUIRE(S > 0 && M > 0 && L > 0 && P > 0, "msda_backward_ws: bad shape");
  HIPIE_REQUIRE(value && spatial_shapes && level_start && grad_value && sampling_loc && attn_weight && grad_output && grad_sampling_loc &&
                    grad_attn_weight, "msda_backward_ws: null pointer");
  const BwdWorkspace w = bwd_workspace(B, S, M, L, Lq, P);
  HIPIE_REQUIRE(w.samples * 4 < (1L << 31) && (long)B * Lq < (1L << 31), "msda_backward_ws: more than 2^31 corner records");
  HIPIE_REQUIRE(workspace && workspace_bytes >= (int64_t)w.bytes, "msda_backward_ws: workspace of %lld bytes, %lld needed (hipie_msda_backward_workspace)",
                (long long)workspace_bytes, (long long)w.bytes);
  HIPIE_REQUIRE(((uintptr_t)workspace & 15) == 0, "msda_backward_ws: workspace must be 16-byte aligned");
  return launch_msda_bwd_gather(value, spatial_shapes, level_start, sampling_loc, attn_weight, grad_output, grad_value, grad_sampling_loc,
                                grad_attn_weight, B, S, M, L, Lq, P, workspace, (hipStream_t)stream);
}

// msda_study.hip -- study copy of msda_d32_kernel<float, float, FUSED = true> (hipie_amd/csrc/msda.hip) with switchable parts, for the
// co-residency fault of DESIGN.md section 9 (wrong (query, head) groups when the workgroups share a CU with gemm_kernel<256>).
// NOT product code: built by tools/ubench/Makefile.msda_study into tools/ubench/libmsda_study.so, driven by tools/msda_study.py.
//
// flag bits (template parameter V):
//   1   DUMP      every lane writes, per point it owns, 20 words to dbg: record (4 offsets, 4 weights), raw x, y, r0..r3, logit, wmax,
//                 winv, H, W, lstart -- so a wrong group can be traced to the load or the arithmetic that produced it
//   2   NOSOFT    logits are taken as ready attention weights (no max / exp / sum / reciprocal)
//   4   NOLOC     offsets are taken as ready sampling locations (no ref reads, no divisions)
//   8   NOTRANS   exp / reciprocal without v_exp_f32 / v_rcp_f32 (polynomial + Newton)
//   16  SHAPELDS  shapes / lstart copied to LDS once per workgroup (behind a block barrier) instead of per-point global loads
//   32  STATICLDS static __shared__ array (L * P == 16) instead of extern
//   64  BLOCKBAR  __syncthreads() between the phases instead of the wave barrier
//   128 TWICE     phase 1 is run twice into two record sets; phase 2 uses the second; differences between the two are counted in dbg
//   256 DPP       group reductions by DPP row operations instead of ds_bpermute
//   512 FLOATMASK corner validity as four 0/1 floats multiplied into the weights (no chain of lane-mask ANDs into VCC)
//   1024 ASMTIGHT the four corner selects as the product's ISA has them, pinned in inline asm: s_and_b64 vcc / v_cndmask_b32 vcc back to back
//   2048 ASMRAW   the same with s_nop 1 between each s_and_b64 vcc and its v_cndmask
//   4096 ASMWAR   the same with s_nop 1 between each v_cndmask and the NEXT s_and_b64 vcc
//   8192 ASMNOVCC four s_and_b64 into four SGPR pairs, four v_cndmask_b32_e64 (VCC not used)
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PointRec {
  int o[4];
  float w[4];
};

#define CHAIN_HEAD "v_cmp_lt_i32_e64 %[m0], -1, %[a]\n\tv_cmp_lt_i32_e64 %[m1], %[a], %[hb]\n\tv_cmp_lt_i32_e64 %[m3], %[b], %[wb]\n\tv_cmp_lt_i32_e64 %[m2], -1, %[b]\n\ts_nop 3\n\t"
#define CHAIN_OUT [r0] "=&v"(r.w[0]), [r1] "=&v"(r.w[1]), [r2] "=&v"(r.w[2]), [r3] "=&v"(r.w[3]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
#define CHAIN_IN [a] "v"(h0), [b] "v"(w0), [hb] "v"(H - 1), [wb] "v"(W - 1), [x0] "v"(p0), [x1] "v"(p1), [x2] "v"(p2), [x3] "v"(p3)
template <int V>
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
  if (V & 512) {                         // validity as floats: one select per condition, no chain of lane-mask ANDs
    const float fh0 = okh0 ? 1.f : 0.f, fh1 = okh1 ? 1.f : 0.f, fw0 = okw0 ? 1.f : 0.f, fw1 = okw1 ? 1.f : 0.f;
    r.w[0] = hh * hw * aw * (fh0 * fw0);
    r.w[1] = hh * lw * aw * (fh0 * fw1);
    r.w[2] = lh * hw * aw * (fh1 * fw0);
    r.w[3] = lh * lw * aw * (fh1 * fw1);
  } else if (V & (1024 | 2048 | 4096 | 8192)) {
    const float awi = inside ? aw : 0.f;
    const float p0 = hh * hw * awi, p1 = hh * lw * awi, p2 = lh * hw * awi, p3 = lh * lw * awi;
    unsigned long m0, m1, m2, m3, t0, t1, t2, t3;
    if (V & 1024)
      asm volatile(CHAIN_HEAD "s_and_b64 vcc, %[m0], %[m2]\n\tv_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\t"
                   "s_and_b64 vcc, %[m1], %[m2]\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\t"
                   "s_and_b64 vcc, %[m0], %[m3]\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\t"
                   "s_and_b64 vcc, %[m1], %[m3]\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t" : CHAIN_OUT : CHAIN_IN : "vcc");
    else if (V & 2048)
      asm volatile(CHAIN_HEAD "s_and_b64 vcc, %[m0], %[m2]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\t"
                   "s_and_b64 vcc, %[m1], %[m2]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\t"
                   "s_and_b64 vcc, %[m0], %[m3]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\t"
                   "s_and_b64 vcc, %[m1], %[m3]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t" : CHAIN_OUT : CHAIN_IN : "vcc");
    else if (V & 4096)
      asm volatile(CHAIN_HEAD "s_and_b64 vcc, %[m0], %[m2]\n\tv_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\ts_nop 1\n\t"
                   "s_and_b64 vcc, %[m1], %[m2]\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\ts_nop 1\n\t"
                   "s_and_b64 vcc, %[m0], %[m3]\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\ts_nop 1\n\t"
                   "s_and_b64 vcc, %[m1], %[m3]\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t" : CHAIN_OUT : CHAIN_IN : "vcc");
    else
      asm volatile(CHAIN_HEAD "s_and_b64 %[t0], %[m0], %[m2]\n\ts_and_b64 %[t2], %[m1], %[m2]\n\ts_and_b64 %[t1], %[m0], %[m3]\n\t"
                   "s_and_b64 %[t3], %[m1], %[m3]\n\tv_cndmask_b32_e64 %[r0], 0, %[x0], %[t0]\n\tv_cndmask_b32_e64 %[r2], 0, %[x2], %[t2]\n\t"
                   "v_cndmask_b32_e64 %[r1], 0, %[x1], %[t1]\n\tv_cndmask_b32_e64 %[r3], 0, %[x3], %[t3]\n\t"
                   : CHAIN_OUT, [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3) : CHAIN_IN);
  } else {
    r.w[0] = (okh0 && okw0) ? hh * hw * aw : 0.f;
    r.w[1] = (okh0 && okw1) ? hh * lw * aw : 0.f;
    r.w[2] = (okh1 && okw0) ? lh * hw * aw : 0.f;
    r.w[3] = (okh1 && okw1) ? lh * lw * aw : 0.f;
  }
  return r;
}

template <int V> __device__ __forceinline__ float xor_lane(float v, int mask) {
  if (V & 256) {
    // quad_perm / row operations inside 8 lanes: xor 1 = quad_perm [1,0,3,2], xor 2 = quad_perm [2,3,0,1], xor 4 = row_half_mirror composed
    int iv = __float_as_int(v), r;
    if (mask == 1) r = __builtin_amdgcn_mov_dpp(iv, 0xB1, 0xF, 0xF, true);
    else if (mask == 2) r = __builtin_amdgcn_mov_dpp(iv, 0x4E, 0xF, 0xF, true);
    else {
      // row_half_mirror (0x141) maps lane i of each 8 to 7 - i; then reverse inside the quad (quad_perm [3,2,1,0] = 0x1B) -> i ^ 4
      r = __builtin_amdgcn_mov_dpp(iv, 0x141, 0xF, 0xF, true);
      r = __builtin_amdgcn_mov_dpp(r, 0x1B, 0xF, 0xF, true);
    }
    return __int_as_float(r);
  }
  return __shfl_xor(v, mask);
}
template <int V> __device__ __forceinline__ float group_max8(float v) {
  v = fmaxf(v, xor_lane<V>(v, 1)); v = fmaxf(v, xor_lane<V>(v, 2)); v = fmaxf(v, xor_lane<V>(v, 4));
  return v;
}
template <int V> __device__ __forceinline__ float group_sum8(float v) {
  v += xor_lane<V>(v, 1); v += xor_lane<V>(v, 2); v += xor_lane<V>(v, 4);
  return v;
}

// exp without v_exp_f32: 2^(x log2 e) = 2^n * 2^f, f in [-0.5, 0.5], degree-6 polynomial (rel. error ~ 2e-7: the study compares a
// variant with ITSELF run alone, so only determinism matters)
__device__ __forceinline__ float exp_notrans(float x) {
  float t = x * 1.44269504088896341f;
  t = fmaxf(t, -126.f);
  const float n = rintf(t);
  const float f = t - n;
  float p = 1.5403530393381609954e-4f;
  p = fmaf(p, f, 1.3333558146428443423e-3f);
  p = fmaf(p, f, 9.6181291076284771619e-3f);
  p = fmaf(p, f, 5.5504108664821579953e-2f);
  p = fmaf(p, f, 2.4022650695910071233e-1f);
  p = fmaf(p, f, 6.9314718055994530942e-1f);
  p = fmaf(p, f, 1.f);
  return __int_as_float(__float_as_int(p) + ((int)n << 23));
}
__device__ __forceinline__ float rcp_notrans(float d) {       // d > 0
  float r = __int_as_float(0x7EF311C7 - __float_as_int(d));
  for (int i = 0; i < 4; ++i) r = r * (2.f - d * r);
  return r;
}

template <int V>
__device__ __forceinline__ void publish(float* rec, int sub, const float* lp, const float* wp, const float* refrow, int ref_dim,
                                        const int64_t* shapes, const int64_t* lstart, const int* sh_lds, int L, int P, long row,
                                        float* dbg) {
  const int LP = L * P;
  float wmax = 0.f, winv = 1.f;
  if (!(V & 2)) {
    float mx = -INFINITY;
    for (int i = sub; i < LP; i += 8) mx = fmaxf(mx, wp[i]);
    wmax = group_max8<V>(mx);
    float sm = 0.f;
    for (int i = sub; i < LP; i += 8) sm += (V & 8) ? exp_notrans(wp[i] - wmax) : expf(wp[i] - wmax);
    const float tot = group_sum8<V>(sm);
    winv = (V & 8) ? rcp_notrans(tot) : 1.f / tot;
  }
  for (int i = sub; i < LP; i += 8) {
    const int l = i / P;
    int H, W;
    long ls;
    if (V & 16) {
      H = sh_lds[3 * l]; W = sh_lds[3 * l + 1]; ls = sh_lds[3 * l + 2];
    } else {
      H = (int)shapes[2 * l]; W = (int)shapes[2 * l + 1]; ls = lstart[l];
    }
    const float xr = lp[2 * i], yr = lp[2 * i + 1], lg = wp[i];
    float x = xr, y = yr, aw, r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
    if (!(V & 4)) {
      const float* r = refrow + l * ref_dim;
      r0 = r[0]; r1 = r[1];
      if (ref_dim == 2) {
        if (V & 8) { x = r0 + x * rcp_notrans((float)W); y = r1 + y * rcp_notrans((float)H); }
        else { x = r0 + x / (float)W; y = r1 + y / (float)H; }
      } else {
        r2 = r[2]; r3 = r[3];
        if (V & 8) { x = r0 + x * rcp_notrans((float)P) * r2 * 0.5f; y = r1 + y * rcp_notrans((float)P) * r3 * 0.5f; }
        else { x = r0 + x / (float)P * r2 * 0.5f; y = r1 + y / (float)P * r3 * 0.5f; }
      }
    }
    if (!(V & 2)) aw = ((V & 8) ? exp_notrans(lg - wmax) : expf(lg - wmax)) * winv;
    else aw = lg;
    const PointRec pr = point_record<V>(x, y, H, W, ls, row, aw);
    *reinterpret_cast<int4*>(rec + i * 8) = make_int4(pr.o[0], pr.o[1], pr.o[2], pr.o[3]);
    *reinterpret_cast<float4*>(rec + i * 8 + 4) = make_float4(pr.w[0], pr.w[1], pr.w[2], pr.w[3]);
    if ((V & 1) && dbg) {
      float* d = dbg + i * 20;
      for (int c = 0; c < 4; ++c) { d[c] = __int_as_float(pr.o[c]); d[4 + c] = pr.w[c]; }
      d[8] = xr; d[9] = yr; d[10] = r0; d[11] = r1; d[12] = r2; d[13] = r3; d[14] = lg; d[15] = wmax; d[16] = winv;
      d[17] = (float)H; d[18] = (float)W; d[19] = (float)ls;
    }
  }
}

template <int V>
__global__ __launch_bounds__(256) void msda_study_kernel(const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                         const int64_t* __restrict__ lstart, const float* __restrict__ off,
                                                         const float* __restrict__ logit, const float* __restrict__ ref,
                                                         float* __restrict__ out, float* __restrict__ dbg, int S, int M, int L, int Lq,
                                                         int P, int ref_dim, long total_groups, long off_stride, long w_stride, long vrow) {
  extern __shared__ __attribute__((aligned(16))) float dyn_recs[];
  __shared__ __attribute__((aligned(16))) float st_recs[(V & 32) ? 32 * (16 * 8 + 4) * ((V & 128) ? 2 : 1) : 4];
  __shared__ int sh_lds[(V & 16) ? 3 * 8 : 1];
  float* recs = (V & 32) ? st_recs : dyn_recs;
  const long nblk = gridDim.x, qn = nblk >> 3, rn = nblk & 7;
  const long xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const long blk = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  if (V & 16) {
    if (threadIdx.x < L) {
      sh_lds[3 * threadIdx.x] = (int)shapes[2 * threadIdx.x];
      sh_lds[3 * threadIdx.x + 1] = (int)shapes[2 * threadIdx.x + 1];
      sh_lds[3 * threadIdx.x + 2] = (int)lstart[threadIdx.x];
    }
    __syncthreads();
  }
  const long chunk = blk / M;
  const int mh = (int)(blk - chunk * M);
  const long bqn = chunk * 32 + (threadIdx.x >> 3);
  const bool live = bqn < total_groups / M;
  const long g = (live ? bqn : total_groups / M - 1) * M + mh;
  const int sub = threadIdx.x & 7;
  const int m = (int)(g % M);
  const long bq = g / M;
  const int b = (int)(bq / Lq);
  const int LP = L * P;
  const long row = vrow;
  const int gstride = LP * 8 + 4;
  float* rec = recs + (threadIdx.x >> 3) * gstride;
  float* dg = (V & 1) && live ? dbg + g * (long)LP * 20 : nullptr;
  publish<V>(rec, sub, off + bq * off_stride + (long)m * (LP * 2), logit + bq * w_stride + (long)m * LP, ref + bq * L * ref_dim, ref_dim,
             shapes, lstart, sh_lds, L, P, row, dg);
  if (V & 128) {
    float* rec2 = rec + 32 * gstride;
    asm volatile("s_sleep 8" ::: "memory");
    publish<(V & ~1)>(rec2, sub, off + bq * off_stride + (long)m * (LP * 2), logit + bq * w_stride + (long)m * LP, ref + bq * L * ref_dim,
                      ref_dim, shapes, lstart, sh_lds, L, P, row, nullptr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int nd = 0;
    for (int i = sub; i < LP; i += 8)
      for (int c = 0; c < 8; ++c) nd += __float_as_int(rec[i * 8 + c]) != __float_as_int(rec2[i * 8 + c]);
    if (nd && dbg) atomicAdd(reinterpret_cast<int*>(dbg + total_groups * (long)LP * 20), nd);   // the word behind the dump counts differing record words
    rec = rec2;
  }
  if (V & 64) __syncthreads();
  else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* vb = value + (long)b * S * row + m * 32 + sub * 4;
#pragma unroll 2
  for (int i = 0; i < LP; ++i) {
    const int4 o = *reinterpret_cast<const int4*>(rec + i * 8);
    const float4 w = *reinterpret_cast<const float4*>(rec + i * 8 + 4);
    const float4 v1 = *reinterpret_cast<const float4*>(vb + o.x);
    const float4 v2 = *reinterpret_cast<const float4*>(vb + o.y);
    const float4 v3 = *reinterpret_cast<const float4*>(vb + o.z);
    const float4 v4 = *reinterpret_cast<const float4*>(vb + o.w);
    acc[0] = fmaf(w.w, v4.x, fmaf(w.z, v3.x, fmaf(w.y, v2.x, fmaf(w.x, v1.x, acc[0]))));
    acc[1] = fmaf(w.w, v4.y, fmaf(w.z, v3.y, fmaf(w.y, v2.y, fmaf(w.x, v1.y, acc[1]))));
    acc[2] = fmaf(w.w, v4.z, fmaf(w.z, v3.z, fmaf(w.y, v2.z, fmaf(w.x, v1.z, acc[2]))));
    acc[3] = fmaf(w.w, v4.w, fmaf(w.z, v3.w, fmaf(w.y, v2.w, fmaf(w.x, v1.w, acc[3]))));
  }
  if (live) *reinterpret_cast<float4*>(out + g * 32 + sub * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

#define CASE(v)                                                                                                                     \
  case v:                                                                                                                           \
    hipLaunchKernelGGL((msda_study_kernel<v>), dim3((unsigned)blocks), dim3(256), ((v) & 32) ? 0 : lds * (((v) & 128) ? 2 : 1), st, value, \
                       shapes, lstart, off, logit, ref, out, dbg, S, M, L, Lq, P, ref_dim, groups, off_stride, w_stride, (long)M * 32);    \
    break;

extern "C" int msda_study(int flags, const float* value, const int64_t* shapes, const int64_t* lstart, const float* ref, const float* off,
                          const float* logit, float* out, float* dbg, int B, int S, int M, int L, int Lq, int P, int ref_dim, void* stream) {
  const long groups = (long)B * Lq * M;
  const long blocks = (((long)B * Lq + 31) / 32) * M;
  const size_t lds = (size_t)32 * (L * P * 8 + 4) * sizeof(float);
  const long off_stride = (long)M * L * P * 2, w_stride = (long)M * L * P;
  hipStream_t st = (hipStream_t)stream;
  switch (flags) {
    CASE(0) CASE(1) CASE(2) CASE(4) CASE(6) CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256) CASE(48) CASE(24) CASE(129)
    CASE(2 + 16) CASE(4 + 16) CASE(8 + 16 + 32 + 256) CASE(512) CASE(1024) CASE(2048) CASE(4096) CASE(8192)
    default: return -1;
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

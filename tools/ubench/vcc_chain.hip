// vcc_chain.hip -- torch-free reproducer for the co-residency fault of hipie_msda_fused.
//
// tools/msda_study.py (DUMP variant) showed WHAT goes wrong in the fused MSDA kernel beside gemm_kernel<256>: for one 16-lane quarter of a
// wave, in one loop iteration, the second corner weight (okh0 && okw1 ? hh * lw * aw : 0) comes out 0 -- every load and every other
// computed value is right.  In the ISA that weight is the third link of
//     s_and_b64 vcc, m, m' ; v_cndmask_b32 w, 0, x, vcc        (x 4, back to back: SALU writes VCC, the next VALU reads it)
// This program runs that chain (inline asm, so the schedule is pinned) in a self-checking loop on one stream while an MFMA kernel runs on
// another, with variants that space the chain out, to find which adjacency the hardware does not interlock when the SIMD co-issues an MFMA
// of another wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/vcc_chain.hip -o tools/ubench/vcc_chain -ldl
//   tools/ubench/vcc_chain [path to libhipie_mi355.so]      (with the library: hipie_gemm K = 256 is a second aggressor)
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- aggressor: MFMA stream, 128 accumulator registers, no LDS: leaves room for victim waves on the same SIMD ------------------------
__global__ __launch_bounds__(256) void mfma_spin(float* sink, int iters) {
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = r; c1[r] = r + 1; c2[r] = r + 2; c3[r] = r + 3; }
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
  }
  if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.f) sink[0] = 1.f;
}


// ---- synthetic aggressors, one instruction class each (which part of gemm_kernel<256> disturbs the victim's packed multiply?) ------------
//   1 v_pk_fma_f32 / v_pk_mul_f32 stream   2 MFMA + packed fp32 interleaved   3 v_permlane32_swap stream   4 LDS-DMA (global_load_lds_dwordx4)
//   5 ds_read_b128 stream   6 v_cvt_pk_f16_f32 stream   7 MFMA + permlane32_swap   8 MFMA + LDS-DMA + ds_read_b128 (the k loop of the GEMM)
template <int K>
__global__ __launch_bounds__(256) void spin(float* sink, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x16 c0, c1;
  for (int r = 0; r < 16; ++r) { c0[r] = r; c1[r] = r + 1; }
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  f32x2 p = {1.0f + threadIdx.x * 1e-3f, 0.5f}, q = {0.999f, 1.001f}, r2 = {0.1f, 0.2f};
  unsigned u0 = threadIdx.x, u1 = threadIdx.x * 3;
  f32x4 acc4 = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
  __syncthreads();
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + (threadIdx.x >> 6) * 1024);
  for (int i = 0; i < iters; ++i) {
    if (K == 1 || K == 2) {
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %2, %2, %1\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %2, %2, %1" : "+v"(p), "+v"(q), "+v"(r2));
    }
    if (K == 2 || K == 7 || K == 8) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    }
    if (K == 3 || K == 7) asm volatile("v_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(u0), "+v"(u1));
    if (K == 4 || K == 8) {
      unsigned save;
      const float* g = src + ((blockIdx.x * 256 + threadIdx.x) * 4 + (i & 63) * 262144) % (1 << 24);
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(save) : "v"(g), "s"(lds_base) : "memory");
    }
    if (K == 5 || K == 8) acc4 += *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + i * 64) & 8188));
    if (K == 6) { asm volatile("v_cvt_pk_f16_f32 %0, %1, %2\n\tv_cvt_pk_f16_f32 %1, %0, %2" : "+v"(u0), "+v"(u1) : "v"(p[0])); }
    if (K == 8 && (i & 7) == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (c0[0] + c1[1] + p[0] + p[1] + r2[0] + (float)u0 + (float)u1 + acc4[0] + acc4[3] == 12345.f) sink[0] = 1.f;
}
static const float* g_src = nullptr;
static void launch_aggressor(int aggressor, hipStream_t sa, float* sink) {
  switch (aggressor) {
    case 0: hipLaunchKernelGGL(mfma_spin, dim3(4096), dim3(256), 0, sa, sink, 600); break;
    case 1: hipLaunchKernelGGL(spin<1>, dim3(4096), dim3(256), 0, sa, sink, g_src, 2400); break;
    case 2: hipLaunchKernelGGL(spin<2>, dim3(4096), dim3(256), 0, sa, sink, g_src, 1000); break;
    case 3: hipLaunchKernelGGL(spin<3>, dim3(4096), dim3(256), 0, sa, sink, g_src, 4000); break;
    case 4: hipLaunchKernelGGL(spin<4>, dim3(4096), dim3(256), 0, sa, sink, g_src, 400); break;
    case 5: hipLaunchKernelGGL(spin<5>, dim3(4096), dim3(256), 0, sa, sink, g_src, 4000); break;
    case 6: hipLaunchKernelGGL(spin<6>, dim3(4096), dim3(256), 0, sa, sink, g_src, 4000); break;
    case 7: hipLaunchKernelGGL(spin<7>, dim3(4096), dim3(256), 0, sa, sink, g_src, 1000); break;
    case 8: hipLaunchKernelGGL(spin<8>, dim3(4096), dim3(256), 0, sa, sink, g_src, 400); break;
  }
}
static const char* AGG[] = {"mfma_spin (MFMA only, no LDS)", "v_pk_fma_f32 / v_pk_mul_f32 stream", "MFMA + packed fp32", "v_permlane32_swap stream",
                            "LDS-DMA stream", "ds_read_b128 stream", "v_cvt_pk_f16_f32 stream", "MFMA + v_permlane32_swap", "MFMA + LDS-DMA + ds_read_b128 + barriers",
                            "hipie_gemm M 174080, N 256, K 256 (gemm_kernel<256>, fp32 rows)"};

// ---- victim ---------------------------------------------------------------------------------------------------------------------------
// err[0..3]: wrong r0..r3 (lanes x iterations); err[4..7]: wrong lanes by quarter of the wave; err[8]: got 0 where a value was expected;
// err[9]: got a value where 0 was expected
template <int V>
__global__ __launch_bounds__(256) void victim(unsigned int* err, int iters, int hb, int wb) {
  const unsigned lane = threadIdx.x & 63;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  unsigned e0 = 0, e1 = 0, e2 = 0, e3 = 0, ez = 0, en = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int a = (int)((s >> 8) % (unsigned)(hb + 3)) - 1;      // -1 .. hb + 1
    const int b = (int)((s >> 16) % (unsigned)(wb + 3)) - 1;
    const float x0 = __uint_as_float(0x3f800000u | (s & 0x7fffffu));        // in [1, 2): never 0
    const float x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    float r0, r1, r2, r3, f0 = x0, f1 = x1;
    unsigned long m0, m1, m2, m3, t0, t1, t2, t3;
    if (V == 0)        // the product's schedule
      asm volatile(
          "v_cmp_lt_i32_e64 %[m0], -1, %[a]\n\tv_cmp_lt_i32_e64 %[m1], %[a], %[hb]\n\tv_cmp_lt_i32_e64 %[m3], %[b], %[wb]\n\t"
          "v_cmp_lt_i32_e64 %[m2], -1, %[b]\n\t"
          "v_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "s_and_b64 vcc, %[m0], %[m2]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "v_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\t"
          "s_and_b64 vcc, %[m1], %[m2]\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\t"
          "s_and_b64 vcc, %[m0], %[m3]\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\t"
          "s_and_b64 vcc, %[m1], %[m3]\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t"
          : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),
            [f0] "+v"(f0), [f1] "+v"(f1)
          : [a] "v"(a), [b] "v"(b), [hb] "v"(hb), [wb] "v"(wb), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3)
          : "vcc");
    else if (V == 1)   // a wait state between each SALU write of VCC and its VALU reader
      asm volatile(
          "v_cmp_lt_i32_e64 %[m0], -1, %[a]\n\tv_cmp_lt_i32_e64 %[m1], %[a], %[hb]\n\tv_cmp_lt_i32_e64 %[m3], %[b], %[wb]\n\t"
          "v_cmp_lt_i32_e64 %[m2], -1, %[b]\n\t"
          "v_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "s_and_b64 vcc, %[m0], %[m2]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "v_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\t"
          "s_and_b64 vcc, %[m1], %[m2]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\t"
          "s_and_b64 vcc, %[m0], %[m3]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\t"
          "s_and_b64 vcc, %[m1], %[m3]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t"
          : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),
            [f0] "+v"(f0), [f1] "+v"(f1)
          : [a] "v"(a), [b] "v"(b), [hb] "v"(hb), [wb] "v"(wb), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3)
          : "vcc");
    else if (V == 2)   // a wait state between each VALU reader of VCC and the next SALU write (WAR)
      asm volatile(
          "v_cmp_lt_i32_e64 %[m0], -1, %[a]\n\tv_cmp_lt_i32_e64 %[m1], %[a], %[hb]\n\tv_cmp_lt_i32_e64 %[m3], %[b], %[wb]\n\t"
          "v_cmp_lt_i32_e64 %[m2], -1, %[b]\n\t"
          "v_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "s_and_b64 vcc, %[m0], %[m2]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "v_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\ts_nop 1\n\t"
          "s_and_b64 vcc, %[m1], %[m2]\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\ts_nop 1\n\t"
          "s_and_b64 vcc, %[m0], %[m3]\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\ts_nop 1\n\t"
          "s_and_b64 vcc, %[m1], %[m3]\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t"
          : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),
            [f0] "+v"(f0), [f1] "+v"(f1)
          : [a] "v"(a), [b] "v"(b), [hb] "v"(hb), [wb] "v"(wb), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3)
          : "vcc");
    else if (V == 3)   // no VCC: four SGPR pairs, e64 selects
      asm volatile(
          "v_cmp_lt_i32_e64 %[m0], -1, %[a]\n\tv_cmp_lt_i32_e64 %[m1], %[a], %[hb]\n\tv_cmp_lt_i32_e64 %[m3], %[b], %[wb]\n\t"
          "v_cmp_lt_i32_e64 %[m2], -1, %[b]\n\t"
          "v_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\tv_mul_f32 %[f0], %[f0], %[f1]\n\tv_mul_f32 %[f1], %[f1], %[f0]\n\t"
          "s_and_b64 %[t0], %[m0], %[m2]\n\ts_and_b64 %[t2], %[m1], %[m2]\n\ts_and_b64 %[t1], %[m0], %[m3]\n\ts_and_b64 %[t3], %[m1], %[m3]\n\t"
          "v_cndmask_b32_e64 %[r0], 0, %[x0], %[t0]\n\tv_cndmask_b32_e64 %[r2], 0, %[x2], %[t2]\n\t"
          "v_cndmask_b32_e64 %[r1], 0, %[x1], %[t1]\n\tv_cndmask_b32_e64 %[r3], 0, %[x3], %[t3]\n\t"
          : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),
            [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2), [t3] "=&s"(t3), [f0] "+v"(f0), [f1] "+v"(f1)
          : [a] "v"(a), [b] "v"(b), [hb] "v"(hb), [wb] "v"(wb), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3));
    else               // V == 4: the v_cmp results are consumed by the s_and IMMEDIATELY (VALU writes SGPR -> SALU reads it)
      asm volatile(
          "v_cmp_lt_i32_e64 %[m0], -1, %[a]\n\tv_cmp_lt_i32_e64 %[m1], %[a], %[hb]\n\tv_cmp_lt_i32_e64 %[m3], %[b], %[wb]\n\t"
          "v_cmp_lt_i32_e64 %[m2], -1, %[b]\n\t"
          "s_and_b64 vcc, %[m0], %[m2]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r0], 0, %[x0], vcc\n\ts_nop 1\n\t"
          "s_and_b64 vcc, %[m1], %[m2]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r2], 0, %[x2], vcc\n\ts_nop 1\n\t"
          "s_and_b64 vcc, %[m0], %[m3]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r1], 0, %[x1], vcc\n\ts_nop 1\n\t"
          "s_and_b64 vcc, %[m1], %[m3]\n\ts_nop 1\n\tv_cndmask_b32_e32 %[r3], 0, %[x3], vcc\n\t"
          : [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3),
            [f0] "+v"(f0), [f1] "+v"(f1)
          : [a] "v"(a), [b] "v"(b), [hb] "v"(hb), [wb] "v"(wb), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3)
          : "vcc");
    // expectation without lane masks: all-ones / all-zeros words from sign bits
    const unsigned h0 = ~(unsigned)(a >> 31), h1 = (unsigned)((a - hb) >> 31), w0 = ~(unsigned)(b >> 31), w1 = (unsigned)((b - wb) >> 31);
    const unsigned q0 = __float_as_uint(x0) & h0 & w0, q2 = __float_as_uint(x2) & h1 & w0, q1 = __float_as_uint(x1) & h0 & w1,
                   q3 = __float_as_uint(x3) & h1 & w1;
    const unsigned g0 = __float_as_uint(r0), g1 = __float_as_uint(r1), g2 = __float_as_uint(r2), g3 = __float_as_uint(r3);
    e0 += g0 != q0; e1 += g1 != q1; e2 += g2 != q2; e3 += g3 != q3;
    ez += (g0 == 0 && q0 != 0) + (g1 == 0 && q1 != 0) + (g2 == 0 && q2 != 0) + (g3 == 0 && q3 != 0);
    en += (g0 != 0 && q0 == 0) + (g1 != 0 && q1 == 0) + (g2 != 0 && q2 == 0) + (g3 != 0 && q3 == 0);
    s ^= __float_as_uint(f0 + f1) & 1u;                     // keep the filler alive
  }
  if (e0) atomicAdd(err + 0, e0);
  if (e1) atomicAdd(err + 1, e1);
  if (e2) atomicAdd(err + 2, e2);
  if (e3) atomicAdd(err + 3, e3);
  if (e0 + e1 + e2 + e3) atomicAdd(err + 4 + (lane >> 4), e0 + e1 + e2 + e3);
  if (ez) atomicAdd(err + 8, ez);
  if (en) atomicAdd(err + 9, en);
}

typedef int (*gemm_fn)(const void*, int64_t, const void*, int64_t, const float*, const float*, int64_t, void*, int64_t, const int32_t*, int, int,
                       int, int, int, int, float, float, void*);

typedef int (*gemm_fn)(const void*, int64_t, const void*, int64_t, const float*, const float*, int64_t, void*, int64_t, const int32_t*, int, int,
                       int, int, int, int, float, float, void*);

// ---- victim 2: v_pk_mul_f32 followed at once by VALU writes of its SOURCE registers (write-after-read) -----------------------------------
// In the failing ISA the corner products are  v_pk_mul_f32 v[18:19], v[0:1], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]  and the next two
// instructions are  v_cndmask_b32 v0, ... ; v_cndmask_b32 v1, ...  (the integer h0 / w0 land in the registers the packed multiply reads).
// If the packed multiply reads its operands late when the SIMD is busy with another wave's MFMA, lo = v0 * v19 sees the INTEGER in v0 --
// a denormal as a float -- and the weight underflows to 0: exactly the observed fault.
//   W 0: as the product (dst pair = src1 pair, WAR writers adjacent)   1: one s_nop 0 between   2: s_nop 1 between
//   3: distinct dst pair, writers adjacent   4: control, two v_mul_f32 instead of the packed multiply, writers adjacent
//   5: v_pk_add_f32 in place of the multiply   6: as 0 with v_mov_b32 writers
// err[0]: lo results wrong, err[1]: hi results wrong, err[2..5] by quarter, err[6]: wrong lo equals (written value) * b1, err[7]: hi likewise
template <int W>
__global__ __launch_bounds__(256) void victim_pk(unsigned int* err, int iters) {
  const unsigned lane = threadIdx.x & 63;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 777u;
  unsigned elo = 0, ehi = 0, klo = 0, khi = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const float a0 = __uint_as_float(0x3f000000u | (s & 0x7fffffu)), a1 = __uint_as_float(0x3f000000u | ((s >> 3) & 0x7fffffu));
    const float b0 = __uint_as_float(0x3f800000u | ((s >> 5) & 0x7fffffu)), b1 = __uint_as_float(0x3f800000u | ((s >> 7) & 0x7fffffu));
    const int k0 = (int)((s >> 9) & 127), k1 = (int)((s >> 17) & 127);           // small integers, as h0 / w0 are
    float c0, c1;
    unsigned long m;
#define PK_SETUP "v_mov_b32 v100, %[a0]\n\tv_mov_b32 v101, %[a1]\n\tv_mov_b32 v102, %[b0]\n\tv_mov_b32 v103, %[b1]\n\tv_cmp_lt_i32_e64 %[m], -1, %[k0]\n\ts_nop 4\n\ts_mov_b64 vcc, %[m]\n\ts_nop 4\n\t"
#define PK_OUT [c0] "=&v"(c0), [c1] "=&v"(c1), [m] "=&s"(m)
#define PK_IN [a0] "v"(a0), [a1] "v"(a1), [b0] "v"(b0), [b1] "v"(b1), [k0] "v"(k0), [k1] "v"(k1)
#define PK_CLOB "vcc", "v100", "v101", "v102", "v103", "v104", "v105"
    if (W == 0)
      asm volatile(PK_SETUP "v_pk_mul_f32 v[102:103], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                   "v_cndmask_b32_e32 v100, 0, %[k0], vcc\n\tv_cndmask_b32_e32 v101, 0, %[k1], vcc\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v102\n\tv_mov_b32 %[c1], v103\n\t" : PK_OUT : PK_IN : PK_CLOB);
    else if (W == 1)
      asm volatile(PK_SETUP "v_pk_mul_f32 v[102:103], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 0\n\t"
                   "v_cndmask_b32_e32 v100, 0, %[k0], vcc\n\tv_cndmask_b32_e32 v101, 0, %[k1], vcc\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v102\n\tv_mov_b32 %[c1], v103\n\t" : PK_OUT : PK_IN : PK_CLOB);
    else if (W == 2)
      asm volatile(PK_SETUP "v_pk_mul_f32 v[102:103], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 1\n\t"
                   "v_cndmask_b32_e32 v100, 0, %[k0], vcc\n\tv_cndmask_b32_e32 v101, 0, %[k1], vcc\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v102\n\tv_mov_b32 %[c1], v103\n\t" : PK_OUT : PK_IN : PK_CLOB);
    else if (W == 3)
      asm volatile(PK_SETUP "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                   "v_cndmask_b32_e32 v100, 0, %[k0], vcc\n\tv_cndmask_b32_e32 v101, 0, %[k1], vcc\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v104\n\tv_mov_b32 %[c1], v105\n\t" : PK_OUT : PK_IN : PK_CLOB);
    else if (W == 4)
      asm volatile(PK_SETUP "v_mul_f32 v104, v100, v103\n\tv_mul_f32 v105, v101, v102\n\t"
                   "v_cndmask_b32_e32 v100, 0, %[k0], vcc\n\tv_cndmask_b32_e32 v101, 0, %[k1], vcc\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v104\n\tv_mov_b32 %[c1], v105\n\t" : PK_OUT : PK_IN : PK_CLOB);
    else if (W == 5)
      asm volatile(PK_SETUP "v_pk_add_f32 v[102:103], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                   "v_cndmask_b32_e32 v100, 0, %[k0], vcc\n\tv_cndmask_b32_e32 v101, 0, %[k1], vcc\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v102\n\tv_mov_b32 %[c1], v103\n\t" : PK_OUT : PK_IN : PK_CLOB);
    else
      asm volatile(PK_SETUP "v_pk_mul_f32 v[102:103], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]\n\t"
                   "v_mov_b32 v100, %[k0]\n\tv_mov_b32 v101, %[k1]\n\ts_nop 7\n\t"
                   "v_mov_b32 %[c0], v102\n\tv_mov_b32 %[c1], v103\n\t" : PK_OUT : PK_IN : PK_CLOB);
    const float q0 = W == 5 ? a0 + b1 : a0 * b1, q1 = W == 5 ? a1 + b0 : a1 * b0;
    const float w0 = W == 5 ? __int_as_float(k0) + b1 : __int_as_float(k0) * b1, w1 = W == 5 ? __int_as_float(k1) + b0 : __int_as_float(k1) * b0;
    const bool blo = __float_as_uint(c0) != __float_as_uint(q0), bhi = __float_as_uint(c1) != __float_as_uint(q1);
    elo += blo; ehi += bhi;
    if (blo) {                                   // keep the first few wrong samples: operands, got, lane
      const unsigned slot = atomicAdd(err + 15, 1u);
      if (slot < 8) {
        unsigned* d = err + 16 + slot * 8;
        d[0] = __float_as_uint(a0); d[1] = __float_as_uint(b1); d[2] = __float_as_uint(a1); d[3] = __float_as_uint(b0);
        d[4] = __float_as_uint(c0); d[5] = (unsigned)k0; d[6] = threadIdx.x; d[7] = blockIdx.x;
      }
    }
    klo += blo && __float_as_uint(c0) == __float_as_uint(w0);
    khi += bhi && __float_as_uint(c1) == __float_as_uint(w1);
  }
  if (elo) atomicAdd(err + 0, elo);
  if (ehi) atomicAdd(err + 1, ehi);
  if (elo + ehi) atomicAdd(err + 2 + (lane >> 4), elo + ehi);
  if (klo) atomicAdd(err + 6, klo);
  if (khi) atomicAdd(err + 7, khi);
}

template <int W> static void run_pk(const char* name, hipStream_t sv, hipStream_t sa, int aggressor, gemm_fn gemm, void* A, void* Wt, void* O,
                                    float* sink, unsigned* err) {
  unsigned h[80];
  for (int beside = 0; beside < 2; ++beside) {
    CK(hipMemset(err, 0, 320));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 60; ++rep) {
      if (beside) {
        if (aggressor < 9) launch_aggressor(aggressor, sa, sink);
        else if (gemm(A, 256, Wt, 512, nullptr, nullptr, 0, O, 256, nullptr, 174080, 256, 256, 0, 0, 0, 1.f, 1.f, sa) != 0) { printf("gemm failed\n"); exit(1); }
      }
      for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((victim_pk<W>), dim3(600), dim3(256), 0, sv, err, 64);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, err, 320, hipMemcpyDeviceToHost));
    for (unsigned i = 0; i < (h[15] < 4 ? h[15] : 4); ++i) {
      const unsigned* d = h + 16 + i * 8;
      float f[5];
      memcpy(f, d, 20);
      printf("      sample: a0 %.9g (%08x) b1 %.9g (%08x) a1 %.9g b0 %.9g -> lo got %.9g (%08x) want %.9g (%08x); a0*b0 %.9g a1*b1 %.9g a1*b0 %.9g; k0 %u thread %u block %u\n",
             f[0], d[0], f[1], d[1], f[2], f[3], f[4], d[4], W == 5 ? f[0] + f[1] : f[0] * f[1], 0u, f[0] * f[3], f[2] * f[1], f[2] * f[3], d[5], d[6], d[7]);
    }
    printf("  %-52s %-7s: wrong lo %u hi %u | by quarter %u %u %u %u | explained by the overwritten source: lo %u hi %u\n", name,
           beside ? "BESIDE" : "alone", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
}

template <int V> static void run_variant(const char* name, hipStream_t sv, hipStream_t sa, int aggressor, gemm_fn gemm, void* A, void* W,
                                         void* O, float* sink, unsigned* err) {
  unsigned h[10];
  for (int beside = 0; beside < 2; ++beside) {
    CK(hipMemset(err, 0, 40));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 60; ++rep) {
      if (beside) {
        if (aggressor < 9) launch_aggressor(aggressor, sa, sink);
        else if (gemm(A, 256, W, 512, nullptr, nullptr, 0, O, 256, nullptr, 174080, 256, 256, /*HIPIE_F32*/ 0, 0, 0, 1.f, 1.f, sa) != 0) { printf("gemm failed\n"); exit(1); }
      }
      for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((victim<V>), dim3(600), dim3(256), 0, sv, err, 64, 127, 127);
    }
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, err, 40, hipMemcpyDeviceToHost));
    printf("  %-44s %-8s: wrong r0 %u r1 %u r2 %u r3 %u | by quarter %u %u %u %u | zero-for-value %u value-for-zero %u\n", name,
           beside ? "BESIDE" : "alone", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
  }
}

int main(int argc, char** argv) {
  hipStream_t sv, sa;
  CK(hipStreamCreate(&sv));
  CK(hipStreamCreate(&sa));
  float* sink;
  unsigned* err;
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&err, 512));
  gemm_fn gemm = nullptr;
  void *A = nullptr, *W = nullptr, *O = nullptr;
  if (argc > 1) {
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    gemm = (gemm_fn)dlsym(lib, "hipie_gemm");
    CK(hipMalloc(&A, (size_t)174080 * 256 * 4));
    CK(hipMalloc(&W, (size_t)256 * 512 * 2));
    CK(hipMalloc(&O, (size_t)174080 * 256 * 4));
    CK(hipMemset(A, 0x3c, (size_t)174080 * 256 * 4));       // 0x3c3c3c3c = 0.0115 as fp32: finite operands
    CK(hipMemset(W, 0x2c, (size_t)256 * 512 * 2));          // fp16 0x2c2c = 0.065
  }
  float* srcbuf;
  CK(hipMalloc(&srcbuf, (size_t)(1 << 24) * 4 + 4096));
  CK(hipMemset(srcbuf, 0x3c, (size_t)(1 << 24) * 4 + 4096));
  g_src = srcbuf;
  const bool chain = getenv("CHAIN") != nullptr;
  for (int aggressor = 0; aggressor < (gemm ? 10 : 9); ++aggressor) {
    printf("aggressor: %s\n", AGG[aggressor]);
    if (chain) {
      run_variant<0>("0 product schedule (tight chain)", sv, sa, aggressor, gemm, A, W, O, sink, err);
      run_variant<3>("3 no VCC: e64 selects on 4 SGPR pairs", sv, sa, aggressor, gemm, A, W, O, sink, err);
    }
    run_pk<0>("pk 0 v_pk_mul_f32 (dst = src1), sources overwritten at once", sv, sa, aggressor, gemm, A, W, O, sink, err);
    run_pk<2>("pk 2 the same, s_nop 1 between", sv, sa, aggressor, gemm, A, W, O, sink, err);
    run_pk<3>("pk 3 distinct dst pair, sources overwritten at once", sv, sa, aggressor, gemm, A, W, O, sink, err);
    run_pk<4>("pk 4 control: two v_mul_f32, sources overwritten", sv, sa, aggressor, gemm, A, W, O, sink, err);
    run_pk<5>("pk 5 v_pk_add_f32 (dst = src1), sources overwritten", sv, sa, aggressor, gemm, A, W, O, sink, err);
  }
  return 0;
}
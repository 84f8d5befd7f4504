// pk_f32_hazard.hip -- torch-free reproducer of the fault behind DESIGN.md section 9's "co-residency corruption of hipie_msda_fused".
//
// Finding (round 6): on gfx950 a PACKED FP32 VALU instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) of one wave can read ZERO for
// one of its source halves in lanes 48-63 when ANOTHER wave on the same SIMD runs the inner loop of an LDS-tiled MFMA GEMM (MFMA +
// global_load_lds + ds_read_b128, with barriers).  Scalar v_mul_f32 / v_add_f32 / v_fma_f32 on the same registers are never wrong.
// hipcc emits v_pk_*_f32 through the SLP vectoriser; in msda_d32_kernel<FUSED> the corner weight hh * lw came out of
//     v_pk_mul_f32 v[18:19], v[0:1], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]
// and was 0 for the last quarter of a wave (two adjacent (query, head) groups) whenever the workgroup shared a CU with gemm_kernel<256>.
//
// This program: a self-checking victim kernel per packed-op FORM (inline asm on fixed registers, so the encoding is pinned) on one stream,
// an aggressor built from switchable instruction classes on another; every wrong result is classified by which source half read as 0
// explains it.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_f32_hazard.hip -o tools/ubench/pk_f32_hazard && tools/ubench/pk_f32_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- aggressor: K is a set of instruction classes ----------------------------------------------------------------------------------------
//   1 MFMA (32x32x16 f16)   2 LDS-DMA (global_load_lds_dwordx4)   4 ds_read_b128   8 s_barrier every 8 iterations   16 packed fp32 VALU
//   32 global_load_dwordx4 to VGPRs   64 ds_write_b128   128 plain fp32 VALU stream
template <int K>
__global__ __launch_bounds__(256) void spin(float* sink, const float* src, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  f32x16 c0, c1;
  for (int r = 0; r < 16; ++r) { c0[r] = r; c1[r] = r + 1; }
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x - j)); }
  f32x2 p = {1.0f + threadIdx.x * 1e-3f, 0.5f}, q = {0.999f, 1.001f}, r2 = {0.1f, 0.2f};
  float s0 = threadIdx.x, s1 = 0.5f;
  f32x4 acc4 = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = i;
  __syncthreads();
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + (threadIdx.x >> 6) * 1024);
  for (int i = 0; i < iters; ++i) {
    if (K & 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %2, %2, %1\n\tv_pk_fma_f32 %0, %0, %1, %2\n\tv_pk_mul_f32 %2, %2, %1" : "+v"(p), "+v"(q), "+v"(r2));
    if (K & 128) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_mul_f32 %1, %1, %0\n\tv_fma_f32 %0, %0, %1, %1\n\tv_mul_f32 %1, %1, %0" : "+v"(s0), "+v"(s1));
    if (K & 1) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
    }
    const float* g = src + ((blockIdx.x * 256 + threadIdx.x) * 4 + (i & 63) * 262144) % (1 << 24);
    if (K & 2) {
      unsigned save;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(save) : "v"(g), "s"(lds_base) : "memory");
    }
    if (K & 32) acc4 += *reinterpret_cast<const f32x4*>(g);
    if (K & 4) acc4 += *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + i * 64) & 8188));
    if (K & 64) *reinterpret_cast<f32x4*>(lds + ((threadIdx.x * 4 + i * 64 + 4096) & 8188)) = acc4;
    if ((K & 8) && (i & 7) == 7) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (c0[0] + c1[1] + p[0] + p[1] + r2[0] + s0 + s1 + acc4[0] + acc4[3] == 12345.f) sink[0] = 1.f;
}

// ---- victim: one packed-op form per instantiation ----------------------------------------------------------------------------------------
// registers: A = v[100:101] (a0, a1), B = v[102:103] (b0, b1), C = v[106:107] (c0, c1), D = v[104:105]
struct Form {
  const char* text;
};
#define FORM_LIST(X) \
  X(0, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel_hi:[0,0]", a0 * b0, a0 * b0) \
  X(1, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel_hi:[0,1]", a0 * b0, a0 * b1) \
  X(2, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel_hi:[1,0]", a0 * b0, a1 * b0) \
  X(3, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103]", a0 * b0, a1 * b1) \
  X(4, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[0,0]", a0 * b1, a0 * b0) \
  X(5, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[0,1]", a0 * b1, a0 * b1) \
  X(6, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]", a0 * b1, a1 * b0) \
  X(7, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[0,1]", a0 * b1, a1 * b1) \
  X(8, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,0] op_sel_hi:[0,0]", a1 * b0, a0 * b0) \
  X(9, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,0] op_sel_hi:[0,1]", a1 * b0, a0 * b1) \
  X(10, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,0] op_sel_hi:[1,0]", a1 * b0, a1 * b0) \
  X(11, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,0]", a1 * b0, a1 * b1) \
  X(12, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,1] op_sel_hi:[0,0]", a1 * b1, a0 * b0) \
  X(13, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,1] op_sel_hi:[0,1]", a1 * b1, a0 * b1) \
  X(14, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,1] op_sel_hi:[1,0]", a1 * b1, a1 * b0) \
  X(15, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[102:103] op_sel:[1,1]", a1 * b1, a1 * b1) \
  X(16, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel_hi:[0,0]", a0 * b0, a0 * b0) \
  X(17, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel_hi:[0,1]", a0 * b0, a0 * b1) \
  X(18, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel_hi:[1,0]", a0 * b0, a1 * b0) \
  X(19, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109]", a0 * b0, a1 * b1) \
  X(20, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1] op_sel_hi:[0,0]", a0 * b1, a0 * b0) \
  X(21, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1] op_sel_hi:[0,1]", a0 * b1, a0 * b1) \
  X(22, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1] op_sel_hi:[1,0]", a0 * b1, a1 * b0) \
  X(23, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[0,1]", a0 * b1, a1 * b1) \
  X(24, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,0] op_sel_hi:[0,0]", a1 * b0, a0 * b0) \
  X(25, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,0] op_sel_hi:[0,1]", a1 * b0, a0 * b1) \
  X(26, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,0] op_sel_hi:[1,0]", a1 * b0, a1 * b0) \
  X(27, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,0]", a1 * b0, a1 * b1) \
  X(28, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,1] op_sel_hi:[0,0]", a1 * b1, a0 * b0) \
  X(29, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,1] op_sel_hi:[0,1]", a1 * b1, a0 * b1) \
  X(30, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,1] op_sel_hi:[1,0]", a1 * b1, a1 * b0) \
  X(31, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v108, %4\n\tv_mov_b32 v109, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[100:101], v[108:109] op_sel:[1,1]", a1 * b1, a1 * b1) \
  X(32, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel_hi:[0,0]", a0 * b0, a0 * b0) \
  X(33, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel_hi:[0,1]", a0 * b0, a0 * b1) \
  X(34, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel_hi:[1,0]", a0 * b0, a1 * b0) \
  X(35, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101]", a0 * b0, a1 * b1) \
  X(36, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[0,1] op_sel_hi:[0,0]", a0 * b1, a0 * b0) \
  X(37, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[0,1] op_sel_hi:[0,1]", a0 * b1, a0 * b1) \
  X(38, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[0,1] op_sel_hi:[1,0]", a0 * b1, a1 * b0) \
  X(39, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[0,1]", a0 * b1, a1 * b1) \
  X(40, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,0] op_sel_hi:[0,0]", a1 * b0, a0 * b0) \
  X(41, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,0] op_sel_hi:[0,1]", a1 * b0, a0 * b1) \
  X(42, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,0] op_sel_hi:[1,0]", a1 * b0, a1 * b0) \
  X(43, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,0]", a1 * b0, a1 * b1) \
  X(44, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,1] op_sel_hi:[0,0]", a1 * b1, a0 * b0) \
  X(45, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,1] op_sel_hi:[0,1]", a1 * b1, a0 * b1) \
  X(46, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,1] op_sel_hi:[1,0]", a1 * b1, a1 * b0) \
  X(47, "v_mov_b32 v102, %2\n\tv_mov_b32 v103, %3\n\tv_mov_b32 v100, %4\n\tv_mov_b32 v101, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_mul_f32 v[104:105], v[102:103], v[100:101] op_sel:[1,1]", a1 * b1, a1 * b1) \
  X(48, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_add_f32 v[104:105], v[100:101], v[102:103] op_sel:[0,1] op_sel_hi:[1,0]", a0 + b1, a1 + b0) \
  X(49, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel_hi:[1,1,0]", fmaf(a0, b0, c0), fmaf(a1, b1, c0)) \
  X(50, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel_hi:[0,1,1]", fmaf(a0, b0, c0), fmaf(a0, b1, c1)) \
  X(51, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel_hi:[1,0,0]", fmaf(a0, b0, c0), fmaf(a1, b0, c0)) \
  X(52, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel_hi:[1,0,1]", fmaf(a0, b0, c0), fmaf(a1, b0, c1)) \
  X(53, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[0,1,0]", fmaf(a0, b1, c0), fmaf(a1, b1, c1)) \
  X(54, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[1,0,0]", fmaf(a1, b0, c0), fmaf(a1, b1, c1)) \
  X(55, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[0,0,1] op_sel_hi:[1,1,0]", fmaf(a0, b0, c1), fmaf(a1, b1, c0)) \
  X(56, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[1,0,0] op_sel_hi:[0,1,0]", fmaf(a1, b0, c0), fmaf(a0, b1, c0)) \
  X(57, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[0,1,0] op_sel_hi:[1,0,1]", fmaf(a0, b1, c0), fmaf(a1, b0, c1)) \
  X(58, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[0,0,1]", fmaf(a0, b0, c1), fmaf(a1, b1, c1)) \
  X(59, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107] op_sel:[0,1,1] op_sel_hi:[1,0,0]", fmaf(a0, b1, c1), fmaf(a1, b0, c0)) \
  X(60, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_pk_fma_f32 v[104:105], v[100:101], v[102:103], v[106:107]", fmaf(a0, b0, c0), fmaf(a1, b1, c1)) \
  X(61, "v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\tv_mov_b32 v106, %6\n\tv_mov_b32 v107, %7", "v_mul_f32 v104, v100, v103\n\tv_mul_f32 v105, v101, v102", a0 * b1, a1 * b0)

constexpr int NFORMS = 62;
template <int F> struct FormT;
#define X(n, mov, txt, lo, hi)                                                                                                                \
  template <> struct FormT<n> {                                                                                                            \
    static constexpr const char* text = txt;                                                                                               \
    static __device__ __forceinline__ void run(float a0, float a1, float b0, float b1, float c0, float c1, float& d0, float& d1) {         \
      asm volatile(mov "\n\ts_nop 7\n\t" txt "\n\ts_nop 7\n\tv_mov_b32 %0, v104\n\tv_mov_b32 %1, v105"                                        \
                   : "=&v"(d0), "=&v"(d1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1)                                           \
                   : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109");                                                      \
    }                                                                                                                                      \
    static __device__ __forceinline__ void ref(float a0, float a1, float b0, float b1, float c0, float c1, float& d0, float& d1) {         \
      d0 = lo; d1 = hi;                                                                                                                    \
    }                                                                                                                                      \
  };
FORM_LIST(X)
#undef X

// err[0] wrong lo, [1] wrong hi, [2..5] wrong results by quarter of the wave, [6..11] wrong results explained by a0/a1/b0/b1/c0/c1 read as 0
template <int F>
__global__ __launch_bounds__(256) void victim(unsigned int* err, int iters) {
  const unsigned lane = threadIdx.x & 63;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 777u;
  unsigned elo = 0, ehi = 0, z[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    float in[6];
    in[0] = __uint_as_float(0x3f000000u | (s & 0x7fffffu)); in[1] = __uint_as_float(0x3f000000u | ((s >> 3) & 0x7fffffu));
    in[2] = __uint_as_float(0x3f800000u | ((s >> 5) & 0x7fffffu)); in[3] = __uint_as_float(0x3f800000u | ((s >> 7) & 0x7fffffu));
    in[4] = __uint_as_float(0x40000000u | ((s >> 2) & 0x7fffffu)); in[5] = __uint_as_float(0x40000000u | ((s >> 6) & 0x7fffffu));
    float d0, d1, q0, q1;
    FormT<F>::run(in[0], in[1], in[2], in[3], in[4], in[5], d0, d1);
    FormT<F>::ref(in[0], in[1], in[2], in[3], in[4], in[5], q0, q1);
    const bool blo = __float_as_uint(d0) != __float_as_uint(q0), bhi = __float_as_uint(d1) != __float_as_uint(q1);
    elo += blo; ehi += bhi;
    if (blo || bhi) {
      for (int k = 0; k < 6; ++k) {
        float t[6];
        for (int j = 0; j < 6; ++j) t[j] = j == k ? 0.f : in[j];
        float w0, w1;
        FormT<F>::ref(t[0], t[1], t[2], t[3], t[4], t[5], w0, w1);
        z[k] += __float_as_uint(d0) == __float_as_uint(w0) && __float_as_uint(d1) == __float_as_uint(w1);
      }
    }
  }
  if (elo) atomicAdd(err + 0, elo);
  if (ehi) atomicAdd(err + 1, ehi);
  if (elo + ehi) atomicAdd(err + 2 + (lane >> 4), elo + ehi);
  for (int k = 0; k < 6; ++k) if (z[k]) atomicAdd(err + 6 + k, z[k]);
}

static float* g_sink;
static float* g_src;
static unsigned* g_err;
static hipStream_t sv, sa;

template <int K> static void aggress(int iters) { hipLaunchKernelGGL(spin<K>, dim3(4096), dim3(256), 0, sa, g_sink, g_src, iters); }
typedef void (*agg_fn)(int);

template <int F> static void run_form(agg_fn agg, int agg_iters) {
  unsigned h[12];
  CK(hipMemset(g_err, 0, 48));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 12; ++rep) {
    if (agg) agg(agg_iters);
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((victim<F>), dim3(600), dim3(256), 0, sv, g_err, 64);
  }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h, g_err, 48, hipMemcpyDeviceToHost));
  char txt[112];
  strncpy(txt, FormT<F>::text, 111);
  txt[111] = 0;
  for (char* c = txt; *c; ++c) if (*c == '\n' || *c == '\t') *c = ' ';
  printf("  %3d %-86s: wrong lo %8u hi %8u | quarters %u %u %u %u | read as 0: a0 %u a1 %u b0 %u b1 %u c0 %u c1 %u\n", F, txt, h[0], h[1], h[2], h[3],
         h[4], h[5], h[6], h[7], h[8], h[9], h[10], h[11]);
}

template <int F> struct RunAll {
  static void go(agg_fn agg, int agg_iters) { RunAll<F - 1>::go(agg, agg_iters); run_form<F>(agg, agg_iters); }
};
template <> struct RunAll<-1> { static void go(agg_fn, int) {} };
static void run_all(const char* name, agg_fn agg, int agg_iters, bool all_forms) {
  printf("aggressor: %s\n", name);
  if (all_forms) RunAll<NFORMS - 1>::go(agg, agg_iters);
  else { run_form<6>(agg, agg_iters); run_form<NFORMS - 1>(agg, agg_iters); }
  fflush(stdout);
}

int main() {
  CK(hipStreamCreate(&sv));
  CK(hipStreamCreate(&sa));
  CK(hipMalloc(&g_sink, 64));
  CK(hipMalloc(&g_err, 64));
  CK(hipMalloc(&g_src, (size_t)(1 << 24) * 4 + 4096));
  CK(hipMemset(g_src, 0x3c, (size_t)(1 << 24) * 4 + 4096));
  run_all("none (victim alone)", nullptr, 0, true);
  run_all("MFMA + LDS-DMA + ds_read_b128 + barrier (the k loop of an LDS-tiled GEMM)", aggress<1 + 2 + 4 + 8>, 400, true);
  // which classes are needed?
  run_all("MFMA + LDS-DMA + ds_read_b128 (no barrier)", aggress<1 + 2 + 4>, 400, false);
  run_all("MFMA + LDS-DMA + barrier", aggress<1 + 2 + 8>, 400, false);
  run_all("MFMA + ds_read_b128 + barrier", aggress<1 + 4 + 8>, 1000, false);
  run_all("LDS-DMA + ds_read_b128 + barrier (no MFMA)", aggress<2 + 4 + 8>, 400, false);
  run_all("MFMA only", aggress<1>, 1000, false);
  run_all("LDS-DMA only", aggress<2>, 400, false);
  run_all("ds_read_b128 only", aggress<4>, 4000, false);
  run_all("ds_read_b128 + barrier", aggress<4 + 8>, 4000, false);
  run_all("barrier only", aggress<8>, 4000, false);
  run_all("MFMA + global_load_dwordx4 + ds_write_b128 + ds_read_b128 + barrier (register-staged GEMM loop)", aggress<1 + 32 + 64 + 4 + 8>, 400, false);
  run_all("global_load_dwordx4 + barrier", aggress<32 + 8>, 400, false);
  run_all("packed fp32 VALU + ds_read_b128 + barrier", aggress<16 + 4 + 8>, 1000, false);
  run_all("plain fp32 VALU + LDS-DMA + ds_read_b128 + barrier", aggress<128 + 2 + 4 + 8>, 400, false);
  return 0;
}

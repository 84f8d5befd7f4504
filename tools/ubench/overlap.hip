// Micro-benchmark: can the two waves that share a SIMD overlap one wave's MFMAs with the other's VALU / LDS work?
// One workgroup of 512 threads per CU: waves w and w + 4 share a SIMD.  Role of each half: 0 idle, 1 MFMA stream,
// 2 VALU stream (v_exp + v_add), 3 mixed (1 MFMA : 5 VALU in ONE wave), 4 LDS read stream.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int ROLE>
__device__ __forceinline__ float run_role(int iters, float seed, float* lds) {
  float acc = 0.f;
  if (ROLE == 1 || ROLE == 3) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed + 1; c2[r] = seed + 2; c3[r] = seed + 3; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed - j); }
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        if (ROLE == 3) { v0 = __builtin_amdgcn_exp2f(v0) + 0.5f; v1 = v1 * 1.0001f + 0.25f; v2 = v2 * 0.999f + v0; __builtin_amdgcn_sched_barrier(0); }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        if (ROLE == 3) { v3 = __builtin_amdgcn_exp2f(v3) + 0.5f; v4 = v4 * 1.0001f + 0.25f; v1 = v1 * 0.999f + v3; __builtin_amdgcn_sched_barrier(0); }
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        if (ROLE == 3) { v0 = __builtin_amdgcn_exp2f(v0) + 0.5f; v1 = v1 * 1.0001f + 0.25f; v2 = v2 * 0.999f + v0; __builtin_amdgcn_sched_barrier(0); }
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
        if (ROLE == 3) { v3 = __builtin_amdgcn_exp2f(v3) + 0.5f; v4 = v4 * 1.0001f + 0.25f; v1 = v1 * 0.999f + v3; __builtin_amdgcn_sched_barrier(0); }
      }
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4;
  } else if (ROLE == 5 || ROLE == 6) {
    // the planned attention stream: per MFMA ~3.3 VALU (a third of them exp2) + ~1.7 LDS reads (role 6: + a b128 write every
    // 6th MFMA and a workgroup barrier every 24 MFMAs)
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed + 1; c2[r] = seed + 2; c3[r] = seed + 3; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed - j); }
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3;
    f32x4 s4 = {0, 0, 0, 0};
    f32x4* p4 = reinterpret_cast<f32x4*>(lds) + (threadIdx.x & 63);
    for (int i = 0; i < iters / 2; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        v0 = __builtin_amdgcn_exp2f(v0) + 0.5f; v1 = v1 * 1.0001f + v0; s4 += p4[64 * ((u + i) & 15)];
        __builtin_amdgcn_sched_barrier(0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        v2 = __builtin_amdgcn_exp2f(v2) + 0.5f; v3 = fmaxf(v3, v2); v1 = fmaxf(v1, v0); s4 += p4[64 * ((u + i + 5) & 15)];
        __builtin_amdgcn_sched_barrier(0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        v0 = v0 * 0.999f + v2; v1 = v1 * 1.0001f + 0.25f; v3 = v3 + v1; s4 += p4[64 * ((u + i + 9) & 15)];
        __builtin_amdgcn_sched_barrier(0);
        if (ROLE == 6 && (u & 1)) p4[64 * ((u + i + 3) & 15)] = s4;
      }
      if (ROLE == 6) __syncthreads();
    }
    acc = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + s4[0];
  } else if (ROLE == 7) {
    // the 64-queries-per-wave stream (one wave per SIMD): per 6 MFMAs ~20 VALU (8 of them exp2) + 4 LDS reads consumed two
    // MFMAs later + 1 wait; 48 MFMAs per "tile"
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x16 c[6];
    for (int q = 0; q < 6; ++q) for (int r = 0; r < 16; ++r) c[q][r] = seed + q;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(seed + j); b[j] = (__bf16)(seed - j); }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    f32x4 l0 = p4[0], l1 = p4[64];
    for (int i = 0; i < iters / 3; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4 n0 = p4[64 * ((u + i) & 15)], n1 = p4[64 * ((u + i + 3) & 15)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          if (q == 0) { a[0] = (__bf16)l0[0]; b[1] = (__bf16)l1[1]; }
          c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[q], 0, 0, 0);
          v[q] = __builtin_amdgcn_exp2f(v[q]) + 0.5f;
          v[(q + 1) & 7] = fmaxf(v[(q + 1) & 7], v[q]);
          if (q < 2) v[6 + q] = __builtin_amdgcn_exp2f(v[6 + q]) * 1.0001f;
          __builtin_amdgcn_sched_barrier(0);
        }
        l0 = n0; l1 = n1;
      }
    }
    for (int q = 0; q < 6; ++q) acc += c[q][q];
    for (int j = 0; j < 8; ++j) acc += v[j];
  } else if (ROLE == 2) {
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = seed + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 10; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j] * 0.001f) + 0.5f;     // 80 x (mul, exp, add) = 240 VALU per iteration
    }
    for (int j = 0; j < 8; ++j) acc += v[j];
  } else if (ROLE == 4) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 s = {0, 0, 0, 0};
    const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) s += p[64 * ((u + i) & 15)];
    }
    acc = s[0] + s[1] + s[2] + s[3];
  }
  return acc;
}

template <int RA, int RB>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, float seed) {
  __shared__ float lds[64 * 4 * 16 + 64];
  for (int i = threadIdx.x; i < 64 * 4 * 16; i += 512) lds[i] = seed;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float r = (wave < 4) ? run_role<RA>(iters, seed, lds) : run_role<RB>(iters, seed, lds);
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int RA, int RB>
static void bench(const char* name, float* d) {
  const int iters = 2000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, iters, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<RA, RB>), dim3(256), dim3(512), 0, 0, d, iters, 0.001f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  printf("%-34s %8.3f ms\n", name, ms);
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  bench<1, 0>("A: MFMA (16/iter)   B: idle", d);
  bench<1, 1>("A: MFMA             B: MFMA", d);
  bench<2, 0>("A: VALU (240/iter)  B: idle", d);
  bench<2, 2>("A: VALU             B: VALU", d);
  bench<1, 2>("A: MFMA             B: VALU", d);
  bench<3, 0>("A: MFMA+5 VALU/gap  B: idle", d);
  bench<3, 3>("A: mixed            B: mixed", d);
  bench<5, 0>("A: attn-like 24 MFMA/iter B: idle", d);
  bench<5, 5>("A: attn-like        B: attn-like", d);
  bench<6, 6>("A: attn-like+wr+bar B: same", d);
  bench<7, 0>("A: 64q stream 48 MFMA/it B: idle", d);
  bench<7, 7>("A: 64q stream       B: same", d);
  bench<4, 0>("A: LDS b128 (16/it) B: idle", d);
  bench<4, 4>("A: LDS              B: LDS", d);
  bench<1, 4>("A: MFMA             B: LDS", d);
  bench<2, 4>("A: VALU             B: LDS", d);
  return 0;
}

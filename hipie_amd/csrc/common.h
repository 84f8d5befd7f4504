// common.h -- shared device/host helpers for libhipie_mi355 (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/hipie_mi355.h"

namespace hipie {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- error plumbing (host) -------------------------------------------------------------------------
extern thread_local char g_err[512];
int set_err(int code, const char* fmt, ...);
int check_launch(const char* what);

#define HIPIE_REQUIRE(cond, ...)                              \
  do {                                                        \
    if (!(cond)) return ::hipie::set_err(HIPIE_EINVAL, __VA_ARGS__); \
  } while (0)

// ---- 16-bit <-> f32 ---------------------------------------------------------------------------------
template <typename T> struct elem;
template <> struct elem<float> {
  static __device__ __forceinline__ float to_f32(float x) { return x; }
  static __device__ __forceinline__ float from_f32(float x) { return x; }
};
template <> struct elem<bf16_t> {
  static __device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
  static __device__ __forceinline__ bf16_t from_f32(float x) { return (bf16_t)x; }
};
template <> struct elem<f16_t> {
  static __device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }
  static __device__ __forceinline__ f16_t from_f32(float x) { return (f16_t)x; }
};

// x -> (hi, lo) fp16 pair with hi = fp16(x), lo = fp16(x - hi)  (HIPIE_HL8).  The value is pinned in a register first: hipcc otherwise
// folds the multiply that PRODUCED x into a v_fma_mix*_f16 for one of its two uses (single rounding of the exact product) while the other
// use converts the fp32-rounded product -- near an fp16 tie the stored hi and the hi the remainder was taken from then differ by one ulp
// (measured: isolated 1-ulp(hi) errors in one output of 1e5).
// Values beyond the fp16 range saturate at +-65504 (an overflowing hi would be inf and its remainder x - inf a NaN).
__device__ __forceinline__ void hl_split(float x, f16_t& h, f16_t& l) {
  x = __builtin_fminf(__builtin_fmaxf(x, -65504.f), 65504.f);
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
  h = (f16_t)x;
  l = (f16_t)(x - (float)h);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// A/B switches of the kernel studies (tile orders, wave counts, ablations: tools/bench_*.py).  The shipped library has ONE path per policy:
// the switches exist only in a study build (make EXTRA=-DHIPIE_STUDY_KNOBS), where they are read from the environment once per process.
#ifdef HIPIE_STUDY_KNOBS
static inline const char* study_env(const char* name) { return getenv(name); }
#else
static inline const char* study_env(const char*) { return nullptr; }
#endif

}  // namespace hipie

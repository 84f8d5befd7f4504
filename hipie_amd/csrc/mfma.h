// mfma.h -- the MFMA / LDS-transpose lane layouts this library relies on (gfx950, wave64), in one place.
// tests/test_gpu_selftest.py pins them on hardware through hipie_selftest().
//
// v_mfma_f32_32x32x16_{bf16,f16}:  D(32x32) += A(32x16) . B(16x32)
//   A operand: lane l holds 8 values  A[i = l & 31][k = kslot(l >> 5, j)], j = 0..7
//   B operand: lane l holds 8 values  B[k = kslot(l >> 5, j)][n = l & 31], j = 0..7
//   (the same kslot(half, j) function for A and B -- the contraction pairs A's (half, j) with B's (half, j), so any
//    code that fills both operands with the same (half, j) -> k convention is correct whatever the hardware's k label)
//   C/D: lane l, register r (0..15):  row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),  col = l & 31
//
// v_mfma_f32_32x32x2_f32:  D(32x32) += A(32x2) . B(2x32);  A: lane l holds A[l & 31][l >> 5];  B: B[l >> 5][l & 31];
//   C/D as above.
//
// ds_read_b64_tr_b16 (per 16-lane group g = l >> 4, li = l & 15): lane li supplies the address of 4 consecutive b16
//   elements; the 16 lanes x 4 elements are regarded as a 4 x 16 matrix M[row = li / 4][col = 4 * (li % 4) + e] and lane
//   li receives column li:  result[j] = M[j][li], j = 0..3.  So pointing lane li at &tile[r0 + li / 4][c0 + 4 * (li % 4)]
//   of a row-major tile returns tile[r0 + j][c0 + li], i.e. 4 consecutive ROWS of one column -- the k-contiguous MFMA
//   operand of a matrix stored with k as the row index.
#pragma once
#include "common.h"

namespace hipie {

__device__ __forceinline__ int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

template <typename T> struct Mfma32;
template <> struct Mfma32<bf16_t> {
  typedef bf16x8 frag;
  typedef bf16x4 half_frag;
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ half_frag tr_read(const bf16_t* lds_ptr) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)lds_ptr));
  }
};
template <> struct Mfma32<f16_t> {
  typedef f16x8 frag;
  typedef f16x4 half_frag;
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ half_frag tr_read(const f16_t* lds_ptr) {
    return __builtin_bit_cast(f16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)lds_ptr));
  }
};

}  // namespace hipie

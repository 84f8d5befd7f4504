// selftest.hip -- on-device probes that pin the lane layouts documented in mfma.h (used by tests/test_gpu_selftest.py).
#include "common.h"
#include "mfma.h"

namespace hipie {

// D(32x32) = A(32x16) . B(16x32), A and B row-major bf16, operands filled exactly as the kernels fill them.
__global__ void selftest_mfma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ D) {
  const int lane = threadIdx.x, li = lane & 31, hi = lane >> 5;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = A[li * 16 + 8 * hi + j];
    b[j] = B[(8 * hi + j) * 32 + li];
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = Mfma32<bf16_t>::mma(a, b, c);
  for (int r = 0; r < 16; ++r) D[crow(r, hi) * 32 + li] = c[r];
}

// ds_read_b64_tr_b16 on an (8 x 32) row-major bf16 tile, addressed exactly like the V^T operand fetch of flash_attn.hip.
// out[lane*4 + j] must equal tile[4*hi + j][16*g1 + l16].
__global__ void selftest_tr_kernel(const bf16_t* __restrict__ tile, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) bf16_t lds[8 * 32];
  const int lane = threadIdx.x;
  for (int i = lane; i < 8 * 32; i += 64) lds[i] = tile[i];
  __syncthreads();
  const int l16 = lane & 15, g1 = (lane >> 4) & 1, hi = lane >> 5;
  const bf16_t* a0 = lds + (4 * hi + (l16 >> 2)) * 32 + 16 * g1 + 4 * (l16 & 3);
  const bf16x4 v = Mfma32<bf16_t>::tr_read(a0);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}

}  // namespace hipie

extern "C" int hipie_selftest(int which, const uint16_t* a, const uint16_t* b, float* out, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(a && out, "selftest: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (which == 0) {
    HIPIE_REQUIRE(b != nullptr, "selftest 0 needs b");
    hipLaunchKernelGGL(selftest_mfma_kernel, dim3(1), dim3(64), 0, st, (const bf16_t*)a, (const bf16_t*)b, out);
  } else if (which == 1) {
    hipLaunchKernelGGL(selftest_tr_kernel, dim3(1), dim3(64), 0, st, (const bf16_t*)a, out);
  } else {
    return set_err(HIPIE_EINVAL, "selftest: unknown probe %d", which);
  }
  return check_launch("selftest");
}

// layernorm.hip -- fused residual-add + LayerNorm + cast (glue kernel, HBM-bound).
//
//   s = x + delta (delta optional);  res_out = s (optional, dtype of x);  norm_out = LN(s) * gamma + beta  (dtype Tn)
//
// Replaces the add -> LayerNorm -> cast chains around every ViT block (hipie/backbone/vit.py:212-230, eps 1e-6) and the
// post-norm residuals of the deformable encoder layers (deformable_transformer_dino.py:384-394), which in eager PyTorch
// are 2-3 separate elementwise passes over the (B, tokens, C) stream.  One wave owns one row: the row is read once (8-byte
// vector loads), statistics are fp32 wave reductions (two-pass: mean, then centred variance -- the same arithmetic as
// torch's layer_norm), and both outputs are written once.  Bytes per row: C * (|x| + |delta| + |res| + |norm|).
#include "common.h"

namespace hipie {

template <typename T> struct V4 {
  static __device__ __forceinline__ void ld(const T* p, float (&v)[4]);
  static __device__ __forceinline__ void st(T* p, const float (&v)[4]);
};
template <> struct V4<float> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const float4 r = *reinterpret_cast<const float4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <> struct V4<bf16_t> {
  static __device__ __forceinline__ void ld(const bf16_t* p, float (&v)[4]) {
    const bf16x4 r = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&v)[4]) {
    bf16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x4*>(p) = r;
  }
};
template <> struct V4<f16_t> {
  static __device__ __forceinline__ void ld(const f16_t* p, float (&v)[4]) {
    const f16x4 r = *reinterpret_cast<const f16x4*>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void st(f16_t* p, const float (&v)[4]) {
    f16x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (f16_t)v[i];
    *reinterpret_cast<f16x4*>(p) = r;
  }
};

// HIPIE_HL8 as an OUTPUT (or addend) type of these kernels.  An HL8 row of C values occupies 4 C bytes like an fp32 row, so the kernels'
// element indexing (row * C + c, 4-byte elements) lands on the right row; inside the row the 4 consecutive values of a lane (c % 8 is 0
// or 4) live at bytes 32 (c / 8) + 2 (c % 8) (hi) and + 16 (lo).  With p = base + 4 c that is p / p + 16 for c % 8 == 0 and p - 8 / p + 8 for
// c % 8 == 4 (bit 4 of p: the rows are 32-byte aligned because C % 8 == 0 and torch allocations are).
struct hl8_t { unsigned int u; };
template <> struct V4<hl8_t> {
  static __device__ __forceinline__ void ld(const hl8_t* p, float (&v)[4]) {
    const char* q = reinterpret_cast<const char*>(p);
    const bool second = (reinterpret_cast<uintptr_t>(q) & 16) != 0;
    const f16x4 h = *reinterpret_cast<const f16x4*>(second ? q - 8 : q);
    const f16x4 l = *reinterpret_cast<const f16x4*>(second ? q + 8 : q + 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)h[i] + (float)l[i];
  }
  static __device__ __forceinline__ void st(hl8_t* p, const float (&v)[4]) {
    char* q = reinterpret_cast<char*>(p);
    const bool second = (reinterpret_cast<uintptr_t>(q) & 16) != 0;
    f16x4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16_t hh, ll;
      hl_split(v[i], hh, ll);
      h[i] = hh;
      l[i] = ll;
    }
    *reinterpret_cast<f16x4*>(second ? q - 8 : q) = h;
    *reinterpret_cast<f16x4*>(second ? q + 8 : q + 16) = l;
  }
};

constexpr int LN_MAXV = 8;     // up to 8 x 4 elements per lane: C <= 2048
// row maps of the call being dispatched (host-side plumbing through the dtype switch; set and cleared by the entry points)
static thread_local const int32_t* g_delta_row = nullptr;
static thread_local const int32_t* g_out_src = nullptr;

template <typename Tx, typename Td, typename Tn>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const Tx* __restrict__ x, const Td* __restrict__ delta,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            Tx* __restrict__ res_out, Tn* __restrict__ norm_out, long rows,
                                                            int C, float eps, const int32_t* __restrict__ delta_row,
                                                            const int32_t* __restrict__ out_src, const Tn* __restrict__ addend,
                                                            Tn* __restrict__ sum_out) {
  // row maps (both optional): the wave owns OUTPUT row `orow`; it normalises x row `row = out_src[orow]` (-1: the output row
  // is padding -> zeros, e.g. the pad tokens of window_partition) and adds delta row `delta_row[row]` (e.g. the window
  // layout the attention wrote) -- window partition / un-partition become index arithmetic of this pass
  const long orow = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (orow >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nv = C / 256;                       // full 4-element vectors per lane (64 lanes x 4)
  const int tail = (C - nv * 256) / 4;          // remaining vectors (< 64), one per lane for lane < tail
  const long row = out_src ? (long)out_src[orow] : orow;
  if (row < 0) {
    const float z[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i <= nv; ++i)
      if ((i < nv) || (lane < tail)) V4<Tn>::st(norm_out + orow * C + i * 256 + lane * 4, z);
    return;
  }
  const long drow = delta_row ? (long)delta_row[row] : row;
  float v[LN_MAXV][4];
  float sum = 0.f;
  const Tx* xr = x + row * C;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool on = (i < nv) || (i == nv && lane < tail);
    if (on) {
      const int c = i * 256 + lane * 4;
      V4<Tx>::ld(xr + c, v[i]);
      if (delta != nullptr) {
        float d[4];
        V4<Td>::ld(delta + drow * C + c, d);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] += d[e];
      }
      if (res_out != nullptr) V4<Tx>::st(res_out + row * C + c, v[i]);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += v[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool on = (i < nv) || (i == nv && lane < tail);
    if (on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool on = (i < nv) || (i == nv && lane < tail);
    if (on) {
      const int c = i * 256 + lane * 4;
      const float4 g = *reinterpret_cast<const float4*>(gamma + c);
      const float4 b = *reinterpret_cast<const float4*>(beta + c);
      float o[4];
      o[0] = (v[i][0] - mean) * rstd * g.x + b.x;
      o[1] = (v[i][1] - mean) * rstd * g.y + b.y;
      o[2] = (v[i][2] - mean) * rstd * g.z + b.z;
      o[3] = (v[i][3] - mean) * rstd * g.w + b.w;
      V4<Tn>::st(norm_out + orow * C + c, o);
      if (sum_out != nullptr) {               // the normalised row plus a second addend (e.g. the position embedding of the next query)
        float a[4];
        V4<Tn>::ld(addend + orow * C + c, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += o[e];
        V4<Tn>::st(sum_out + orow * C + c, a);
      }
    }
  }
}

static thread_local const void* g_ln_addend = nullptr;     // optional extra output of the call being dispatched (as the row maps)
static thread_local void* g_ln_sum_out = nullptr;

template <typename Tx, typename Td, typename Tn>
static int launch_ln(const void* x, const void* d, const float* g, const float* b, void* r, void* n, long rows, int C,
                     float eps, hipStream_t st) {
  hipLaunchKernelGGL((add_layernorm_kernel<Tx, Td, Tn>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, (const Tx*)x,
                     (const Td*)d, g, b, (Tx*)r, (Tn*)n, rows, C, eps, g_delta_row, g_out_src, (const Tn*)g_ln_addend, (Tn*)g_ln_sum_out);
  return check_launch("add_layernorm");
}

template <typename Tx, typename Td>
static int ln_out(int nd, const void* x, const void* d, const float* g, const float* b, void* r, void* n, long rows, int C,
                  float eps, hipStream_t st) {
  switch (nd) {
    case HIPIE_F32: return launch_ln<Tx, Td, float>(x, d, g, b, r, n, rows, C, eps, st);
    case HIPIE_F16: return launch_ln<Tx, Td, f16_t>(x, d, g, b, r, n, rows, C, eps, st);
    case HIPIE_BF16: return launch_ln<Tx, Td, bf16_t>(x, d, g, b, r, n, rows, C, eps, st);
    case HIPIE_HL8:
      if (C % 8 != 0 || (reinterpret_cast<uintptr_t>(n) & 31) != 0) return set_err(HIPIE_EINVAL, "add_layernorm: HL8 output needs C %% 8 == 0 and a 32-byte aligned buffer");
      return launch_ln<Tx, Td, hl8_t>(x, d, g, b, r, n, rows, C, eps, st);
    default: return set_err(HIPIE_EINVAL, "add_layernorm: bad norm dtype %d", nd);
  }
}

template <typename Tx>
static int ln_delta(int dd, int nd, const void* x, const void* d, const float* g, const float* b, void* r, void* n,
                    long rows, int C, float eps, hipStream_t st) {
  switch (dd) {
    case HIPIE_F32: return ln_out<Tx, float>(nd, x, d, g, b, r, n, rows, C, eps, st);
    case HIPIE_F16: return ln_out<Tx, f16_t>(nd, x, d, g, b, r, n, rows, C, eps, st);
    case HIPIE_BF16: return ln_out<Tx, bf16_t>(nd, x, d, g, b, r, n, rows, C, eps, st);
    default: return set_err(HIPIE_EINVAL, "add_layernorm: bad delta dtype %d", dd);
  }
}


// Post-norm residual of the DINO decoder layers with an fp32 query stream and 16-bit GEMMs: the LayerNorm output leaves once in
// fp32 (the next residual) and, in the same pass, as the 16-bit operands of the GEMMs that follow: a plain copy (value / FFN /
// box-head input) and / or the copy with the positional query added (the query of the next attention).  Replaces the
// add -> cast chains between the launches of a decoder layer (deformable_transformer_dino.py:418-450).
template <typename Td, typename Ta>
__global__ __launch_bounds__(256) void add_layernorm_dec_kernel(const float* __restrict__ x, const Td* __restrict__ delta,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ norm_out, Ta* __restrict__ norm16,
                                                                const Ta* __restrict__ addend, Ta* __restrict__ sum16, long rows,
                                                                int C, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int nv = C / 256, tail = (C - nv * 256) / 4;
  float v[LN_MAXV][4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool on = (i < nv) || (i == nv && lane < tail);
    if (on) {
      const long c = row * C + i * 256 + lane * 4;
      float d[4];
      V4<float>::ld(x + c, v[i]);
      V4<Td>::ld(delta + c, d);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[i][e] += d[e]; sum += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = 0.f;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool on = (i < nv) || (i == nv && lane < tail);
    if (on) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = rsqrtf(sq / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const bool on = (i < nv) || (i == nv && lane < tail);
    if (on) {
      const int cc = i * 256 + lane * 4;
      const long c = row * C + cc;
      const float4 g = *reinterpret_cast<const float4*>(gamma + cc);
      const float4 b = *reinterpret_cast<const float4*>(beta + cc);
      float o[4];
      o[0] = (v[i][0] - mean) * rstd * g.x + b.x;
      o[1] = (v[i][1] - mean) * rstd * g.y + b.y;
      o[2] = (v[i][2] - mean) * rstd * g.z + b.z;
      o[3] = (v[i][3] - mean) * rstd * g.w + b.w;
      V4<float>::st(norm_out + c, o);
      if (norm16 != nullptr) V4<Ta>::st(norm16 + c, o);
      if (sum16 != nullptr) {
        float a[4];
        V4<Ta>::ld(addend + c, a);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += o[e];
        V4<Ta>::st(sum16 + c, a);
      }
    }
  }
}

template <typename Td, typename Ta>
static int launch_ln_dec(const float* x, const void* d, const float* g, const float* b, float* n, void* n16, const void* add,
                         void* s16, long rows, int C, float eps, hipStream_t st) {
  hipLaunchKernelGGL((add_layernorm_dec_kernel<Td, Ta>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, (const Td*)d, g, b,
                     n, (Ta*)n16, (const Ta*)add, (Ta*)s16, rows, C, eps);
  return check_launch("add_layernorm_dec");
}

// out = (Ta)(a + b): the positional query added to the fp32 stream, rounded once to the GEMM operand type
template <typename Ta>
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const Ta* __restrict__ b, Ta* __restrict__ out, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float x[4], y[4];
  V4<float>::ld(a + 4 * i, x);
  V4<Ta>::ld(b + 4 * i, y);
#pragma unroll
  for (int e = 0; e < 4; ++e) x[e] += y[e];
  V4<Ta>::st(out + 4 * i, x);
}

}  // namespace hipie

extern "C" int hipie_add_layernorm_rows(const void* x, const void* delta, const float* gamma, const float* beta,
                                        void* res_out, void* norm_out, int64_t out_rows, int C, float eps, int x_dtype,
                                        int delta_dtype, int norm_dtype, const int32_t* delta_row, const int32_t* out_src,
                                        void* stream) {
  hipie::g_delta_row = delta_row;
  hipie::g_out_src = out_src;
  const int rc = hipie_add_layernorm(x, delta, gamma, beta, res_out, norm_out, out_rows, C, eps, x_dtype, delta_dtype,
                                     norm_dtype, stream);
  hipie::g_delta_row = nullptr;
  hipie::g_out_src = nullptr;
  return rc;
}

extern "C" int hipie_add_layernorm(const void* x, const void* delta, const float* gamma, const float* beta, void* res_out,
                                   void* norm_out, int64_t rows, int C, float eps, int x_dtype, int delta_dtype,
                                   int norm_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(x && gamma && beta && norm_out, "add_layernorm: null pointer");
  HIPIE_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && C <= LN_MAXV * 256, "add_layernorm: C=%d must be a multiple of 4 and <= %d", C, LN_MAXV * 256);
  if (rows == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (x_dtype) {
    case HIPIE_F32: return ln_delta<float>(delta_dtype, norm_dtype, x, delta, gamma, beta, res_out, norm_out, rows, C, eps, st);
    case HIPIE_F16: return ln_delta<f16_t>(delta_dtype, norm_dtype, x, delta, gamma, beta, res_out, norm_out, rows, C, eps, st);
    case HIPIE_BF16: return ln_delta<bf16_t>(delta_dtype, norm_dtype, x, delta, gamma, beta, res_out, norm_out, rows, C, eps, st);
    default: return set_err(HIPIE_EINVAL, "add_layernorm: bad x dtype %d", x_dtype);
  }
}

extern "C" int hipie_add_layernorm_dec(const float* x, const void* delta, const float* gamma, const float* beta, float* norm_out,
                                       void* norm16_out, const void* addend, void* sum16_out, int64_t rows, int C, float eps,
                                       int delta_dtype, int aux_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(x && delta && gamma && beta && norm_out, "add_layernorm_dec: null pointer");
  HIPIE_REQUIRE((sum16_out == nullptr) || (addend != nullptr), "add_layernorm_dec: sum16_out needs addend");
  HIPIE_REQUIRE(rows >= 0 && C > 0 && C % 4 == 0 && C <= LN_MAXV * 256, "add_layernorm_dec: C=%d must be a multiple of 4 and <= %d", C, LN_MAXV * 256);
  HIPIE_REQUIRE(aux_dtype == HIPIE_F16 || aux_dtype == HIPIE_BF16 || aux_dtype == HIPIE_HL8, "add_layernorm_dec: aux dtype must be f16, bf16 or HL8");
  HIPIE_REQUIRE(aux_dtype != HIPIE_HL8 || (C % 8 == 0 && (((uintptr_t)norm16_out | (uintptr_t)addend | (uintptr_t)sum16_out) & 31) == 0),
                "add_layernorm_dec: HL8 outputs need C %% 8 == 0 and 32-byte aligned buffers");
  if (rows == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
#define HIPIE_LND(Td)                                                                                                          \
  return aux_dtype == HIPIE_F16                                                                                                \
             ? launch_ln_dec<Td, f16_t>(x, delta, gamma, beta, norm_out, norm16_out, addend, sum16_out, rows, C, eps, st)      \
         : aux_dtype == HIPIE_HL8                                                                                              \
             ? launch_ln_dec<Td, hl8_t>(x, delta, gamma, beta, norm_out, norm16_out, addend, sum16_out, rows, C, eps, st)      \
             : launch_ln_dec<Td, bf16_t>(x, delta, gamma, beta, norm_out, norm16_out, addend, sum16_out, rows, C, eps, st);
  switch (delta_dtype) {
    case HIPIE_F32: HIPIE_LND(float)
    case HIPIE_F16: HIPIE_LND(f16_t)
    case HIPIE_BF16: HIPIE_LND(bf16_t)
    default: return set_err(HIPIE_EINVAL, "add_layernorm_dec: bad delta dtype %d", delta_dtype);
  }
#undef HIPIE_LND
}

extern "C" int hipie_add_cast(const float* a, const void* b, void* out, int64_t n, int dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(a && b && out, "add_cast: null pointer");
  HIPIE_REQUIRE(n >= 0 && n % 4 == 0, "add_cast: n must be a multiple of 4");
  HIPIE_REQUIRE(dtype == HIPIE_F16 || dtype == HIPIE_BF16 || dtype == HIPIE_HL8, "add_cast: dtype must be f16, bf16 or HL8");
  HIPIE_REQUIRE(dtype != HIPIE_HL8 || (n % 8 == 0 && (((uintptr_t)b | (uintptr_t)out) & 31) == 0), "add_cast: HL8 needs n %% 8 == 0 and 32-byte aligned buffers");
  if (n == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  const long n4 = n / 4;
  const unsigned grid = (unsigned)((n4 + 255) / 256);
  if (dtype == HIPIE_HL8)
    hipLaunchKernelGGL((add_cast_kernel<hl8_t>), dim3(grid), dim3(256), 0, st, a, (const hl8_t*)b, (hl8_t*)out, n4);
  else if (dtype == HIPIE_F16)
    hipLaunchKernelGGL((add_cast_kernel<f16_t>), dim3(grid), dim3(256), 0, st, a, (const f16_t*)b, (f16_t*)out, n4);
  else
    hipLaunchKernelGGL((add_cast_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a, (const bf16_t*)b, (bf16_t*)out, n4);
  return check_launch("add_cast");
}

extern "C" int hipie_add_layernorm_sum(const void* x, const void* delta, const float* gamma, const float* beta, void* res_out,
                                       void* norm_out, const void* addend, void* sum_out, int64_t rows, int C, float eps,
                                       int x_dtype, int delta_dtype, int norm_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE((addend == nullptr) == (sum_out == nullptr), "add_layernorm_sum: addend and sum_out go together");
  g_ln_addend = addend;
  g_ln_sum_out = sum_out;
  const int rc = hipie_add_layernorm(x, delta, gamma, beta, res_out, norm_out, rows, C, eps, x_dtype, delta_dtype, norm_dtype, stream);
  g_ln_addend = nullptr;
  g_ln_sum_out = nullptr;
  return rc;
}

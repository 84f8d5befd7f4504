// groupnorm.hip -- GroupNorm(32 groups of 8 channels) of the conv + GN blocks after the backbone (input_proj of both heads,
// the FPN lateral / output convs and the mask_features head of the MaskDINO pixel decoder: deformable_detr.py:139-160,
// maskdino_encoder.py:262-300), with the pieces eager PyTorch runs as separate passes folded in:
//   * channels-last input is normalised in place of layout (the library path first copies NHWC -> NCHW),
//   * an optional per-channel pre-bias (the bias of a bias-less library transposed conv), and an optional ReLU,
//   * output in the same layout, any of f32 / f16 / bf16.
// Two launches: partial (sum, sum of squares) per (image, group, chunk) in fp32, then the apply pass, which first reduces the
// partials of its groups (deterministic: no atomics).  HBM-bound: 2 reads + 1 write of the tensor.
// The sums are SHIFTED: every value has K = (first value of its group, pre-bias included) subtracted before it is accumulated --
// var = E[(a-K)^2] - (E[a-K])^2 is then free of the cancellation that E[a^2] - mean^2 suffers when |mean| >> std (real
// convolution biases folded in as pre-bias), at the cost of one extra scalar load per block.
#include "common.h"

namespace hipie {

constexpr int GN_CPG = 8;          // channels per group
constexpr int GN_MAXK = 512;       // partials per (image, group)

template <typename T> struct V8 {
  static __device__ __forceinline__ void ld(const T* p, float (&v)[8]) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    const t8 r = *reinterpret_cast<const t8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)r[i];
  }
  static __device__ __forceinline__ void st(T* p, const float (&v)[8]) {
    typedef T t8 __attribute__((ext_vector_type(8)));
    t8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (T)v[i];
    *reinterpret_cast<t8*>(p) = r;
  }
};

// ---- channels-last: x (B, HW, C), C = 8 G.  thread -> (group = t % G, pixel lane = t / G) ----
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ prebias,
                                                            float* __restrict__ part, int HW, int G, int nchunk, int per) {
  __shared__ float red[2][256];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int g = threadIdx.x % G, pl = threadIdx.x / G, npl = 256 / G;
  const int C = G * GN_CPG;
  float pb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) pb[i] = prebias ? prebias[g * 8 + i] : 0.f;
  const int p1 = min(HW, (chunk + 1) * per);
  const float K = (float)x[(long)b * HW * C + g * 8] + pb[0];          // the group's shift: its first value
  float s = 0.f, ss = 0.f;
  for (int p = chunk * per + pl; p < p1; p += npl) {
    float v[8];
    V8<T>::ld(x + ((long)b * HW + p) * C + g * 8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float a = v[i] + pb[i] - K; s += a; ss = fmaf(a, a, ss); }
  }
  red[0][threadIdx.x] = s;
  red[1][threadIdx.x] = ss;
  __syncthreads();
  if (threadIdx.x < G) {
    for (int k = 1; k < npl; ++k) { s += red[0][threadIdx.x + k * G]; ss += red[1][threadIdx.x + k * G]; }
    float* o = part + (((long)b * G + g) * nchunk + chunk) * 2;
    o[0] = s; o[1] = ss;
  }
}

template <typename T, typename OutT>
__global__ __launch_bounds__(256) void gn_apply_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ prebias,
                                                            const float* __restrict__ part, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, OutT* __restrict__ out, int HW, int G,
                                                            int nchunk, int per, float eps, int relu) {
  __shared__ float stat[2][64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int g = threadIdx.x % G, pl = threadIdx.x / G, npl = 256 / G;
  const int C = G * GN_CPG;
  if (threadIdx.x < G) {
    float s = 0.f, ss = 0.f;
    const float* pp = part + ((long)b * G + threadIdx.x) * nchunk * 2;
    for (int k = 0; k < nchunk; ++k) { s += pp[2 * k]; ss += pp[2 * k + 1]; }
    const float n = (float)HW * GN_CPG;
    const float ms = s / n;                                              // mean of the shifted values
    const float K = (float)x[(long)b * HW * C + threadIdx.x * 8] + (prebias ? prebias[threadIdx.x * 8] : 0.f);
    stat[0][threadIdx.x] = K + ms;
    stat[1][threadIdx.x] = rsqrtf(fmaxf(ss / n - ms * ms, 0.f) + eps);
  }
  __syncthreads();
  const float mean = stat[0][g], rstd = stat[1][g];
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = rstd * gamma[g * 8 + i];
    sh[i] = beta[g * 8 + i] + ((prebias ? prebias[g * 8 + i] : 0.f) - mean) * sc[i];
  }
  const int p1 = min(HW, (chunk + 1) * per);
  for (int p = chunk * per + pl; p < p1; p += npl) {
    const long off = ((long)b * HW + p) * C + g * 8;
    float v[8];
    V8<T>::ld(x + off, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = fmaf(v[i], sc[i], sh[i]); if (relu) v[i] = fmaxf(v[i], 0.f); }
    V8<OutT>::st(out + off, v);
  }
}

// ---- NHWC in, NCHW out (channels_last = 2): the normalised map leaves pixel-fastest -- the operand layout of the mask contraction
// (hipie_mask_einsum) -- without the 2 x (B, C, HW) transposing copy that followed the channels-last pass.  A workgroup owns 64 pixels of
// one image: the channels-last rows are read as the plain kernel reads them, the normalised values cross an LDS tile [channel][pixel]
// and leave as 256-byte runs of one channel.  HW % 64 == 0.
constexpr int GN_TP = 64;             // pixels per tile
template <typename T, typename OutT>
__global__ __launch_bounds__(256) void gn_apply_nhwc_to_nchw_kernel(const T* __restrict__ x, const float* __restrict__ prebias,
                                                                    const float* __restrict__ part, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, OutT* __restrict__ out, int HW, int G,
                                                                    int nchunk, float eps, int relu) {
  extern __shared__ __attribute__((aligned(16))) float gn_tile[];      // [C][GN_TP + 1]
  __shared__ float stat[2][64];
  const int b = blockIdx.y, p0 = blockIdx.x * GN_TP;
  const int g = threadIdx.x % G, pl = threadIdx.x / G, npl = 256 / G;
  const int C = G * GN_CPG;
  if (threadIdx.x < G) {
    float s = 0.f, ss = 0.f;
    const float* pp = part + ((long)b * G + threadIdx.x) * nchunk * 2;
    for (int k = 0; k < nchunk; ++k) { s += pp[2 * k]; ss += pp[2 * k + 1]; }
    const float n = (float)HW * GN_CPG;
    const float ms = s / n;
    const float K = (float)x[(long)b * HW * C + threadIdx.x * 8] + (prebias ? prebias[threadIdx.x * 8] : 0.f);
    stat[0][threadIdx.x] = K + ms;
    stat[1][threadIdx.x] = rsqrtf(fmaxf(ss / n - ms * ms, 0.f) + eps);
  }
  __syncthreads();
  const float mean = stat[0][g], rstd = stat[1][g];
  float sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    sc[i] = rstd * gamma[g * 8 + i];
    sh[i] = beta[g * 8 + i] + ((prebias ? prebias[g * 8 + i] : 0.f) - mean) * sc[i];
  }
  for (int p = pl; p < GN_TP; p += npl) {
    float v[8];
    V8<T>::ld(x + ((long)b * HW + p0 + p) * C + g * 8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      v[i] = fmaf(v[i], sc[i], sh[i]);
      if (relu) v[i] = fmaxf(v[i], 0.f);
      gn_tile[(g * 8 + i) * (GN_TP + 1) + p] = v[i];
    }
  }
  __syncthreads();
  struct alignas(sizeof(OutT) * 4) Pack { OutT v[4]; };
  const int q = threadIdx.x & 15, cr = threadIdx.x >> 4;                 // 16 threads x 4 pixels = one channel's 64 pixels; 16 channels per pass
  for (int c = cr; c < C; c += 16) {
    Pack o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o.v[i] = elem<OutT>::from_f32(gn_tile[c * (GN_TP + 1) + 4 * q + i]);
    *reinterpret_cast<Pack*>(out + ((long)b * C + c) * HW + p0 + 4 * q) = o;
  }
}

// ---- NCHW: x (B, C, HW).  one block per (chunk, image * channel); 8 elements per thread per step ----
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_nchw_kernel(const T* __restrict__ x, const float* __restrict__ prebias,
                                                            float* __restrict__ part, int HW, int C, int nchunk, int per) {
  __shared__ float red[2][4];
  const int bc = blockIdx.y, chunk = blockIdx.x;
  const int c = bc % C, b = bc / C;
  const float pb = prebias ? prebias[c] : 0.f;
  const T* xp = x + (long)bc * HW;
  const int e1 = min(HW, (chunk + 1) * per);
  const int c0 = (c / GN_CPG) * GN_CPG;                                  // the group's shift: the first value of its first channel
  const float K = (float)x[((long)b * C + c0) * HW] + (prebias ? prebias[c0] : 0.f);
  float s = 0.f, ss = 0.f;
  for (int e = chunk * per + threadIdx.x * 8; e < e1; e += 256 * 8) {
    float v[8];
    V8<T>::ld(xp + e, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float a = v[i] + pb - K; s += a; ss = fmaf(a, a, ss); }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = ss; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int G = C / GN_CPG, g = c / GN_CPG, K = nchunk * GN_CPG;
    float* o = part + (((long)b * G + g) * K + (c % GN_CPG) * nchunk + chunk) * 2;
    o[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    o[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}

template <typename T, typename OutT>
__global__ __launch_bounds__(256) void gn_apply_nchw_kernel(const T* __restrict__ x, const float* __restrict__ prebias,
                                                            const float* __restrict__ part, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, OutT* __restrict__ out, int HW, int C,
                                                            int nchunk, int per, float eps, int relu) {
  __shared__ float stat[2];
  const int bc = blockIdx.y, chunk = blockIdx.x;
  const int c = bc % C, b = bc / C;
  if (threadIdx.x < 64) {
    const int G = C / GN_CPG, g = c / GN_CPG, K = nchunk * GN_CPG;
    const float* pp = part + ((long)b * G + g) * K * 2;
    float s = 0.f, ss = 0.f;
    for (int k = threadIdx.x; k < K; k += 64) { s += pp[2 * k]; ss += pp[2 * k + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if (threadIdx.x == 0) {
      const float n = (float)HW * GN_CPG;
      const float ms = s / n;
      const int c0 = g * GN_CPG;
      const float K = (float)x[((long)b * C + c0) * HW] + (prebias ? prebias[c0] : 0.f);
      stat[0] = K + ms;
      stat[1] = rsqrtf(fmaxf(ss / n - ms * ms, 0.f) + eps);
    }
  }
  __syncthreads();
  const float sc = stat[1] * gamma[c];
  const float sh = beta[c] + ((prebias ? prebias[c] : 0.f) - stat[0]) * sc;
  const T* xp = x + (long)bc * HW;
  OutT* op = out + (long)bc * HW;
  const int e1 = min(HW, (chunk + 1) * per);
  for (int e = chunk * per + threadIdx.x * 8; e < e1; e += 256 * 8) {
    float v[8];
    V8<T>::ld(xp + e, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = fmaf(v[i], sc, sh); if (relu) v[i] = fmaxf(v[i], 0.f); }
    V8<OutT>::st(op + e, v);
  }
}

template <typename T, typename OutT>
static int launch_gn(const void* x, const float* prebias, const float* gamma, const float* beta, void* out, float* part, int B,
                     int C, int HW, int nhwc, float eps, int relu, hipStream_t st) {
  const int G = C / GN_CPG;
  if (nhwc) {
    const int npl = 256 / G;
    int nchunk = max(1, min(64, HW / (npl * 16)));
    const int per = (HW + nchunk - 1) / nchunk;
    nchunk = (HW + per - 1) / per;
    hipLaunchKernelGGL((gn_stats_nhwc_kernel<T>), dim3(nchunk, B), dim3(256), 0, st, (const T*)x, prebias, part, HW, G, nchunk, per);
    if (nhwc == 2) {                         // channels-last in, NCHW out
      const size_t lds = (size_t)C * (GN_TP + 1) * sizeof(float);
      auto kern = gn_apply_nhwc_to_nchw_kernel<T, OutT>;
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, dim3(HW / GN_TP, B), dim3(256), lds, st, (const T*)x, prebias, part, gamma, beta, (OutT*)out, HW, G, nchunk,
                         eps, relu);
      return check_launch("group_norm");
    }
    hipLaunchKernelGGL((gn_apply_nhwc_kernel<T, OutT>), dim3(nchunk, B), dim3(256), 0, st, (const T*)x, prebias, part, gamma, beta,
                       (OutT*)out, HW, G, nchunk, per, eps, relu);
  } else {
    int nchunk = max(1, min(GN_MAXK / GN_CPG, HW / 8192));
    int per = (HW + nchunk - 1) / nchunk;
    per = (per + 7) / 8 * 8;
    nchunk = (HW + per - 1) / per;
    hipLaunchKernelGGL((gn_stats_nchw_kernel<T>), dim3(nchunk, B * C), dim3(256), 0, st, (const T*)x, prebias, part, HW, C, nchunk, per);
    hipLaunchKernelGGL((gn_apply_nchw_kernel<T, OutT>), dim3(nchunk, B * C), dim3(256), 0, st, (const T*)x, prebias, part, gamma,
                       beta, (OutT*)out, HW, C, nchunk, per, eps, relu);
  }
  return check_launch("group_norm");
}

template <typename T>
static int gn_out(int od, const void* x, const float* prebias, const float* gamma, const float* beta, void* out, float* part, int B,
                  int C, int HW, int nhwc, float eps, int relu, hipStream_t st) {
  switch (od) {
    case HIPIE_F32: return launch_gn<T, float>(x, prebias, gamma, beta, out, part, B, C, HW, nhwc, eps, relu, st);
    case HIPIE_F16: return launch_gn<T, f16_t>(x, prebias, gamma, beta, out, part, B, C, HW, nhwc, eps, relu, st);
    case HIPIE_BF16: return launch_gn<T, bf16_t>(x, prebias, gamma, beta, out, part, B, C, HW, nhwc, eps, relu, st);
    default: return set_err(HIPIE_EINVAL, "group_norm: bad out dtype %d", od);
  }
}

}  // namespace hipie

extern "C" int hipie_group_norm(const void* x, const float* prebias, const float* gamma, const float* beta, void* out,
                                float* workspace, int B, int C, int HW, int groups, int channels_last, float eps, int relu,
                                int x_dtype, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(x && gamma && beta && out && workspace, "group_norm: null pointer");
  HIPIE_REQUIRE(B >= 0 && C > 0 && HW > 0, "group_norm: bad shape");
  HIPIE_REQUIRE(groups > 0 && C == groups * GN_CPG && groups <= 64 && 256 % groups == 0,
                "group_norm: %d channels in %d groups unsupported (8 channels per group, 256 %% groups == 0)", C, groups);
  HIPIE_REQUIRE(channels_last || HW % 8 == 0, "group_norm: NCHW needs H*W %% 8 == 0 (got %d)", HW);
  HIPIE_REQUIRE(channels_last >= 0 && channels_last <= 2 && (channels_last != 2 || HW % 64 == 0),
                "group_norm: channels_last %d (0 NCHW, 1 channels-last, 2 channels-last in / NCHW out: H*W %% 64 == 0, got %d)", channels_last, HW);
  if (B == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (x_dtype) {
    case HIPIE_F32: return gn_out<float>(out_dtype, x, prebias, gamma, beta, out, workspace, B, C, HW, channels_last, eps, relu, st);
    case HIPIE_F16: return gn_out<f16_t>(out_dtype, x, prebias, gamma, beta, out, workspace, B, C, HW, channels_last, eps, relu, st);
    case HIPIE_BF16: return gn_out<bf16_t>(out_dtype, x, prebias, gamma, beta, out, workspace, B, C, HW, channels_last, eps, relu, st);
    default: return set_err(HIPIE_EINVAL, "group_norm: bad x dtype %d", x_dtype);
  }
}

// postprocess.hip -- device kernels of the post-processing row: batched NMS and instance-mask finalisation.
//
// hipie_batched_nms replaces the per-image torchvision.ops.batched_nms call of HIPIE_IMG.inference
// (projects/HIPIE/hipie/hipie_img.py:626-629): one workgroup per image keeps the whole suppression matrix
// (Q x Q bits, 128 KB at Q = 1024) in LDS, so the IoU pass and the greedy scan are a single launch with no global
// scratch and no host round trip.  Integer result: bit-exact against the oracle (explicit _rn arithmetic, no FMA
// contraction in the IoU).
// hipie_mask_finalize fuses F.interpolate(x4, bilinear) -> sigmoid -> "> mask_thres" -> crop (hipie_img.py:693-699) with
// the nearest resize to the output resolution of segmentation_postprocess (models/ddetrs.py:1065-1070): the (n, H, W)
// fp32 intermediate (420 MB per image at 1024^2) never exists; reads the stride-4 logits, writes the final bytes.
#include "common.h"

namespace hipie {

// ---------------------------------------------------------------------------------------------------- NMS
constexpr int NMS_THREADS = 1024;
constexpr int NMS_MAXQ = 1024;

__device__ __forceinline__ bool iou_gt(const float* a, const float* b, float aa, float ab, float thr) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float w = fmaxf(__fsub_rn(right, left), 0.f), h = fmaxf(__fsub_rn(bottom, top), 0.f);
  if (w <= 0.f || h <= 0.f) return false;       // disjoint (the vast majority of pairs): 0 / union > thr is false; skips the division
  const float inter = __fmul_rn(w, h);
  const float uni = __fsub_rn(__fadd_rn(aa, ab), inter);
  return __fdiv_rn(inter, uni) > thr;
}

__global__ __launch_bounds__(NMS_THREADS) void batched_nms_kernel(
    const float* __restrict__ boxes, const int64_t* __restrict__ classes, const int32_t* __restrict__ order,
    int32_t* __restrict__ keep, int32_t* __restrict__ count, int Q, float thr, int trick) {
  extern __shared__ unsigned char smem_raw[];
  const int W = (Q + 63) >> 6;
  float* sbox = reinterpret_cast<float*>(smem_raw);                 // (Q, 4) offset xyxy in sorted order
  float* sarea = sbox + 4 * NMS_MAXQ;                               // (Q)
  int* scls = reinterpret_cast<int*>(sarea + NMS_MAXQ);             // (Q)
  float* sred = reinterpret_cast<float*>(scls + NMS_MAXQ);          // (16) per-wave maxima
  unsigned long long* smask = reinterpret_cast<unsigned long long*>(sred + 16);   // (Q, W)
  const int b = blockIdx.x, t = threadIdx.x;
  boxes += (size_t)b * Q * 4;
  classes += (size_t)b * Q;
  order += (size_t)b * Q;
  keep += (size_t)b * Q;

  // 1. sorted xyxy (box_cxcywh_to_xyxy, util/box_ops.py:17-21; 0.5*w is exact so cx - 0.5*w has one rounding)
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  float mx = -INFINITY;
  int cls = 0;
  if (t < Q) {
    const int q = order[t];
    const float4 c = *reinterpret_cast<const float4*>(boxes + 4 * q);
    x[0] = __fsub_rn(c.x, 0.5f * c.z);
    x[1] = __fsub_rn(c.y, 0.5f * c.w);
    x[2] = __fadd_rn(c.x, 0.5f * c.z);
    x[3] = __fadd_rn(c.y, 0.5f * c.w);
    mx = fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3]));
    cls = (int)classes[q];
  }
  // 2. max coordinate of the image (torchvision _batched_nms_coordinate_trick: offsets = idx * (max + 1))
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((t & 63) == 0) sred[t >> 6] = mx;
  __syncthreads();
  mx = sred[0];
  for (int i = 1; i < NMS_THREADS / 64; ++i) mx = fmaxf(mx, sred[i]);
  if (t < Q) {
    if (trick) {
      const float off = __fmul_rn((float)cls, __fadd_rn(mx, 1.0f));
      for (int i = 0; i < 4; ++i) x[i] = __fadd_rn(x[i], off);
    }
    for (int i = 0; i < 4; ++i) sbox[4 * t + i] = x[i];
    sarea[t] = __fmul_rn(__fsub_rn(x[2], x[0]), __fsub_rn(x[3], x[1]));
    scls[t] = cls;
  }
  __syncthreads();
  // 3. suppression bits: smask[i][w] bit j  <=>  j > i (sorted positions) and IoU(i, j) > thr
  for (int item = t; item < Q * W; item += NMS_THREADS) {
    const int i = item / W, w = item - i * W;
    unsigned long long bits = 0ull;
    const int j0 = w << 6;
    if (j0 + 63 > i) {
      const float* a = sbox + 4 * i;
      const float aa = sarea[i];
      const int ci = scls[i];
      const int jend = min(64, Q - j0);
      for (int jj = 0; jj < jend; ++jj) {
        const int j = j0 + jj;
        if (j <= i) continue;
        if (!trick && scls[j] != ci) continue;
        if (iou_gt(a, sbox + 4 * j, aa, sarea[j], thr)) bits |= 1ull << jj;
      }
    }
    smask[item] = bits;
  }
  __syncthreads();
  // 4. greedy scan by wave 0: lane w owns word w of the removed set
  if (t < 64) {
    unsigned long long remv = 0ull;
    int n = 0;
    for (int i = 0; i < Q; ++i) {
      const unsigned long long word = __shfl(remv, i >> 6);
      if (!((word >> (i & 63)) & 1ull)) {
        if (t == 0) keep[n] = order[i];
        ++n;
        if (t < W) remv |= smask[i * W + t];
      }
    }
    for (int i = n + t; i < Q; i += 64) keep[i] = -1;
    if (t == 0) count[b] = n;
  }
}

// ---------------------------------------------------------------------------------------------------- mask finalize
template <typename T>
__global__ __launch_bounds__(256) void mask_finalize_kernel(
    const T* __restrict__ masks, const int32_t* __restrict__ qidx, int hm, int wm, int up, int crop_h, int crop_w,
    int out_h, int out_w, float thr, uint8_t* __restrict__ out) {
  const int n = blockIdx.y;
  const int w4 = (out_w + 3) >> 2;
  const int item = blockIdx.x * 256 + threadIdx.x;          // (oy, ox0 / 4) flattened
  if (item >= out_h * w4) return;
  const int oy = item / w4;
  const int ox0 = (item - oy * w4) * 4;
  const T* m = masks + (size_t)(qidx ? qidx[n] : n) * hm * wm;
  // nearest (legacy) source index in the cropped x`up` grid: min(floor(dst * in/out), in - 1)
  const float sy = (float)crop_h / (float)out_h, sx = (float)crop_w / (float)out_w;
  const int uy = min((int)floorf(oy * sy), crop_h - 1);
  // bilinear, align_corners False: src = (dst + 0.5) / up - 0.5, clamped at 0
  const float inv = 1.0f / (float)up;
  float fy = fmaxf(((float)uy + 0.5f) * inv - 0.5f, 0.f);
  const int y0 = (int)fy, y1 = y0 + (y0 < hm - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1;
  uint8_t res[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ox = ox0 + k;
    if (ox >= out_w) break;
    const int ux = min((int)floorf(ox * sx), crop_w - 1);
    float fx = fmaxf(((float)ux + 0.5f) * inv - 0.5f, 0.f);
    const int x0 = (int)fx, x1 = x0 + (x0 < wm - 1 ? 1 : 0);
    const float lx1 = fx - (float)x0, lx0 = 1.f - lx1;
    const float a = elem<T>::to_f32(m[y0 * wm + x0]), bq = elem<T>::to_f32(m[y0 * wm + x1]);
    const float c = elem<T>::to_f32(m[y1 * wm + x0]), d = elem<T>::to_f32(m[y1 * wm + x1]);
    const float v = ly0 * (lx0 * a + lx1 * bq) + ly1 * (lx0 * c + lx1 * d);
    res[k] = (1.f / (1.f + expf(-v))) > thr ? 1 : 0;
  }
  uint8_t* o = out + ((size_t)n * out_h + oy) * out_w + ox0;
  if (ox0 + 3 < out_w && (((size_t)n * out_h + oy) * out_w + ox0) % 4 == 0) {
    *reinterpret_cast<uint32_t*>(o) = (uint32_t)res[0] | ((uint32_t)res[1] << 8) | ((uint32_t)res[2] << 16) | ((uint32_t)res[3] << 24);
  } else {
    for (int k = 0; k < 4 && ox0 + k < out_w; ++k) o[k] = res[k];
  }
}

}  // namespace hipie

using namespace hipie;

extern "C" int hipie_batched_nms(const float* boxes, const int64_t* classes, const int32_t* order, int32_t* keep,
                                 int32_t* count, int B, int Q, float iou_threshold, int coordinate_trick, void* stream) {
  HIPIE_REQUIRE(B >= 0 && Q >= 0, "batched_nms: negative size");
  HIPIE_REQUIRE(Q <= NMS_MAXQ, "batched_nms: Q=%d > %d (suppression matrix must fit the 160 KB LDS)", Q, NMS_MAXQ);
  if (B == 0) return HIPIE_OK;
  HIPIE_REQUIRE(count, "batched_nms: null pointer");
  if (Q == 0) {
    (void)hipMemsetAsync(count, 0, sizeof(int32_t) * B, (hipStream_t)stream);
    return check_launch("batched_nms(memset)");
  }
  HIPIE_REQUIRE(boxes && classes && order && keep, "batched_nms: null pointer");
  const int W = (Q + 63) >> 6;
  const size_t smem = sizeof(float) * (4 * NMS_MAXQ + NMS_MAXQ) + sizeof(int) * NMS_MAXQ + sizeof(float) * 16 +
                      sizeof(unsigned long long) * (size_t)Q * W;
  static bool attr_set[64] = {false};               // the attribute is per device
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(batched_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return set_err(HIPIE_ELAUNCH, "batched_nms: cannot raise the dynamic LDS limit");
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  batched_nms_kernel<<<B, NMS_THREADS, smem, (hipStream_t)stream>>>(boxes, classes, order, keep, count, Q, iou_threshold,
                                                                   coordinate_trick);
  return check_launch("batched_nms");
}

extern "C" int hipie_mask_finalize(const void* masks, int dtype, const int32_t* qidx, int n, int hm, int wm, int up,
                                   int crop_h, int crop_w, int out_h, int out_w, float threshold, uint8_t* out,
                                   void* stream) {
  HIPIE_REQUIRE(n >= 0 && hm > 0 && wm > 0 && up > 0, "mask_finalize: bad geometry");
  if (n == 0) return HIPIE_OK;
  HIPIE_REQUIRE(masks && out, "mask_finalize: null pointer");
  HIPIE_REQUIRE(crop_h > 0 && crop_w > 0 && crop_h <= hm * up && crop_w <= wm * up, "mask_finalize: crop outside the mask");
  HIPIE_REQUIRE(out_h > 0 && out_w > 0 && n <= 65535, "mask_finalize: bad output size");
  if (n == 0) return HIPIE_OK;
  dim3 grid(ceil_div(out_h * ceil_div(out_w, 4), 256), n);
  hipStream_t st = (hipStream_t)stream;
  switch (dtype) {
    case HIPIE_F32:
      mask_finalize_kernel<float><<<grid, 256, 0, st>>>((const float*)masks, qidx, hm, wm, up, crop_h, crop_w, out_h, out_w, threshold, out);
      break;
    case HIPIE_BF16:
      mask_finalize_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)masks, qidx, hm, wm, up, crop_h, crop_w, out_h, out_w, threshold, out);
      break;
    case HIPIE_F16:
      mask_finalize_kernel<f16_t><<<grid, 256, 0, st>>>((const f16_t*)masks, qidx, hm, wm, up, crop_h, crop_w, out_h, out_w, threshold, out);
      break;
    default:
      return set_err(HIPIE_EINVAL, "mask_finalize: unknown dtype %d", dtype);
  }
  return check_launch("mask_finalize");
}

// attn_f32.hip -- EXACT fp32 softmax attention for the SMALL attentions of the path: BERT self-attention (<= 512 tokens per chunk, 12
// heads x 64; transformers' BertSelfAttention behind models/deformable_detr/bert_model.py:54-58) and the decoders' query self-attention
// (900 + 10 / 300 queries, 8 heads x 32; nn.MultiheadAttention in deformable_transformer_dino.py:418-432 and dino_decoder.py:222-240).
//
// Why not the MFMA flash kernel: at the headline configuration (full ViT-H, 1024^2) the model amplifies operand rounding by two to three
// orders of magnitude through the six decoder layers (tests/study/prec_sim.py on tests/golden/e2e_full.npz): single-fp16 q / k / v in the BERT
// attention alone moves pred_masks by 4.2e-3, in the decoder self-attention by 5e-4 -- and these attentions are 0.3 % of the step's
// flops.  So they run as the reference runs them: fp32 operands, fp32 products.
//
// One THREAD per query row: q (hd registers) and the output accumulator (hd registers) never leave the lane; key / value tiles of 32 keys
// are staged in LDS by the whole workgroup (coalesced) and read back as LDS broadcasts (every lane reads the same address: conflict
// free); per tile: 32 scores -> tile maximum -> one rescale -> 32 exp2 and the P.V FMAs.  fp32 FMA chains in key order -- no reduction
// across lanes, no atomics: deterministic.
#include "common.h"

namespace hipie {

struct AFParams {
  const float *q, *k, *v;
  const unsigned char* key_mask;      // (B, Nk), 1 = attend; or null
  const unsigned char* query_mask;    // (B, Nq, Nk), 1 = attend: a mask per QUERY row (MaskCLIP's mask tokens); or null
  float* out;                         // (B, Nq, H * HD)
  int B, H, Nq, Nk;
  long q_sb, q_st, k_sb, k_st, v_sb, v_st;      // batch / token strides in elements; a head is HD contiguous elements at h * HD
  float scale_log2e;
};

template <int HD>
__global__ __launch_bounds__(128) void attn_f32_kernel(const AFParams p) {
  constexpr int TK = 32;
  __shared__ __attribute__((aligned(16))) float ks[TK * HD];
  __shared__ __attribute__((aligned(16))) float vs[TK * HD];
  __shared__ float km[TK];
  const int tid = threadIdx.x;
  const int qtiles = (p.Nq + 127) / 128;
  const int bh = blockIdx.x / qtiles, qt = blockIdx.x % qtiles;
  const int b = bh / p.H, h = bh % p.H;
  const int qi = qt * 128 + tid;
  const bool live = qi < p.Nq;
  const float* qrow = p.q + b * p.q_sb + (long)min(qi, p.Nq - 1) * p.q_st + h * HD;
  const unsigned char* qm = p.query_mask ? p.query_mask + ((long)b * p.Nq + min(qi, p.Nq - 1)) * p.Nk : nullptr;
  float q[HD], o[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const float4 t = *reinterpret_cast<const float4*>(qrow + d);
    q[d] = t.x * p.scale_log2e; q[d + 1] = t.y * p.scale_log2e; q[d + 2] = t.z * p.scale_log2e; q[d + 3] = t.w * p.scale_log2e;
    o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;
  const float* kb = p.k + b * p.k_sb + h * HD;
  const float* vb = p.v + b * p.v_sb + h * HD;
  for (int k0 = 0; k0 < p.Nk; k0 += TK) {
    __syncthreads();                                   // the previous tile has been consumed
    for (int i = tid; i < TK * HD / 4; i += 128) {     // 16-byte pieces, coalesced along the head dimension
      const int j = i / (HD / 4), d = 4 * (i % (HD / 4));
      const int kj = min(k0 + j, p.Nk - 1);
      *reinterpret_cast<float4*>(ks + j * HD + d) = *reinterpret_cast<const float4*>(kb + (long)kj * p.k_st + d);
      *reinterpret_cast<float4*>(vs + j * HD + d) = *reinterpret_cast<const float4*>(vb + (long)kj * p.v_st + d);
    }
    if (tid < TK) km[tid] = (k0 + tid < p.Nk && (p.key_mask == nullptr || p.key_mask[(long)b * p.Nk + k0 + tid] != 0)) ? 0.f : -INFINITY;
    __syncthreads();
    float s[TK];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      float a0 = 0.f, a1 = 0.f;                        // two chains: enough ILP for the FMA latency, fixed order
#pragma unroll
      for (int d = 0; d < HD; d += 8) {
        const float4 k4 = *reinterpret_cast<const float4*>(ks + j * HD + d), k5 = *reinterpret_cast<const float4*>(ks + j * HD + d + 4);
        a0 = fmaf(q[d], k4.x, a0); a0 = fmaf(q[d + 1], k4.y, a0); a0 = fmaf(q[d + 2], k4.z, a0); a0 = fmaf(q[d + 3], k4.w, a0);
        a1 = fmaf(q[d + 4], k5.x, a1); a1 = fmaf(q[d + 5], k5.y, a1); a1 = fmaf(q[d + 6], k5.z, a1); a1 = fmaf(q[d + 7], k5.w, a1);
      }
      s[j] = (a0 + a1) + km[j];
      if (qm && k0 + j < p.Nk && qm[k0 + j] == 0) s[j] = -INFINITY;             // this row may not see key k0 + j
      mx = fmaxf(mx, s[j]);
    }
    if (mx > m_run) {                                  // per-lane branch: the rescale is rare after the first tiles
      const float alpha = __builtin_amdgcn_exp2f(m_run - mx);      // m_run = -inf -> 0
      m_run = mx;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] *= alpha;
    }
    if (m_run == -INFINITY) continue;                  // every key so far masked: nothing to add (and no -inf - -inf)
#pragma unroll
    for (int j = 0; j < TK; ++j) {
      const float pj = __builtin_amdgcn_exp2f(s[j] - m_run);
      l_run += pj;
#pragma unroll
      for (int d = 0; d < HD; d += 4) {
        const float4 v4 = *reinterpret_cast<const float4*>(vs + j * HD + d);
        o[d] = fmaf(pj, v4.x, o[d]); o[d + 1] = fmaf(pj, v4.y, o[d + 1]); o[d + 2] = fmaf(pj, v4.z, o[d + 2]); o[d + 3] = fmaf(pj, v4.w, o[d + 3]);
      }
    }
  }
  if (live) {
    const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
    float* orow = p.out + ((long)b * p.Nq + qi) * (p.H * HD) + h * HD;
#pragma unroll
    for (int d = 0; d < HD; d += 4) *reinterpret_cast<float4*>(orow + d) = make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv);
  }
}

}  // namespace hipie

static int attn_f32_launch(const float* q, const float* k, const float* v, const unsigned char* key_mask, const unsigned char* query_mask,
                           float* out, int B, int H, int Nq, int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st,
                           int64_t v_sb, int64_t v_st, float scale, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(q && k && v && out, "attn_f32: null pointer");
  HIPIE_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attn_f32: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
  HIPIE_REQUIRE(head_dim == 32 || head_dim == 64, "attn_f32: head_dim 32 / 64 only (got %d)", head_dim);
  HIPIE_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0 && ((q_sb | q_st | k_sb | k_st | v_sb | v_st) & 3) == 0,
                "attn_f32: pointers must be 16-byte aligned and strides multiples of 4 elements");
  AFParams p;
  p.q = q; p.k = k; p.v = v; p.key_mask = key_mask; p.query_mask = query_mask; p.out = out;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_sb = q_sb; p.q_st = q_st; p.k_sb = k_sb; p.k_st = k_st; p.v_sb = v_sb; p.v_st = v_st;
  p.scale_log2e = scale * 1.4426950408889634f;
  const unsigned grid = (unsigned)(B * H * ((Nq + 127) / 128));
  if (head_dim == 32) hipLaunchKernelGGL(attn_f32_kernel<32>, dim3(grid), dim3(128), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_f32_kernel<64>, dim3(grid), dim3(128), 0, (hipStream_t)stream, p);
  return check_launch("attn_f32");
}

extern "C" int hipie_attn_f32(const float* q, const float* k, const float* v, const unsigned char* key_mask, float* out, int B, int H, int Nq,
                              int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st,
                              float scale, void* stream) {
  return attn_f32_launch(q, k, v, key_mask, nullptr, out, B, H, Nq, Nk, head_dim, q_sb, q_st, k_sb, k_st, v_sb, v_st, scale, stream);
}

extern "C" int hipie_attn_f32_rows(const float* q, const float* k, const float* v, const unsigned char* query_mask, float* out, int B, int H,
                                   int Nq, int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb,
                                   int64_t v_st, float scale, void* stream) {
  return attn_f32_launch(q, k, v, nullptr, query_mask, out, B, H, Nq, Nk, head_dim, q_sb, q_st, k_sb, k_st, v_sb, v_st, scale, stream);
}

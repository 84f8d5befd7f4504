// attn_split.hip -- the SMALL softmax attentions of the path on the matrix pipe at fp32-class accuracy (round 4): BERT self-attention
// (<= 512 tokens per chunk, 12 heads x 64; transformers' BertSelfAttention behind models/deformable_detr/bert_model.py:54-58) and the
// decoders' query self-attention (900 + 10 / 300 queries, 8 heads x 32; nn.MultiheadAttention in deformable_transformer_dino.py:418-432
// and maskdino/transformer_decoder/dino_decoder.py:222-240).
//
// Round 3 ran them as exact fp32 FMA chains, one thread per query row (attn_f32.hip: 27 launches, 4.4 ms per step): single-fp16 operands
// are not an option here (BERT attention alone moves pred_masks by 4e-3 at the headline configuration).  This kernel
// uses the arithmetic of hipie_vit_attn_split instead: q (pre-multiplied by scale * log2 e), k and v are split IN THE KERNEL into fp16 pairs
// x = hi + lo (22 mantissa bits), every logit is the three-product sum  q_lo.k_hi + q_hi.k_lo + q_hi.k_hi  with fp32 accumulation, the
// probabilities are an fp16 pair as well and  O += V_hi^T.(P_hi + P_lo) + V_lo^T.P_hi.  Same operands and output as hipie_attn_f32 (fp32
// views with strides), same semantics (key mask, fully masked rows give zeros): tests hold it to the exact kernel at 2e-6.
//
// Structure: swapped products as in vit_attn_split.hip -- S^T = K.Q^T (keys are the 32 MFMA rows, a lane owns one query), online softmax
// per lane over its 16 rows + one cross-half exchange, O^T = V^T.P^T with P used in place as the B operand and V^T fetched by
// ds_read_b64_tr_b16.  A workgroup = 4 waves x 32 queries of one (batch, head); key / value tiles of 64 keys are converted and staged in
// LDS by the whole workgroup (single buffer, two barriers per tile: these launches are 10 .. 50 us of latency, not throughput).
#include "common.h"
#include "mfma.h"

namespace hipie {

struct ASParams {
  const float *q, *k, *v;
  const unsigned char* key_mask;      // (B, Nk), 1 = attend; or null
  const unsigned char* query_mask;    // (B, Nq, Nk), 1 = attend: a mask per QUERY row (MaskCLIP's mask tokens); or null
  float* out;                         // (B, Nq, H * HD)
  int B, H, Nq, Nk;
  long q_sb, q_st, k_sb, k_st, v_sb, v_st;
  float scale_log2e;
};

__device__ __forceinline__ float as_xhalf_max(float x) {
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  const unsigned int u = __builtin_bit_cast(unsigned int, x);
  const u32x2 r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return fmaxf(__builtin_bit_cast(float, (unsigned int)r[0]), __builtin_bit_cast(float, (unsigned int)r[1]));
}

template <int HD>
__global__ __launch_bounds__(256) void attn_split_kernel(const ASParams p) {
  typedef f16_t T;
  typedef Mfma32<T>::frag frag;
  typedef Mfma32<T>::half_frag hfrag;
  constexpr int KT = 64, NB = 2;               // keys per tile, 32-key MFMA blocks per tile
  constexpr int KS = HD / 16;                  // k16 steps of Q.K^T
  constexpr int DB = HD / 32;                  // 32-row d blocks of O^T
  constexpr int KSTR = HD + 8;                 // K plane row stride (elements): conflict-free b128 reads (vit_attn.hip)
  constexpr int VSTR = (HD == 32) ? 32 : HD + 32;
  __shared__ __attribute__((aligned(16))) T Kh[KT * KSTR], Kl[KT * KSTR], Vh[KT * VSTR], Vl[KT * VSTR];
  __shared__ float km[KT];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int qtiles = (p.Nq + 127) / 128;
  const int bh = blockIdx.x / qtiles, qt = blockIdx.x % qtiles;
  const int b = bh / p.H, h = bh % p.H;
  const int qi = qt * 128 + wave * 32 + li;
  const float* qrow = p.q + b * p.q_sb + (long)min(qi, p.Nq - 1) * p.q_st + h * HD;
  const unsigned char* qm = p.query_mask ? p.query_mask + ((long)b * p.Nq + min(qi, p.Nq - 1)) * p.Nk : nullptr;     // this lane's query row

  // ---- Q fragments (B operand): lane (query li, half hi) holds head-dim group 2 ks + hi, pre-scaled into the exp2 domain, as an fp16 pair ----
  frag qh[KS], ql[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float4 a = *reinterpret_cast<const float4*>(qrow + 16 * ks + 8 * hi), c = *reinterpret_cast<const float4*>(qrow + 16 * ks + 8 * hi + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      T hh, ll;
      hl_split(x[e] * p.scale_log2e, hh, ll);
      qh[ks][e] = hh;
      ql[ks][e] = ll;
    }
  }

  f32x16 O[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const float* kb = p.k + b * p.k_sb + h * HD;
  const float* vb = p.v + b * p.v_sb + h * HD;
  const int l16 = lane & 15, g1 = (lane >> 4) & 1;
  const int vlane = (4 * hi + (l16 >> 2)) * VSTR + 16 * g1 + 4 * (l16 & 3);   // this lane's V^T read offset inside a (16-key step, d block)
  const int klane = li * KSTR + 8 * hi;

  for (int k0 = 0; k0 < p.Nk; k0 += KT) {
    __syncthreads();                                   // the previous tile has been consumed
    // ---- stage the tile: fp32 rows -> hi / lo planes (keys beyond Nk repeat the last row; their scores are masked) ----
    for (int i = tid; i < KT * HD / 4; i += 256) {
      const int j = i / (HD / 4), d = 4 * (i % (HD / 4));
      const int kj = min(k0 + j, p.Nk - 1);
      const float4 kk = *reinterpret_cast<const float4*>(kb + (long)kj * p.k_st + d);
      const float4 vv = *reinterpret_cast<const float4*>(vb + (long)kj * p.v_st + d);
      const float kx[4] = {kk.x, kk.y, kk.z, kk.w}, vx[4] = {vv.x, vv.y, vv.z, vv.w};
      f16x4 a, bq, c, dq;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T hh, ll;
        hl_split(kx[e], hh, ll);
        a[e] = hh; bq[e] = ll;
        hl_split(vx[e], hh, ll);
        c[e] = hh; dq[e] = ll;
      }
      *reinterpret_cast<f16x4*>(Kh + j * KSTR + d) = a;
      *reinterpret_cast<f16x4*>(Kl + j * KSTR + d) = bq;
      *reinterpret_cast<f16x4*>(Vh + j * VSTR + d) = c;
      *reinterpret_cast<f16x4*>(Vl + j * VSTR + d) = dq;
    }
    if (tid < KT) km[tid] = (k0 + tid < p.Nk && (p.key_mask == nullptr || p.key_mask[(long)b * p.Nk + k0 + tid] != 0)) ? 0.f : -INFINITY;
    __syncthreads();

    // ---- S^T = K . Q'^T, three products per k-step; + mask ----
    f32x16 S[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int r = 0; r < 16; ++r) S[blk][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const frag kfh = *reinterpret_cast<const frag*>(Kh + klane + 32 * blk * KSTR + 16 * ks);
        const frag kfl = *reinterpret_cast<const frag*>(Kl + klane + 32 * blk * KSTR + 16 * ks);
        S[blk] = Mfma32<T>::mma(kfl, qh[ks], S[blk]);
        S[blk] = Mfma32<T>::mma(kfh, ql[ks], S[blk]);
        S[blk] = Mfma32<T>::mma(kfh, qh[ks], S[blk]);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kk = 32 * blk + crow(r, hi);
        S[blk][r] += km[kk];
        if (qm && k0 + kk < p.Nk && qm[k0 + kk] == 0) S[blk][r] = -INFINITY;                 // this row may not see key k0 + kk
        mx = fmaxf(mx, S[blk][r]);
      }
    mx = as_xhalf_max(mx);
    if (mx > m_run) {                                  // per-lane (= per query) rescale
      const float alpha = __builtin_amdgcn_exp2f(m_run - mx);      // m_run = -inf -> 0
      m_run = mx;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[d][r] *= alpha;
    }
    // every key so far masked (m_run = -inf): the probabilities of this tile are exp2(-inf - 0) = 0
    const float mref = (m_run == -INFINITY) ? 0.f : m_run;

    // ---- P = exp2(S - m) as an fp16 PAIR;  O^T += V_hi^T.(P_hi + P_lo)^T + V_lo^T.P_hi^T, one 16-key step at a time ----
#pragma unroll
    for (int step = 0; step < 2 * NB; ++step) {
      const int blk = step >> 1, s2 = step & 1;
      frag pf, pfl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float pv = __builtin_amdgcn_exp2f(S[blk][8 * s2 + j] - mref);
        l_run += pv;
        const T ph = (T)pv;
        pf[j] = ph;
        pfl[j] = (T)(pv - (float)ph);
      }
#pragma unroll
      for (int d = 0; d < DB; ++d) {
        const T* vp = Vh + (16 * step) * VSTR + vlane + 32 * d;
        const T* vq = Vl + (16 * step) * VSTR + vlane + 32 * d;
        const hfrag a0 = Mfma32<T>::tr_read(vp), a1 = Mfma32<T>::tr_read(vp + 8 * VSTR);
        const hfrag b0 = Mfma32<T>::tr_read(vq), b1 = Mfma32<T>::tr_read(vq + 8 * VSTR);
        frag vfh, vfl;
#pragma unroll
        for (int j = 0; j < 4; ++j) { vfh[j] = a0[j]; vfh[4 + j] = a1[j]; vfl[j] = b0[j]; vfl[4 + j] = b1[j]; }
        O[d] = Mfma32<T>::mma(vfh, pf, O[d]);
        O[d] = Mfma32<T>::mma(vfh, pfl, O[d]);
        O[d] = Mfma32<T>::mma(vfl, pf, O[d]);
      }
    }
  }

  // ---- epilogue: O^T rows = head-dim index, lane = query; a lane owns 4 consecutive d per register quad ----
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
  if (qi < p.Nq) {
    float* orow = p.out + ((long)b * p.Nq + qi) * (p.H * HD) + h * HD;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
        *reinterpret_cast<float4*>(orow + 32 * d + 8 * rr + 4 * hi) =
            make_float4(O[d][4 * rr] * inv, O[d][4 * rr + 1] * inv, O[d][4 * rr + 2] * inv, O[d][4 * rr + 3] * inv);
  }
}

}  // namespace hipie

static int attn_split_launch(const float* q, const float* k, const float* v, const unsigned char* key_mask, const unsigned char* query_mask,
                             float* out, int B, int H, int Nq,
                                int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st,
                                float scale, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(q && k && v && out, "attn_split: null pointer");
  HIPIE_REQUIRE(B > 0 && H > 0 && Nq > 0 && Nk > 0, "attn_split: bad shape B=%d H=%d Nq=%d Nk=%d", B, H, Nq, Nk);
  HIPIE_REQUIRE(head_dim == 32 || head_dim == 64, "attn_split: head_dim 32 / 64 only (got %d)", head_dim);
  HIPIE_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) & 15) == 0 && ((q_sb | q_st | k_sb | k_st | v_sb | v_st) & 3) == 0,
                "attn_split: pointers must be 16-byte aligned and strides multiples of 4 elements");
  ASParams p;
  p.q = q; p.k = k; p.v = v; p.key_mask = key_mask; p.query_mask = query_mask; p.out = out;
  p.B = B; p.H = H; p.Nq = Nq; p.Nk = Nk;
  p.q_sb = q_sb; p.q_st = q_st; p.k_sb = k_sb; p.k_st = k_st; p.v_sb = v_sb; p.v_st = v_st;
  p.scale_log2e = scale * 1.4426950408889634f;
  const unsigned grid = (unsigned)(B * H * ((Nq + 127) / 128));
  if (head_dim == 32) hipLaunchKernelGGL(attn_split_kernel<32>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(attn_split_kernel<64>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("attn_split");
}

extern "C" int hipie_attn_split(const float* q, const float* k, const float* v, const unsigned char* key_mask, float* out, int B, int H, int Nq,
                                int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb, int64_t v_st,
                                float scale, void* stream) {
  return attn_split_launch(q, k, v, key_mask, nullptr, out, B, H, Nq, Nk, head_dim, q_sb, q_st, k_sb, k_st, v_sb, v_st, scale, stream);
}

extern "C" int hipie_attn_split_rows(const float* q, const float* k, const float* v, const unsigned char* query_mask, float* out, int B, int H,
                                     int Nq, int Nk, int head_dim, int64_t q_sb, int64_t q_st, int64_t k_sb, int64_t k_st, int64_t v_sb,
                                     int64_t v_st, float scale, void* stream) {
  return attn_split_launch(q, k, v, nullptr, query_mask, out, B, H, Nq, Nk, head_dim, q_sb, q_st, k_sb, k_st, v_sb, v_st, scale, stream);
}

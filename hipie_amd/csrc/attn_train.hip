// attn_train.hip -- fused attention forward + backward for the TRAINING step's global ViT blocks, split-fp16 operands.
//
// What it replaces: Attention.forward between the qkv and proj Linears, hipie/backbone/vit.py:69-80, with add_decomposed_rel_pos
// (hipie/backbone/utils.py:96-125) folded into the operands by the caller (hipie_amd/training/net.vit_attention):
//     q' = [scale q, rel_h(q, :), rel_w(q, :), 0..]   k' = [k, onehot(key row), onehot(key column), 0..]      (224 columns)
//     O = softmax(q' k'^T) v          and its gradients dq' (whose columns 80.. are d rel_h / d rel_w), dk (the first 80 columns of dk'), dv
// so the kernels are a PLAIN attention with d_qk = 224 and d_v = 80 -- no (heads, N, N) tensor in HBM, forward or backward (the materialised
// formulation is 12.3 ms per ViT-H block at two 1024^2 images and keeps 2 GB per block alive for the backward).
//
// Arithmetic: every operand is an fp16 pair (hi = fp16(x), lo = fp16(x - hi)) and every product is three v_mfma_f32_16x16x32_f16 (hi hi,
// hi lo, lo hi) accumulated in fp32 -- the library's split form (hipie_gemm, vit_attn_split.hip), 2.7 x the rate of the fp32 matrix pipe
// the library GEMMs of the materialised formulation run on.  P and dS are split again in registers.  The caller scales dO by a power of two
// into fp16's normal range and scales the gradients back.
//
// Lane layout of v_mfma_f32_16x16x32_f16 (c = lane & 15, g = lane >> 4): A: row c, 8 k-values of k-group g; B: column c, the same 8
// k-values; C/D: rows 4 g + i (i = 0..3), column c.  The contraction index is a dummy: A and B only have to agree on which value sits in
// slot (g, j).  Two C tiles of the logits therefore ARE one operand of the next product over keys / queries (slots j = 0..3 from tile t,
// 4..7 from tile t + 1: keys 16 t + 4 g + j), and the other operand reads the same 2 x 4 ROWS of a row-major LDS tile with ds_read_b64_tr_b16.
//
//   forward  (grid N/128 x BH, 8 waves x 16 queries):  S^T = K' Q'^T per 32-key tile, online softmax per query column, O^T += V^T P^T
//   backward 1 (grid N/128 x BH, 8 waves x 16 keys):   per 32-query tile  S = Q' K'^T, P = exp(S - lse), dP = dO V^T, dS = P (dP - delta),
//                                                      dV += P^T dO, dK += dS^T Q'[:, :80]
//   backward 2 (grid N/128 x BH, 8 waves x 16 queries): per 32-key tile    S^T, P^T, dP^T = V dO^T, dS^T, dQ'^T += K'^T dS^T
// All three are software-pipelined: the next tile travels global -> registers while the current one is computed on, then into the other LDS
// buffer; one barrier per tile.
#include "common.h"
#include "mfma.h"

namespace hipie {

constexpr int AT_DQ = 224;          // columns of q' / k' (7 MFMA k-steps of 32)
constexpr int AT_DV = 80;           // columns of v / O
constexpr int AT_DVP = 96;          // v / dO columns as a contraction (3 k-steps), zero padded by the caller
constexpr int AT_KS = AT_DQ + 8;    // row stride (halfs) of a row-major q' / k' tile in LDS
constexpr int AT_VS = AT_DV + 8;    // row stride (halfs) of the forward's row-major v tile in LDS
constexpr int AT_DS = AT_DVP + 8;   // row stride (halfs) of a row-major v / dO tile in LDS

constexpr float kAtShift = 8.317766166719343f;     // 12 ln 2: probabilities enter the P . v / P^T . dO products as 2^12 p (p <= 1 leaves fp16's
                                                    // normal range at 6e-5, and a row of 4096 keys has p ~ 2e-4: the lo half would be a subnormal)
#ifndef AT_WAVES
#define AT_WAVES 8                  // waves per workgroup: 16 queries (or keys) each
#endif
#ifndef AT_TR
#define AT_TR 32                    // rows of the tile that streams through LDS per step (64: 8.3 ms per block instead of 7.5)
#endif
constexpr int AT_THREADS = 64 * AT_WAVES, AT_WG_ROWS = 16 * AT_WAVES, AT_NT = AT_TR / 16;

constexpr float kAtDsScale = 16.f;                 // dS = P (dP - delta) is split as 16 dS: with max |dO| in [8, 16) a row of 4096 keys has
                                                    // |dS| ~ 5e-3, whose lo half would be a subnormal; |dS| <= P (1 - P) range(dP) keeps 16 dS in range
typedef f16x8 at_frag;

__device__ __forceinline__ f32x4 at_mma3(at_frag ah, at_frag al, at_frag bh, at_frag bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
}

// 2 x 4 values (two C tiles) -> one hi / lo operand
__device__ __forceinline__ void at_split8(const f32x4& a, const f32x4& b, at_frag& h, at_frag& l) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f16_t hh, ll;
    hl_split(a[i], hh, ll);
    h[i] = hh; l[i] = ll;
    hl_split(b[i], hh, ll);
    h[4 + i] = hh; l[4 + i] = ll;
  }
}

// The operand of a product that contracts over the ROWS of a row-major LDS tile (keys or queries): lane (c, g) needs tile[r + j][col + c] for
// the rows r = 16 tp + 4 g + j and 16 (tp + 1) + 4 g + j, j = 0..3 -- two ds_read_b64_tr_b16 (mfma.h: lane c of a 16-lane group points at
// &tile[r0 + c / 4][c0 + 4 (c % 4)] and receives tile[r0 + j][c0 + c]), no transposed copy of the tile.
__device__ __forceinline__ at_frag at_rows8(const f16_t* tile, int ls, int tp, int col, int c, int g) {
  const f16_t* p0 = tile + (16 * tp + 4 * g + (c >> 2)) * ls + col + 4 * (c & 3);
  const f16x4 a = Mfma32<f16_t>::tr_read(p0), b = Mfma32<f16_t>::tr_read(p0 + 16 * ls);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// workgroup -> (head bh, 128-row block): with BH a multiple of 8, the workgroups of ONE head run on ONE XCD (the dispatcher places workgroup
// id on XCD id % 8): the 32 workgroups of a 4096-token head stream the same 5 MB of k' / v (or q' / dO) tiles, which then live in that
// XCD's 4 MB L2 instead of eight heads' tiles competing for it.  A placement hint only.
__device__ __forceinline__ void at_block(int BH, int per_head, long& bh, int& blk) {
  const int id = blockIdx.x;
  if ((BH & 7) == 0) {
    const int w = id >> 3;
    bh = (long)(w / per_head) * 8 + (id & 7);
    blk = w % per_head;
  } else {
    bh = id / per_head;
    blk = id % per_head;
  }
}

// Staging of a streamed tile (AT_TR rows x COLS halfs of a row-major global matrix -> row-major LDS tile with a padded row stride) in two halves,
// for the software pipeline of all three kernels: global -> registers (in flight while the current tile is
// computed on), registers -> the OTHER LDS buffer, one barrier per tile.  (Keeping each thread's LDS offsets in registers instead of
// recomputing idx / chunks-per-row every tile removes 130 VALU instructions per tile and is 15 % SLOWER -- measured on one box; not kept.)
template <int COLS> struct AtStage {
  static constexpr int kChunks = AT_TR * (COLS / 8), kPer = (kChunks + AT_THREADS - 1) / AT_THREADS;
  at_frag r[kPer];
  // Branch-free on purpose: threads beyond the tile's last chunk re-load and re-store that chunk (same value, same address).  With the
  // loads under `if (idx < kChunks)` the compiler put an s_waitcnt vmcnt(0) between them -- six serialised L2 round trips at the top of
  // every tile instead of six loads in flight behind the compute.
  __device__ __forceinline__ void fetch(const f16_t* src, long ld, int tid) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int idx = min(tid + k * AT_THREADS, kChunks - 1);
      r[k] = *reinterpret_cast<const at_frag*>(src + (idx / (COLS / 8)) * ld + (idx % (COLS / 8)) * 8);
    }
  }
  __device__ __forceinline__ void commit(f16_t* dst, int ls, int tid) const {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int idx = min(tid + k * AT_THREADS, kChunks - 1);
      *reinterpret_cast<at_frag*>(dst + (idx / (COLS / 8)) * ls + (idx % (COLS / 8)) * 8) = r[k];
    }
  }
};

__global__ __launch_bounds__(AT_THREADS) void attn_train_fwd_kernel(const f16_t* __restrict__ Qh, const f16_t* __restrict__ Ql,
                                                             const f16_t* __restrict__ Kh, const f16_t* __restrict__ Kl,
                                                             const f16_t* __restrict__ Vh, const f16_t* __restrict__ Vl,
                                                             float* __restrict__ O, float* __restrict__ LSE, int N, int BH) {
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  constexpr int kBuf = 2 * AT_TR * AT_KS + 2 * AT_TR * AT_VS;                   // halfs per buffer: k' pair, v pair
  f16_t* sbase = reinterpret_cast<f16_t*>(at_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  long bh;
  int blk;
  at_block(BH, N / AT_WG_ROWS, bh, blk);
  const int q0 = blk * AT_WG_ROWS + wave * 16;
  at_frag qh[7], ql[7];
  {
    const long row = (bh * N + q0 + c) * AT_DQ + 8 * g;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      qh[s] = *reinterpret_cast<const at_frag*>(Qh + row + 32 * s);
      ql[s] = *reinterpret_cast<const at_frag*>(Ql + row + 32 * s);
    }
  }
  f32x4 o[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) o[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  AtStage<AT_DQ> fkh, fkl;
  AtStage<AT_DV> fvh, fvl;
  auto fetch = [&](int kb) {
    fkh.fetch(Kh + (bh * N + kb) * AT_DQ, AT_DQ, tid);
    fkl.fetch(Kl + (bh * N + kb) * AT_DQ, AT_DQ, tid);
    fvh.fetch(Vh + (bh * N + kb) * AT_DV, AT_DV, tid);
    fvl.fetch(Vl + (bh * N + kb) * AT_DV, AT_DV, tid);
  };
  auto commit = [&](int buf) {
    f16_t* b = sbase + buf * kBuf;
    fkh.commit(b, AT_KS, tid);
    fkl.commit(b + AT_TR * AT_KS, AT_KS, tid);
    fvh.commit(b + 2 * AT_TR * AT_KS, AT_VS, tid);
    fvl.commit(b + 2 * AT_TR * AT_KS + AT_TR * AT_VS, AT_VS, tid);
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int kb = 0, it = 0; kb < N; kb += AT_TR, ++it) {
    const bool more = kb + AT_TR < N;
    if (more) fetch(kb + AT_TR);
    const f16_t* sKh = sbase + (it & 1) * kBuf;
    const f16_t* sKl = sKh + AT_TR * AT_KS;
    const f16_t* sVh = sKl + AT_TR * AT_KS;
    const f16_t* sVl = sVh + AT_TR * AT_VS;
    f32x4 acc[AT_NT];
#pragma unroll
    for (int t = 0; t < AT_NT; ++t) {
      acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int off = (16 * t + c) * AT_KS + 8 * g;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const at_frag kh = *reinterpret_cast<const at_frag*>(sKh + off + 32 * s), kl = *reinterpret_cast<const at_frag*>(sKl + off + 32 * s);
        acc[t] = at_mma3(kh, kl, qh[s], ql[s], acc[t]);                   // S^T: rows = keys 16 t + 4 g + i, column = query c
      }
    }
    float mx = acc[0][0];
#pragma unroll
    for (int t = 0; t < AT_NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[t][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = __expf(m - mn);
    m = mn;
    lsum *= alpha;
#pragma unroll
    for (int j = 0; j < 5; ++j) o[j] *= alpha;
#pragma unroll
    for (int t = 0; t < AT_NT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[t][i] = __expf(acc[t][i] - mn + kAtShift);                       // 2^12 p: its fp16 pair is exact to 2^-22 down to p = 2^-16
        lsum += acc[t][i];
      }
#pragma unroll
    for (int tp = 0; tp < AT_NT; tp += 2) {
      at_frag ph, pl;
      at_split8(acc[tp], acc[tp + 1], ph, pl);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const at_frag vh = at_rows8(sVh, AT_VS, tp, 16 * j, c, g), vl = at_rows8(sVl, AT_VS, tp, 16 * j, c, g);
        o[j] = at_mma3(vh, vl, ph, pl, o[j]);                             // O^T: rows = d 16 j + 4 g + i, column = query c
      }
    }
    if (more) commit((it & 1) ^ 1);
    __syncthreads();
  }
  float ltot = lsum;
  ltot += __shfl_xor(ltot, 16);
  ltot += __shfl_xor(ltot, 32);
  const float inv = 1.f / ltot;
  float* orow = O + (bh * N + q0 + c) * AT_DV + 4 * g;
#pragma unroll
  for (int j = 0; j < 5; ++j) *reinterpret_cast<float4*>(orow + 16 * j) = make_float4(o[j][0] * inv, o[j][1] * inv, o[j][2] * inv, o[j][3] * inv);
  if (g == 0) LSE[bh * N + q0 + c] = m + logf(ltot) - kAtShift;
}

// ---- backward 1: dK (the first 80 columns of dK') and dV; a wave owns 16 keys, the workgroup 128, the queries stream through LDS ----------
__global__ __launch_bounds__(AT_THREADS) void attn_train_bwd_kv_kernel(const f16_t* __restrict__ Qh, const f16_t* __restrict__ Ql,
                                                                const f16_t* __restrict__ Kh, const f16_t* __restrict__ Kl,
                                                                const f16_t* __restrict__ Vh, const f16_t* __restrict__ Vl,
                                                                const f16_t* __restrict__ Dh, const f16_t* __restrict__ Dl,
                                                                const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                                float* __restrict__ dK, float* __restrict__ dV, int N, int BH) {
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  constexpr int kBuf = 2 * AT_TR * AT_KS + 2 * AT_TR * AT_DS + 4 * AT_TR;       // halfs per buffer: q' pair, dO pair, lse + delta (floats)
  f16_t* sbase = reinterpret_cast<f16_t*>(at_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  long bh;
  int blk;
  at_block(BH, N / AT_WG_ROWS, bh, blk);
  const int k0 = blk * AT_WG_ROWS + wave * 16;
  at_frag kh[7], kl[7], vh[3], vl[3];
  {
    const long row = (bh * N + k0 + c) * AT_DQ + 8 * g;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      kh[s] = *reinterpret_cast<const at_frag*>(Kh + row + 32 * s);
      kl[s] = *reinterpret_cast<const at_frag*>(Kl + row + 32 * s);
    }
    const long vrow = (bh * N + k0 + c) * AT_DVP + 8 * g;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      vh[s] = *reinterpret_cast<const at_frag*>(Vh + vrow + 32 * s);
      vl[s] = *reinterpret_cast<const at_frag*>(Vl + vrow + 32 * s);
    }
  }
  f32x4 dv[5], dk[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) dv[j] = dk[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  AtStage<AT_DQ> fqh, fql;
  AtStage<AT_DVP> fdh, fdl;
  float fstat = 0.f;
  auto fetch = [&](int qb) {
    const long qrow = (bh * N + qb) * AT_DQ, drow = (bh * N + qb) * AT_DVP;
    fqh.fetch(Qh + qrow, AT_DQ, tid);
    fql.fetch(Ql + qrow, AT_DQ, tid);
    fdh.fetch(Dh + drow, AT_DVP, tid);
    fdl.fetch(Dl + drow, AT_DVP, tid);
    {
      const int i = min(tid, 2 * AT_TR - 1);                                // branch-free: threads beyond 2 AT_TR repeat the last entry
      fstat = (i < AT_TR ? LSE : DELTA)[bh * N + qb + (i < AT_TR ? i : i - AT_TR)];
    }
  };
  auto commit = [&](int buf) {
    f16_t* b = sbase + buf * kBuf;
    fqh.commit(b, AT_KS, tid);
    fql.commit(b + AT_TR * AT_KS, AT_KS, tid);
    fdh.commit(b + 2 * AT_TR * AT_KS, AT_DS, tid);
    fdl.commit(b + 2 * AT_TR * AT_KS + AT_TR * AT_DS, AT_DS, tid);
    reinterpret_cast<float*>(b + 2 * AT_TR * AT_KS + 2 * AT_TR * AT_DS)[min(tid, 2 * AT_TR - 1)] = fstat;
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int qb = 0, it = 0; qb < N; qb += AT_TR, ++it) {
    const bool more = qb + AT_TR < N;
    if (more) fetch(qb + AT_TR);                                          // in flight while this tile is computed on
    const f16_t* sQh = sbase + (it & 1) * kBuf;
    const f16_t* sQl = sQh + AT_TR * AT_KS;
    const f16_t* sDh = sQl + AT_TR * AT_KS;
    const f16_t* sDl = sDh + AT_TR * AT_DS;
    const float* sLse = reinterpret_cast<const float*>(sDl + AT_TR * AT_DS);
    const float* sDel = sLse + AT_TR;
    f32x4 p[AT_NT], ds[AT_NT];
#pragma unroll
    for (int t = 0; t < AT_NT; ++t) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      const int off = (16 * t + c) * AT_KS + 8 * g;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const at_frag ah = *reinterpret_cast<const at_frag*>(sQh + off + 32 * s), al = *reinterpret_cast<const at_frag*>(sQl + off + 32 * s);
        acc = at_mma3(ah, al, kh[s], kl[s], acc);                         // S: rows = queries 16 t + 4 g + i, column = key c
      }
      const int doff = (16 * t + c) * AT_DS + 8 * g;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const at_frag ah = *reinterpret_cast<const at_frag*>(sDh + doff + 32 * s), al = *reinterpret_cast<const at_frag*>(sDl + doff + 32 * s);
        dp = at_mma3(ah, al, vh[s], vl[s], dp);                           // dP = dO v^T, same layout
      }
      const float4 l4 = *reinterpret_cast<const float4*>(sLse + 16 * t + 4 * g), d4 = *reinterpret_cast<const float4*>(sDel + 16 * t + 4 * g);
      const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        p[t][i] = __expf(acc[i] - lq[i]);
        ds[t][i] = p[t][i] * (dp[i] - dq_[i]) * kAtDsScale;
      }
    }
#pragma unroll
    for (int tp = 0; tp < AT_NT; tp += 2) {
      at_frag ph, pl, sh, sl;
      at_split8(p[tp] * 4096.f, p[tp + 1] * 4096.f, ph, pl);
      at_split8(ds[tp], ds[tp + 1], sh, sl);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        dv[j] = at_mma3(ph, pl, at_rows8(sDh, AT_DS, tp, 16 * j, c, g), at_rows8(sDl, AT_DS, tp, 16 * j, c, g), dv[j]);   // dV: rows = keys 4 g + i, column = d 16 j + c
        dk[j] = at_mma3(sh, sl, at_rows8(sQh, AT_KS, tp, 16 * j, c, g), at_rows8(sQl, AT_KS, tp, 16 * j, c, g), dk[j]);   // dK: rows = keys, column = dim 16 j + c
      }
    }
    if (more) commit((it & 1) ^ 1);                                       // the other buffer: everyone left it at the previous barrier
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long at = (bh * N + k0 + 4 * g + i) * AT_DV + 16 * j + c;
      dV[at] = dv[j][i] * (1.f / 4096.f);
      dK[at] = dk[j][i] * (1.f / kAtDsScale);
    }
}

// ---- backward 2: dQ' (all 224 columns); a wave owns 16 queries, the keys stream through LDS ---------------------------------------------
__global__ __launch_bounds__(AT_THREADS) void attn_train_bwd_q_kernel(const f16_t* __restrict__ Qh, const f16_t* __restrict__ Ql,
                                                               const f16_t* __restrict__ Kh, const f16_t* __restrict__ Kl,
                                                               const f16_t* __restrict__ Vh, const f16_t* __restrict__ Vl,
                                                               const f16_t* __restrict__ Dh, const f16_t* __restrict__ Dl,
                                                               const float* __restrict__ LSE, const float* __restrict__ DELTA,
                                                               float* __restrict__ dQ, int N, int BH) {
  extern __shared__ __attribute__((aligned(16))) char at_smem[];
  constexpr int kBuf = 2 * AT_TR * AT_KS + 2 * AT_TR * AT_DS;                   // halfs per buffer: k' pair, v pair
  f16_t* sbase = reinterpret_cast<f16_t*>(at_smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
  long bh;
  int blk;
  at_block(BH, N / AT_WG_ROWS, bh, blk);
  const int q0 = blk * AT_WG_ROWS + wave * 16;
  at_frag qh[7], ql[7], dh[3], dl[3];
  {
    const long row = (bh * N + q0 + c) * AT_DQ + 8 * g;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      qh[s] = *reinterpret_cast<const at_frag*>(Qh + row + 32 * s);
      ql[s] = *reinterpret_cast<const at_frag*>(Ql + row + 32 * s);
    }
    const long drow = (bh * N + q0 + c) * AT_DVP + 8 * g;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      dh[s] = *reinterpret_cast<const at_frag*>(Dh + drow + 32 * s);
      dl[s] = *reinterpret_cast<const at_frag*>(Dl + drow + 32 * s);
    }
  }
  const float lse = LSE[bh * N + q0 + c], delta = DELTA[bh * N + q0 + c];
  f32x4 dq[14];
#pragma unroll
  for (int j = 0; j < 14; ++j) dq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  AtStage<AT_DQ> fkh, fkl;
  AtStage<AT_DVP> fvh, fvl;
  auto fetch = [&](int kb) {
    const long krow = (bh * N + kb) * AT_DQ, vrow = (bh * N + kb) * AT_DVP;
    fkh.fetch(Kh + krow, AT_DQ, tid);
    fkl.fetch(Kl + krow, AT_DQ, tid);
    fvh.fetch(Vh + vrow, AT_DVP, tid);
    fvl.fetch(Vl + vrow, AT_DVP, tid);
  };
  auto commit = [&](int buf) {
    f16_t* b = sbase + buf * kBuf;
    fkh.commit(b, AT_KS, tid);
    fkl.commit(b + AT_TR * AT_KS, AT_KS, tid);
    fvh.commit(b + 2 * AT_TR * AT_KS, AT_DS, tid);
    fvl.commit(b + 2 * AT_TR * AT_KS + AT_TR * AT_DS, AT_DS, tid);
  };
  fetch(0);
  commit(0);
  __syncthreads();
  for (int kb = 0, it = 0; kb < N; kb += AT_TR, ++it) {
    const bool more = kb + AT_TR < N;
    if (more) fetch(kb + AT_TR);
    const f16_t* sKh = sbase + (it & 1) * kBuf;
    const f16_t* sKl = sKh + AT_TR * AT_KS;
    const f16_t* sVh = sKl + AT_TR * AT_KS;
    const f16_t* sVl = sVh + AT_TR * AT_DS;
    f32x4 ds[AT_NT];
#pragma unroll
    for (int t = 0; t < AT_NT; ++t) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      const int off = (16 * t + c) * AT_KS + 8 * g;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const at_frag ah = *reinterpret_cast<const at_frag*>(sKh + off + 32 * s), al = *reinterpret_cast<const at_frag*>(sKl + off + 32 * s);
        acc = at_mma3(ah, al, qh[s], ql[s], acc);                         // S^T: rows = keys 16 t + 4 g + i, column = query c
      }
      const int voff = (16 * t + c) * AT_DS + 8 * g;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const at_frag ah = *reinterpret_cast<const at_frag*>(sVh + voff + 32 * s), al = *reinterpret_cast<const at_frag*>(sVl + voff + 32 * s);
        dp = at_mma3(ah, al, dh[s], dl[s], dp);                           // dP^T = v dO^T, same layout
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) ds[t][i] = __expf(acc[i] - lse) * (dp[i] - delta) * kAtDsScale;
    }
#pragma unroll
    for (int tp = 0; tp < AT_NT; tp += 2) {
      at_frag sh, sl;
      at_split8(ds[tp], ds[tp + 1], sh, sl);
#pragma unroll
      for (int j = 0; j < 14; ++j)
        dq[j] = at_mma3(at_rows8(sKh, AT_KS, tp, 16 * j, c, g), at_rows8(sKl, AT_KS, tp, 16 * j, c, g), sh, sl, dq[j]);   // dQ'^T: rows = dims 16 j + 4 g + i, column = query c
    }
    if (more) commit((it & 1) ^ 1);
    __syncthreads();
  }
  float* out = dQ + (bh * N + q0 + c) * AT_DQ + 4 * g;
#pragma unroll
  for (int j = 0; j < 14; ++j)
    *reinterpret_cast<float4*>(out + 16 * j) = make_float4(dq[j][0] * (1.f / kAtDsScale), dq[j][1] * (1.f / kAtDsScale), dq[j][2] * (1.f / kAtDsScale),
                                                           dq[j][3] * (1.f / kAtDsScale));
}

// fp32 rows -> the two fp16 planes of the operands above, zero-padded to Cp columns and optionally scaled by a DEVICE scalar (dO's power of
// two): one pass instead of pad + clamp + two casts + a subtraction
__global__ __launch_bounds__(256) void to_f16_pair_kernel(const float* __restrict__ x, long ldx, f16_t* __restrict__ hi, f16_t* __restrict__ lo,
                                                          long rows, int C, int Cp, const float* __restrict__ scale) {
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  const int cpr = Cp / 8;
  if (gid >= rows * cpr) return;
  const long r = gid / cpr;
  const int c0 = (int)(gid - r * cpr) * 8;
  const float sc = scale ? *scale : 1.f;
  const float* src = x + r * ldx + c0;
  at_frag h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    f16_t hh, ll;
    hl_split(c0 + e < C ? src[e] * sc : 0.f, hh, ll);
    h[e] = hh;
    l[e] = ll;
  }
  *reinterpret_cast<at_frag*>(hi + r * Cp + c0) = h;
  *reinterpret_cast<at_frag*>(lo + r * Cp + c0) = l;
}

constexpr size_t kAtBwdKvLds = 2 * ((size_t)(2 * AT_TR * AT_KS + 2 * AT_TR * AT_DS) * sizeof(f16_t) + 2 * AT_TR * sizeof(float));   // two buffers
constexpr size_t kAtBwdQLds = 2 * (size_t)(2 * AT_TR * AT_KS + 2 * AT_TR * AT_DS) * sizeof(f16_t);
constexpr size_t kAtFwdLds = 2 * (size_t)(2 * AT_TR * AT_KS + 2 * AT_TR * AT_VS) * sizeof(f16_t);                                          // two buffers

}  // namespace hipie

using namespace hipie;

extern "C" int hipie_attn_train_forward(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* v_hi, const void* v_lo,
                                        void* out, void* lse, int BH, int N, void* stream) {
  HIPIE_REQUIRE(q_hi && q_lo && k_hi && k_lo && v_hi && v_lo && out && lse, "attn_train_forward: null pointer");
  HIPIE_REQUIRE(BH > 0 && N > 0 && N % 128 == 0 && (long)BH * (N / 128) < (1L << 31), "attn_train_forward: BH=%d N=%d (N a multiple of 128)", BH, N);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_train_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAtFwdLds);
    attr = true;
  }
  hipLaunchKernelGGL(attn_train_fwd_kernel, dim3((unsigned)((N / AT_WG_ROWS) * BH)), dim3(AT_THREADS), kAtFwdLds, (hipStream_t)stream, (const f16_t*)q_hi,
                     (const f16_t*)q_lo, (const f16_t*)k_hi, (const f16_t*)k_lo, (const f16_t*)v_hi, (const f16_t*)v_lo, (float*)out, (float*)lse, N, BH);
  return check_launch("attn_train_forward");
}

extern "C" int hipie_attn_train_backward(const void* q_hi, const void* q_lo, const void* k_hi, const void* k_lo, const void* v_hi, const void* v_lo,
                                         const void* do_hi, const void* do_lo, const void* lse, const void* delta, void* dq, void* dk, void* dv,
                                         int BH, int N, void* stream) {
  HIPIE_REQUIRE(q_hi && q_lo && k_hi && k_lo && v_hi && v_lo && do_hi && do_lo && lse && delta && dq && dk && dv, "attn_train_backward: null pointer");
  HIPIE_REQUIRE(BH > 0 && N > 0 && N % 128 == 0 && (long)BH * (N / 128) < (1L << 31), "attn_train_backward: BH=%d N=%d (N a multiple of 128)", BH, N);
  static_assert(kAtBwdKvLds <= 160 * 1024 && kAtBwdQLds <= 160 * 1024, "LDS budget");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_train_bwd_kv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAtBwdKvLds);
    (void)hipFuncSetAttribute((const void*)attn_train_bwd_q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kAtBwdQLds);
    attr = true;
  }
  const dim3 grid((unsigned)((N / AT_WG_ROWS) * BH)), block(AT_THREADS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(attn_train_bwd_kv_kernel, grid, block, kAtBwdKvLds, st, (const f16_t*)q_hi, (const f16_t*)q_lo, (const f16_t*)k_hi,
                     (const f16_t*)k_lo, (const f16_t*)v_hi, (const f16_t*)v_lo, (const f16_t*)do_hi, (const f16_t*)do_lo, (const float*)lse,
                     (const float*)delta, (float*)dk, (float*)dv, N, BH);
  hipLaunchKernelGGL(attn_train_bwd_q_kernel, grid, block, kAtBwdQLds, st, (const f16_t*)q_hi, (const f16_t*)q_lo, (const f16_t*)k_hi,
                     (const f16_t*)k_lo, (const f16_t*)v_hi, (const f16_t*)v_lo, (const f16_t*)do_hi, (const f16_t*)do_lo, (const float*)lse,
                     (const float*)delta, (float*)dq, N, BH);
  return check_launch("attn_train_backward");
}

extern "C" int hipie_to_f16_pair(const void* x, int64_t ldx, void* hi, void* lo, int64_t rows, int C, int Cp, const void* scale, void* stream) {
  HIPIE_REQUIRE(x && hi && lo && rows > 0 && C > 0 && Cp >= C && Cp % 8 == 0 && ldx >= C, "to_f16_pair: rows=%ld C=%d Cp=%d ldx=%ld", (long)rows, C, Cp, (long)ldx);
  HIPIE_REQUIRE(((uintptr_t)hi % 16) == 0 && ((uintptr_t)lo % 16) == 0, "to_f16_pair: planes must be 16-byte aligned");
  const long n = rows * (Cp / 8);
  hipLaunchKernelGGL(to_f16_pair_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (long)ldx, (f16_t*)hi,
                     (f16_t*)lo, (long)rows, C, Cp, (const float*)scale);
  return check_launch("to_f16_pair");
}

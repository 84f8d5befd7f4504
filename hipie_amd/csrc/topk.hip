// topk.hip -- row-wise top-k (k <= 1024) of fp32 scores: the two-stage query selections of the path
// (deformable_transformer_dino.py:222-230: 900 of Nv = 21760 encoder tokens; maskdino_decoder.py:413-426: 300) and the
// (query, class) instance selection of the post-processing (hipie_img.py:640-648: 100 of Q * C).
//
// Why a kernel: torch.topk's multi-block radix select is 12 launches per call on this stack AND is not hipGraph-replay safe
// (a captured graph that contains nothing but torch.topk((8, 21760), 900) ends in a GPU memory fault after ~10 replays -- the
// cause of the whole-forward replay fault of round 2, tools/graph_fault.py).  One workgroup per row: a 4-pass 8-bit radix select
// on order-preserving integer keys finds the k-th largest key T (the row is 87 KB: it is re-read from L2, not staged); elements
// above T are compacted in any order, elements equal to T are taken in INDEX order (ballot-ranked scan); the k winners are sorted
// by a bitonic network on (key, ~index) -- descending value, ascending index among equal values: a deterministic total order
// (torch.topk leaves the order of ties open).  NaN sorts as the largest value, as in torch.
#include "common.h"

namespace hipie {

// order-preserving key.  Canonical forms first: every NaN (either sign bit: 0 * -inf gives a negative one) is the largest key, as in
// torch; -0.0 is folded into +0.0 so that equal values tie and the tie is broken by index.
__device__ __forceinline__ unsigned int tk_key(float v) {
  if (v != v) return 0xFFFFFFFFu;
  v += 0.0f;
  const unsigned int b = __builtin_bit_cast(unsigned int, v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__global__ __launch_bounds__(1024) void topk_kernel(const float* __restrict__ x, long row_stride, int n, int k, long* __restrict__ idx_out,
                                                    float* __restrict__ val_out) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned int s_prefix, s_krem, s_ngt;
  __shared__ unsigned int wave_cnt[16];
  __shared__ unsigned long long sel[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = x + (long)blockIdx.x * row_stride;

  // ---- radix select of the k-th largest key ----
  unsigned int prefix = 0u, mask = 0u, krem = (unsigned int)k;
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
      const unsigned int kk = tk_key(row[i]);
      if ((kk & mask) == prefix) atomicAdd(&hist[(kk >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int cum = 0u;
      int d = 255;
      for (; d > 0; --d) {
        if (cum + hist[d] >= krem) break;
        cum += hist[d];
      }
      s_prefix = prefix | ((unsigned int)d << shift);
      s_krem = krem - cum;
    }
    __syncthreads();
    prefix = s_prefix;
    krem = s_krem;
    mask |= 0xFFu << shift;
  }
  const unsigned int T = prefix;                 // key of the k-th largest element; krem (>= 1) elements equal to T are taken
  const unsigned int G = (unsigned int)k - krem; // elements strictly above T

  // ---- compaction: > T in any order (sorted below), == T in index order ----
  if (tid == 0) s_ngt = 0u;
  sel[tid] = 0ull;
  __syncthreads();
  const int seg = ((n + 16 * 64 - 1) / (16 * 64)) * 64;          // indices per wave, a multiple of 64
  const int beg = wave * seg, end = min(n, beg + seg);
  unsigned int cnt = 0u;
  for (int i0 = beg; i0 < end; i0 += 64) {
    const int i = i0 + lane;
    const unsigned int kk = (i < end) ? tk_key(row[i]) : 0u;
    const bool gt = (i < end) && kk > T, eq = (i < end) && kk == T;
    if (gt) {
      const unsigned int slot = atomicAdd(&s_ngt, 1u);
      sel[slot] = ((unsigned long long)kk << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i);
    }
    cnt += (unsigned int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(eq));
  }
  if (lane == 0) wave_cnt[wave] = cnt;
  __syncthreads();
  unsigned int running = 0u;
  for (int w = 0; w < wave; ++w) running += wave_cnt[w];
  for (int i0 = beg; i0 < end && running < krem; i0 += 64) {
    const int i = i0 + lane;
    const bool eq = (i < end) && tk_key(row[i]) == T;
    const unsigned long long b = __builtin_amdgcn_ballot_w64(eq);
    if (eq) {
      const unsigned int rank = running + (unsigned int)__builtin_popcountll(b & ((1ull << lane) - 1ull));
      if (rank < krem) sel[G + rank] = ((unsigned long long)T << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)i);
    }
    running += (unsigned int)__builtin_popcountll(b);
  }
  __syncthreads();

  // ---- bitonic sort of the 1024 slots, descending (unused slots hold 0 = below every real composite) ----
  for (int size = 2; size <= 1024; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (partner > tid) {
        const unsigned long long a = sel[tid], b = sel[partner];
        const bool desc = (tid & size) == 0;
        if (desc ? (a < b) : (a > b)) { sel[tid] = b; sel[partner] = a; }
      }
      __syncthreads();
    }
  }
  if (tid < k) {
    const unsigned int i = 0xFFFFFFFFu - (unsigned int)(sel[tid] & 0xFFFFFFFFull);
    idx_out[(long)blockIdx.x * k + tid] = (long)i;
    if (val_out != nullptr) val_out[(long)blockIdx.x * k + tid] = row[i];
  }
}

}  // namespace hipie

extern "C" int hipie_topk(const float* x, int64_t row_stride, int rows, int n, int k, int64_t* idx_out, float* val_out, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(x && idx_out, "topk: null pointer");
  HIPIE_REQUIRE(rows > 0 && n > 0 && k > 0 && k <= 1024 && k <= n && row_stride >= n, "topk: rows=%d n=%d k=%d (k <= min(n, 1024))", rows, n, k);
  hipLaunchKernelGGL(topk_kernel, dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, x, (long)row_stride, n, k, (long*)idx_out, val_out);
  return check_launch("topk");
}

// gemm_k256.hip -- thin-K form of the split GEMM: out[M, N] = X[M, 256] . W[N, 256]^T + bias, fp32 out, for the K = 256 linears over ALL
// pyramid tokens (value / offset / output projections of the deformable encoder layers, deformable_transformer_dino.py:378-394 and
// ops/modules/ms_deform_attn.py:93-114: M = 174080 at bs 8, N = 256 or 384).
//
// Why a second kernel: with K = 256 the 256 x 256 tile of gemm_kernel is eight k-steps between a 256 KB load and a 256 KB store, one
// workgroup per CU, and the three phases of a round do not overlap (0.109 ms per N = 256 launch where the k loop alone is ~0.05 and the
// HBM time 0.065).  Here the structure of ffn_fused.hip's first product: a wave keeps the X fragments of its 32 tokens in registers for
// its whole life (K = 256 as fp16 pairs: 128 registers, read from HBM exactly once -- no barrier on that side), and walks the N / 32
// chunks of 32 output features:  y^T = W[chunk] . X^T  = 16 k-steps x 3 products into two interleaved accumulators.  The weight chunk
// tiles (32 rows x 1 KB) go L2 -> LDS by LDS-DMA, double buffered, one barrier per chunk, the DMA instructions riding in the k-steps;
// the 16-byte units of a row are stored at c ^ (row & 15): conflict-free ds_read_b128.  The C tile is (feature rows x token columns): a
// lane owns 16 features of ITS token and writes them as four 16-byte pieces of its output row; the stores of chunk c are issued behind
// the barrier of chunk c + 1, so the wait for the DMA (vmcnt counts stores too) does not wait for stores that have just been issued.
// 4 waves = 128 tokens per workgroup, two workgroups per CU: one loads X while the other multiplies.
// X rows may be HL8 (as add_layernorm_dec writes them) or plain fp32 (the residual stream itself: split here, once per wave).
//
// How X gets into the registers (second version).  "lane = token" means a lane needs its WHOLE 1 KB row: read straight from global that is
// sixty-four 16-byte pieces per lane, 1 KB apart across lanes -- 32 cache lines per load instruction and a 32 KB L1 working set per
// wave; the first version did that and was no faster than the tile kernel (tools/bench_gemm_k256.py: 0.128 against 0.123 ms at N = 256).
// Now every wave brings ITS 32 rows in by LDS-DMA, two half rows (2 x 512 B, contiguous per instruction) at a time into a private 16 KB
// slice of the weight buffers, and reads the fragments back with conflict-free ds_read_b128 (units swizzled by row & 15 on the DMA's
// source side) -- coalesced 512-byte requests, no block barrier (a wave reads only what it fetched itself), twice for the two K halves.
#include <stdlib.h>

#include "common.h"
#include "mfma.h"

namespace hipie {

struct TKParams {
  const char* X; const char* W; const float* bias; float* out;
  long ldx_b, ldw_b, ldo;               // row strides: X / W in bytes, out in floats
  int M, N;
};

__device__ __forceinline__ void tk_dma16(const char* sbase, unsigned int voff, unsigned int lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}

constexpr int TK_K = 256, TK_C = 32;                         // K, output features per chunk
constexpr int TK_TILE = TK_C * TK_K * 4;                     // bytes of a weight chunk tile (32 rows x 1 KB of fp16 pairs)

template <bool XF32>
__global__ __launch_bounds__(256, 2) void gemm_k256_kernel(const TKParams p) {
  typedef f16_t T;
  typedef Mfma32<T>::frag frag;
  constexpr int KS = TK_K / 16;                 // 16 k-steps
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + li;
  const int mc = min(m, p.M - 1);
  const int nch = p.N / TK_C;

  // ---- DMA plan: instruction i = wave + 4 q fills LDS row i (1 KB) of the tile: lane l -> position l, logical unit l ^ (i & 15) ----
  unsigned int dv[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = wave + 4 * q;
    dv[q] = (unsigned int)((long)i * p.ldw_b + 16 * (lane ^ (i & 15)));
  }
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  auto dma_chunk = [&](const int c, const int buf, const int q) __attribute__((always_inline)) {
    const int i = wave + 4 * q;
    tk_dma16(p.W + (long)c * TK_C * p.ldw_b, dv[q], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(buf * TK_TILE + 1024 * i)));
  };

  // ---- X fragments (B operand): lane (token li, half hi) holds k group 2 ks + hi of its row, both halves.  HL8 rows: 16 bytes of
  //      hi parts + 16 bytes of lo parts per group of 8; fp32 rows: the same 32 bytes are the 8 values themselves.
  //      Per K half: DMA instruction j (0 .. 15) of this wave moves the half rows 2 j, 2 j + 1 of its 32 tokens (lane l: half row
  //      l >> 5, LDS unit l & 31 <- logical unit (l & 31) ^ (row & 15)) into its 16 KB slice; then 8 k-steps x 2 fragment reads ----
  frag xh[KS], xl[KS];
  {
    const int m0w = blockIdx.x * 128 + wave * 32;
    const unsigned int xslice = lds0 + (unsigned int)(wave * 16384);
    const char* xs = smem + wave * 16384 + li * 512;
    unsigned int xv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int r = 2 * j + (lane >> 5);
      xv[j] = (unsigned int)((long)min(m0w + r, p.M - 1) * p.ldx_b + 16 * ((lane & 31) ^ (r & 15)));
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): the fragment reads of the first half are done before it is overwritten
#pragma unroll
      for (int j = 0; j < 16; ++j) tk_dma16(p.X + 512 * half, xv[j], __builtin_amdgcn_readfirstlane(xslice + 1024 * j));
      __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): landed (only this wave reads this slice)
#pragma unroll
      for (int ksl = 0; ksl < KS / 2; ++ksl) {
        const int ks = half * (KS / 2) + ksl;
        const int u = 2 * (2 * ksl + hi);
        const frag a = *reinterpret_cast<const frag*>(xs + 16 * (u ^ (li & 15)));
        const frag b = *reinterpret_cast<const frag*>(xs + 16 * ((u + 1) ^ (li & 15)));
        if (XF32) {
          const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
          const float v[8] = {fa[0], fa[1], fa[2], fa[3], fb[0], fb[1], fb[2], fb[3]};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            T h, l;
            hl_split(v[e], h, l);
            xh[ks][e] = h;
            xl[ks][e] = l;
          }
        } else {
          xh[ks] = a;
          xl[ks] = b;
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
    __syncthreads();                              // every wave has its fragments: the slices become the weight buffers
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) dma_chunk(0, 0, q);

  const int wrow = li * 1024, wsw = li & 15;                   // tile row li, unit u at byte 16 * (u ^ wsw)
  float* orow = p.out + (long)mc * p.ldo + 4 * hi;
  float4 prev[4];                                             // the finished chunk, stored one barrier later
#pragma unroll
  for (int g = 0; g < 4; ++g) prev[g] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int c = 0; c < nch; ++c) {
    const int buf = c & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's pieces of chunk c have landed (and the stores of chunk c - 2)
    __syncthreads();                            // ... everybody's; all reads of chunk c - 1 (the other buffer) are done
    if (c > 0 && m < p.M) {
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(orow + 32 * (c - 1) + 8 * g) = prev[g];
    }
    const char* bs = smem + buf * TK_TILE;
    const bool more = c + 1 < nch;
    f32x16 h0, h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      const frag ah0 = *reinterpret_cast<const frag*>(bs + wrow + 16 * ((2 * (2 * ks + hi)) ^ wsw));
      const frag al0 = *reinterpret_cast<const frag*>(bs + wrow + 16 * ((2 * (2 * ks + hi) + 1) ^ wsw));
      const frag ah1 = *reinterpret_cast<const frag*>(bs + wrow + 16 * ((2 * (2 * ks + 2 + hi)) ^ wsw));
      const frag al1 = *reinterpret_cast<const frag*>(bs + wrow + 16 * ((2 * (2 * ks + 2 + hi) + 1) ^ wsw));
      h0 = Mfma32<T>::mma(al0, xh[ks], h0);
      h1 = Mfma32<T>::mma(al1, xh[ks + 1], h1);
      h0 = Mfma32<T>::mma(ah0, xl[ks], h0);
      h1 = Mfma32<T>::mma(ah1, xl[ks + 1], h1);
      h0 = Mfma32<T>::mma(ah0, xh[ks], h0);
      h1 = Mfma32<T>::mma(ah1, xh[ks + 1], h1);
      if (more) dma_chunk(c + 1, buf ^ 1, ks >> 1);
    }
    // registers 4 g .. 4 g + 3 of the C tile = features 32 c + 8 g + 4 hi + (0 .. 3) of this lane's token
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) b4 = *reinterpret_cast<const float4*>(p.bias + 32 * c + 8 * g + 4 * hi);
      prev[g] = make_float4(h0[4 * g] + h1[4 * g] + b4.x, h0[4 * g + 1] + h1[4 * g + 1] + b4.y, h0[4 * g + 2] + h1[4 * g + 2] + b4.z,
                            h0[4 * g + 3] + h1[4 * g + 3] + b4.w);
    }
  }
  if (m < p.M) {
#pragma unroll
    for (int g = 0; g < 4; ++g) *reinterpret_cast<float4*>(orow + 32 * (nch - 1) + 8 * g) = prev[g];
  }
}

// X rows: HL8 (x_f32 = 0) or plain fp32 (x_f32 = 1), W rows HL8; row strides of X / W in BYTES, of out in floats.  Called by gemm_impl
// (gemm.hip) for the shapes this form is built for; everything is validated there.
int launch_gemm_k256(const void* X, long ldx_b, int x_f32, const void* W, long ldw_b, const float* bias, float* out, long ldo, int M, int N,
                     hipStream_t st) {
  TKParams p;
  p.X = (const char*)X; p.W = (const char*)W; p.bias = bias; p.out = out;
  p.ldx_b = ldx_b; p.ldw_b = ldw_b; p.ldo = ldo;
  p.M = M; p.N = N;
  const size_t lds = (size_t)2 * TK_TILE;
  static bool lds_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)gemm_k256_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)gemm_k256_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = true;
  }
  const dim3 grid((unsigned)((M + 127) / 128));
  if (x_f32) hipLaunchKernelGGL(gemm_k256_kernel<true>, grid, dim3(256), lds, st, p);
  else hipLaunchKernelGGL(gemm_k256_kernel<false>, grid, dim3(256), lds, st, p);
  return check_launch("gemm_k256");
}

}  // namespace hipie

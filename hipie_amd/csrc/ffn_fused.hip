// ffn_fused.hip -- the FFN of the deformable encoder layers (linear1 256 -> 2048, ReLU, linear2 2048 -> 256; deformable_transformer_dino.py
// :378-394 and maskdino/pixel_decoder/maskdino_encoder.py:142-157, 12 layers on 174080 tokens per bs-8 step) as ONE kernel at the split
// policy's fp32-class accuracy: the 1.43 GB hidden tensor (HL8) that linear1 wrote and linear2 read back per layer never exists.
//
// Why it fits now and did not as a tile of the GEMM kernel: the hidden activations of 32 tokens x 32 hidden units are exactly one MFMA
// C tile (16 registers), and the C layout of  h^T = W1c . X^T  (lane = token, 16 rows) IS a valid B operand of the next product after
// ReLU + hi / lo split -- two k16 steps whose k order (rows 0-3, 8-11 | 4-7, 12-15 per lane half) is folded into a column permutation
// of W2 done once on the host.  So a wave keeps, in registers, the X fragments of its 32 tokens (K = 256 as fp16 pairs: 128 registers),
// the output accumulators (256 features x 32 tokens: 128 registers) and one hidden tile; it walks the 64 hidden chunks of 32 units:
//     h^T  = W1[chunk] . X^T           16 k-steps x 3 products  (two interleaved accumulators: no dependent back-to-back MFMAs)
//     H    = split(relu(h + b1))       fp16 pair, in place as the B operand
//     out^T += W2'[:, chunk] . H       8 feature blocks x 2 k-steps x 3 products
// 96 MFMAs per chunk and wave against 64 KB of weights per chunk and WORKGROUP (4 waves = 128 tokens): the weight tiles go L2 -> LDS by
// LDS-DMA, double buffered, one barrier per chunk; W1 rows (1 KB) are stored with 16-byte chunk c at position c ^ (row & 15), W2 rows
// (128 B) with c ^ ((row >> 1) & 7): conflict-free ds_read_b128 for every lane group.  One wave per SIMD (a 512-register kernel): measured
// on this chip, ONE wave per SIMD saturates the power-limited matrix pipe as long as it has a hardware barrier and deep prefetch.
#include "common.h"
#include "mfma.h"

namespace hipie {

struct FFParams {
  const char* X; const char* W1; const char* W2; const float* b1; const float* b2; float* out;
  long ldx_b, ldw1_b, ldw2_b, ldo;      // row strides: X / W1 / W2 in bytes, out in floats
  int M;
};

__device__ __forceinline__ void ff_dma16(const char* sbase, unsigned int voff, unsigned int lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}

constexpr int FF_D = 256, FF_F = 2048, FF_C = 32;            // model width, hidden width, hidden units per chunk
constexpr int FF_W1B = FF_C * FF_D * 4;                        // bytes of a W1 chunk tile (32 rows x 1 KB)
constexpr int FF_W2B = FF_D * FF_C * 4;                        // bytes of a W2 chunk tile (256 rows x 128 B)
constexpr int FF_BUF = FF_W1B + FF_W2B;

__global__ __launch_bounds__(256, 1) void ffn_fused_kernel(const FFParams p) {
  typedef f16_t T;
  typedef Mfma32<T>::frag frag;
  constexpr int KS = FF_D / 16;                 // 16 k-steps of the first product
  constexpr int NB = FF_D / 32;                 // 8 output feature blocks
  constexpr int NCH = FF_F / FF_C;              // 64 hidden chunks
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* sb1 = reinterpret_cast<float*>(smem + 2 * FF_BUF);       // all of b1 (8 KB)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + li;
  const int mc = min(m, p.M - 1);

  for (int i = tid; i < FF_F / 4; i += 256) reinterpret_cast<float4*>(sb1)[i] = reinterpret_cast<const float4*>(p.b1)[i];

  // ---- X fragments (B operand): lane (token li, half hi) holds k group 2 ks + hi of its row, both halves ----
  frag xh[KS], xl[KS];
  {
    const char* xr = p.X + (long)mc * p.ldx_b + 32 * hi;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      xh[ks] = *reinterpret_cast<const frag*>(xr + 64 * ks);
      xl[ks] = *reinterpret_cast<const frag*>(xr + 64 * ks + 16);
    }
  }

  // ---- DMA plan.  W1 tile: instruction i = wave + 4 q fills LDS row i (1 KB): lane l -> position l, logical chunk l ^ (i & 15).
  //      W2 tile: instruction j = wave + 4 q fills rows 8 j .. 8 j + 7 (128 B each): lane -> (row rl = l >> 3, position l & 7), logical
  //      chunk (l & 7) ^ ((row >> 1) & 7).  Per chunk c the W1 source rows move by 32 rows, the W2 source columns by 128 bytes. ----
  unsigned int dv1[8], dv2[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int i = wave + 4 * q;
    dv1[q] = (unsigned int)((long)i * p.ldw1_b + 16 * (lane ^ (i & 15)));
    const int f = 8 * i + (lane >> 3);
    dv2[q] = (unsigned int)((long)f * p.ldw2_b + 16 * ((lane & 7) ^ ((f >> 1) & 7)));
  }
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  auto dma_chunk = [&](const int c, const int buf, const int q) {
    const int i = wave + 4 * q;
    ff_dma16(p.W1 + (long)c * FF_C * p.ldw1_b, dv1[q], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(buf * FF_BUF + 1024 * i)));
    ff_dma16(p.W2 + (long)c * (FF_C * 4), dv2[q], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(buf * FF_BUF + FF_W1B + 1024 * i)));
  };

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

#pragma unroll
  for (int q = 0; q < 8; ++q) dma_chunk(0, 0, q);

  // fragment addresses inside a buffer (W1 tile, then W2 tile)
  const int w1row = li * 1024, w1sw = li & 15;                  // W1 tile: row li, chunk index c at byte 16 * (c ^ w1sw)
  const int w2row = FF_W1B + li * 128, w2sw = (li >> 1) & 7;    // W2 tile: row 32 j + li at + j * 4096, chunk c at byte 16 * (c ^ w2sw)

  // Schedule: plain -- first product, ReLU / split, second product, one barrier per chunk.  Two software-pipelined forms (the next chunk's
  // first product issued around the split; explicit fragment double buffers) measured 4-8 % SLOWER on this 450-register kernel: the
  // extra live accumulator tile turns into v_accvgpr traffic (tools/bench_ffn_fused.py: 1.075 ms this form, 1.118 / 1.160 ms those).
  for (int c = 0; c < NCH; ++c) {
    const int buf = c & 1;
    __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): this wave's pieces of chunk c have landed
    __syncthreads();                            // ... everybody's; all reads of chunk c - 1 (the other buffer) are done
    const char* bs = smem + buf * FF_BUF;
    const bool more = c + 1 < NCH;

    // ---- h^T = W1[chunk] . X^T: two accumulators (even / odd k-steps); the DMA instructions of chunk c + 1 ride in the k-steps ----
    f32x16 h0, h1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { h0[r] = 0.f; h1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ks += 2) {
      const frag ah0 = *reinterpret_cast<const frag*>(bs + w1row + 16 * ((2 * (2 * ks + hi)) ^ w1sw));
      const frag al0 = *reinterpret_cast<const frag*>(bs + w1row + 16 * ((2 * (2 * ks + hi) + 1) ^ w1sw));
      const frag ah1 = *reinterpret_cast<const frag*>(bs + w1row + 16 * ((2 * (2 * ks + 2 + hi)) ^ w1sw));
      const frag al1 = *reinterpret_cast<const frag*>(bs + w1row + 16 * ((2 * (2 * ks + 2 + hi) + 1) ^ w1sw));
      h0 = Mfma32<T>::mma(al0, xh[ks], h0);
      h1 = Mfma32<T>::mma(al1, xh[ks + 1], h1);
      h0 = Mfma32<T>::mma(ah0, xl[ks], h0);
      h1 = Mfma32<T>::mma(ah1, xl[ks + 1], h1);
      h0 = Mfma32<T>::mma(ah0, xh[ks], h0);
      h1 = Mfma32<T>::mma(ah1, xh[ks + 1], h1);
      if (more) dma_chunk(c + 1, buf ^ 1, ks >> 1);
    }
    // ---- H = split(relu(h + b1)): registers 8 s .. 8 s + 7 of the C tile are k-step s of the next product ----
    frag Hh[2], Hl[2];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *reinterpret_cast<const float4*>(sb1 + c * FF_C + 8 * g + 4 * hi);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        T hh, ll;
        hl_split(fmaxf(h0[r] + h1[r] + bb[e], 0.f), hh, ll);
        Hh[r >> 3][r & 7] = hh;
        Hl[r >> 3][r & 7] = ll;
      }
    }
    // ---- out^T += W2'[:, chunk] . H: product-major order, 8 independent accumulators between two uses of one ----
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      frag wh[NB], wl[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        wh[j] = *reinterpret_cast<const frag*>(bs + w2row + j * 4096 + 16 * ((2 * (2 * s2 + hi)) ^ w2sw));
        wl[j] = *reinterpret_cast<const frag*>(bs + w2row + j * 4096 + 16 * ((2 * (2 * s2 + hi) + 1) ^ w2sw));
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = Mfma32<T>::mma(wl[j], Hh[s2], acc[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = Mfma32<T>::mma(wh[j], Hl[s2], acc[j]);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = Mfma32<T>::mma(wh[j], Hh[s2], acc[j]);
    }
  }

  // ---- epilogue: + b2, fp32 rows (lane = token, 4 consecutive features per accumulator quad) ----
  if (m < p.M) {
    float* orow = p.out + (long)m * p.ldo;
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * j + 8 * g + 4 * hi;
        const float4 b4 = *reinterpret_cast<const float4*>(p.b2 + n);
        *reinterpret_cast<float4*>(orow + n) = make_float4(acc[j][4 * g] + b4.x, acc[j][4 * g + 1] + b4.y, acc[j][4 * g + 2] + b4.z, acc[j][4 * g + 3] + b4.w);
      }
  }
}

}  // namespace hipie

extern "C" int hipie_ffn_fused(const void* x, int64_t ldx, const void* w1, const float* b1, const void* w2p, const float* b2, float* out,
                               int64_t ldo, int M, int D, int F, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(x && w1 && b1 && w2p && b2 && out, "ffn_fused: null pointer");
  HIPIE_REQUIRE(D == FF_D && F == FF_F, "ffn_fused: model width %d / hidden width %d (built for 256 / 2048)", D, F);
  HIPIE_REQUIRE(M > 0 && ldx >= 2 * D && ldx % 8 == 0 && ldo >= D && ldo % 4 == 0, "ffn_fused: M=%d ldx=%ld ldo=%ld", M, (long)ldx, (long)ldo);
  HIPIE_REQUIRE((((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2p | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)out) & 15) == 0,
                "ffn_fused: pointers must be 16-byte aligned");
  FFParams p;
  p.X = (const char*)x; p.W1 = (const char*)w1; p.W2 = (const char*)w2p; p.b1 = b1; p.b2 = b2; p.out = out;
  p.ldx_b = ldx * 2; p.ldw1_b = (long)2 * D * 2; p.ldw2_b = (long)2 * F * 2; p.ldo = ldo;
  p.M = M;
  const size_t lds = (size_t)2 * FF_BUF + FF_F * sizeof(float);
  static bool lds_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)ffn_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) lds_set[dev] = true;
  }
  hipLaunchKernelGGL(ffn_fused_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), lds, (hipStream_t)stream, p);
  return check_launch("ffn_fused");
}

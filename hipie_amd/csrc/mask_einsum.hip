// mask_einsum.hip -- mask-logit contraction out[b,q,p] = sum_c embed[b,q,c] * feats[b,c,p]   (SURVEY row a21).
//
// Shape at 1024^2: Q = 300, C = 256, P = H*W = 65536 per image; 10.07 GFLOP but 67 MB of features in and 79 MB (f32)
// of logits out -> AI = 69 flop/B (f32 out), far below the MI355X ridge: HBM-bound, the output write dominates.
//
// Decomposition (fp32 features): a wave owns one 32-wide MFMA column block of pixels and ALL queries of a pass: its whole
// (320 x 32) f32 accumulator lives in registers (10 x 16 = 160).  So every feature element is read from HBM exactly once and
// every logit written exactly once; `embed` (307 KB) is re-read per workgroup from L2 and staged through LDS in K-chunks of
// 16 channels.
//   B operand (features): feats is (C, P) with p fastest, lanes run along p -> each lane's 8 k-values are 8 different
//     rows, each row load is a 128-byte coalesced segment per half-wave; no LDS, no transpose needed because no other
//     wave ever uses these elements.
//   A operand (embed): LDS tile [320 q][16 c], row stride padded to keep ds_read_b128/b32 conflict-free.
//   C tile: register r of a 32x32 block is one query row and 32 consecutive pixels per half-wave: full 128-byte line stores.
// Precision modes:
//   0  v_mfma_f32_32x32x2_f32: exact fp32 (an fmaf chain) -- MFMA-f32-bound at 157 TF (about 65 us / image); the plain
//      two-barrier kernel (mask_einsum_kernel), 8 waves per workgroup;
//   1  bf16x3: a = a_hi + a_lo, b = b_hi + b_lo in bf16, out += a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (f32 accumulate):
//      ~2^-16 relative error at 3/16 of the f32-MFMA time -> back under the HBM roof;
//   2  single bf16 MFMA (2^-8 relative inputs).
//   Modes 1 and 2 run software-pipelined kernels: 4 waves = 128 pixels per workgroup, TWO workgroups per CU (one covers the
//   other's prologue / store burst), features prefetched K-chunks ahead straight into registers, the embed chunk one ahead
//   into the other half of a double-buffered LDS tile, ONE barrier per chunk.
//     mask_einsum_x3_kernel (hipie_mask_einsum, no workspace): the embed chunk is fetched as fp32, split in registers and
//       written to LDS by every workgroup -- about as much VALU work per chunk as the chunk's 30 MFMAs take (0.44 ms per bs-8
//       call against 0.46 ms of the old two-barrier form: the 3-product contraction is MFMA-issue bound once the loads are
//       hidden, 258 GFLOP of bf16 MFMA = 0.19 ms at the power-limited rate, and the split competes with it);
//     mask_einsum_dma_kernel (hipie_mask_einsum_ws): the embedding is split ONCE per call into a workspace that already has
//       the LDS tile layout (me_split_embed_kernel, a few microseconds) and the chunks arrive by LDS-DMA: no staging
//       registers, no conversions, no ds_write in the loop.
#include <stdlib.h>

#include "common.h"
#include "mfma.h"

namespace hipie {

constexpr int ME_QB = 10;        // 32-row query blocks per pass (Q <= 320 per pass)
constexpr int ME_KC = 16;        // channels per K chunk (one k16 MFMA step / eight k2 steps)
constexpr int ME_WAVES = 8;      // waves per workgroup of the fp32 kernel (two per SIMD, 256 registers each)
constexpr int ME_TP = ME_WAVES * 32;   // pixels per workgroup
constexpr int ME_F32_STRIDE = 17;   // floats per LDS row (mode 0): bank = (17 q + c) % 32 distinct over q
constexpr int ME_B16_STRIDE = 24;   // bf16 per LDS row (modes 1,2): 48 B rows -> 12 q % 64 banks, conflict-free b128

// ---- mode 0: exact fp32 products on v_mfma_f32_32x32x2_f32 (MFMA-f32 bound; two barriers per chunk are not what limits it) ----
template <typename OutT>
__global__ __launch_bounds__(ME_WAVES * 64) void mask_einsum_f32_kernel(const float* __restrict__ embed,
                                                                 const float* __restrict__ feats, OutT* __restrict__ out,
                                                                 int Q, int C, int P) {
  __shared__ __attribute__((aligned(16))) float e32[ME_QB * 32 * ME_F32_STRIDE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const long p0 = (long)blockIdx.x * ME_TP + wave * 32;
  const float* E = embed + (long)b * Q * C;
  const float* F = feats + (long)b * C * P;
  OutT* O = out + (long)b * Q * P;
  const long pc = min(p0 + li, (long)P - 1);       // the tail workgroup clamps its loads and predicates its stores
  const bool pv = p0 + li < P;

  for (int q0 = 0; q0 < Q; q0 += ME_QB * 32) {
    const int nq = min(Q - q0, ME_QB * 32);
    const int nqb = (nq + 31) / 32;
    f32x16 acc[ME_QB];
#pragma unroll
    for (int i = 0; i < ME_QB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int kc = 0; kc < C; kc += ME_KC) {
      // ---- B operands for the whole chunk straight from HBM (issued first: they overlap the LDS staging) ----
      float bv[ME_KC / 2];                          // k2 step kk: lane holds F[kc + 2 kk + hi][p]
#pragma unroll
      for (int kk = 0; kk < ME_KC / 2; ++kk) bv[kk] = F[(long)(kc + 2 * kk + hi) * P + pc];
      // ---- stage embed[q0 .. q0+320, kc .. kc+16) into LDS ----
      __syncthreads();   // previous chunk's readers are done
      for (int i = tid; i < ME_QB * 32 * (ME_KC / 4); i += ME_WAVES * 64) {
        const int q = i / (ME_KC / 4), c4 = (i % (ME_KC / 4)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nq) v = *reinterpret_cast<const float4*>(E + (long)(q0 + q) * C + kc + c4);
        float* d = e32 + q * ME_F32_STRIDE + c4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < ME_KC / 2; ++kk) {
#pragma unroll
        for (int qb = 0; qb < ME_QB; ++qb) {
          if (qb < nqb) {
            const float a = e32[(qb * 32 + li) * ME_F32_STRIDE + 2 * kk + hi];
            acc[qb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[kk], acc[qb], 0, 0, 0);
          }
        }
      }
    }
    // ---- epilogue: each register r of a block is one row q, 32 consecutive pixels per half-wave ----
#pragma unroll
    for (int qb = 0; qb < ME_QB; ++qb) {
      if (qb < nqb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qb * 32 + crow(r, hi);
          if (q < nq && pv) O[(long)(q0 + q) * P + pc] = elem<OutT>::from_f32(acc[qb][r]);
        }
      }
    }
  }
}

// ---- modes 1 / 2: software-pipelined bf16 split contraction (see the file header) --------------------------------------
constexpr int MX_WAVES = 4;                       // waves per workgroup, one 32-pixel column block each
constexpr int MX_TP = MX_WAVES * 32;              // pixels per workgroup
constexpr int MX_EPT = ME_QB * 32 * (ME_KC / 4) / (MX_WAVES * 64);   // float4 of the embed chunk staged per thread (5)

template <int PREC, typename OutT>
__global__ __launch_bounds__(MX_WAVES * 64, 2) void mask_einsum_x3_kernel(const float* __restrict__ embed,
                                                                         const float* __restrict__ feats,
                                                                         const float* __restrict__ row_bias,
                                                                         OutT* __restrict__ out, int Q, int C, int P) {
  constexpr int PARTS = PREC == 1 ? 2 : 1;
  constexpr int EB = ME_QB * 32 * ME_B16_STRIDE;                 // bf16 elements of one part (hi or lo) of one buffer
  __shared__ __attribute__((aligned(16))) bf16_t esm[2 * PARTS * EB];
  __shared__ float rbs[ME_QB * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const long p0 = (long)blockIdx.x * MX_TP + wave * 32;
  const float* E = embed + (long)b * Q * C;
  OutT* O = out + (long)b * Q * P;
  const long pc = min(p0 + li, (long)P - 1);       // the tail workgroup clamps its loads and predicates its stores
  const bool pv = p0 + li < P;
  // this lane's feature column: k-step of chunk kc reads F[(kc + 8 hi + j) * P + pc], j = 0 .. 7
  const float* Fl = feats + (long)b * C * P + (long)(8 * hi) * P + pc;
  const int nk = C / ME_KC;

  for (int q0 = 0; q0 < Q; q0 += ME_QB * 32) {
    const int nq = min(Q - q0, ME_QB * 32);
    const int nqb = (nq + 31) / 32;
    // staging role: float4 number i = tid + 256 j of the (320 x 16) chunk: row tid / 4 + 64 j, channels 4 (tid % 4) ..
    const int sq = tid >> 2, sc4 = (tid & 3) * 4;
    const float* Eq = E + (long)q0 * C + sc4;
    int eoff[MX_EPT];                                           // rows past the last query re-read it and are stored as zeros
#pragma unroll
    for (int j = 0; j < MX_EPT; ++j) eoff[j] = min(sq + 64 * j, nq - 1) * C;
    // split to bf16 hi (+ lo) and park in LDS buffer `buf`; !live: zeros (the padding steps of the K loop, below)
    auto stage = [&](const float4* ev, int buf, bool live) __attribute__((always_inline)) {
      bf16_t* dh = esm + buf * PARTS * EB + sq * ME_B16_STRIDE + sc4;
#pragma unroll
      for (int j = 0; j < MX_EPT; ++j) {
        const bool ok = live && sq + 64 * j < nq;
        const float f[4] = {ok ? ev[j].x : 0.f, ok ? ev[j].y : 0.f, ok ? ev[j].z : 0.f, ok ? ev[j].w : 0.f};
        bf16x4 h, l;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          h[t] = (bf16_t)f[t];
          l[t] = (bf16_t)(f[t] - (float)h[t]);
        }
        *reinterpret_cast<bf16x4*>(dh + 64 * j * ME_B16_STRIDE) = h;
        if (PREC == 1) *reinterpret_cast<bf16x4*>(dh + EB + 64 * j * ME_B16_STRIDE) = l;
      }
    };
    f32x16 acc[ME_QB];
#pragma unroll
    for (int i = 0; i < ME_QB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // ---- prologue: chunk 0 of both operands, chunk 1 of the features ----
    float4 ev[MX_EPT];
    float fa[8], fb[8], fc[8];                     // features of three consecutive chunks; the roles rotate by NAME (below)
#pragma unroll
    for (int j = 0; j < MX_EPT; ++j) ev[j] = *reinterpret_cast<const float4*>(Eq + eoff[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) fa[j] = Fl[(long)j * P];
    {
      const int c1 = min(1, nk - 1) * ME_KC;
#pragma unroll
      for (int j = 0; j < 8; ++j) fb[j] = Fl[(long)(c1 + j) * P];
    }
    for (int i = tid; i < ME_QB * 32; i += MX_WAVES * 64) rbs[i] = (row_bias && i < nq) ? row_bias[(long)b * Q + q0 + i] : 0.f;
    stage(ev, 0, true);
    __syncthreads();

    // One K-chunk: `cur` holds its features (loaded two steps ago), `nxt2` receives those of chunk k + 2.  The embed loads are
    // issued BEFORE the feature loads: memory returns in order, so the wait for them at the end of the step leaves the feature
    // loads of chunk k + 2 in flight across the barrier.  The body is branch-free: chunk numbers past the end are clamped
    // (re-reading the last chunk) and their embed tile is staged as zeros -- such a padding step adds nothing.
    auto step = [&](int k, const float* cur, float* nxt2) __attribute__((always_inline)) {
      {
        const int c1 = min(k + 1, nk - 1) * ME_KC;
#pragma unroll
        for (int j = 0; j < MX_EPT; ++j) ev[j] = *reinterpret_cast<const float4*>(Eq + eoff[j] + c1);
        __builtin_amdgcn_sched_barrier(0);         // keep the issue order (the scheduler moved an embed load behind the features)
        const int c2 = min(k + 2, nk - 1) * ME_KC;
#pragma unroll
        for (int j = 0; j < 8; ++j) nxt2[j] = Fl[(long)(c2 + j) * P];
        __builtin_amdgcn_sched_barrier(0);
      }
      bf16x8 bh, bl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bh[j] = (bf16_t)cur[j];
        bl[j] = (bf16_t)(cur[j] - (float)bh[j]);
      }
      const bf16_t* eh = esm + (k & 1) * PARTS * EB + li * ME_B16_STRIDE + 8 * hi;
      // every query block of the pass, whatever nq: rows past the last query are zeros in LDS, and ONE straight MFMA stream
      // (no per-block branch).  Reading block qb + 1 ahead of the products of block qb was tried: out of registers, the
      // allocator then funnels every fragment through one quad and waits after each read.
#pragma unroll
      for (int qb = 0; qb < ME_QB; ++qb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(eh + qb * 32 * ME_B16_STRIDE);
        if (PREC == 1) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(eh + EB + qb * 32 * ME_B16_STRIDE);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[qb], 0, 0, 0);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[qb], 0, 0, 0);
        }
        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
      }
      // the other buffer was last read in step k - 1, i.e. before the barrier that ended it.  Fenced off from the MFMA stream:
      // the scheduler otherwise starts the split right behind the first MFMA and the wave then waits for the embed loads it
      // issued a moment ago instead of at the end of the step
      __builtin_amdgcn_sched_barrier(0);
      stage(ev, (k + 1) & 1, k + 1 < nk);
      __syncthreads();
    };
    // The three feature register sets take turns by name, three steps per trip: a rolled loop of one step rotates them with
    // register copies, and a copy of a just-issued load waits for it (vmcnt(0) at the top of every step -- seen in the ISA);
    // a remainder after the loop (or a fully unrolled K loop) made the register allocator spill the accumulators.  So the trip
    // count is rounded UP to a multiple of three with padding steps: 18 for the 16 chunks of C = 256, 12 % more MFMA issue in a
    // kernel that waits for HBM.
    for (int k = 0; k < nk; k += 3) {
      step(k, fa, fc);
      step(k + 1, fb, fa);
      step(k + 2, fc, fb);
    }
    // ---- epilogue: register r of block qb is query row 32 qb + crow(r, hi), 32 consecutive pixels per half-wave ----
#pragma unroll
    for (int qb = 0; qb < ME_QB; ++qb) {
      if (qb < nqb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qb * 32 + crow(r, hi);
          if (q < nq && pv) O[(long)(q0 + q) * P + pc] = elem<OutT>::from_f32(acc[qb][r] + rbs[q]);
        }
      }
    }
    __syncthreads();                                // rbs / the LDS tile are rewritten by the next pass
  }
}

template <int PREC, typename OutT>
static int launch_mx(const float* e, const float* f, const float* rb, void* out, int B, int Q, int C, int P, hipStream_t st) {
  hipLaunchKernelGGL((mask_einsum_x3_kernel<PREC, OutT>), dim3((P + MX_TP - 1) / MX_TP, B), dim3(MX_WAVES * 64), 0, st, e, f, rb, (OutT*)out, Q, C, P);
  return check_launch("mask_einsum");
}

template <int PREC>
static int dispatch_mx(const float* e, const float* f, const float* rb, void* out, int B, int Q, int C, int P, int odt, hipStream_t st) {
  switch (odt) {
    case HIPIE_F32: return launch_mx<PREC, float>(e, f, rb, out, B, Q, C, P, st);
    case HIPIE_F16: return launch_mx<PREC, f16_t>(e, f, rb, out, B, Q, C, P, st);
    case HIPIE_BF16: return launch_mx<PREC, bf16_t>(e, f, rb, out, B, Q, C, P, st);
    default: return set_err(HIPIE_EINVAL, "mask_einsum: bad out_dtype %d", odt);
  }
}


// ---- modes 1 / 2 with a workspace: the embedding pre-split into LDS-tile images, staged by LDS-DMA ----------------------------
// One part (hi or lo) of one chunk image: 320 rows x 32 B (16 bf16), NO row padding -- every workgroup streams the whole image
// of its batch item from L2, so its size is traffic (the 48-byte rows of the register-staged kernel would be +50 %).  The two
// 16-byte units of a row are swapped in rows 8 .. 15 of every 16: a 16-lane pass of ds_read_b128 (rows r .. r + 15, same unit)
// then covers all 64 banks once.
constexpr int MD_ROW = 32;
constexpr int MD_PART = ME_QB * 32 * MD_ROW;              // 10 x 1 KB
constexpr int MD_CHUNK = 2 * MD_PART;                     // hi + lo: the image of one K chunk of one pass of one batch item
constexpr int MD_PERIOD = 4;                              // feature register sets (the K loop runs in multiples of this)

__device__ __forceinline__ void me_dma16(const char* sbase, unsigned int voff, unsigned int lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
  unsigned int keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
#endif
}
constexpr int me_vmcnt(int n) { return 0x0F70 | (n & 15) | ((n >> 4) << 14); }   // s_waitcnt vmcnt(n) only (gfx9 encoding)

// workspace image: [batch][pass][chunk 0 .. nk4)[part][320 rows][2 units of 8 bf16, swizzled]; rows past Q and chunks past C / 16 = 0
__global__ __launch_bounds__(256) void me_split_embed_kernel(const float* __restrict__ embed, char* __restrict__ ws, int Q, int C,
                                                             int npass, int nk4, long total) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= total) return;
  const int g = (int)(t & 3), row = (int)((t >> 2) % (ME_QB * 32));
  const long ck = t / (4 * ME_QB * 32);                    // (batch * npass + pass) * nk4 + chunk
  const int k = (int)(ck % nk4), pass = (int)((ck / nk4) % npass);
  const long b = ck / ((long)nk4 * npass);
  const int q = pass * ME_QB * 32 + row, c = k * ME_KC + 4 * g;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < Q && c < C) v = *reinterpret_cast<const float4*>(embed + ((long)b * Q + q) * C + c);
  const float f[4] = {v.x, v.y, v.z, v.w};
  bf16x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    h[i] = (bf16_t)f[i];
    l[i] = (bf16_t)(f[i] - (float)h[i]);
  }
  char* d = ws + ck * MD_CHUNK + row * MD_ROW + 16 * ((g >> 1) ^ ((row >> 3) & 1)) + 8 * (g & 1);
  *reinterpret_cast<bf16x4*>(d) = h;
  *reinterpret_cast<bf16x4*>(d + MD_PART) = l;
}

template <int PREC, typename OutT>
__global__ __launch_bounds__(MX_WAVES * 64, 2) void mask_einsum_dma_kernel(const char* __restrict__ ws, const float* __restrict__ feats,
                                                                          const float* __restrict__ row_bias, OutT* __restrict__ out,
                                                                          int Q, int C, int P, int npass, int nk4, int tiles, int xmap) {
  constexpr int PARTS = PREC == 1 ? 2 : 1;
  constexpr int NDMA = PARTS * (MD_PART / 1024);                 // 1 KB LDS-DMA instructions per chunk and workgroup (20 | 10)
  constexpr int DPW = (NDMA + MX_WAVES - 1) / MX_WAVES;          // ... per wave (the last ones predicated by wave)
  constexpr int NBUF = 3;                                        // LDS tiles: chunk kk lives in tile kk % 3, fetched two steps ahead
  __shared__ __attribute__((aligned(1024))) char esm[NBUF * PARTS * MD_PART];
  __shared__ float rbs[ME_QB * 32];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  // Workgroup -> (batch item, pixel tile).  Every workgroup streams the embedding image of its batch item (C / 16 chunks x 20 KB)
  // through L2: with the hardware's round-robin placement (workgroup id % 8 = XCD) and batch-major ids every XCD's L2 would hold
  // the images of ALL items beside the feature stream and lose them to it.  xmap (B a multiple of 8): XCD x serves the items
  // x, x + 8, .. only, one after the other -- its L2 keeps one 320 KB image at a time.
  int b, tile;
  if (xmap) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    b = xcd + 8 * (slot / tiles);
    tile = slot % tiles;
  } else {
    b = blockIdx.x / tiles;
    tile = blockIdx.x % tiles;
  }
  const long p0 = (long)tile * MX_TP + wave * 32;
  OutT* O = out + (long)b * Q * P;
  const long pc = min(p0 + li, (long)P - 1);       // the tail workgroup clamps its loads and predicates its stores
  const bool pv = p0 + li < P;
  const float* Fl = feats + (long)b * C * P + (long)(8 * hi) * P + pc;
  const int nk = C / ME_KC;
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)esm);
  const unsigned int dvoff = 16 * lane;

  for (int pass = 0; pass < npass; ++pass) {
    const int q0 = pass * ME_QB * 32;
    const int nq = min(Q - q0, ME_QB * 32);
    const int nqb = (nq + 31) / 32;
    const char* img = ws + ((long)b * npass + pass) * nk4 * MD_CHUNK;
    // chunk kk of this pass -> LDS tile kk % 3: instruction i = wave + 4 j moves bytes [1 KB i, 1 KB (i + 1)) of the image
    auto dma = [&](int kk) __attribute__((always_inline)) {
      const char* src = img + (long)min(kk, nk4 - 1) * MD_CHUNK;
      const unsigned int dst = lds0 + (kk % NBUF) * PARTS * MD_PART;
#pragma unroll
      for (int j = 0; j < DPW; ++j) {               // NDMA is not a multiple of 4: the last round repeats the last piece (same bytes,
        const int i = min(wave + MX_WAVES * j, NDMA - 1);   // same place) rather than branch -- every wave issues exactly DPW
        me_dma16(src + 1024 * i, dvoff, __builtin_amdgcn_readfirstlane(dst + 1024 * i));
      }
    };
    f32x16 acc[ME_QB];
#pragma unroll
    for (int i = 0; i < ME_QB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    for (int i = tid; i < ME_QB * 32; i += MX_WAVES * 64) rbs[i] = (row_bias && i < nq) ? row_bias[(long)b * Q + q0 + i] : 0.f;
    // ---- prologue: chunks 0, 1 of the embedding, chunks 0 .. 2 of the features ----
    float f0[8], f1[8], f2[8], f3[8];                // features of four consecutive chunks; the roles rotate by NAME (below)
    dma(0);
    dma(1);
    {
      const int c1 = min(1, nk - 1) * ME_KC, c2 = min(2, nk - 1) * ME_KC;
#pragma unroll
      for (int j = 0; j < 8; ++j) f0[j] = Fl[(long)j * P];
#pragma unroll
      for (int j = 0; j < 8; ++j) f1[j] = Fl[(long)(c1 + j) * P];
#pragma unroll
      for (int j = 0; j < 8; ++j) f2[j] = Fl[(long)(c2 + j) * P];
    }
    __builtin_amdgcn_s_waitcnt(me_vmcnt(24));       // the DMA pieces are older than the 24 feature loads (vmcnt counts in order)
    __syncthreads();

    // One K-chunk: `cur` holds its features (loaded three steps ago), `nxt3` receives those of chunk k + 3; the image of chunk
    // k + 2 is DMAed into the LDS tile step k - 1 read (everybody passed the barrier that ended it).  Memory returns in order:
    // the queue at the end of step k is .. DMA(k+1) | features(k+2) | DMA(k+2) | features(k+3), so waiting until 8 + DPW + 8
    // operations are outstanding means "the image of chunk k + 1 has landed" and leaves two steps of loads in flight across
    // the barrier.  Branch-free; chunk numbers past the end are clamped and their image is zero.
    auto step = [&](int k, const float* cur, float* nxt3) __attribute__((always_inline)) {
      dma(k + 2);
      {
        const int c3 = min(k + 3, nk - 1) * ME_KC;
#pragma unroll
        for (int j = 0; j < 8; ++j) nxt3[j] = Fl[(long)(c3 + j) * P];
      }
      __builtin_amdgcn_sched_barrier(0);
      bf16x8 bh, bl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bh[j] = (bf16_t)cur[j];
        bl[j] = (bf16_t)(cur[j] - (float)bh[j]);
      }
      const char* eh = esm + (k % NBUF) * PARTS * MD_PART + li * MD_ROW + 16 * (hi ^ ((li >> 3) & 1));
#pragma unroll
      for (int qb = 0; qb < ME_QB; ++qb) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(eh + qb * 32 * MD_ROW);
        if (PREC == 1) {
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(eh + MD_PART + qb * 32 * MD_ROW);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[qb], 0, 0, 0);
          acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[qb], 0, 0, 0);
        }
        acc[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[qb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_waitcnt(me_vmcnt(16 + DPW));   // this wave's pieces of chunk k + 1 have landed ...
      __syncthreads();                              // ... everybody's; all reads of chunk k are done
    };
    // four steps per trip, the feature sets taking turns by name (a register rotation would wait for the loads it copies); the
    // trip count is rounded up with zero chunks of the workspace image (none for C = 256)
    for (int k = 0; k < nk4; k += MD_PERIOD) {
      step(k, f0, f3);
      step(k + 1, f1, f0);
      step(k + 2, f2, f1);
      step(k + 3, f3, f2);
    }
    // ---- epilogue: register r of block qb is query row 32 qb + crow(r, hi), 32 consecutive pixels per half-wave ----
#pragma unroll
    for (int qb = 0; qb < ME_QB; ++qb) {
      if (qb < nqb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = qb * 32 + crow(r, hi);
          if (q < nq && pv) O[(long)(q0 + q) * P + pc] = elem<OutT>::from_f32(acc[qb][r] + rbs[q]);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(me_vmcnt(0));        // the clamped DMA of the last step targets the buffer the next pass fills
    __syncthreads();                                // rbs / the LDS tile are rewritten by the next pass
  }
}

template <int PREC, typename OutT>
static int launch_md(const float* e, const float* f, const float* rb, void* out, char* ws, int B, int Q, int C, int P, hipStream_t st) {
  const int npass = (Q + ME_QB * 32 - 1) / (ME_QB * 32), nk = C / ME_KC, nk4 = (nk + MD_PERIOD - 1) / MD_PERIOD * MD_PERIOD;
  const long total = (long)B * npass * nk4 * ME_QB * 32 * 4;
  hipLaunchKernelGGL(me_split_embed_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, e, ws, Q, C, npass, nk4, total);
  const int tiles = (P + MX_TP - 1) / MX_TP;
  static const int xenv = [] { const char* v = study_env("HIPIE_ME_XMAP"); return v ? atoi(v) : 1; }();    // 0: batch-major ids (A/B timing)
  hipLaunchKernelGGL((mask_einsum_dma_kernel<PREC, OutT>), dim3((unsigned)((long)B * tiles)), dim3(MX_WAVES * 64), 0, st, (const char*)ws, f, rb,
                     (OutT*)out, Q, C, P, npass, nk4, tiles, (B % 8 == 0 && xenv) ? 1 : 0);
  return check_launch("mask_einsum_ws");
}

template <int PREC>
static int dispatch_md(const float* e, const float* f, const float* rb, void* out, char* ws, int B, int Q, int C, int P, int odt, hipStream_t st) {
  switch (odt) {
    case HIPIE_F32: return launch_md<PREC, float>(e, f, rb, out, ws, B, Q, C, P, st);
    case HIPIE_F16: return launch_md<PREC, f16_t>(e, f, rb, out, ws, B, Q, C, P, st);
    case HIPIE_BF16: return launch_md<PREC, bf16_t>(e, f, rb, out, ws, B, Q, C, P, st);
    default: return set_err(HIPIE_EINVAL, "mask_einsum_ws: bad out_dtype %d", odt);
  }
}

template <typename OutT>
static int launch_me(const float* e, const float* f, void* out, int B, int Q, int C, int P, hipStream_t st) {
  hipLaunchKernelGGL((mask_einsum_f32_kernel<OutT>), dim3((P + ME_TP - 1) / ME_TP, B), dim3(ME_WAVES * 64), 0, st, e, f, (OutT*)out, Q, C, P);
  return check_launch("mask_einsum");
}

static int dispatch_me(const float* e, const float* f, void* out, int B, int Q, int C, int P, int odt, hipStream_t st) {
  switch (odt) {
    case HIPIE_F32: return launch_me<float>(e, f, out, B, Q, C, P, st);
    case HIPIE_F16: return launch_me<f16_t>(e, f, out, B, Q, C, P, st);
    case HIPIE_BF16: return launch_me<bf16_t>(e, f, out, B, Q, C, P, st);
    default: return set_err(HIPIE_EINVAL, "mask_einsum: bad out_dtype %d", odt);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// 16-bit form (the fast policy): features arrive in the activation dtype (f16 / bf16), logits leave in 16 bit (or f32).
//   algorithmic bytes per image at 1024^2: 33.5 MB of features in + 39.3 MB of logits out = 73 MB (AI 138 flop/B) -- HBM bound.
// The product is computed TRANSPOSED, out^T = feats^T . embed^T: pixels are the MFMA rows, queries the columns.  Then
//   * the A operand of a lane is (pixel, 8 channels): with TWO interleaved pixel blocks per wave (block hb = pixels 2 i + hb)
//     one 4-byte load per channel row fetches a lane's pixel pair -- 128 contiguous bytes per half-wave and channel;
//   * in the accumulator a lane owns ONE query row and registers 4g .. 4g+3 of the two blocks are 8 CONSECUTIVE pixels of it:
//     the epilogue is one 16-byte store per lane (16-bit output) instead of sixteen 2-byte ones;
//   * the B operand (embed, pre-split by the host into 16-bit hi + lo parts: two MFMAs per product keep the query embedding
//     at ~2^-22; lo == NULL: single product) is a plain 16-byte load per lane from the L2-resident (Q, C) matrix.
// Workgroup = 8 waves = 256 pixels x 320 queries: wave w owns pixel group w >> 1 (64 pixels) and query blocks 5 (w & 1) ..
template <typename T, typename OutT, bool LO, int ABL>
__global__ __launch_bounds__(512, 2) void mask_einsum16_kernel(const T* __restrict__ ehi, const T* __restrict__ elo,
                                                               const T* __restrict__ feats, const float* __restrict__ row_bias,
                                                               OutT* __restrict__ out, int Q, int C, int P) {
  typedef typename Mfma32<T>::frag frag;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // embed k-chunk (16 channels of all 320 query rows, hi and lo parts) staged in LDS once per workgroup and k-step, double
  // buffered: every wave reads its 5 (x2) B fragments from there -- fetched per wave from L2 they were 5x the feature bytes
  constexpr int ESTR = 24;                               // elements per LDS row: 48 B -> conflict-free ds_read_b128
  constexpr int EBUF = 320 * ESTR;                       // one part (hi or lo) of one buffer
  __shared__ __attribute__((aligned(16))) T esm[2 * (LO ? 2 : 1) * EBUF];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int b = blockIdx.y;
  const int qh = wave & 1;
  const long p0 = (long)blockIdx.x * 256 + (wave >> 1) * 64;
  const T* F = feats + (long)b * C * P;
  const T* EH = ehi + (long)b * Q * C;
  const T* EL = LO ? elo + (long)b * Q * C : nullptr;
  OutT* O = out + (long)b * Q * P;
  const long pl = min(p0 + 2 * li, (long)P - 2);          // this lane's pixel pair (clamped: the tail predicates its stores)

  f32x16 acc[5][2];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging role of this thread: 640 16-byte chunks per part and k-step (row q = chunk / 2, half = chunk % 2)
  constexpr int NCH = 640, PARTS = LO ? 2 : 1, CPT = (NCH * PARTS + 511) / 512;
  const T* esrc[CPT];
  int edst[CPT];
  bool eval_[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const int idx = c * 512 + tid;
    eval_[c] = idx < NCH * PARTS;
    const int part = min(idx / NCH, PARTS - 1), ch = idx % NCH, q = min(ch >> 1, Q - 1), hf = ch & 1;
    esrc[c] = (part ? EL : EH) + (long)q * C + 8 * hf;
    edst[c] = part * EBUF + (ch >> 1) * ESTR + 8 * hf;
  }
  // Operand pipeline, two k-steps deep (one workgroup per CU: nothing else hides the L2 / HBM latency): at the top of step k
  // the loads of step k + 2 are issued; the embed chunk of step k + 1 (loaded one step ago) goes to LDS at the end of step k.
  u32x4 e1[CPT], e2[CPT];           // embed chunks of steps k + 1, k + 2 (registers)
  unsigned int fw[8], f1[8], f2[8]; // feature words of steps k, k + 1, k + 2
  const int nk = C / 16;
#pragma unroll
  for (int c = 0; c < CPT; ++c) e1[c] = *reinterpret_cast<const u32x4*>(esrc[c]);
#pragma unroll
  for (int j = 0; j < 8; ++j) fw[j] = *reinterpret_cast<const unsigned int*>(F + (long)(8 * hi + j) * P + pl);
#pragma unroll
  for (int c = 0; c < CPT; ++c)
    if (eval_[c]) *reinterpret_cast<u32x4*>(esm + edst[c]) = e1[c];
  if (nk > 1) {
#pragma unroll
    for (int c = 0; c < CPT; ++c) e1[c] = *reinterpret_cast<const u32x4*>(esrc[c] + 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) f1[j] = *reinterpret_cast<const unsigned int*>(F + (long)(16 + 8 * hi + j) * P + pl);
  }
  __syncthreads();

  for (int k = 0; k < nk; ++k) {
    const T* ecur = esm + (k & 1) * PARTS * EBUF;
    if (k + 2 < nk) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) e2[c] = *reinterpret_cast<const u32x4*>(esrc[c] + 16 * (k + 2));
#pragma unroll
      for (int j = 0; j < 8; ++j) f2[j] = *reinterpret_cast<const unsigned int*>(F + (long)(16 * (k + 2) + 8 * hi + j) * P + pl);
    }
    u32x4 a0, a1;                                        // even-pixel / odd-pixel A fragments: 8 channels x 16 bit
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a0[j] = __builtin_amdgcn_perm(fw[2 * j + 1], fw[2 * j], 0x05040100u);      // lo halves of (fw[2j], fw[2j+1])
      a1[j] = __builtin_amdgcn_perm(fw[2 * j + 1], fw[2 * j], 0x07060302u);      // hi halves
    }
    const frag af[2] = {__builtin_bit_cast(frag, a0), __builtin_bit_cast(frag, a1)};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = 32 * (5 * qh + i) + li;
      const frag bh = *reinterpret_cast<const frag*>(ecur + row * ESTR + 8 * hi);
      if (LO) {
        const frag bl = *reinterpret_cast<const frag*>(ecur + EBUF + row * ESTR + 8 * hi);
#pragma unroll
        for (int hb = 0; hb < 2; ++hb) acc[i][hb] = Mfma32<T>::mma(af[hb], bl, acc[i][hb]);
      }
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        if (ABL == 1) { acc[i][hb][0] += (float)af[hb][0] + (float)bh[0]; }
        else acc[i][hb] = Mfma32<T>::mma(af[hb], bh, acc[i][hb]);
      }
    }
    if (k + 1 < nk) {
      T* enext = esm + ((k + 1) & 1) * PARTS * EBUF;     // last read in step k - 1, before the previous barrier
#pragma unroll
      for (int c = 0; c < CPT; ++c)
        if (eval_[c]) *reinterpret_cast<u32x4*>(enext + edst[c]) = e1[c];
#pragma unroll
      for (int c = 0; c < CPT; ++c) e1[c] = e2[c];
#pragma unroll
      for (int j = 0; j < 8; ++j) { fw[j] = f1[j]; f1[j] = f2[j]; }
    }
    __syncthreads();
  }
  // ---- epilogue: lane = query row; registers 4g .. 4g+3 of the two blocks = pixels p0 + 16 g + 8 hi + (0 .. 7) ----
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int q = 32 * (5 * qh + i) + li;
    if (q >= Q) continue;
    if (ABL == 2 && acc[i][0][0] != 12345.f) continue;        // timing ablation: no stores
    OutT* orow = O + (long)q * P + p0;
    const float rb = row_bias ? row_bias[(long)b * Q + q] : 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long px = p0 + 16 * g + 8 * hi;
      if (px + 8 <= P) {
        typedef OutT o8 __attribute__((ext_vector_type(8)));
        o8 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = elem<OutT>::from_f32(acc[i][0][4 * g + e] + rb);
          v[2 * e + 1] = elem<OutT>::from_f32(acc[i][1][4 * g + e] + rb);
        }
        *reinterpret_cast<o8*>(orow + 16 * g + 8 * hi) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (px + 2 * e < P) orow[16 * g + 8 * hi + 2 * e] = elem<OutT>::from_f32(acc[i][0][4 * g + e] + rb);
          if (px + 2 * e + 1 < P) orow[16 * g + 8 * hi + 2 * e + 1] = elem<OutT>::from_f32(acc[i][1][4 * g + e] + rb);
        }
      }
    }
  }
}

template <typename T, typename OutT>
static int launch_me16(const void* eh, const void* el, const void* f, const float* rb, void* out, int B, int Q, int C, int P, hipStream_t st) {
  dim3 grid((P + 255) / 256, B);
#ifdef HIPIE_VA_ABLATIONS
  { const char* e = study_env("HIPIE_ME_ABL"); const int a = e ? atoi(e) : 0;
    if (a == 1) { hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, false, 1>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P); return check_launch("me16"); }
    if (a == 2) { hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, false, 2>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P); return check_launch("me16"); } }
#endif
  if (el != nullptr)
    hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, true, 0>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P);
  else
    hipLaunchKernelGGL((mask_einsum16_kernel<T, OutT, false, 0>), grid, dim3(512), 0, st, (const T*)eh, (const T*)el, (const T*)f, rb, (OutT*)out, Q, C, P);
  return check_launch("mask_einsum16");
}

}  // namespace hipie

extern "C" int hipie_mask_einsum(const float* embed, const float* feats, void* out, int B, int Q, int C, int HW,
                                 int precision, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(embed && feats && out, "mask_einsum: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q > 0 && C > 0 && HW > 0, "mask_einsum: bad shape");
  HIPIE_REQUIRE(C % ME_KC == 0, "mask_einsum: C=%d must be a multiple of %d", C, ME_KC);
  if (B == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (precision) {
    case 0: return dispatch_me(embed, feats, out, B, Q, C, HW, out_dtype, st);
    case 1: return dispatch_mx<1>(embed, feats, nullptr, out, B, Q, C, HW, out_dtype, st);
    case 2: return dispatch_mx<2>(embed, feats, nullptr, out, B, Q, C, HW, out_dtype, st);
    default: return set_err(HIPIE_EINVAL, "mask_einsum: bad precision %d", precision);
  }
}

extern "C" int64_t hipie_mask_einsum_workspace(int B, int Q, int C) {
  using namespace hipie;
  if (B <= 0 || Q <= 0 || C <= 0) return 0;
  const long npass = (Q + ME_QB * 32 - 1) / (ME_QB * 32), nk = (C + ME_KC - 1) / ME_KC, nk4 = (nk + MD_PERIOD - 1) / MD_PERIOD * MD_PERIOD;
  return (int64_t)B * npass * nk4 * MD_CHUNK;
}

extern "C" int hipie_mask_einsum_ws(const float* embed, const float* feats, const float* row_bias, void* out, void* workspace,
                                    int64_t workspace_bytes, int B, int Q, int C, int HW, int precision, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(embed && feats && out, "mask_einsum_ws: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q > 0 && C > 0 && HW > 0 && (long)B * ((HW + 127) / 128) < (1L << 31), "mask_einsum_ws: bad shape");
  HIPIE_REQUIRE(C % ME_KC == 0, "mask_einsum_ws: C=%d must be a multiple of %d", C, ME_KC);
  HIPIE_REQUIRE(precision == 1 || precision == 2, "mask_einsum_ws: precision must be 1 (bf16 x 3) or 2 (bf16), got %d", precision);
  if (B == 0) return HIPIE_OK;
  const int64_t need = hipie_mask_einsum_workspace(B, Q, C);
  HIPIE_REQUIRE(workspace != nullptr && workspace_bytes >= need, "mask_einsum_ws: workspace of %ld bytes needed (hipie_mask_einsum_workspace), got %ld",
                (long)need, (long)workspace_bytes);
  HIPIE_REQUIRE((((uintptr_t)workspace | (uintptr_t)embed) & 15) == 0, "mask_einsum_ws: workspace / embed must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return precision == 1 ? dispatch_md<1>(embed, feats, row_bias, out, (char*)workspace, B, Q, C, HW, out_dtype, st)
                        : dispatch_md<2>(embed, feats, row_bias, out, (char*)workspace, B, Q, C, HW, out_dtype, st);
}

extern "C" int hipie_mask_einsum16(const void* embed_hi, const void* embed_lo, const void* feats, const float* row_bias, void* out,
                                   int B, int Q, int C, int HW, int dtype, int out_dtype, void* stream) {
  using namespace hipie;
  HIPIE_REQUIRE(embed_hi && feats && out, "mask_einsum16: null pointer");
  HIPIE_REQUIRE(B >= 0 && Q > 0 && Q <= 320 && C > 0 && HW > 0, "mask_einsum16: bad shape (Q=%d must be in 1..320)", Q);
  HIPIE_REQUIRE(C % 16 == 0 && HW % 2 == 0, "mask_einsum16: C=%d must be a multiple of 16 and HW=%d even", C, HW);
  HIPIE_REQUIRE((((uintptr_t)embed_hi | (uintptr_t)embed_lo | (uintptr_t)feats | (uintptr_t)out) & 15) == 0, "mask_einsum16: pointers must be 16-byte aligned");
  HIPIE_REQUIRE(out_dtype == dtype || out_dtype == HIPIE_F32, "mask_einsum16: out_dtype must be the feature dtype or f32");
  if (B == 0) return HIPIE_OK;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == HIPIE_F16) {
    if (out_dtype == HIPIE_F16) { HIPIE_REQUIRE(HW % 8 == 0, "mask_einsum16: HW %% 8"); return launch_me16<f16_t, f16_t>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st); }
    return launch_me16<f16_t, float>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st);
  }
  if (dtype == HIPIE_BF16) {
    if (out_dtype == HIPIE_BF16) { HIPIE_REQUIRE(HW % 8 == 0, "mask_einsum16: HW %% 8"); return launch_me16<bf16_t, bf16_t>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st); }
    return launch_me16<bf16_t, float>(embed_hi, embed_lo, feats, row_bias, out, B, Q, C, HW, st);
  }
  return set_err(HIPIE_EINVAL, "mask_einsum16: dtype must be f16 or bf16 (got %d)", dtype);
}

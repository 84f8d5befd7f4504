// gemm_overlap_study.h -- round-4 timing study (moved out of hipie_amd/csrc in round 5), compiled ONLY into `make -C hipie_amd/csrc EXTRA=-DHIPIE_GEMM_VARIANTS` builds (included by gemm.hip).
// Three attempts to run the tile epilogue of the split GEMM beside another tile's k loop: gemm2 (two free-running 4-wave workgroups per
// CU), gemm3 (one persistent workgroup whose two 4-wave groups ping-pong) and gemm4 (gemm2 made persistent, the second workgroup of a CU
// started half a tile late).  All produce correct results (tests/test_gpu_gemm.py passes with HIPIE_GEMM2=1|2|4) and none beats gemm_kernel on the ViT shapes; the measurements and what they established are in
// DESIGN.md section 9 and profiles/r04_gemm_overlap_study.md.  Not part of the product path.
#pragma once
__device__ unsigned int g_cu_arrivals[2048];       // per-CU arrival counters of gemm4 (zero at module load, only ever grow)
// ------------------------------------------------------------------------------------------------------------------------------
// gemm2: the SPLIT product as TWO independent 4-wave workgroups per CU (round 4).
//
// The 8-wave kernel above keeps the matrix pipe idle while all its waves run the tile epilogue together (15 % of the K = 1280
// shapes, almost half of the K = 256 shapes of the deformable encoders), and 144 KB of LDS leave no room for a second workgroup.
// Here a workgroup is 4 waves stacked along M -- tile 256 tokens x BN = 32 NJ features (160 | 128), a wave owns 64 tokens x BN
// = 2 x NJ MFMA tiles exactly as before -- with a THREE-slot LDS ring of k16 steps (64 bytes per operand row: 3 x 26 KB), so two
// workgroups share a CU, one wave of each per SIMD: while one workgroup stores its tile (or waits at its stage barrier) the
// other one has the matrix pipe to itself.  The hardware arbitrates; nothing is synchronised across the two.
//   * LDS rows of 64 B: chunk c (16 B) of tile row r sits at chunk position c ^ ((r >> 2) & 3) -- every ds_read_b128 lane group
//     (MI355X_MICROARCH.md LDS table) then covers the 64 banks exactly once; one LDS-DMA instruction fills 16 rows (1 KB);
//   * A rows are private to their wave (its own 64 tokens): only the W tile needs the stage barrier; DMA of step t + 2 is issued
//     during step t, vmcnt(<issued this step>) + one barrier per step;
//   * phase: two workgroups that start together would run in lock step (and store together).  The workgroups of the first
//     dispatch round's second half (blocks 256..511) therefore run their tile at low priority: their partner finishes first
//     and every later pair on that CU stays out of phase by construction.
template <int NJ, int VAR>
__global__ __launch_bounds__(256, 2) void gemm2_kernel(const GemmParams pin) {
  GemmParams p = pin;
  if (gridDim.y > 1) {
    const int bo = blockIdx.y / p.nbi, bi = blockIdx.y - bo * p.nbi;
    p.A += bo * p.a_bo + bi * p.a_bi;
    p.W += bo * p.w_bo + bi * p.w_bi;
    p.out += bo * p.o_bo + bi * p.o_bi;
  }
  constexpr int BM = 256, BN = 32 * NJ;
  constexpr int ROWS = BM + BN;
  constexpr int STAGE = ROWS * 64;             // bytes of one k16 step: A rows then W rows
  constexpr int NWI = (BN + 63) / 64;          // W DMA instructions per wave (instruction i: W rows 16 (4 i + wave) ..+15, if < BN)
  typedef Mfma32<f16_t>::frag frag;

  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, hi = lane >> 5;

#if defined(__HIP_DEVICE_COMPILE__)
  if (p.prio_mode == 1) {
    if (blockIdx.x >= 256 && blockIdx.x < 512) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
  } else if (p.prio_mode == 2) {
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
  } else if (p.prio_mode == 3) {
    if (((blockIdx.x >> 8) ^ (blockIdx.x >> 3)) & 1) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
  } else if (p.prio_mode == 4) {
    if (blockIdx.x < 512 && (((blockIdx.x >> 8) ^ (blockIdx.x >> 3)) & 1)) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
  }
#endif

  int tm, tn;
  {
    const int nblk = p.tiles_m * p.tiles_n;
    const int id = blockIdx.x, xcd = id & 7, j = id >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    tm = v / p.tiles_n;
    tn = v - tm * p.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
#ifdef HIPIE_GEMM_VARIANTS
  if (p.variant == 6 && blockIdx.x < 512) {        // 6: random start skew of the first dispatch round (0 .. ~100 us)
    const int n = (int)((blockIdx.x * 2654435761u) >> 27);       // 0..31
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
  if (p.variant >= 10 && p.variant < 100 && blockIdx.x < 512) {   // 1x: blocks 256..511 sleep (variant - 10) * 5.4 us; 5x: odd (b >> 3)
    const bool sel = p.variant < 50 ? (blockIdx.x >= 256) : (((blockIdx.x >> 3) & 1) != 0);
    const int n = p.variant == 35 || p.variant == 36 ? 5 : (p.variant < 50 ? p.variant - 10 : p.variant - 50);
    if (sel) for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif

  // ---- DMA plan: lane -> (row rl of the 16-row block, chunk position cp); the logical chunk stored there is cp ^ ((rl >> 2) & 3) ----
  unsigned int dvA[4], dvW[NWI];
  {
    const int rl = lane >> 2, c = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int i = 0; i < 4; ++i) dvA[i] = (unsigned int)((long)min(wave * 64 + 16 * i + rl, p.M - 1 - m0) * p.lda_b + 16 * c);
#pragma unroll
    for (int i = 0; i < NWI; ++i) dvW[i] = (unsigned int)((long)min(16 * (4 * i + wave) + rl, p.N - 1 - n0) * p.ldw_b + 16 * c);
  }
  const char* abase = p.A + (long)m0 * p.lda_b;
  const char* wbase = p.W + (long)n0 * p.ldw_b;
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem);
  const bool w_last = 16 * (4 * (NWI - 1) + wave) < BN;           // does this wave issue the last W instruction (BN = 160: waves 0, 1)
  const int nd = 4 + (NWI - 1) + (w_last ? 1 : 0);                // DMA instructions per step of this wave

  auto dma_a = [&](const int i, const int kt, const int slot) {
    gm_dma16(abase + (long)kt * 64, dvA[i], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(slot * STAGE + 1024 * (4 * wave + i))));
  };
  auto dma_w = [&](const int i, const int kt, const int slot) {
    gm_dma16(wbase + (long)kt * 64, dvW[i], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(slot * STAGE + BM * 64 + 1024 * (4 * i + wave))));
  };
  auto dma_step = [&](const int kt, const int slot) {
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_a(i, kt, slot);
#pragma unroll
    for (int i = 0; i < NWI; ++i)
      if (i + 1 < NWI || w_last) dma_w(i, kt, slot);
  };
  auto wait_keep = [&](const bool keep) {       // wait until at most this step's own DMA instructions (issued last) are in flight
    if (!keep) __builtin_amdgcn_s_waitcnt(0x0F70);
    else if (w_last) __builtin_amdgcn_s_waitcnt(0x0F70 | (4 + NWI));
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (4 + NWI - 1));
  };

  // ---- fragment addresses ----
  const int swz = (li >> 2) & 3;
  const char* xrow = smem + (wave * 64 + li) * 64;          // + t * 32 * 64
  const char* wrow = smem + (BM + li) * 64;                 // + j * 32 * 64
  const int ch0 = 16 * ((2 * hi) ^ swz), ch1 = 16 * ((2 * hi + 1) ^ swz);     // hi | lo piece of this lane's k group

  f32x16 acc[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

#ifdef HIPIE_GEMM_VARIANTS
  const int nkt = p.variant == 4 ? 1 : p.nkt;        // 4: the epilogue alone (one k step)
#else
  const int nkt = p.nkt;
#endif
  dma_step(0, 0);
  if (nkt > 1) dma_step(1, 1);
  wait_keep(nkt > 1);
  __syncthreads();

  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 2 < nkt;
    int nslot = slot + 2;
    if (nslot >= 3) nslot -= 3;
    const char* xs = xrow + slot * STAGE;
    const char* ws = wrow + slot * STAGE;
    frag xh[2], xl[2], wa[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      xh[t] = *reinterpret_cast<const frag*>(xs + t * 2048 + ch0);
      xl[t] = *reinterpret_cast<const frag*>(xs + t * 2048 + ch1);
    }
    wa[0][0] = *reinterpret_cast<const frag*>(ws + ch0);
    wa[0][1] = *reinterpret_cast<const frag*>(ws + ch1);
    if (VAR == 2) {
      // fp32 A rows: the two 16-byte pieces hold x0..x3 / x4..x7 of the lane's k group; split them here
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const f32x4 a = __builtin_bit_cast(f32x4, xh[t]), b = __builtin_bit_cast(f32x4, xl[t]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f16_t hh, ll;
          hl_split(a[e], hh, ll);
          xh[t][e] = hh; xl[t][e] = ll;
          hl_split(b[e], hh, ll);
          xh[t][4 + e] = hh; xl[t][4 + e] = ll;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      if (j + 1 < NJ) {
        wa[(j + 1) & 1][0] = *reinterpret_cast<const frag*>(ws + (j + 1) * 2048 + ch0);
        wa[(j + 1) & 1][1] = *reinterpret_cast<const frag*>(ws + (j + 1) * 2048 + ch1);
      }
      const frag wh = wa[j & 1][0], wl = wa[j & 1][1];
      acc[j][0] = Mfma32<f16_t>::mma(wl, xh[0], acc[j][0]);
      acc[j][1] = Mfma32<f16_t>::mma(wl, xh[1], acc[j][1]);
      acc[j][0] = Mfma32<f16_t>::mma(wh, xl[0], acc[j][0]);
      acc[j][1] = Mfma32<f16_t>::mma(wh, xl[1], acc[j][1]);
      acc[j][0] = Mfma32<f16_t>::mma(wh, xh[0], acc[j][0]);
      acc[j][1] = Mfma32<f16_t>::mma(wh, xh[1], acc[j][1]);
#ifdef HIPIE_GEMM_VARIANTS
      if (more && p.variant != 2) {
#else
      if (more) {
#endif
        // the step's DMA instructions ride in its first sub-steps: A rows 2 per sub-step, then the W rows
        if (j == 0) { dma_a(0, kt + 2, nslot); dma_a(1, kt + 2, nslot); }
        if (j == 1) { dma_a(2, kt + 2, nslot); dma_a(3, kt + 2, nslot); }
        if (j == 2) {
#pragma unroll
          for (int i = 0; i < NWI; ++i)
            if (i + 1 < NWI || w_last) dma_w(i, kt + 2, nslot);
        }
      }
    }
#ifdef HIPIE_GEMM_VARIANTS
    if (p.variant != 3)
#endif
    {
      wait_keep(more);
      __syncthreads();
    }
    slot = slot + 1 == 3 ? 0 : slot + 1;
  }

  // ---- epilogue: lane = token, registers = features (as in gemm_kernel; the wave owns all BN features of its 64 tokens) ----
#ifdef HIPIE_GEMM_VARIANTS
  if ((p.variant == 1 || p.variant == 36) && p.alpha != 12345.f) return;
#endif
  const bool has_res = p.resid != nullptr;
  const int act = p.act, ofmt = p.out_fmt;
  const float alpha = p.alpha, osc = p.oscale;
  float* sbias = reinterpret_cast<float*>(smem);
  if (tid < BN) sbias[tid] = (p.bias != nullptr && n0 + tid < p.N) ? p.bias[n0 + tid] : 0.f;
  __syncthreads();
  float4 rq[2][4];
  long orow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int m = m0 + wave * 64 + t * 32 + li;
    orow[t] = (m < p.M) ? (p.out_row != nullptr ? (long)p.out_row[m] : (long)m) : -1;
#ifdef HIPIE_GEMM_VARIANTS
    if ((p.variant == 5 || p.variant == 35) && p.alpha != 12345.f) orow[t] = -1;       // 5 | 35: the epilogue's arithmetic without its stores
#endif
  }
  auto load_res = [&](const int t, const int j, float4 (&dst)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + j * 32 + 8 * g + 4 * hi;
      dst[g] = (has_res && orow[t] >= 0 && n < p.N) ? *reinterpret_cast<const float4*>(p.resid + orow[t] * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_res(0, 0, rq[0]);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const long m = orow[t];
    const bool mok = m >= 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int blk = t * NJ + j;
      if (blk + 1 < 2 * NJ) load_res((blk + 1) / NJ, (blk + 1) % NJ, rq[(blk + 1) & 1]);
      const int nb = n0 + j * 32;
      gm_epi_quads<0, 4>(acc[j][t], rq[blk & 1], sbias + (nb - n0), m, mok, nb, hi, p, has_res);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------------------
// gemm3: PING-PONG form of the split product (round 4) -- one persistent 8-wave workgroup per CU whose two 4-wave GROUPS work on
// different tiles, half a tile apart in time, so that one group's tile epilogue (bias / activation / split / stores: the matrix
// pipe idles through it in gemm_kernel, 24 % of the K = 1280 shapes) runs beside the other group's k loop.
//   Measured basis (tools/bench_gemm_var.py): ONE 4-wave group per CU already saturates the matrix pipe to
// the chip's power-limited rate (gemm2 with one workgroup per CU, epilogue skipped: 1.42 PFLOP/s of MFMA issue against 1.37 with
// two), so nothing is lost while the partner stores; two free-running workgroups per CU (gemm2) fall into lock step and store
// together, and a start-up skew of half a tile between them recovered 12 % -- the groups here keep that skew by construction:
//   * every group owns a three-slot LDS ring of k16 steps (rows of 64 B, chunk c of row r at position c ^ ((r >> 2) & 3), as gemm2)
//     and walks its own tile list (slot 2 b + g of the XCD's contiguous tile range, stride = slots per XCD);
//   * the two groups never synchronise with each other after the start: a hardware s_barrier would couple all 8 waves at every k
//     step (measured: 19 % slower main loop -- both waves of every SIMD then sit in the LDS-read latency behind the barrier at the
//     same time).  Each group has its own SOFTWARE barrier instead -- a monotonic LDS counter: every wave adds 1 after its own
//     vmcnt wait and polls until the count reaches 4 x epoch; while a wave polls, the other group's wave on that SIMD issues MFMAs;
//   * a compute step = one k16 step (30 MFMAs per wave + the LDS-DMA of the step two ahead; the fetch stream runs across tile
//     boundaries: the next tile's first two steps are fetched during the last two of the current one); the epilogue of a tile is
//     free-running per wave (one group barrier publishes the tile's bias values in LDS);
//   * phase: group 1 starts when group 0's counter shows it is half way through its first tile -- until then group 0 has the pipe
//     alone at full rate, so the offset costs nothing.  Two free-running groups keep their phase difference (while one stores,
//     the other runs at twice its shared speed -- symmetric over a period), so the offset holds for the launch.
template <int NJ, int VAR>
__global__ __launch_bounds__(512, 2) void gemm3_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 32 * NJ;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int RING = 3 * STAGE;
  constexpr int NWI = (BN + 63) / 64;
  constexpr int EP = 4 * NJ;                   // epilogue steps per tile: (2 token tiles x NJ feature blocks) x 2 halves
  typedef Mfma32<f16_t>::frag frag;

  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wv = wave & 3;
  const int li = lane & 31, hi = lane >> 5;
  char* gsm = smem + grp * RING;
  float* sbias = reinterpret_cast<float*>(smem + 2 * RING) + grp * BN;
  unsigned int* gcnt = reinterpret_cast<unsigned int*>(smem + 2 * RING + 2 * BN * 4);      // [2] group barrier counters (16 bytes apart)
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)gsm);

  // ---- tile lists: XCD x owns the contiguous tile range [base, base + cnt); its 2 * nbx group slots walk it with stride 2 * nbx ----
  const int nblk = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, S = 2 * (int)(gridDim.x >> 3);
  int base, cnt;
  {
    const int q = nblk >> 3, r = nblk & 7;
    base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    cnt = q + (xcd < r ? 1 : 0);
  }
  const int js = 2 * jb + grp;
  const int nt_a = (2 * jb < cnt) ? (cnt - 2 * jb + S - 1) / S : 0;
  const int nt_b = (2 * jb + 1 < cnt) ? (cnt - 2 * jb - 1 + S - 1) / S : 0;
#ifdef HIPIE_GEMM_VARIANTS
  const int nt = grp ? ((p.variant == 8 || p.variant == 9) ? 0 : nt_b) : nt_a;       // 8 | 9: group 0 alone (half of the tiles)
#else
  const int nt = grp ? nt_b : nt_a;
#endif
  const int nkt = p.nkt;

  // ---- fetch stream (LDS-DMA), two k steps ahead of the compute stream ----
  const int rl = lane >> 2, cch = 16 * ((lane & 3) ^ ((lane >> 4) & 3));
  const bool w_last = 16 * (4 * (NWI - 1) + wv) < BN;
  unsigned int dvA[4], dvW[NWI];
  const char* fa = p.A;
  const char* fw = p.W;
  int f_i = 0, f_k = 0, f_slot = 0;
  auto fetch_tile = [&](const int i) {          // descriptors of this group's tile i
    const int v = base + js + i * S;
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    fa = p.A + (long)m0 * p.lda_b;
    fw = p.W + (long)n0 * p.ldw_b;
#pragma unroll
    for (int k = 0; k < 4; ++k) dvA[k] = (unsigned int)((long)min(wv * 64 + 16 * k + rl, p.M - 1 - m0) * p.lda_b + cch);
#pragma unroll
    for (int k = 0; k < NWI; ++k) dvW[k] = (unsigned int)((long)min(16 * (4 * k + wv) + rl, p.N - 1 - n0) * p.ldw_b + cch);
  };
  auto dma_a = [&](const int k) {
    gm_dma16(fa + (long)f_k * 64, dvA[k], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(f_slot * STAGE + 1024 * (4 * wv + k))));
  };
  auto dma_w = [&](const int k) {
    if (k + 1 < NWI || w_last)
      gm_dma16(fw + (long)f_k * 64, dvW[k], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(f_slot * STAGE + BM * 64 + 1024 * (4 * k + wv))));
  };
  auto fetch_advance = [&]() {
    f_slot = f_slot == 2 ? 0 : f_slot + 1;
    if (++f_k == nkt) {
      f_k = 0;
      if (++f_i < nt) fetch_tile(f_i);
    }
  };
  auto wait_keep = [&](const bool keep) {       // at most the DMA instructions issued in this step stay in flight
    if (!keep) __builtin_amdgcn_s_waitcnt(0x0F70);
    else if (w_last) __builtin_amdgcn_s_waitcnt(0x0F70 | (4 + NWI));
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (4 + NWI - 1));
  };

  const int swz = (li >> 2) & 3;
  const char* xrow = gsm + (wv * 64 + li) * 64;
  const char* wrow = gsm + (BM + li) * 64;
  const int ch0 = 16 * ((2 * hi) ^ swz), ch1 = 16 * ((2 * hi + 1) ^ swz);

  f32x16 acc[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  // ---- prologue: the first two k steps of the group's first tile ----
  if (nt > 0) {
    fetch_tile(0);
#pragma unroll 1
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dma_a(k);
#pragma unroll
      for (int k = 0; k < NWI; ++k) dma_w(k);
      fetch_advance();
    }
  }
  if (tid < 2) gcnt[4 * tid] = 0u;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();                             // the only workgroup barrier: counters zeroed, everybody's first two k steps landed

  // ---- group barrier: monotonic LDS counter, 4 arrivals per epoch ----
  volatile unsigned int* mycnt = gcnt + 4 * grp;
  unsigned int epoch = 0;
  auto gbar = [&]() {
    ++epoch;
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(const_cast<unsigned int*>(mycnt), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const unsigned int want = 4u * epoch;
    while (__builtin_amdgcn_readfirstlane((int)*mycnt) < (int)want) {}
    asm volatile("" ::: "memory");             // the tile reads below stay below the poll
  };
#ifdef HIPIE_GEMM_VARIANTS
  if (grp == 1 && nt_a > 0 && nt > 0 && p.variant != 7) {
#else
  if (grp == 1 && nt_a > 0 && nt > 0) {
#endif
    // start half a tile behind group 0 (which runs alone, at full rate, until then)
    volatile unsigned int* other = gcnt;
    const unsigned int want = 4u * (unsigned int)(nkt / 2);
    while (__builtin_amdgcn_readfirstlane((int)*other) < (int)want) __builtin_amdgcn_s_sleep(8);
  }

  // ---- compute / epilogue ----
  int slot = 0;
  const bool has_res = p.resid != nullptr;
#pragma unroll 1
  for (int ti = 0; ti < nt; ++ti) {
    int m0, n0;
    {
      const int v = base + js + ti * S;
      const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
      m0 = tm * BM;
      n0 = tn * BN;
    }
    // ================= nkt compute steps: one k16 step of the tile each =================
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = f_i < nt;
      const char* xs = xrow + slot * STAGE;
      const char* ws = wrow + slot * STAGE;
      frag xh[2], xl[2], wa[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xh[t] = *reinterpret_cast<const frag*>(xs + t * 2048 + ch0);
        xl[t] = *reinterpret_cast<const frag*>(xs + t * 2048 + ch1);
      }
      wa[0][0] = *reinterpret_cast<const frag*>(ws + ch0);
      wa[0][1] = *reinterpret_cast<const frag*>(ws + ch1);
      if (VAR == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 a = __builtin_bit_cast(f32x4, xh[t]), b = __builtin_bit_cast(f32x4, xl[t]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f16_t hh, ll;
            hl_split(a[e], hh, ll);
            xh[t][e] = hh; xl[t][e] = ll;
            hl_split(b[e], hh, ll);
            xh[t][4 + e] = hh; xl[t][4 + e] = ll;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) {
          wa[(j + 1) & 1][0] = *reinterpret_cast<const frag*>(ws + (j + 1) * 2048 + ch0);
          wa[(j + 1) & 1][1] = *reinterpret_cast<const frag*>(ws + (j + 1) * 2048 + ch1);
        }
        const frag wh = wa[j & 1][0], wl = wa[j & 1][1];
        acc[j][0] = Mfma32<f16_t>::mma(wl, xh[0], acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wl, xh[1], acc[j][1]);
        acc[j][0] = Mfma32<f16_t>::mma(wh, xl[0], acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wh, xl[1], acc[j][1]);
        acc[j][0] = Mfma32<f16_t>::mma(wh, xh[0], acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wh, xh[1], acc[j][1]);
        if (more) {
          if (j == 0) { dma_a(0); dma_a(1); }
          if (j == 1) { dma_a(2); dma_a(3); }
          if (j == 2) {
#pragma unroll
            for (int k = 0; k < NWI; ++k) dma_w(k);
          }
        }
      }
      if (more) fetch_advance();
      slot = slot == 2 ? 0 : slot + 1;
      wait_keep(more);
      gbar();
    }
#ifdef HIPIE_GEMM_VARIANTS
    if ((p.variant == 1 || p.variant == 9) && p.alpha != 12345.f) continue;          // timing: no epilogue
#endif
    // ================= epilogue: EP half blocks (8 values per lane, 2 stores each), free-running per wave =================
    {
      const int tg = tid & 255;
      if (tg < BN) sbias[tg] = (p.bias != nullptr && n0 + tg < p.N) ? p.bias[n0 + tg] : 0.f;
    }
    long orow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = m0 + wv * 64 + t * 32 + li;
      orow[t] = (m < p.M) ? (p.out_row != nullptr ? (long)p.out_row[m] : (long)m) : -1;
#ifdef HIPIE_GEMM_VARIANTS
      if (p.variant == 5 && p.alpha != 12345.f) orow[t] = -1;       // 5: the epilogue's arithmetic without its stores
#endif
    }
    // compact code: ONE copy of the block epilogue, the 8 accumulator values of step e are selected by a switch (the fully unrolled
    // form is ~100 KB of instructions -- measured 150 us per tile of instruction-cache misses, which also evict the other group's k loop)
    float4 rq[2], rn[2];
    auto load_res = [&](const int e, float4 (&dst)[2]) {
      const int blk = e >> 1, t = blk >= NJ ? 1 : 0, j = blk - t * NJ;
      const long m = t ? orow[1] : orow[0];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int n = n0 + j * 32 + 8 * (2 * (e & 1) + g) + 4 * hi;
        dst[g] = (has_res && m >= 0 && n < p.N) ? *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    load_res(0, rq);
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's bias values are in LDS
    gbar();
    if (p.prio_mode == 1) __builtin_amdgcn_s_setprio(3);
    else if (p.prio_mode == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int e = 0; e < EP; ++e) {
      if (e + 1 < EP) load_res(e + 1, rn);
      float x[2][4];
      switch (e) {
#define HIPIE_G3_CASE(J, T, H)                                                                         \
  case 2 * ((T) * NJ + (J)) + (H):                                                                     \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) x[k >> 2][k & 3] = acc[(J) < NJ ? (J) : 0][T][8 * (H) + k]; \
    break;
#define HIPIE_G3_BLOCK(J, T) HIPIE_G3_CASE(J, T, 0) HIPIE_G3_CASE(J, T, 1)
        HIPIE_G3_BLOCK(0, 0) HIPIE_G3_BLOCK(1, 0) HIPIE_G3_BLOCK(2, 0) HIPIE_G3_BLOCK(3, 0)
        HIPIE_G3_BLOCK(0, 1) HIPIE_G3_BLOCK(1, 1) HIPIE_G3_BLOCK(2, 1) HIPIE_G3_BLOCK(3, 1)
        default: {
          // j = 4 (NJ = 5 only): e = 2 * (t * 5 + 4) + h
          const bool t1 = e >= 2 * NJ, h1 = (e & 1) != 0;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            x[k >> 2][k & 3] = t1 ? (h1 ? acc[NJ - 1][1][8 + k] : acc[NJ - 1][1][k]) : (h1 ? acc[NJ - 1][0][8 + k] : acc[NJ - 1][0][k]);
        } break;
#undef HIPIE_G3_BLOCK
#undef HIPIE_G3_CASE
      }
      const int blk = e >> 1, t = blk >= NJ ? 1 : 0, j = blk - t * NJ;
      const long m = t ? orow[1] : orow[0];
      gm_epi_vals<2>(x, 2 * (e & 1), rq, sbias + j * 32, m, m >= 0, n0 + j * 32, hi, p, has_res);
      rq[0] = rn[0];
      rq[1] = rn[1];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
    if (p.prio_mode != 0) __builtin_amdgcn_s_setprio(0);
    // the next tile's bias values overwrite sbias only behind this group's next barrier epochs (nkt compute steps away): every wave
    // has long finished the reads above by then
  }
}

template <int NJ, int VAR>
__global__ __launch_bounds__(256, 2) void gemm4_kernel(const GemmParams p) {
  constexpr int BM = 256, BN = 32 * NJ;
  constexpr int STAGE = (BM + BN) * 64;
  constexpr int RING = 3 * STAGE;
  constexpr int NWI = (BN + 63) / 64;
  constexpr int EP = 4 * NJ;                   // epilogue steps per tile: (2 token tiles x NJ feature blocks) x 2 halves
  typedef Mfma32<f16_t>::frag frag;

  extern __shared__ __attribute__((aligned(128))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = wave;
  const int li = lane & 31, hi = lane >> 5;
  char* gsm = smem;
  float* sbias = reinterpret_cast<float*>(smem + RING);
  unsigned int* sflag = reinterpret_cast<unsigned int*>(smem + RING + BN * 4);
  const unsigned int lds0 = (unsigned int)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)gsm);

  // ---- tile lists: XCD x owns the contiguous tile range [base, base + cnt); its 2 * nbx group slots walk it with stride 2 * nbx ----
  const int nblk = p.tiles_m * p.tiles_n;
  const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3, S = (int)(gridDim.x >> 3);
  int base, cnt;
  {
    const int q = nblk >> 3, r = nblk & 7;
    base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    cnt = q + (xcd < r ? 1 : 0);
  }
  const int js = jb;
  const int nt = (js < cnt) ? (cnt - js + S - 1) / S : 0;
  const int nkt = p.nkt;

  // ---- fetch stream (LDS-DMA), two k steps ahead of the compute stream ----
  const int rl = lane >> 2, cch = 16 * ((lane & 3) ^ ((lane >> 4) & 3));
  const bool w_last = 16 * (4 * (NWI - 1) + wv) < BN;
  unsigned int dvA[4], dvW[NWI];
  const char* fa = p.A;
  const char* fw = p.W;
  int f_i = 0, f_k = 0, f_slot = 0;
  auto fetch_tile = [&](const int i) {          // descriptors of this group's tile i
    const int v = base + js + i * S;
    const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    fa = p.A + (long)m0 * p.lda_b;
    fw = p.W + (long)n0 * p.ldw_b;
#pragma unroll
    for (int k = 0; k < 4; ++k) dvA[k] = (unsigned int)((long)min(wv * 64 + 16 * k + rl, p.M - 1 - m0) * p.lda_b + cch);
#pragma unroll
    for (int k = 0; k < NWI; ++k) dvW[k] = (unsigned int)((long)min(16 * (4 * k + wv) + rl, p.N - 1 - n0) * p.ldw_b + cch);
  };
  auto dma_a = [&](const int k) {
    gm_dma16(fa + (long)f_k * 64, dvA[k], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(f_slot * STAGE + 1024 * (4 * wv + k))));
  };
  auto dma_w = [&](const int k) {
    if (k + 1 < NWI || w_last)
      gm_dma16(fw + (long)f_k * 64, dvW[k], __builtin_amdgcn_readfirstlane(lds0 + (unsigned int)(f_slot * STAGE + BM * 64 + 1024 * (4 * k + wv))));
  };
  auto fetch_advance = [&]() {
    f_slot = f_slot == 2 ? 0 : f_slot + 1;
    if (++f_k == nkt) {
      f_k = 0;
      if (++f_i < nt) fetch_tile(f_i);
    }
  };
  auto wait_keep = [&](const bool keep) {       // at most the DMA instructions issued in this step stay in flight
    if (!keep) __builtin_amdgcn_s_waitcnt(0x0F70);
    else if (w_last) __builtin_amdgcn_s_waitcnt(0x0F70 | (4 + NWI));
    else __builtin_amdgcn_s_waitcnt(0x0F70 | (4 + NWI - 1));
  };

  const int swz = (li >> 2) & 3;
  const char* xrow = gsm + (wv * 64 + li) * 64;
  const char* wrow = gsm + (BM + li) * 64;
  const int ch0 = 16 * ((2 * hi) ^ swz), ch1 = 16 * ((2 * hi + 1) ^ swz);

  f32x16 acc[NJ][2];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  // ---- prologue: the first two k steps of the group's first tile ----
  if (nt > 0) {
    fetch_tile(0);
#pragma unroll 1
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dma_a(k);
#pragma unroll
      for (int k = 0; k < NWI; ++k) dma_w(k);
      fetch_advance();
    }
  }
  // ---- which of the CU's two resident workgroups am I?  arrival order on this CU (HW_ID: xcc, se, cu), counted in a device-global table
  //      that only ever grows (two arrivals per CU and launch keep the parity); the SECOND one starts half a tile (alone-rate) late ----
  if (tid == 0) {
    const unsigned int hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);          // HW_REG_HW_ID[15:0]
    const unsigned int xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);         // HW_REG_XCC_ID[3:0]
    // HW_ID[15:8] = SE_ID (15:13), SH_ID (12), CU_ID (11:8): together with the XCC id a unique CU index
    sflag[0] = atomicAdd(&g_cu_arrivals[(xcc & 7) * 256 + ((hw >> 8) & 0xFF)], 1u) & 1u;
#ifdef HIPIE_GEMM_VARIANTS
    if (p.variant == 11) sflag[0] = blockIdx.x >= gridDim.x / 2 ? 1u : 0u;         // 11: the dispatch-order guess instead of the CU id
    if (p.variant == 12) sflag[0] = (blockIdx.x >> 3) & 1u;
#endif
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();                             // everybody's first two k steps landed; the slot is known
#ifdef HIPIE_GEMM_VARIANTS
  if (sflag[0] == 1u && nt > 0 && p.variant != 7) {
#else
  if (sflag[0] == 1u && nt > 0) {
#endif
    const int n = (nkt * 480 + 4064) / 8128;   // half a tile at the alone rate (nkt x 30 MFMAs x 32 cycles / 2), in s_sleep 127 units
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
  auto gbar = [&]() { __builtin_amdgcn_s_barrier(); };

  // ---- compute / epilogue ----
  int slot = 0;
  const bool has_res = p.resid != nullptr;
#pragma unroll 1
  for (int ti = 0; ti < nt; ++ti) {
    int m0, n0;
    {
      const int v = base + js + ti * S;
      const int tm = v / p.tiles_n, tn = v - tm * p.tiles_n;
      m0 = tm * BM;
      n0 = tn * BN;
    }
    // ================= nkt compute steps: one k16 step of the tile each =================
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
      const bool more = f_i < nt;
      const char* xs = xrow + slot * STAGE;
      const char* ws = wrow + slot * STAGE;
      frag xh[2], xl[2], wa[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        xh[t] = *reinterpret_cast<const frag*>(xs + t * 2048 + ch0);
        xl[t] = *reinterpret_cast<const frag*>(xs + t * 2048 + ch1);
      }
      wa[0][0] = *reinterpret_cast<const frag*>(ws + ch0);
      wa[0][1] = *reinterpret_cast<const frag*>(ws + ch1);
      if (VAR == 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const f32x4 a = __builtin_bit_cast(f32x4, xh[t]), b = __builtin_bit_cast(f32x4, xl[t]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            f16_t hh, ll;
            hl_split(a[e], hh, ll);
            xh[t][e] = hh; xl[t][e] = ll;
            hl_split(b[e], hh, ll);
            xh[t][4 + e] = hh; xl[t][4 + e] = ll;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        if (j + 1 < NJ) {
          wa[(j + 1) & 1][0] = *reinterpret_cast<const frag*>(ws + (j + 1) * 2048 + ch0);
          wa[(j + 1) & 1][1] = *reinterpret_cast<const frag*>(ws + (j + 1) * 2048 + ch1);
        }
        const frag wh = wa[j & 1][0], wl = wa[j & 1][1];
        acc[j][0] = Mfma32<f16_t>::mma(wl, xh[0], acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wl, xh[1], acc[j][1]);
        acc[j][0] = Mfma32<f16_t>::mma(wh, xl[0], acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wh, xl[1], acc[j][1]);
        acc[j][0] = Mfma32<f16_t>::mma(wh, xh[0], acc[j][0]);
        acc[j][1] = Mfma32<f16_t>::mma(wh, xh[1], acc[j][1]);
        if (more) {
          if (j == 0) { dma_a(0); dma_a(1); }
          if (j == 1) { dma_a(2); dma_a(3); }
          if (j == 2) {
#pragma unroll
            for (int k = 0; k < NWI; ++k) dma_w(k);
          }
        }
      }
      if (more) fetch_advance();
      slot = slot == 2 ? 0 : slot + 1;
      wait_keep(more);
      gbar();
    }
#ifdef HIPIE_GEMM_VARIANTS
    if ((p.variant == 1 || p.variant == 9) && p.alpha != 12345.f) continue;          // timing: no epilogue
#endif
    // ================= epilogue: EP half blocks (8 values per lane, 2 stores each), free-running per wave =================
    {
      const int tg = tid & 255;
      if (tg < BN) sbias[tg] = (p.bias != nullptr && n0 + tg < p.N) ? p.bias[n0 + tg] : 0.f;
    }
    long orow[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int m = m0 + wv * 64 + t * 32 + li;
      orow[t] = (m < p.M) ? (p.out_row != nullptr ? (long)p.out_row[m] : (long)m) : -1;
#ifdef HIPIE_GEMM_VARIANTS
      if (p.variant == 5 && p.alpha != 12345.f) orow[t] = -1;       // 5: the epilogue's arithmetic without its stores
#endif
    }
    // compact code: ONE copy of the block epilogue, the 8 accumulator values of step e are selected by a switch (the fully unrolled
    // form is ~100 KB of instructions -- measured 150 us per tile of instruction-cache misses, which also evict the other group's k loop)
    float4 rq[2], rn[2];
    auto load_res = [&](const int e, float4 (&dst)[2]) {
      const int blk = e >> 1, t = blk >= NJ ? 1 : 0, j = blk - t * NJ;
      const long m = t ? orow[1] : orow[0];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int n = n0 + j * 32 + 8 * (2 * (e & 1) + g) + 4 * hi;
        dst[g] = (has_res && m >= 0 && n < p.N) ? *reinterpret_cast<const float4*>(p.resid + m * p.ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    load_res(0, rq);
    __builtin_amdgcn_s_waitcnt(0xC07F);        // lgkmcnt(0): this wave's bias values are in LDS
    gbar();
    if (p.prio_mode == 1) __builtin_amdgcn_s_setprio(3);
    else if (p.prio_mode == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
    for (int e = 0; e < EP; ++e) {
      if (e + 1 < EP) load_res(e + 1, rn);
      float x[2][4];
      switch (e) {
#define HIPIE_G3_CASE(J, T, H)                                                                         \
  case 2 * ((T) * NJ + (J)) + (H):                                                                     \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) x[k >> 2][k & 3] = acc[(J) < NJ ? (J) : 0][T][8 * (H) + k]; \
    break;
#define HIPIE_G3_BLOCK(J, T) HIPIE_G3_CASE(J, T, 0) HIPIE_G3_CASE(J, T, 1)
        HIPIE_G3_BLOCK(0, 0) HIPIE_G3_BLOCK(1, 0) HIPIE_G3_BLOCK(2, 0) HIPIE_G3_BLOCK(3, 0)
        HIPIE_G3_BLOCK(0, 1) HIPIE_G3_BLOCK(1, 1) HIPIE_G3_BLOCK(2, 1) HIPIE_G3_BLOCK(3, 1)
        default: {
          // j = 4 (NJ = 5 only): e = 2 * (t * 5 + 4) + h
          const bool t1 = e >= 2 * NJ, h1 = (e & 1) != 0;
#pragma unroll
          for (int k = 0; k < 8; ++k)
            x[k >> 2][k & 3] = t1 ? (h1 ? acc[NJ - 1][1][8 + k] : acc[NJ - 1][1][k]) : (h1 ? acc[NJ - 1][0][8 + k] : acc[NJ - 1][0][k]);
        } break;
#undef HIPIE_G3_BLOCK
#undef HIPIE_G3_CASE
      }
      const int blk = e >> 1, t = blk >= NJ ? 1 : 0, j = blk - t * NJ;
      const long m = t ? orow[1] : orow[0];
      gm_epi_vals<2>(x, 2 * (e & 1), rq, sbias + j * 32, m, m >= 0, n0 + j * 32, hi, p, has_res);
      rq[0] = rn[0];
      rq[1] = rn[1];
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;
    if (p.prio_mode != 0) __builtin_amdgcn_s_setprio(0);
    // the next tile's bias values overwrite sbias only behind this group's next barrier epochs (nkt compute steps away): every wave
    // has long finished the reads above by then
  }
}



template <int NJ, int VAR>
static int launch_gemm4(GemmParams& p, hipStream_t st) {
  constexpr int BN = 32 * NJ;
  constexpr size_t lds = (size_t)3 * (256 + BN) * 64 + BN * 4 + 16;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nkt = p.K / 16;
  auto kern = gemm4_kernel<NJ, VAR>;
  static bool lds_set[64] = {false};
  static int ncu[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return set_err(HIPIE_EINVAL, "gemm4: device %d", dev);
  if (!lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_set[dev] = true;
  }
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, dev);
    ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  int blocks = ntiles < 2 * ncu[dev] ? ntiles : 2 * ncu[dev];
  blocks = (blocks + 7) & ~7;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, p);
  return check_launch("gemm4");
}

template <int NJ, int VAR>
static int launch_gemm3(GemmParams& p, hipStream_t st) {
  constexpr int BN = 32 * NJ;
  constexpr size_t lds = (size_t)6 * (256 + BN) * 64 + 2 * BN * 4 + 32;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nkt = p.K / 16;
  auto kern = gemm3_kernel<NJ, VAR>;
  static bool lds_set[64] = {false};
  static int ncu[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) return set_err(HIPIE_EINVAL, "gemm3: device %d", dev);
  if (!lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    lds_set[dev] = true;
  }
  if (!ncu[dev]) {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, dev);
    ncu[dev] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  const int ntiles = p.tiles_m * p.tiles_n;
  int blocks = (ntiles + 1) / 2;
  if (blocks > ncu[dev]) blocks = ncu[dev];
  blocks = (blocks + 7) & ~7;                  // whole XCD rounds (a block without tiles only runs the barriers)
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), lds, st, p);
  return check_launch("gemm3");
}

template <int NJ, int VAR>
static int launch_gemm2(GemmParams& p, hipStream_t st, int batches = 1) {
  constexpr int BN = 32 * NJ;
  size_t lds = (size_t)3 * (256 + BN) * 64;
#ifdef HIPIE_GEMM_VARIANTS
  { const char* e = getenv("HIPIE_GEMM2_LDS"); if (e) lds = (size_t)atol(e); }       // > 80 KB: one workgroup per CU (timing experiment)
#endif
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + BN - 1) / BN;
  p.nkt = p.K / 16;
  auto kern = gemm2_kernel<NJ, VAR>;
  static bool lds_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !lds_set[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds > 81920 ? lds : 81920));
    if (dev >= 0 && dev < 64) lds_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.tiles_m * p.tiles_n), (unsigned)batches), dim3(256), lds, st, p);
  return check_launch("gemm2");
}

// HIPIE_GEMM2 = 0 selects the round-3 8-wave kernel for the split product (A/B measurements); HIPIE_GEMM2_PRIO = 0 | 1 | 2
static int gemm2_mode() {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("HIPIE_GEMM2"); mode = e ? atoi(e) : 0; }
  return mode;
}
static int gemm2_prio() {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("HIPIE_GEMM2_PRIO"); mode = e ? atoi(e) : 1; }
  return mode;
}


// Micro-benchmark: per-CU fill rate of K/V tiles from L2 in the access pattern of the ViT attention kernels.
//   layout 0: packed qkv (token stride 3*C*2 = 7680 B, 160 contiguous bytes per (token, head))   -- what the kernels read today
//   layout 1: per-head packed tiles (160 B rows back to back)
// Every workgroup (512 threads) of a (batch, head) streams all 64 tiles (64 keys x 160 B x {K, V}) like the attention loop
// does, 16 workgroups per head, 8 heads per XCD in flight -- the data is L2 resident.  mode 0: global_load_dwordx4 into
// registers (xor-reduced), mode 1: LDS-DMA.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fill.hip -o tools/ubench/fill
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int LAYOUT, int MODE, int NTHR>
__global__ __launch_bounds__(NTHR) void k(const char* base, unsigned* out, int nqt, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int id = blockIdx.x;
  const int bh = (id & 7) + 8 * ((id >> 3) / nqt);
  const int b = bh / 16, h = bh % 16;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  u32x4 acc = {0, 0, 0, 0};
  // tile = 64 keys x 10 chunks x {K, V} = 1280 chunks of 16 B
  for (int it = 0; it < iters; ++it)
    for (int t = 0; t < 64; ++t) {
      for (int c = tid; c < 1280; c += NTHR) {
        const int kv = c / 640, cc = c % 640, row = cc / 10, col = cc % 10;
        const char* src;
        if (LAYOUT == 0) src = base + ((long)b * 4096 + t * 64 + row) * 7680 + (1 + kv) * 2560 + h * 160 + col * 16;
        else src = base + (((long)(bh * 2 + kv) * 4096) + t * 64 + row) * 160 + col * 16;
        if (MODE == 0) acc ^= *reinterpret_cast<const u32x4*>(src);
        else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                              (__attribute__((address_space(3))) void*)(smem + (c / 64) * 1024), 16, 0, 0);
      }
      if (MODE == 1) { __builtin_amdgcn_s_waitcnt(0x0F70); }
    }
  if (acc[0] == 0x12345678u) out[tid] = acc[1] ^ acc[2] ^ acc[3];
}

template <int LAYOUT, int MODE, int NTHR>
static void bench(const char* name, const char* d, unsigned* o) {
  const int nqt = 4096 / (NTHR / 2), iters = 1;
  const int grid = 128 * nqt;
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k<LAYOUT, MODE, NTHR>), dim3(grid), dim3(NTHR), 32768, 0, d, o, nqt, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k<LAYOUT, MODE, NTHR>), dim3(grid), dim3(NTHR), 32768, 0, d, o, nqt, iters);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)grid * 64 * 1280 * 16;
  printf("%-52s %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.0GHz\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.0e9);
}

int main() {
  char* d; unsigned* o;
  (void)hipMalloc(&d, (size_t)8 * 4096 * 7680);
  (void)hipMemset(d, 1, (size_t)8 * 4096 * 7680);
  (void)hipMalloc(&o, 4096);
  bench<0, 0, 256>("qkv layout, loads->regs, 256-thr WGs (2/CU)", d, o);
  bench<1, 0, 256>("packed layout, loads->regs, 256-thr WGs", d, o);
  bench<0, 0, 512>("qkv layout, loads->regs, 512-thr WGs", d, o);
  bench<1, 0, 512>("packed layout, loads->regs, 512-thr WGs", d, o);
  bench<0, 1, 512>("qkv layout, LDS-DMA, 512-thr WGs", d, o);
  bench<1, 1, 512>("packed layout, LDS-DMA, 512-thr WGs", d, o);
  return 0;
}

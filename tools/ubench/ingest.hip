// How many bytes per clock can one CU take in?  Every many-row kernel of this library (skinny2, the tiled GEMM, the split-plane
// convs) tops out near 55-75 GB/s per CU = ~30 B/clk when it streams operands that are L2-resident, against the 64 B/clk a
// CU's vector memory path is specified for (MI355X_MICROARCH.md: L2 ~34.5 TB/s aggregate = 135 GB/s per CU).  This
// microbenchmark separates the hypotheses: waves per CU, loads in flight per wave, where the bytes live (a set every CU
// shares — a filter — or a private set per CU — a window; L2-sized or MALL-sized), load flavour (plain global_load_dwordx4,
// nt, LDS-DMA global_load_lds_dwordx4), and whether anybody consumes LDS / the matrix pipe next to the stream.
//   hipcc --offload-arch=gfx950 -O3 -o ingest ingest.hip && ./ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_dma16(const void* gptr, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_addr) : "memory", "m0");
}

// MODE 0: plain loads, 1: nt loads, 2: LDS-DMA into a per-wave 16 KiB ring (hand-counted waits)
// Each wave walks `bytes_per_wave` of its region in 1 KiB wave-instructions, U in flight, `reps` times.
template <int MODE, int U>
__global__ __launch_bounds__(1024) void ingest_k(const char* __restrict__ base, size_t region_stride, int wg_per_region, size_t bytes_per_wave,
                                                 int reps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // region of this workgroup: shared set (region_stride == 0) or one set per `wg_per_region` workgroups
  const char* reg = base + (size_t)(blockIdx.x / wg_per_region) * region_stride;
  const char* p0 = reg + ((size_t)(blockIdx.x % wg_per_region) * nw + wave) * bytes_per_wave + lane * 16;
  const int n = (int)(bytes_per_wave / 1024);
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int r = 0; r < reps; ++r) {
    if constexpr (MODE == 2) {
      const unsigned slot0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + (unsigned)wave * U * 1024u;
      for (int i = 0; i < n; i += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) lds_dma16(p0 + (size_t)min(i + u, n - 1) * 1024, __builtin_amdgcn_readfirstlane(slot0 + u * 1024u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      acc.x += *reinterpret_cast<unsigned*>(lds + wave * U * 1024 + lane * 4);
    } else {
      for (int i = 0; i < n; i += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const u32x4* q = reinterpret_cast<const u32x4*>(p0 + (size_t)min(i + u, n - 1) * 1024);
          v[u] = MODE == 1 ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
      }
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int MODE, int U>
static int run(const char* name, const char* buf, int wgs, int waves, size_t region_stride, int wg_per_region, size_t bytes_per_wave, int reps,
               unsigned* sink, int cus, double clk_ghz) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const size_t smem = MODE == 2 ? (size_t)waves * U * 1024 : 0;
  if (smem > 160 * 1024) return 0;
  auto k = ingest_k<MODE, U>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), smem, 0, buf, region_stride, wg_per_region, bytes_per_wave, 1, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), smem, 0, buf, region_stride, wg_per_region, bytes_per_wave, reps, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)wgs * waves * bytes_per_wave * reps;
  const double per_cu = bytes / (ms * 1e-3) / cus;
  printf("%-34s wgs %4d x %2d waves, U %2d, %6.0f KiB/wave x %3d reps: %8.3f ms  %7.2f TB/s  %6.1f GB/s per CU  %5.1f B/clk/CU\n", name, wgs, waves, U,
         bytes_per_wave / 1024.0, reps, ms, bytes / (ms * 1e-3) / 1e12, per_cu / 1e9, per_cu / (clk_ghz * 1e9));
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate / 1e6;   // GHz
  printf("%s: %d CUs, %.2f GHz\n", prop.name, cus, clk);
  const size_t total = 1ull << 30;            // 1 GiB arena
  char* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, total));
  CK(hipMemset(buf, 1, total));
  CK(hipMalloc(&sink, 4));
  // A. a SHARED L2-hot set (a filter: 128 KiB read by every workgroup), by waves per CU and loads in flight
  printf("-- shared 128 KiB set (every workgroup reads the same bytes: the conv filter / GEMM weight case), one workgroup per CU\n");
  for (int waves : {4, 8, 16})
    for (int reps : {200}) {
      const size_t bpw = 128 * 1024 / waves;
      run<0, 4>("plain, shared", buf, cus, waves, 0, 1, bpw, reps, sink, cus, clk);
      run<0, 16>("plain, shared", buf, cus, waves, 0, 1, bpw, reps, sink, cus, clk);
    }
  run<0, 8>("plain, shared, 2 WG/CU", buf, 2 * cus, 8, 0, 1, 16 * 1024, 200, sink, cus, clk);
  run<0, 8>("plain, shared, 4 WG/CU", buf, 4 * cus, 4, 0, 1, 32 * 1024, 200, sink, cus, clk);
  run<2, 4>("LDS-DMA, shared", buf, cus, 4, 0, 1, 32 * 1024, 200, sink, cus, clk);
  run<2, 8>("LDS-DMA, shared", buf, cus, 8, 0, 1, 16 * 1024, 200, sink, cus, clk);
  run<2, 8>("LDS-DMA, shared", buf, cus, 16, 0, 1, 8 * 1024, 200, sink, cus, clk);
  // B. a PRIVATE set per CU, L2-sized in total (256 x 64 KiB = 16 MiB) and re-read: per-CU streams that hit L2
  printf("-- private 64 KiB set per workgroup, re-read (16 MiB in total: L2-resident), one workgroup per CU\n");
  for (int waves : {4, 8, 16}) {
    const size_t bpw = 64 * 1024 / waves;
    run<0, 8>("plain, private L2", buf, cus, waves, 64 * 1024, 1, bpw, 400, sink, cus, clk);
    run<1, 8>("nt, private L2", buf, cus, waves, 64 * 1024, 1, bpw, 400, sink, cus, clk);
  }
  run<2, 8>("LDS-DMA, private L2", buf, cus, 8, 64 * 1024, 1, 8 * 1024, 400, sink, cus, clk);
  // C. private 2 MiB per CU (512 MiB in total: beyond L2 and MALL -> HBM stream), read once per rep
  printf("-- private 2 MiB per workgroup (512 MiB in total: HBM stream)\n");
  for (int waves : {8, 16}) {
    const size_t bpw = 2 * 1024 * 1024 / waves;
    run<0, 16>("plain, HBM", buf, cus, waves, 2 * 1024 * 1024, 1, bpw, 4, sink, cus, clk);
    run<1, 16>("nt, HBM", buf, cus, waves, 2 * 1024 * 1024, 1, bpw, 4, sink, cus, clk);
  }
  // D. private 256 KiB per CU (64 MiB in total: MALL-resident, beyond L2)
  printf("-- private 256 KiB per workgroup, re-read (64 MiB in total: beyond the 32 MiB of L2, inside the MALL)\n");
  run<0, 16>("plain, MALL", buf, cus, 8, 256 * 1024, 1, 32 * 1024, 100, sink, cus, clk);
  run<0, 16>("plain, MALL", buf, cus, 16, 256 * 1024, 1, 16 * 1024, 100, sink, cus, clk);
  return 0;
}

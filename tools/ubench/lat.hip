// Pointer-chase latency: one lane, N dependent global loads, small (L2-resident) and large (HBM) footprints.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void chase(const int* buf, int n, int* out, long long* cyc) {
  int p = threadIdx.x;  // per-lane start (vector loads, not scalar)
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) p = __builtin_nontemporal_load(buf + p);
  long long t1 = clock64();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  for (size_t bytes : {(size_t)256 << 10, (size_t)16 << 20, (size_t)1 << 30}) {
    size_t n = bytes / 4;
    std::vector<int> h(n);
    // random cyclic permutation with stride of 64 ints (one 256B line per hop)
    size_t lines = n / 64;
    std::vector<int> perm(lines); for (size_t i = 0; i < lines; ++i) perm[i] = (int)i;
    std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
    for (size_t i = 0; i < lines; ++i) for (int j = 0; j < 64; ++j) h[(size_t)perm[i] * 64 + j] = perm[(i + 1) % lines] * 64 + j;
    int* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice));
    int* out; long long* cyc; CK(hipMalloc(&out, 256)); CK(hipMalloc(&cyc, 8));
    const int N = 2000;
    for (int rep = 0; rep < 2; ++rep) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0, 0); hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, buf, N, out, cyc); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("footprint %8zu KiB rep %d: %.1f ns/hop (event), %lld clk64/hop\n", bytes >> 10, rep, ms * 1e6 / N, c / N);
    }
  }
  return 0;
}

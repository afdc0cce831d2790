// Microbenchmark behind the "persistent per-layer kernel" decision: what does a grid-wide barrier cost on
// MI355X (256 workgroups, 8 XCDs), and how does a chain of GEMV-like phases — stream W_p, combine with a small vector every
// workgroup of the previous phase contributed to, publish a slice of the next vector — run as (a) one kernel per phase
// replayed from a hipGraph (what the decode frame does today) against (b) one persistent kernel whose workgroups request the
// NEXT phase's weights before they wait at the barrier.  Phase sizes are one trunk layer of the bench model (bf16 bytes).
// Every spin is bounded: a lost barrier sets an error flag and the kernel drains instead of hanging the box.
//   hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip && ./gridbar
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256, kThreads = 1024, kMaxLoads = 24;    // 24 x 16 B per thread = 100.7 MB over 256 x 1024 threads
constexpr long kSpinLimit = 20 * 1000 * 1000;

// workgroup barrier that does not drain vmcnt (the compiler puts s_waitcnt vmcnt(0) in front of an s_barrier it can see)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int* err) {
  lds_barrier();
  bool ok = true;
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (++spins > kSpinLimit) { *err = 1; ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  lds_barrier();
  return ok;
}

__global__ __launch_bounds__(1024) void barrier_only_k(unsigned* ctr, int iters, int* err) {
  for (int it = 0; it < iters; ++it)
    if (!grid_barrier(ctr, (unsigned)(it + 1) * gridDim.x, err)) return;
}

// Two-level form: the 256 arrivals on ONE address serialise at the memory side (what the flat form above measures); here the
// workgroups of an XCD (ids congruent mod 8) arrive on their own counter (64-byte apart) and the last of each group bumps the
// global one: 32 + 8 serialised atomics instead of 256.
__device__ __forceinline__ bool grid_barrier2(unsigned* ctr, unsigned epoch, int* err) {
  lds_barrier();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned per = gridDim.x / 8, grp = blockIdx.x & 7;
    const unsigned old = __hip_atomic_fetch_add(ctr + 16 * (1 + grp), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * 8) {
      if (++spins > kSpinLimit) { *err = 1; ok = false; break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  lds_barrier();
  return ok;
}
// Relaxed form: no release / acquire on the counter, i.e. no L2 write-back / invalidate per arrival and per poll — legal only if
// every datum exchanged through the barrier is itself written and read with agent-scope (write-through / L2-bypassing) accesses
// and the arrival is ordered behind those stores by an explicit s_waitcnt vmcnt(0).
__device__ __forceinline__ bool grid_barrier3(unsigned* ctr, unsigned epoch, int* err) {
  lds_barrier();
  bool ok = true;
  if (threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned per = gridDim.x / 8, grp = blockIdx.x & 7;
    const unsigned old = __hip_atomic_fetch_add(ctr + 16 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old + 1 == epoch * per) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * 8) {
      if (++spins > kSpinLimit) { *err = 1; ok = false; break; }
    }
  }
  lds_barrier();
  return ok;
}
__global__ __launch_bounds__(1024) void barrier3_only_k(unsigned* ctr, int iters, int* err) {
  for (int it = 0; it < iters; ++it)
    if (!grid_barrier3(ctr, (unsigned)(it + 1), err)) return;
}
__global__ __launch_bounds__(1024) void barrier2_only_k(unsigned* ctr, int iters, int* err) {
  for (int it = 0; it < iters; ++it)
    if (!grid_barrier2(ctr, (unsigned)(it + 1), err)) return;
}

// One phase: this thread's `nl` 16-byte pieces of W_p (workgroup-contiguous, non-temporal), a 12 KiB vector read with
// agent-scope loads (it was written by other workgroups, possibly on other XCDs), a reduction, one published float.
__device__ __forceinline__ void request(u32x4 (&w)[kMaxLoads], const u32x4* base, int nl) {
#pragma unroll
  for (int i = 0; i < kMaxLoads; ++i)
    if (i < nl) w[i] = __builtin_nontemporal_load(base + (size_t)i * kThreads);
}
__device__ __forceinline__ float consume(const u32x4 (&w)[kMaxLoads], int nl, const float* vec_in, float* red) {
  const float xv = __hip_atomic_load(vec_in + (threadIdx.x & 255) * 12, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned acc = 0;
#pragma unroll
  for (int i = 0; i < kMaxLoads; ++i)
    if (i < nl) acc += w[i][0] ^ w[i][1] ^ w[i][2] ^ w[i][3];
  float v = xv * 1e-3f + (float)(acc & 0xff) * 1e-6f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  lds_barrier();
  float s = 0.f;
  if (threadIdx.x == 0) for (int wv = 0; wv < kThreads / 64; ++wv) s += red[wv];
  lds_barrier();
  return s;
}

__global__ __launch_bounds__(kThreads) void phase_k(const u32x4* w, int nl, const float* vec_in, float* vec_out) {
  __shared__ float red[kThreads / 64];
  u32x4 wr[kMaxLoads];
  request(wr, w + (size_t)blockIdx.x * nl * kThreads + threadIdx.x, nl);
  const float s = consume(wr, nl, vec_in, red);
  if (threadIdx.x == 0) {
    for (int j = 0; j < 12; ++j) vec_out[blockIdx.x * 12 + j] = s + j;
  }
}

struct Chain { const u32x4* w[8]; int nl[8]; int n; };

// the same chain of phases, `reps` times, in one launch; weights of phase p+1 requested before the barrier that ends phase p
template <bool PREFETCH>
__global__ __launch_bounds__(kThreads) void persistent_k(Chain c, int reps, float* vec_a, float* vec_b, unsigned* ctr, unsigned epoch0, int* err) {
  __shared__ float red[kThreads / 64];
  u32x4 wr[kMaxLoads];
  const int total = reps * c.n;
  if (PREFETCH) request(wr, c.w[0] + (size_t)blockIdx.x * c.nl[0] * kThreads + threadIdx.x, c.nl[0]);
  for (int it = 0; it < total; ++it) {
    const int p = it % c.n, pn = (it + 1) % c.n;
    const float* vin = (it & 1) ? vec_b : vec_a;
    float* vout = (it & 1) ? vec_a : vec_b;
    if (!PREFETCH) request(wr, c.w[p] + (size_t)blockIdx.x * c.nl[p] * kThreads + threadIdx.x, c.nl[p]);
    const float s = consume(wr, c.nl[p], vin, red);
    // wave 0 carries the barrier: its release-atomic waits for everything the wave has in flight, so its own share of the
    // next weights goes out after the barrier; the other 15 waves request theirs before it
    const bool pre = PREFETCH && it + 1 < total;
    if (pre && threadIdx.x >= 64) request(wr, c.w[pn] + (size_t)blockIdx.x * c.nl[pn] * kThreads + threadIdx.x, c.nl[pn]);
    if (threadIdx.x == 0)
      for (int j = 0; j < 12; ++j) __hip_atomic_store(vout + blockIdx.x * 12 + j, s + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_barrier3(ctr, epoch0 + (unsigned)(it + 1), err)) return;
    if (pre && threadIdx.x < 64) request(wr, c.w[pn] + (size_t)blockIdx.x * c.nl[pn] * kThreads + threadIdx.x, c.nl[pn]);
  }
}

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* ctr; int* err; CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&err, 64));
  float *va, *vb; CK(hipMalloc(&va, kWG * 12 * 4)); CK(hipMalloc(&vb, kWG * 12 * 4));
  CK(hipMemset(va, 0, kWG * 12 * 4)); CK(hipMemset(vb, 0, kWG * 12 * 4)); CK(hipMemset(err, 0, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  int herr = 0;

  // 1. barrier alone
  for (int threads : {256, 1024}) {
    const int iters = 2000;
    CK(hipMemsetAsync(ctr, 0, 64, s));
    hipLaunchKernelGGL(barrier_only_k, dim3(kWG), dim3(threads), 0, s, ctr, 10, err);     // warm
    CK(hipMemsetAsync(ctr, 0, 64, s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(barrier_only_k, dim3(kWG), dim3(threads), 0, s, ctr, iters, err);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("grid barrier, %d workgroups x %4d threads : %.2f us per barrier%s\n", kWG, threads, ms * 1000.f / iters, herr ? "  (TIMED OUT)" : "");
    CK(hipMemsetAsync(ctr, 0, 4096, s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(barrier2_only_k, dim3(kWG), dim3(threads), 0, s, ctr, iters, err);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("two-level barrier, %d x %4d threads         : %.2f us per barrier%s\n", kWG, threads, ms * 1000.f / iters, herr ? "  (TIMED OUT)" : "");
    CK(hipMemsetAsync(ctr, 0, 4096, s));
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(barrier3_only_k, dim3(kWG), dim3(threads), 0, s, ctr, iters, err);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("two-level, relaxed atomics, %d x %4d threads : %.2f us per barrier%s\n", kWG, threads, ms * 1000.f / iters, herr ? "  (TIMED OUT)" : "");
  }

  // 2. one trunk layer's weight streams: qkv 31.5 MB, o 18.9 MB, swiglu 100.7 MB, down 50.3 MB -> 16-byte pieces per thread
  const int nl[4] = {8, 5, 24, 12};   // x 256 workgroups x 1024 threads x 16 B = 33.6, 21.0, 100.7, 50.3 MB
  const int layers = 12;              // distinct weight sets so that every phase streams cold bytes
  size_t per_layer = 0;
  for (int p = 0; p < 4; ++p) per_layer += (size_t)nl[p] * kWG * kThreads * 16;
  u32x4* wall; CK(hipMalloc(&wall, per_layer * layers)); CK(hipMemset(wall, 1, per_layer * layers));
  // (a) one kernel per phase, graph replay
  {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    int it = 0;
    for (int l = 0; l < layers; ++l) {
      size_t off = (size_t)l * per_layer / 16;
      for (int p = 0; p < 4; ++p, ++it) {
        hipLaunchKernelGGL(phase_k, dim3(kWG), dim3(kThreads), 0, s, wall + off, nl[p], (it & 1) ? vb : va, (it & 1) ? va : vb);
        off += (size_t)nl[p] * kWG * kThreads;
      }
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("(a) kernel per phase, hipGraph              : %.2f us per layer (4 phases, %.1f MB) = %.2f TB/s\n", ms * 1000.f / (5 * layers),
           per_layer / 1e6, per_layer / (ms * 1e-3 / (5 * layers)) / 1e12);
  }
  // (b) persistent: the layer chain is walked `layers` times over the SAME 4 weight sets of layer l = it / 4 is not expressible
  // with a fixed Chain, so the persistent kernel streams layer 0's sets repeatedly interleaved with the others through reps=1
  // launches per layer group: use a chain of 8 phases (two layers) and alternate groups to keep the bytes cold.
  for (int pre = 0; pre < 2; ++pre) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemsetAsync(ctr, 0, 4096, s));
      CK(hipEventRecord(e0, s));
      unsigned base = 0;                                 // the counter runs on across launches: no reset between them
      for (int l = 0; l < layers; l += 2) {
        Chain c2; c2.n = 8;
        size_t off = (size_t)l * per_layer / 16;
        for (int q = 0; q < 8; ++q) { c2.w[q] = wall + off; c2.nl[q] = nl[q & 3]; off += (size_t)nl[q & 3] * kWG * kThreads; }
        if (pre) hipLaunchKernelGGL(persistent_k<true>, dim3(kWG), dim3(kThreads), 0, s, c2, 1, va, vb, ctr, base, err);
        else hipLaunchKernelGGL(persistent_k<false>, dim3(kWG), dim3(kThreads), 0, s, c2, 1, va, vb, ctr, base, err);
        base += 8;
      }
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("(b) persistent, 8 phases per launch, %s : %.2f us per layer = %.2f TB/s%s\n", pre ? "next weights before the barrier" : "weights after the barrier     ",
           best * 1000.f / layers, per_layer / (best * 1e-3 / layers) / 1e12, herr ? "  (BARRIER TIMED OUT)" : "");
  }
  return 0;
}

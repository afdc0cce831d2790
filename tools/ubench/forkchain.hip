// Microbenchmark for VERDICT r2 item 5(b): can the per-launch fixed cost of the B = 1 frame (a launch costs ~3.5 us + bytes /
// 7.5 TB/s; 324 dependent launches per frame) be hidden by letting kernel N+1 START before kernel N has finished — issue its
// weight burst, then wait on a device flag that kernel N's workgroups raise after publishing their outputs — instead of
// waiting at the kernel boundary?  The chain is one trunk layer's four weight streams (33.6 / 21.0 / 100.7 / 50.3 MB of bf16),
// 12 distinct weight sets so that every phase streams cold bytes, exactly as tools/ubench/gridbar.hip.
//   (a)  one kernel per phase on one stream, captured into a hipGraph            (what the frame does today)
//   (a') the same with the flag protocol added (publish with agent-scope stores, s_waitcnt vmcnt(0), relaxed arrival;
//        wait = one lane polling)                                                 (what the protocol itself costs)
//   (c)  phases alternate between two captured streams: kernel N+1 depends on kernel N-1 only (stream order) and on kernel N
//        through the flag, so its workgroups take the CU slots kernel N's workgroups free and start their burst early.
// Geometry: 256 workgroups x 512 threads, two resident per CU: the device holds TWO whole kernels at once, so a waiting kernel
// can never keep its producer's workgroups off the CUs whatever order the two queues are served in.  (With grids that fill
// the device — 512 x 512 — the first run deadlocked until the bounded waits gave up: kernel N+1 and kernel N become eligible
// at the same moment at the head of the graph, N+1's waiters took slots N still needed.)  Every spin is bounded; a timeout
// raises an error flag and the kernel drains.
//   hipcc --offload-arch=gfx950 -O3 -o forkchain forkchain.hip && ./forkchain
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kWG = 256, kThreads = 512, kMaxLoads = 24;    // 24 x 16 B in flight per thread (96 VGPRs: two workgroups per CU); larger phases take two rounds
constexpr long kSpinLimit = 20 * 1000;      // ~20-40 ms: a lost wait fails fast

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// FLAG = 0: plain phase.  FLAG = 1: wait for `prev` (all kWG arrivals) after the burst is out, raise `mine` after publishing.
template <int FLAG>
__global__ __launch_bounds__(kThreads, 4) void phase_k(const u32x4* w, int nl, const float* vec_in, float* vec_out, unsigned* prev, unsigned* mine, int* err) {
  __shared__ float red[kThreads / 64];
  u32x4 wr[kMaxLoads];
  unsigned carry = 0;
  const u32x4* wbase = w + (size_t)blockIdx.x * nl * kThreads + threadIdx.x;
  while (nl > kMaxLoads) {                    // all but the last round: stream and fold (no dependence on the previous kernel)
    request(wr, wbase, kMaxLoads);
#pragma unroll
    for (int i = 0; i < kMaxLoads; ++i) carry += wr[i][0] ^ wr[i][3];
    wbase += (size_t)kMaxLoads * kThreads;
    nl -= kMaxLoads;
  }
  request(wr, wbase, nl);
  wr[0][0] ^= carry;
  if (FLAG && prev) {
    if (threadIdx.x == 64) {                 // a lane of a wave whose own loads are few: its poll loads queue behind them only
      long spins = 0;
      while (__hip_atomic_load(prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kWG) {
        if (++spins > kSpinLimit) { *err = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    lds_barrier();
  }
  const float s = consume(wr, nl, vec_in, red);
  if (threadIdx.x == 0) {
    const int per = 3072 / kWG;
    for (int j = 0; j < per; ++j) __hip_atomic_store(vec_out + blockIdx.x * per + j, s + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (FLAG) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main() {
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const int nl[4] = {16, 10, 48, 24};   // x 256 workgroups x 512 threads x 16 B = 33.6, 21.0, 100.7, 50.3 MB
  const int layers = 12, nk = layers * 4;
  size_t per_layer = 0;
  for (int p = 0; p < 4; ++p) per_layer += (size_t)nl[p] * kWG * kThreads * 16;
  u32x4* wall; CK(hipMalloc(&wall, per_layer * layers)); CK(hipMemset(wall, 1, per_layer * layers));
  float *va, *vb; CK(hipMalloc(&va, 3072 * 4)); CK(hipMalloc(&vb, 3072 * 4)); CK(hipMemset(va, 0, 3072 * 4)); CK(hipMemset(vb, 0, 3072 * 4));
  unsigned* ctr; int* err; CK(hipMalloc(&ctr, (nk + 1) * 64)); CK(hipMalloc(&err, 64)); CK(hipMemset(err, 0, 64));
  hipEvent_t e0, e1, ef, ej; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  float ms; int herr = 0;

  for (int mode = 0; mode < 3; ++mode) {     // 0 = (a), 1 = (a'), 2 = (c)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    CK(hipMemsetAsync(ctr, 0, (nk + 1) * 64, s0));
    if (mode == 2) { CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0)); }
    int it = 0;
    for (int l = 0; l < layers; ++l) {
      size_t off = (size_t)l * per_layer / 16;
      for (int p = 0; p < 4; ++p, ++it) {
        hipStream_t st = (mode == 2 && (it & 1)) ? s1 : s0;
        unsigned* prev = it ? ctr + 16 * (it - 1) : nullptr;
        unsigned* mine = ctr + 16 * it;
        const float* vin = (it & 1) ? vb : va;
        float* vout = (it & 1) ? va : vb;
        if (mode == 0) hipLaunchKernelGGL(phase_k<0>, dim3(kWG), dim3(kThreads), 0, st, wall + off, nl[p], vin, vout, prev, mine, err);
        else hipLaunchKernelGGL(phase_k<1>, dim3(kWG), dim3(kThreads), 0, st, wall + off, nl[p], vin, vout, prev, mine, err);
        off += (size_t)nl[p] * kWG * kThreads;
      }
    }
    if (mode == 2) { CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0)); }
    CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s0)); CK(hipStreamSynchronize(s0));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipEventRecord(e0, s0));
      for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s0));
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    const char* name[3] = {"(a)  kernel per phase, one stream, hipGraph           ", "(a') + flag protocol (publish / arrive / poll), one stream",
                           "(c)  two captured streams, kernel N+1 starts under N   "};
    printf("%s : %.2f us per layer (4 phases, %.1f MB) = %.2f TB/s%s\n", name[mode], best * 1000.f / (3 * layers), per_layer / 1e6,
           per_layer / (best * 1e-3 / (3 * layers)) / 1e12, herr ? "  (A WAIT TIMED OUT)" : "");
    fflush(stdout);
    CK(hipMemset(err, 0, 64));
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
  }
  // (d) the same two-stream chain WITHOUT a graph: plain launches on two HIP streams (two hardware queues)
  {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemsetAsync(ctr, 0, (nk + 1) * 64, s0));
      CK(hipEventRecord(ef, s0)); CK(hipStreamWaitEvent(s1, ef, 0));
      CK(hipEventRecord(e0, s0));
      int it = 0;
      for (int l = 0; l < layers; ++l) {
        size_t off = (size_t)l * per_layer / 16;
        for (int p = 0; p < 4; ++p, ++it) {
          hipStream_t st = (it & 1) ? s1 : s0;
          hipLaunchKernelGGL(phase_k<1>, dim3(kWG), dim3(kThreads), 0, st, wall + off, nl[p], (it & 1) ? vb : va, (it & 1) ? va : vb,
                             it ? ctr + 16 * (it - 1) : nullptr, ctr + 16 * it, err);
          off += (size_t)nl[p] * kWG * kThreads;
        }
      }
      CK(hipEventRecord(ej, s1)); CK(hipStreamWaitEvent(s0, ej, 0));
      CK(hipEventRecord(e1, s0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("(d)  two plain streams (no graph), kernel N+1 starts under N : %.2f us per layer = %.2f TB/s%s\n", best * 1000.f / layers,
           per_layer / (best * 1e-3 / layers) / 1e12, herr ? "  (A WAIT TIMED OUT)" : "");
  }
  return 0;
}

// Latency probe for ua2_attn (single-pass mode) at the decode shapes, graph replay.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/ua2hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  struct Cfg { int nh, nkv, hs, pos, maxp; } cfgs[] = {{32, 8, 64, 7, 1}, {24, 8, 128, 100, 32}, {24, 8, 128, 530, 32}};
  for (auto c : cfgs) {
    const int L = 40;  // layers (distinct pools so data is cold-ish)
    size_t pool = (size_t)c.maxp * c.nkv * 64 * c.hs * 2;
    std::vector<void*> kp(L), vp(L);
    for (int l = 0; l < L; ++l) { CK(hipMalloc(&kp[l], pool)); CK(hipMalloc(&vp[l], pool)); CK(hipMemset(kp[l], 0, pool)); CK(hipMemset(vp[l], 0, pool)); }
    float *q, *y; CK(hipMalloc(&q, c.nh * c.hs * 4)); CK(hipMalloc(&y, c.nh * c.hs * 4)); CK(hipMemset(q, 0, c.nh * c.hs * 4));
    int *pos, *pt; CK(hipMalloc(&pos, 4)); CK(hipMalloc(&pt, c.maxp * 4));
    std::vector<int> hpt(c.maxp); for (int i = 0; i < c.maxp; ++i) hpt[i] = i;
    CK(hipMemcpy(pt, hpt.data(), c.maxp * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(pos, &c.pos, 4, hipMemcpyHostToDevice));
    for (int dbg = 0; dbg < 4; ++dbg) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int l = 0; l < L; ++l) {
        ua2_attn_args a; memset(&a, 0, sizeof(a));
        a.dtype = UA2_BF16; a.R = 1; a.q = q; a.row_pos = pos; a.row_seq = nullptr; a.y = y; a.grid_pages = dbg;
        a.kv.k_pool = kp[l]; a.kv.v_pool = vp[l]; a.kv.page_table = pt; a.kv.max_pages = c.maxp; a.kv.n_kv = c.nkv; a.kv.n_head = c.nh; a.kv.head_size = c.hs;
        if (ua2_attn(&a, s)) { printf("err %s\n", ua2_last_error()); return 1; }
      }
      CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipGraphLaunch(ge, s); hipStreamSynchronize(s);
      hipEventRecord(e0, s); for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("hs=%d pos=%d dbg=%d : %.2f us/launch\n", c.hs, c.pos, dbg, ms * 1000 / (10 * L));
    }
  }
  return 0;
}

#!/usr/bin/env python3
"""One depth-decoder pass at B = 1 (4 layers x (qkv, o, SwiGLU, down) + audio head: 17 weight-streaming ops, 536 MB bf16) two ways:
  chain  — 17 ua2_linear launches of the production scaled plan (what ua2_stage3.hip's run_gpt issues for the decoder at one row;
           attention stand-in: o-proj consumes bf16(q)), back to back on one stream, and replayed as one HIP graph;
  engine — tools/ubench/engine.hip: ONE persistent launch, LDS-DMA loader + MFMA consumer + gather waves per CU, granule hand-offs.
Checks every op's fp32 output of the engine against the chain BIT FOR BIT, then times both.
Build first: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include -I uniaudio2_amd/csrc tools/ubench/engine.hip -o tools/ubench/libengine.so
Usage: python tools/ubench/engine_run.py [--layers 4] [--iters 50] [--flags 0] [--no-head]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from uniaudio2_amd import ops
from uniaudio2_amd._lib import EPI_RESIDUAL, EPI_STORE, EPI_SWIGLU, PRO_CAST

PRO_SCALED = 4
ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--no-head", action="store_true")
ap.add_argument("--timeout-ms", type=int, default=50)
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--evt-op", type=int, default=2)
ap.add_argument("--first-only", action="store_true")
opt = ap.parse_args()

dev, dt = torch.device("cuda"), torch.bfloat16
torch.manual_seed(0)
C_, QN, NQKV, INTER, VA, EPS = 2048, 2048, 3072, 8192, 12296, 1e-5
L = opt.layers


class EngOp(C.Structure):
    _fields_ = [("w0", C.c_void_p), ("w1", C.c_void_p), ("nw", C.c_void_p), ("gin", C.c_void_p), ("gssq_in", C.c_void_p),
                ("gout", C.c_void_p), ("gssq_out", C.c_void_p), ("y", C.c_void_p), ("N", C.c_int), ("K", C.c_int), ("ranges", C.c_int),
                ("kind", C.c_int), ("xsel", C.c_int), ("pub_n", C.c_int), ("eps", C.c_float), ("pad", C.c_int)]


class EngArgs(C.Structure):
    _fields_ = [("op", EngOp * 20), ("x0", C.c_void_p), ("err", C.c_void_p), ("stamps", C.c_void_p), ("nops", C.c_int), ("ncu", C.c_int),
                ("timeout_ticks", C.c_int), ("flags", C.c_int), ("evt_op", C.c_int), ("pad", C.c_int)]


eng = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("ENG_LIB", "libengine.so")))
eng.eng_launch.argtypes = [C.POINTER(EngArgs), C.c_void_p, C.c_size_t, C.c_void_p]
eng.eng_timed.argtypes = [C.POINTER(EngArgs), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.POINTER(C.c_float)]


def rand_w(n, k):
    return (torch.randn(n, k, device=dev) * 0.02).to(dt)


def eng_pack(w):
    """[N, K] bf16 -> [N/8 units][K/32][4 (k group)][8 (column)][8 bf16]: a unit's bytes contiguous, half-chunks of 512 B"""
    n, k = w.shape
    return w.view(n // 8, 8, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous()


p = lambda t: t.data_ptr() if t is not None else None

# ---- weights
layers = []
for l in range(L):
    d = dict(qkv=rand_w(NQKV, C_), o=rand_w(C_, QN), g=rand_w(INTER, C_), u=rand_w(INTER, C_), d=rand_w(C_, INTER),
             n1=(1.0 + 0.1 * torch.randn(C_, device=dev)), n2=(1.0 + 0.1 * torch.randn(C_, device=dev)))
    layers.append(d)
lnf = 1.0 + 0.1 * torch.randn(C_, device=dev)
head = rand_w(VA, C_)
x0 = torch.randn(1, C_, device=dev)
prod = [{k: ops.pack_linear(v.float(), dt) for k, v in d.items() if k in ("qkv", "o", "g", "u", "d")} for d in layers]
prod_head = ops.pack_linear(head.float(), dt)
epk = [{k: eng_pack(v) for k, v in d.items() if k in ("qkv", "o", "g", "u", "d")} for d in layers]
epk_head = eng_pack(head)
wbytes = sum(v.numel() * 2 for d in epk for v in d.values()) + (0 if opt.no_head else epk_head.numel() * 2)

# ---- the chain (production kernels)
xh = [torch.empty(1, C_, dtype=dt, device=dev) for _ in range(2 * L + 1)]          # hand-over rows: [2l] -> qkv, [2l+1] -> SwiGLU
ssq = [torch.empty(1, C_ // 16, device=dev) for _ in range(2 * L + 1)]
xs = [torch.empty(1, C_, device=dev) for _ in range(2 * L + 1)]                    # residual stream after each RESIDUAL op
xs[0].copy_(x0)
qkv_c = [torch.empty(1, NQKV, device=dev) for _ in range(L)]
act_c = [torch.empty(1, INTER, device=dev) for _ in range(L)]
logits_c = torch.empty(1, VA, device=dev)
# entry hand-over by a production producer: out = 0 * W + x0
zw = ops.pack_linear(torch.zeros(C_, 32, device=dev), dt)
tmp = torch.empty(1, C_, device=dev)
ops.linear(dtype=dt, M=1, N=C_, K=32, w0=zw, prologue=PRO_CAST, x=torch.zeros(1, 32, device=dev), epilogue=EPI_RESIDUAL, resid=x0, y=tmp,
           y_norm_w=layers[0]["n1"], y_h=xh[0], y_ssq=ssq[0])
torch.cuda.synchronize()
assert torch.equal(tmp, x0)
chain = []
for l in range(L):
    w, d = prod[l], layers[l]
    nxt = layers[l + 1]["n1"] if l + 1 < L else lnf
    chain.append(ops.linear(dtype=dt, M=1, N=NQKV, K=C_, w0=w["qkv"], prologue=PRO_SCALED, x_h=xh[2 * l], x_ssq=ssq[2 * l], eps=EPS,
                            epilogue=EPI_STORE, y=qkv_c[l], launch=False))
    chain.append(ops.linear(dtype=dt, M=1, N=C_, K=QN, w0=w["o"], prologue=PRO_CAST, x=qkv_c[l], ldx=NQKV, epilogue=EPI_RESIDUAL,
                            resid=xs[2 * l], y=xs[2 * l + 1], y_norm_w=d["n2"], y_h=xh[2 * l + 1], y_ssq=ssq[2 * l + 1], launch=False))
    chain.append(ops.linear(dtype=dt, M=1, N=INTER, K=C_, w0=w["g"], w1=w["u"], prologue=PRO_SCALED, x_h=xh[2 * l + 1],
                            x_ssq=ssq[2 * l + 1], eps=EPS, epilogue=EPI_SWIGLU, y=act_c[l], launch=False))
    chain.append(ops.linear(dtype=dt, M=1, N=C_, K=INTER, w0=w["d"], prologue=PRO_CAST, x=act_c[l], epilogue=EPI_RESIDUAL,
                            resid=xs[2 * l + 1], y=xs[2 * l + 2], y_norm_w=nxt, y_h=xh[2 * l + 2], y_ssq=ssq[2 * l + 2], launch=False))
if not opt.no_head:
    chain.append(ops.linear(dtype=dt, M=1, N=VA, K=C_, w0=prod_head, prologue=PRO_SCALED, x_h=xh[2 * L], x_ssq=ssq[2 * L], eps=EPS,
                            epilogue=EPI_STORE, y=logits_c, launch=False))
ops.linear_chain_timed(chain, 1)
torch.cuda.synchronize()

# ---- the engine
G = torch.zeros(L * (1024 + 1024 + 256 + 4096 + 1024 + 256), dtype=torch.int64, device=dev)      # zeroed in front of every launch
cur = [0]


def carve(n):
    t = G[cur[0]:cur[0] + n]
    cur[0] += n
    return t


tag = 1 << 32
g_h0 = (xh[0].view(torch.int16).view(-1).to(torch.int64) & 0xFFFF)
g_h0 = (g_h0[0::2] | (g_h0[1::2] << 16) | tag).contiguous()                                     # entry edge, written once by the host
s0 = torch.zeros(256, device=dev)
s0[0::2] = ssq[0].view(-1)                                                                       # lo = the tile's partial, hi = 0: lo + hi is exact
g_s0 = ((s0.view(torch.int32).to(torch.int64) & 0xFFFFFFFF) | tag).contiguous()
qkv_e = [torch.zeros(NQKV, device=dev) for _ in range(L)]
act_e = [torch.zeros(INTER, device=dev) for _ in range(L)]
xs_e = [torch.zeros(C_, device=dev) for _ in range(2 * L + 1)]
logits_e = torch.zeros(VA, device=dev)
err = torch.zeros(16, dtype=torch.int32, device=dev)
a = EngArgs()
keep = []
n = 0
gh, gs = g_h0, g_s0
for l in range(L):
    e, d = epk[l], layers[l]
    nxt = layers[l + 1]["n1"] if l + 1 < L else lnf
    gq, gh2, gs2, gact, ghn, gsn = carve(1024), carve(1024), carve(256), carve(4096), carve(1024), carve(256)
    keep += [gq, gh2, gs2, gact, ghn, gsn]
    for kw in (dict(w0=p(e["qkv"]), gin=p(gh), gssq_in=p(gs), gout=p(gq), y=p(qkv_e[l]), N=NQKV, K=C_, ranges=16, kind=0, xsel=0, pub_n=QN),
               dict(w0=p(e["o"]), nw=p(d["n2"]), gin=p(gq), gout=p(gh2), gssq_out=p(gs2), y=p(xs_e[2 * l + 1]), N=C_, K=QN, ranges=16, kind=1, xsel=1),
               dict(w0=p(e["g"]), w1=p(e["u"]), gin=p(gh2), gssq_in=p(gs2), gout=p(gact), y=p(act_e[l]), N=INTER, K=C_, ranges=8, kind=2, xsel=0),
               dict(w0=p(e["d"]), nw=p(nxt), gin=p(gact), gout=p(ghn), gssq_out=p(gsn), y=p(xs_e[2 * l + 2]), N=C_, K=INTER, ranges=16, kind=1, xsel=2)):
        for k, v in kw.items():
            setattr(a.op[n], k, v)
        a.op[n].eps = EPS
        n += 1
    gh, gs = ghn, gsn
if not opt.no_head:
    for k, v in dict(w0=p(epk_head), gin=p(gh), gssq_in=p(gs), y=p(logits_e), N=VA, K=C_, ranges=8, kind=0, xsel=0, pub_n=0).items():
        setattr(a.op[n], k, v)
    a.op[n].eps = EPS
    n += 1
a.x0, a.err, a.nops, a.ncu, a.timeout_ticks, a.flags = p(x0), p(err), n, 256, opt.timeout_ms * 100000, opt.flags
stream = ops.stream()
rc = eng.eng_launch(C.byref(a), G.data_ptr(), G.numel() * 8, stream)
torch.cuda.synchronize()
code = int(err[0].item())
print(f"engine: launch rc {rc}, give-up code {code:#x}, {n} ops, {wbytes / 1e6:.1f} MB of weights, LDS {eng.eng_lds_bytes()} B", flush=True)
if opt.first_only:
    sys.exit(0)
if (rc or code) and not (opt.flags & (64 | 128 | 256)):
    sys.exit(1)
if opt.flags & (48 | 64 | 128 | 256):
    ms = C.c_float(0.0)
    eng.eng_timed(C.byref(a), G.data_ptr(), G.numel() * 8, 5, stream, C.byref(ms))
    eng.eng_timed(C.byref(a), G.data_ptr(), G.numel() * 8, opt.iters, stream, C.byref(ms))
    print(f"knock-out flags {opt.flags}: engine {ms.value * 1e3:.1f} us (wrong results by construction)")
    sys.exit(0)


def cmp(name, e, c):
    e, c = e.view(-1), c.view(-1)
    same = torch.equal(e, c)
    dmax = (e - c).abs().max().item()
    print(f"  {name:12s} {'bit-identical' if same else 'DIFFERENT'}  max |diff| {dmax:.3e}  (|chain| max {c.abs().max().item():.3e})")
    return same


ok = True
for l in range(L):
    ok &= cmp(f"L{l} qkv", qkv_e[l], qkv_c[l])
    ok &= cmp(f"L{l} x+attn", xs_e[2 * l + 1], xs[2 * l + 1])
    ok &= cmp(f"L{l} act", act_e[l], act_c[l])
    ok &= cmp(f"L{l} x+mlp", xs_e[2 * l + 2], xs[2 * l + 2])
if not opt.no_head:
    ok &= cmp("logits", logits_e, logits_c)
print("engine == chain, every op, bit for bit" if ok else "MISMATCH", flush=True)

# ---- timing
ops.linear_chain_timed(chain, 5)
t_chain = ops.linear_chain_timed(chain, opt.iters) * len(chain) * 1e3
gr = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ops.linear_chain_timed(chain, 1)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=side):
        from uniaudio2_amd._lib import lib, check
        for c in chain:
            check(lib.ua2_linear(C.byref(c), ops.stream()), "ua2_linear")
for _ in range(5):
    gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(opt.iters):
    gr.replay()
e1.record()
torch.cuda.synchronize()
t_graph = e0.elapsed_time(e1) / opt.iters * 1e3
ms = C.c_float(0.0)
eng.eng_timed(C.byref(a), G.data_ptr(), G.numel() * 8, 5, stream, C.byref(ms))
rc = eng.eng_timed(C.byref(a), G.data_ptr(), G.numel() * 8, opt.iters, stream, C.byref(ms))
torch.cuda.synchronize()
code = int(err[0].item())
t_eng = ms.value * 1e3
if opt.stamps:
    st = torch.zeros(256 * 64 + 96 * 4, dtype=torch.int64, device=dev)
    st[:256 * 64].view(256, 64)[:, 0:51:3] = 1 << 62          # start / ready: atomicMin over the consumer waves
    st[:256 * 64].view(256, 64)[:, 1:52:3] = 1 << 62
    a.stamps, a.evt_op = st.data_ptr(), opt.evt_op
    eng.eng_launch(C.byref(a), G.data_ptr(), G.numel() * 8, stream)
    torch.cuda.synchronize()
    a.stamps = None
    ev = st[256 * 64:].cpu().view(4, 96)
    st = st[:256 * 64].cpu().view(256, 64)
    print(f"CU 0, op {opt.evt_op}: consumer events (us since launch; L slot landed, D slot done, P unit's partials in, E epilogue done)")
    for w in range(3):
        print(f"  wave {w}: " + " ".join(f"{'?LDPE'[int(v) >> 56]}{(int(v) & ((1 << 56) - 1)) / 100:.2f}" for v in ev[w] if int(v)))
    t0 = st[:, 62].min().item()
    names = (["qkv", "o", "swiglu", "down"] * L + ["head"])[:n]
    print("per op (us, 100 MHz wall clock; over the 256 CUs): start of the op on the consumer | operand ready | done   -> median / max")
    for q in range(n):
        f = lambda c: (st[:, 3 * q + c] - t0).float() / 100.0
        print(f"  {q:2d} {names[q]:7s} start {f(0).median():7.2f} / {f(0).max():7.2f}   ready {f(1).median():7.2f} / {f(1).max():7.2f}   done {f(2).median():7.2f} / {f(2).max():7.2f}"
              f"   edge wait (ready - start) median {(f(1) - f(0)).median():6.2f}")
    print(f"  consumer waiting for landed slots: median {st[:, 60].float().median() / 100:.2f} us, max {st[:, 60].max() / 100:.2f}; "
          f"loader waiting for free slots: median {st[:, 61].float().median() / 100:.2f} us, max {st[:, 61].max() / 100:.2f}")
print(f"chain  ({len(chain)} launches back to back): {t_chain:7.1f} us  = {wbytes / t_chain / 1e6:.2f} TB/s")
print(f"chain  (one HIP graph replay)       : {t_graph:7.1f} us  = {wbytes / t_graph / 1e6:.2f} TB/s")
print(f"engine (memset + one launch), rc {rc} code {code:#x}: {t_eng:7.1f} us  = {wbytes / t_eng / 1e6:.2f} TB/s   engine / graph chain = {t_eng / t_graph:.3f}")

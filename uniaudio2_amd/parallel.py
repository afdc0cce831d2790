"""Data-parallel sharding of utterances over the GPUs of one node.

The reference decodes utterances one by one in a Python `for` (multi_task_inference.py:298,510,540);
they are independent, so each rank keeps a full weight replica and its own KV pool, takes the
utterances `i % world == rank` of a longest-first ordering, and the only exchange of the path is one
all-gather of the output token tensors at the end of the shard (RCCL over xGMI: `torch.distributed`
backend "nccl" on ROCm; "gloo" in the CPU tests).  Payload <= 2 x 8 x 500 x 4 B = 32 KB per
utterance: latency-bound, ONE fixed-shape padded int32 tensor (lengths, failure messages and a per-rank fatal flag ride in
its header words), no all-reduce.
"""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

MAX_FRAMES = 500          # evaluation/tts_task.py:222


MSG_BYTES = 96           # a failed utterance's reason travels with the gather (truncated UTF-8)


class GenerationFailed(RuntimeError):
    """(A RuntimeError, as the reference's own torch.stack failure is.)  An utterance-level failure of the generation LOGIC (e.g. PhaseSplitter.result: the model produced no frames of a phase —
    the reference's torch.stack error, evaluation/tts_task.py:283-284).  Only this is turned into a `Failed` slot by the
    sharded runners: device / library errors (HIP faults, out of memory, ua2_* status codes) are NOT utterance-level — after a
    sticky device fault every later utterance would "fail" too — and propagate."""


class Failed:
    """Result slot of an utterance whose generation raised on its rank.  It travels through the all-gather as the
    sentinel T_r = T_s = -1 (a rank that raised BEFORE the collective would leave the others waiting in it forever)."""

    def __init__(self, message: str = "generation failed on its rank"):
        self.message = message

    def __repr__(self):
        return f"Failed({self.message!r})"


def utterance_seed(seed: int, index: int) -> int:
    """Sampler key of utterance `index` (global, i.e. independent of the sharding): distinct, well-mixed 64-bit keys
    (splitmix64 finaliser) so that ranks never walk the same Philox stream."""
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(index) + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def shard_indices(lengths: Sequence[int], world: int, rank: int) -> List[int]:
    """Longest-first round-robin: utterance order[i] goes to rank i % world (balances decode time)."""
    order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
    return [order[i] for i in range(rank, len(order), world)]


HDR = 4 + MSG_BYTES // 4  # int32 header words per slot: (global index, T_r, T_s, reserved) + the message bytes
SLOT_WORDS = HDR + 2 * 8 * MAX_FRAMES


def _msg_words(message: str) -> torch.Tensor:
    raw = message.encode("utf-8", "replace")[:MSG_BYTES]
    buf = torch.zeros(MSG_BYTES, dtype=torch.uint8)
    buf[:len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
    return buf.view(torch.int32)


def _msg_text(words: torch.Tensor) -> str:
    return bytes(words.contiguous().view(torch.uint8).tolist()).rstrip(b"\0").decode("utf-8", "replace")


def pack_local(results: Dict[int, Tuple[torch.Tensor, torch.Tensor]], n_local_max: int, device, n_cb: int = 8, fatal: str = ""):
    """results: global index -> (reason (8,T_r), semantic (8,T_s)) int32.  ONE fixed-shape int32 buffer for the all-gather
    (north_star: a single all-gather of the token tensors): row = [global index, T_r, T_s, 0, message (96 B), tokens
    (2, 8, 500)]; one extra row per rank carries that rank's fatal-error flag + message.  Built on the host and moved with
    one copy (no per-utterance host->device traffic)."""
    assert n_cb == 8
    buf = torch.zeros(n_local_max + 1, SLOT_WORDS, dtype=torch.int32)
    buf[:, 0] = -1
    toks = [t for res in results.values() if not isinstance(res, Failed) for t in res]
    if toks and any(t.is_cuda for t in toks):
        torch.cuda.current_stream().synchronize()         # results come back to the host once, after the shard is done
    for slot, (gi, res) in enumerate(sorted(results.items(), key=lambda kv: kv[0])):
        if not isinstance(res, Failed) and max(res[0].shape[1], res[1].shape[1]) > MAX_FRAMES:
            # never raise between the ranks' collectives (the peers would wait forever): an over-long result is a failed slot
            res = Failed(f"result longer than MAX_FRAMES={MAX_FRAMES}: T_r={res[0].shape[1]} T_s={res[1].shape[1]}")
        row = buf[slot]
        if isinstance(res, Failed):
            row[0], row[1], row[2] = gi, -1, -1
            row[4:HDR] = _msg_words(res.message)
            continue
        r, s = res
        row[0], row[1], row[2] = gi, r.shape[1], s.shape[1]
        tok = row[HDR:].view(2, n_cb, MAX_FRAMES)
        tok[0, :, :r.shape[1]] = r.to(device="cpu", dtype=torch.int32)
        tok[1, :, :s.shape[1]] = s.to(device="cpu", dtype=torch.int32)
    if fatal:
        buf[n_local_max, 1] = 1
        buf[n_local_max, 4:HDR] = _msg_words(fatal)
    return buf.to(device)


def gather_results(results: Dict[int, Tuple[torch.Tensor, torch.Tensor]], n_total: int, device=None, fatal: str = ""):
    """All ranks end up with every utterance's (reason, semantic) tensors, keyed by global index.  `fatal`: this rank hit a
    non-utterance error (device fault, ua2_* status, a bug) while working on its shard — it still enters the collective (the
    peers would otherwise wait in it until the backend's time-out) and EVERY rank raises after it."""
    # No process group (a plain `python` run): nothing to exchange.  WITH a process group the collective runs at every world size,
    # one rank included — `torchrun --nproc-per-node 1` then takes exactly the pack -> device -> all-gather -> parse path of an
    # 8-rank job (a few tens of microseconds), so the one-GPU box exercises the product's RCCL exchange, not a short-cut around it.
    if not (dist.is_available() and dist.is_initialized()):
        if fatal:
            raise RuntimeError(fatal)
        return dict(results)
    world = dist.get_world_size()
    n_local_max = (n_total + world - 1) // world
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    buf = pack_local(results, n_local_max, device, fatal=fatal)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)          # the path's only collective
    out = {}
    fatals = []
    for rk, b in enumerate(bufs):
        hdr = b[:, :HDR].cpu()
        if int(hdr[n_local_max, 1]) == 1:
            fatals.append(f"rank {rk}: {_msg_text(hdr[n_local_max, 4:HDR])}")
        for slot in range(n_local_max):
            gi, tr, ts = (int(v) for v in hdr[slot, :3])
            if gi >= 0:
                if tr < 0:
                    out[gi] = Failed(_msg_text(hdr[slot, 4:HDR]) or "generation failed on its rank")
                else:
                    tok = b[slot, HDR:].view(2, 8, MAX_FRAMES)
                    out[gi] = (tok[0, :, :tr].clone(), tok[1, :, :ts].clone())
    if fatals:
        raise RuntimeError("sharded generation aborted — " + "; ".join(fatals))
    return out


def _shard_loop(body):
    """Runs `body()` (this rank's shard); a non-utterance exception is carried INTO the gather instead of past it."""
    try:
        body()
        return "", None
    except GenerationFailed:
        raise
    except Exception as e:                               # noqa: BLE001 — re-raised on every rank after the collective
        return f"{type(e).__name__}: {e}"[:MSG_BYTES], e


def run_sharded(items: Sequence, lengths: Sequence[int], generate_fn) -> Dict[int, Tuple[torch.Tensor, torch.Tensor]]:
    """generate_fn(item) -> (reason, semantic); runs this rank's shard, then gathers everything everywhere."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    local = {}

    def body():
        for i in shard_indices(lengths, world, rank):
            try:
                local[i] = generate_fn(items[i])
            except GenerationFailed as e:                # e.g. PhaseSplitter.result: no frames of a phase were produced
                local[i] = Failed(f"{type(e).__name__}: {e}")

    fatal, exc = _shard_loop(body)
    if exc is not None and world == 1:
        raise exc
    try:
        return gather_results(local, len(items), fatal=fatal)
    except RuntimeError as e:
        if exc is not None:
            raise e from exc
        raise


def run_sharded_batched(items: Sequence, lengths: Sequence[int], generate_batch_fn, batch_size: int):
    """As run_sharded, but this rank's shard is processed `batch_size` items at a time by
    generate_batch_fn(list of items) -> list of (reason, semantic) (continuous batching on one GPU)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = list(shard_indices(lengths, world, rank))
    local = {}

    def body():
        for s0 in range(0, len(mine), max(1, batch_size)):
            chunk = mine[s0:s0 + max(1, batch_size)]
            try:
                for i, res in zip(chunk, generate_batch_fn([items[i] for i in chunk])):
                    local[i] = res
            except GenerationFailed as e:
                for i in chunk:
                    local.setdefault(i, Failed(f"{type(e).__name__}: {e}"))

    fatal, exc = _shard_loop(body)
    if exc is not None and world == 1:
        raise exc
    try:
        return gather_results(local, len(items), fatal=fatal)
    except RuntimeError as e:
        if exc is not None:
            raise e from exc
        raise

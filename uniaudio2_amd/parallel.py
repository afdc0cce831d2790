"""Data-parallel sharding of utterances over the GPUs of one node.

The reference decodes utterances one by one in a Python `for` (multi_task_inference.py:298,510,540);
they are independent, so each rank keeps a full weight replica and its own KV pool, takes the
utterances `i % world == rank` of a longest-first ordering, and the only exchange of the path is one
all-gather of the output token tensors at the end of the shard (RCCL over xGMI: `torch.distributed`
backend "nccl" on ROCm; "gloo" in the CPU tests).  Payload <= 2 x 8 x 500 x 4 B = 32 KB per
utterance: latency-bound, one fixed-shape padded int32 tensor + a lengths tensor, no all-reduce.
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


def pack_local(results: Dict[int, Tuple[torch.Tensor, torch.Tensor]], n_local_max: int, device, n_cb: int = 8):
    """results: global index -> (reason (8,T_r), semantic (8,T_s)) int32.  Fixed-shape buffers for the all-gather."""
    tok = torch.zeros(n_local_max, 2, n_cb, MAX_FRAMES, dtype=torch.int32, device=device)
    meta = torch.full((n_local_max, 3), -1, dtype=torch.int32, device=device)      # (global index, T_r, T_s)
    msg = torch.zeros(n_local_max, MSG_BYTES, dtype=torch.uint8, device=device)
    for slot, (gi, res) in enumerate(sorted(results.items(), key=lambda kv: kv[0])):
        if not isinstance(res, Failed) and max(res[0].shape[1], res[1].shape[1]) > MAX_FRAMES:
            # never raise between the ranks' collectives (the peers would wait forever): an over-long result is a failed slot
            res = Failed(f"result longer than MAX_FRAMES={MAX_FRAMES}: T_r={res[0].shape[1]} T_s={res[1].shape[1]}")
        if isinstance(res, Failed):
            meta[slot] = torch.tensor([gi, -1, -1], dtype=torch.int32)
            raw = res.message.encode("utf-8", "replace")[:MSG_BYTES]
            msg[slot, :len(raw)] = torch.tensor(list(raw), dtype=torch.uint8)
            continue
        r, s = res
        tok[slot, 0, :, :r.shape[1]] = r.to(device=device, dtype=torch.int32)
        tok[slot, 1, :, :s.shape[1]] = s.to(device=device, dtype=torch.int32)
        meta[slot] = torch.tensor([gi, r.shape[1], s.shape[1]], dtype=torch.int32)
    return tok, meta, msg


def gather_results(results: Dict[int, Tuple[torch.Tensor, torch.Tensor]], n_total: int, device=None):
    """All ranks end up with every utterance's (reason, semantic) tensors, keyed by global index."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dict(results)
    world = dist.get_world_size()
    n_local_max = (n_total + world - 1) // world
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    tok, meta, msg = pack_local(results, n_local_max, device)
    toks = [torch.empty_like(tok) for _ in range(world)]
    metas = [torch.empty_like(meta) for _ in range(world)]
    msgs = [torch.empty_like(msg) for _ in range(world)]
    dist.all_gather(toks, tok)          # the path's only collective (+ its two small side-cars)
    dist.all_gather(metas, meta)
    dist.all_gather(msgs, msg)
    out = {}
    for t, m, g in zip(toks, metas, msgs):
        m = m.cpu()
        for slot in range(n_local_max):
            gi, tr, ts = (int(v) for v in m[slot])
            if gi >= 0:
                if tr < 0:
                    raw = bytes(g[slot].cpu().tolist()).rstrip(b"\0")
                    out[gi] = Failed(raw.decode("utf-8", "replace") or "generation failed on its rank")
                else:
                    out[gi] = (t[slot, 0, :, :tr].clone(), t[slot, 1, :, :ts].clone())
    return out


def run_sharded(items: Sequence, lengths: Sequence[int], generate_fn) -> Dict[int, Tuple[torch.Tensor, torch.Tensor]]:
    """generate_fn(item) -> (reason, semantic); runs this rank's shard, then gathers everything everywhere."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    local = {}
    for i in shard_indices(lengths, world, rank):
        try:
            local[i] = generate_fn(items[i])
        except GenerationFailed as e:                    # e.g. PhaseSplitter.result: no frames of a phase were produced
            local[i] = Failed(f"{type(e).__name__}: {e}")
    return gather_results(local, len(items))


def run_sharded_batched(items: Sequence, lengths: Sequence[int], generate_batch_fn, batch_size: int):
    """As run_sharded, but this rank's shard is processed `batch_size` items at a time by
    generate_batch_fn(list of items) -> list of (reason, semantic) (continuous batching on one GPU)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = list(shard_indices(lengths, world, rank))
    local = {}
    for s0 in range(0, len(mine), max(1, batch_size)):
        chunk = mine[s0:s0 + max(1, batch_size)]
        try:
            for i, res in zip(chunk, generate_batch_fn([items[i] for i in chunk])):
                local[i] = res
        except GenerationFailed as e:
            for i in chunk:
                local.setdefault(i, Failed(f"{type(e).__name__}: {e}"))
    return gather_results(local, len(items))

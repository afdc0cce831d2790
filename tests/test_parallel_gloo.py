"""Multi-process CPU test of the data-parallel driver (gloo, world_size 2): sharding covers every
utterance exactly once and the all-gather returns every rank the full, correctly keyed result set."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def fake_generate(i):
    g = torch.Generator().manual_seed(100 + i)
    tr, ts = 3 + i % 5, 7 + (i * 3) % 11
    return (torch.randint(0, 4096, (8, tr), generator=g, dtype=torch.int32),
            torch.randint(0, 8192, (8, ts), generator=g, dtype=torch.int32))


def _worker(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uniaudio2_amd.parallel import run_sharded, shard_indices
    lengths = [(i * 7) % 13 + 1 for i in range(n)]
    mine = shard_indices(lengths, world, rank)
    out = run_sharded(list(range(n)), lengths, fake_generate)
    ok = sorted(out) == list(range(n))
    for i in range(n):
        r, s = fake_generate(i)
        ok = ok and torch.equal(out[i][0], r) and torch.equal(out[i][1], s)
    ret[rank] = (ok, mine)
    dist.destroy_process_group()


def test_sharded_generation_allgather_world2():
    n, world = 7, 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, 29591, n, ret), nprocs=world, join=True)
        assert all(ret[r][0] for r in range(world))
        covered = sorted(ret[0][1] + ret[1][1])
        assert covered == list(range(n))                      # every utterance exactly once
        assert abs(len(ret[0][1]) - len(ret[1][1])) <= 1


def _worker_batched(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uniaudio2_amd.parallel import run_sharded_batched
    lengths = [(i * 5) % 11 + 1 for i in range(n)]
    calls = []

    def fake_batch(idx):
        calls.append(list(idx))
        return [fake_generate(i) for i in idx]

    out = run_sharded_batched(list(range(n)), lengths, fake_batch, batch_size=3)
    ok = sorted(out) == list(range(n)) and all(len(c) <= 3 for c in calls)
    for i in range(n):
        r, s = fake_generate(i)
        ok = ok and torch.equal(out[i][0], r) and torch.equal(out[i][1], s)
    ret[rank] = (ok, [i for c in calls for i in c])
    dist.destroy_process_group()


def test_sharded_batched_generation_world2():
    """--batch_size path: each rank walks its shard three utterances at a time; the gathered result is complete and
    correctly keyed on every rank."""
    n, world = 11, 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_batched, args=(world, 29593, n, ret), nprocs=world, join=True)
        assert all(ret[r][0] for r in range(world))
        assert sorted(ret[0][1] + ret[1][1]) == list(range(n))


def test_shard_indices_longest_first():
    from uniaudio2_amd.parallel import shard_indices
    lengths = [5, 50, 7, 40, 30]
    assert shard_indices(lengths, 2, 0) == [1, 4, 0] and shard_indices(lengths, 2, 1) == [3, 2]
    assert shard_indices(lengths, 1, 0) == [1, 3, 4, 2, 0]


def _worker_failing(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uniaudio2_amd.parallel import Failed, GenerationFailed, run_sharded

    def gen(i):
        if i == 2:
            raise GenerationFailed("no semantic frames were produced")      # what PhaseSplitter.result raises
        if i == 3:                                                          # over-long result: a failed slot, not a raise between collectives
            return torch.zeros(8, 501, dtype=torch.int32), torch.zeros(8, 3, dtype=torch.int32)
        return fake_generate(i)

    out = run_sharded(list(range(n)), [1] * n, gen)
    ok = sorted(out) == list(range(n)) and isinstance(out[2], Failed) and isinstance(out[3], Failed)
    # the reason survives the gather on EVERY rank, whichever rank the utterance ran on
    ok = ok and "no semantic frames" in out[2].message and "MAX_FRAMES" in out[3].message
    for i in range(n):
        if i not in (2, 3):
            ok = ok and torch.equal(out[i][0], fake_generate(i)[0])
    ret[rank] = ok
    dist.destroy_process_group()


def test_device_errors_are_not_swallowed_as_failed_utterances():
    """A RuntimeError that is not the generation logic's own (HIP fault, OOM, ua2_* status) propagates."""
    from uniaudio2_amd.parallel import run_sharded

    def gen(i):
        raise RuntimeError("HIP error: an illegal memory access was encountered")

    with pytest.raises(RuntimeError):
        run_sharded([0, 1], [1, 1], gen)


def test_failed_utterance_travels_as_sentinel_world2():
    """One utterance raising on its rank must not leave the other rank waiting in the all-gather: it is gathered as
    the sentinel T = -1 and surfaces as parallel.Failed on every rank."""
    n, world = 5, 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_failing, args=(world, 29595, n, ret), nprocs=world, join=True)
        assert all(ret[r] for r in range(world))


def test_utterance_seeds_are_distinct_and_sharding_independent():
    from uniaudio2_amd.parallel import utterance_seed
    keys = {utterance_seed(888, i) for i in range(512)} | {utterance_seed(889, i) for i in range(512)}
    assert len(keys) == 1024 and all(0 <= k < 2 ** 64 for k in keys)


def _worker_fatal(rank, world, port, n, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from uniaudio2_amd.parallel import run_sharded, run_sharded_batched

    def gen(i):
        if i == 1:                                             # lands on exactly one rank
            raise ValueError("ua2_linear: status -3 (a library error, not an utterance-level failure)")
        return fake_generate(i)

    got = []
    for runner in (lambda: run_sharded(list(range(n)), [1] * n, gen),
                   lambda: run_sharded_batched(list(range(n)), [1] * n, lambda idx: [gen(i) for i in idx], batch_size=2)):
        try:
            runner()
            got.append("returned")
        except RuntimeError as e:
            got.append(str(e))
    ret[rank] = got
    dist.destroy_process_group()


def test_fatal_error_on_one_rank_raises_on_every_rank_world2():
    """a non-utterance exception on one rank must not leave the peers blocked in the all-gather — the rank enters
    the collective with a fatal flag and every rank raises after it (the failing rank chains the original exception)."""
    n, world = 6, 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_fatal, args=(world, 29597, n, ret), nprocs=world, join=True)
        for r in range(world):
            assert len(ret[r]) == 2
            for msg in ret[r]:
                assert "aborted" in msg and "status -3" in msg and "rank" in msg, ret[r]


class _FakeCodec:
    """Stand-in for the ReasoningTokenizer in stage 2's driver: a wave that encodes its codes (so a file proves which codes
    produced it), and a log of how the driver called it."""
    sample_rate = 24000

    def __init__(self):
        self.calls = []

    def _wave(self, c):
        n = int(c.shape[-1] / 12.5 * 24000)
        return (torch.arange(n, dtype=torch.float32)[None] % 7) / 70.0 + float(c.sum() % 5) / 50.0

    def detokenize_no_reason(self, c, return_reasoning_text=False, steps=50):
        self.calls.append(1)
        return self._wave(c)

    def detokenize_no_reason_batch(self, cs, steps=50, max_batch=8):
        self.calls.append(len(cs))
        return [self._wave(c) for c in cs]


def _worker_stage2(rank, world, port, tok_dir, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import argparse
    from uniaudio2_amd.multi_task_inference import decode_token_dir
    codec = _FakeCodec()
    args = argparse.Namespace(token_dir=tok_dir, output_dir=tok_dir, wav_dir=os.path.join(tok_dir, f"wavs_rank{rank}"), codec_steps=2, codec_batch=3)
    decode_token_dir(codec, args, torch.device("cpu"))
    wrote = sorted(f[:-4] for f in os.listdir(args.wav_dir))
    everyone = [None] * world
    dist.all_gather_object(everyone, wrote)
    ret[rank] = (wrote, everyone, codec.calls)
    dist.destroy_process_group()


def test_stage2_shards_utterances_over_ranks_world2(tmp_path):
    """stage 2 shards by utterance.  Two ranks decode one token directory (a stand-in codec, CPU): the union of
    what they wrote is every utterance with both token files, the shards are disjoint and balanced, the orphan is skipped, and
    each rank walked its shard --codec_batch utterances at a time."""
    names = [f"utt_{i:02d}" for i in range(9)]
    for i, n in enumerate(names):
        torch.save(torch.randint(0, 4096, (8, 5 + i), dtype=torch.int32), tmp_path / f"{n}_reason.pt")
        torch.save(torch.randint(0, 8192, (8, 20 + 3 * i), dtype=torch.int32), tmp_path / f"{n}_semantic.pt")
    torch.save(torch.zeros(8, 3, dtype=torch.int32), tmp_path / "orphan_reason.pt")
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_stage2, args=(world, 29597, str(tmp_path), ret), nprocs=world, join=True)
        a, b = ret[0][0], ret[1][0]
        assert sorted(a + b) == names and not set(a) & set(b)
        assert abs(len(a) - len(b)) <= 1
        assert ret[0][1] == [a, b] == ret[1][1]                 # both ranks saw the same picture through the collective
        for r in range(world):
            assert all(c <= 3 for c in ret[r][2]) and sum(ret[r][2]) == len(ret[r][0])


def test_stage2_shard_is_a_partition():
    from uniaudio2_amd.multi_task_inference import stage2_shard
    names = [f"n{i}" for i in range(23)]
    for world in (1, 2, 4, 8):
        shards = [stage2_shard(names, world, r) for r in range(world)]
        assert sorted(sum(shards, [])) == sorted(names)
        assert max(map(len, shards)) - min(map(len, shards)) <= 1


def _worker_one_rank(rank, world, port, backend, n, ret):
    """One rank WITH a process group: the product's gather must run its collective (pack -> device -> all_gather -> parse), not
    return early (at world 1 the path had only ever executed under gloo at world 2)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=0, world_size=1)
    from uniaudio2_amd import parallel
    seen = []
    real = dist.all_gather

    def spy(bufs, buf, *a, **k):
        seen.append((str(buf.device), tuple(buf.shape), buf.dtype))
        return real(bufs, buf, *a, **k)

    dist.all_gather = spy
    try:
        dev = torch.device("cuda", 0) if backend == "nccl" else torch.device("cpu")
        gen = lambda i: tuple(t.to(dev) for t in fake_generate(i))
        out = parallel.run_sharded(list(range(n)), [1 + i % 3 for i in range(n)], gen)
        failed = {0: fake_generate(0), 1: parallel.Failed("no semantic phase")}
        out2 = parallel.gather_results(failed, 2)
    finally:
        dist.all_gather = real
    ok = sorted(out) == list(range(n))
    for i in range(n):
        r, s = fake_generate(i)
        ok = ok and torch.equal(out[i][0].cpu(), r) and torch.equal(out[i][1].cpu(), s)
    ok = ok and isinstance(out2[1], parallel.Failed) and out2[1].message == "no semantic phase" and torch.equal(out2[0][0].cpu(), fake_generate(0)[0])
    ret[0] = (ok, seen, str(out[0][0].device))
    dist.destroy_process_group()


def test_one_rank_process_group_still_runs_the_collective_gloo():
    n = 5
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_one_rank, args=(1, 29599, "gloo", n, ret), nprocs=1, join=True)
        ok, seen, _ = ret[0]
        assert ok
        from uniaudio2_amd import parallel
        assert len(seen) == 2 and seen[0] == ("cpu", (n + 1, parallel.SLOT_WORDS), torch.int32)


@pytest.mark.gpu
def test_one_rank_nccl_group_runs_the_products_gather_on_the_device():
    """The product's own exchange over RCCL on this box: pack_local's tensor goes to the device, through dist.all_gather on the
    `nccl` backend, and is parsed back (parallel.py)."""
    n = 6
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker_one_rank, args=(1, 29601, "nccl", n, ret), nprocs=1, join=True)
        ok, seen, out_dev = ret[0]
        assert ok
        from uniaudio2_amd import parallel
        assert len(seen) == 2 and seen[0] == ("cuda:0", (n + 1, parallel.SLOT_WORDS), torch.int32)
        assert out_dev.startswith("cuda")

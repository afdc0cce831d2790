#!/usr/bin/env python3
"""bench.py — audio tokens/s of greedy TTS decoding on MI355X (BASELINE.json metric, configs[1]).

Workload: Model_stage3 at the named sizes (Llama-3.2-3B backbone,
3-layer understanding expert, 2-layer generation expert, 4-layer 2048-d local decoder run 8x per
frame, V_a = 12296), random-init seeded weights (no public checkpoint), bf16 kernels, B = 1.
One STEP = one utterance = 33-token prompt (32-row prefill) + 74 greedy frames (8 audio tokens +
1 text token each, EOS ignored so the length is fixed).  value = audio tokens of all ranks / time.

Multi-GPU (weak scaling): utterances are independent; each rank decodes its own
utterance per step with a full replica, and the only exchange is one RCCL all-gather of the
(2, 8, T) output token tensors per step, inside the timed region.

Extra objects in the JSON line: "roofline" for the dominant kernel (the fused RMSNorm + fc_1/fc_2
+ SwiGLU weight-streaming GEMM) from HIP-event timing of exactly the launch mix one frame issues;
"cpu_baseline" = the CPU oracle (fp32 port of the reference algorithm as shipped) on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

PROMPT_LEN = 33
FRAMES = 74
SEM_CARD, REASON_CARD = 8196, 4100          # V_a = 12296 (placeholder sizes; the real ones live in a yaml not in the repo)
HBM_PEAK_GBS = 8000.0                       # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def model_args():
    from uniaudio2_amd.llm_models.model_new import ModelArgs
    return ModelArgs(llm_name="Llama-3.2-3B", decoder_name="Llama-3.2-300M", llm_pretrained_model="",
                     audio_embeddings_path="", audio_understanding_expert_path="",
                     audio_semantic_vocab_size=SEM_CARD, audio_reason_vocab_size=REASON_CARD, audio_num_codebooks=8)


def build_model(device, seed=0):
    """Random-init at the real sizes directly on the device: N(0, 0.02) for Linear / Embedding /
    audio_head (lit_model.py:74-81 convention), norm weights = 1."""
    from uniaudio2_amd.llm_models.model_new import Model_stage3
    torch.manual_seed(seed)
    model = Model_stage3(model_args(), device=device)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.dim() == 1:
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02)
    return model


def make_prompt(device, seed):
    g = torch.Generator().manual_seed(seed)
    t = torch.zeros(1, PROMPT_LEN, 9, dtype=torch.long)
    t[0, :, -1] = torch.randint(0, 128000, (PROMPT_LEN,), generator=g)
    m = torch.zeros(1, PROMPT_LEN, 9, dtype=torch.bool)
    m[0, :, -1] = True
    return t.to(device), m.to(device)


def utterance(model, tokens, mask, frames=FRAMES):
    """prefill + `frames` frames, all on device; returns the (frames, 1, 9) id log (device)."""
    L = tokens.size(1)
    model.reset_caches()
    pos = torch.arange(L, device=tokens.device).unsqueeze(0)
    model.forward_prefix(tokens[:, :-1], tokens_mask=mask, input_pos=pos[:, :-1])
    model.begin_decode(tokens[:, -1:], mask[:, -1:], torch.tensor([L - 1], device=tokens.device))
    return model.generate_frames(frames, 1, 0, reason_eos=-1, reason_card=REASON_CARD, max_pos=L + frames)


def roofline_leg(model):
    """HIP-event timing of the dominant kernel — gemv_kernel<bf16, NORM, SWIGLU, CPW=4, multi-round>
    (rocprof symbol `gemv_kernel<1, 1, 2, 4, true>`): fused RMSNorm + fc_1/fc_2 + SwiGLU at 3072 -> 2x8192 —
    over exactly its launches in one frame: the 33 layers of the three 3072-d GPTs, each with its own
    100.7 MB of weights (so every launch streams cold data).  The 2048-d local decoder's SwiGLU is a
    different instantiation and is reported by tools/ubench/gemv_shapes.py, not here."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_SWIGLU, PRO_NORM
    dev = model.projection.weight.device
    args, bytes_total = [], 0
    keep = []

    def add(gpt, reps):
        nonlocal bytes_total
        cfg, p = gpt.config, gpt.plan
        x = torch.randn(1, cfg.n_embd, device=dev)
        y = torch.empty(1, cfg.intermediate_size, device=dev)
        keep.extend([x, y])
        for _ in range(reps):
            for l in range(cfg.n_layer):
                args.append(ops.linear(dtype=torch.bfloat16, M=1, N=cfg.intermediate_size, K=cfg.n_embd,
                                       w0=p["fc1"][l], w1=p["fc2"][l], prologue=PRO_NORM, epilogue=EPI_SWIGLU, x=x,
                                       norm_w=p["norm2"][l], eps=cfg.norm_eps, y=y, launch=False))
                # algorithmic bytes: both weight matrices once (bf16) + x, norm weight (fp32) + y (fp32)
                bytes_total += 2 * cfg.intermediate_size * cfg.n_embd * 2 + 2 * cfg.n_embd * 4 + cfg.intermediate_size * 4

    add(model.audio_understanding_expert, 1)
    add(model.backbone, 1)
    add(model.audio_generation_expert, 1)
    ops.linear_chain_timed(args, 2)                       # warm
    ms = ops.linear_chain_timed(args, 10)
    per_launch_bytes = bytes_total / len(args)
    achieved = per_launch_bytes / (ms * 1e-3) / 1e9
    return {"kernel": "gemv_kernel<1, 1, 2, 4, true> (RMSNorm + fc_1/fc_2 + SwiGLU GEMV, 3072 -> 2x8192, bf16)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            # HBM read bytes per launch from a separate PMC pass (profiles/r1_pmc_swiglu.txt: FETCH_SIZE x 1024 x 2, re-measured on the final kernel,
            # the gfx950 half-count correction of MI355X_MICROARCH.md §HBM); 1.003x the algorithmic bytes
            "traffic": 100995000, "launches_per_frame": len(args), "avg_launch_us": round(ms * 1e3, 2),
            "algorithmic_bytes_per_launch": int(per_launch_bytes)}


def codec_leg(dev):
    """Codec side of the metric ("codec RTF"): the in-scope deterministic stage-2 sub-graph on one 20-s
    window — ScalarModel.decode of a (1, 136, 500) latent -> 480 000 samples — and the RVQ search of a
    10-s clip (125 frames x (1+1+6) levels of 8192 x 32).  Channel widths / strides live in the
    reference's sqcodec_config.yaml, which is not in the repo: placeholder config with hop 960,
    latent 136, init_channel 32."""
    from uniaudio2_amd import ops
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    torch.manual_seed(1)
    sq = ScalarModel(num_bands=1, sample_rate=24000, causal=True, num_samples=2, downsample_factors=[2, 4, 4, 5, 3],
                     downsample_kernel_sizes=[4, 8, 8, 10, 6], upsample_factors=[3, 5, 4, 4, 2],
                     upsample_kernel_sizes=[6, 10, 8, 8, 4], latent_hidden_dim=136, default_kernel_size=7,
                     delay_kernel_size=5, init_channel=32, res_kernel_size=7).to(dev).prepare()
    lat = torch.tanh(torch.randn(1, 136, 500, device=dev))
    # algorithmic work of one decode: every ua2_conv1d launch's 2 * B * Cout * Tout * Cin * K flops and its
    # activation + weight bytes (counted by wrapping the op for one call)
    work = {"flop": 0.0, "bytes": 0.0, "launches": 0}
    real_conv1d = ops.conv1d

    def counting_conv1d(x, w_packed, K, Cout, **kw):
        y = real_conv1d(x, w_packed, K, Cout, **kw)
        B_, Cin, Tin = x.shape
        work["flop"] += 2.0 * B_ * Cout * y.shape[-1] * Cin * K
        work["bytes"] += 4.0 * (x.numel() + y.numel() + (y.numel() if kw.get("residual") is not None else 0)) + 4.0 * w_packed.numel()
        work["launches"] += 1
        return y

    ops.conv1d = counting_conv1d
    try:
        wav = sq.decode(lat)
    finally:
        ops.conv1d = real_conv1d
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        wav = sq.decode(lat)
    e1.record(); torch.cuda.synchronize()
    dec_ms = e0.elapsed_time(e1) / 3
    x = torch.randn(125, 32, device=dev)
    emb = torch.randn(6, 8192, 32, device=dev)
    embT = emb.transpose(1, 2).contiguous()
    ops.rvq_encode(x, emb, embT); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.rvq_encode(x, emb, embT)
    e1.record(); torch.cuda.synchronize()
    return {"scalar_decode_ms_per_20s_window": round(dec_ms, 3), "scalar_decode_rtf": round(dec_ms / 1e3 / (wav.shape[-1] / 24000.0), 6),
            # exact-fp32 implicit GEMM on v_mfma_f32_16x16x4_f32: dense f32 MFMA peak = 1/16 of the bf16 one (2.5 PFLOP/s / 16)
            "scalar_decode_conv_launches": work["launches"], "scalar_decode_gflop": round(work["flop"] / 1e9, 2),
            "scalar_decode_tflops": round(work["flop"] / (dec_ms * 1e-3) / 1e12, 2),
            "scalar_decode_frac_f32_mfma_peak": round(work["flop"] / (dec_ms * 1e-3) / (2.5e15 / 16), 4),
            "scalar_decode_algorithmic_GBps": round(work["bytes"] / (dec_ms * 1e-3) / 1e9, 1),
            "rvq_encode_us_125x6x8192x32": round(e0.elapsed_time(e1) / 10 * 1e3, 1), "config": "placeholder init_channel=32, hop 960"}


def batched_leg(model, dev, B=64, frames=24):
    """Information beside the B = 1 headline: one GPU decoding B = 64 sequences together
    (32..33-token prompts, greedy, same kernels; rows bit-identical to their B = 1 runs, tests/test_gpu_configs.py).
    Re-plans the caches for 64 sequences, so it runs last."""
    model.setup_caches(B, dtype=torch.bfloat16, max_seq_length=2048, max_rows=B * PROMPT_LEN, log_frames=frames + 8)
    g = torch.Generator().manual_seed(99)
    t = torch.zeros(B, PROMPT_LEN, 9, dtype=torch.long)
    t[:, :, -1] = torch.randint(0, 128000, (B, PROMPT_LEN), generator=g)
    m = torch.zeros(B, PROMPT_LEN, 9, dtype=torch.bool)
    m[:, :, -1] = True
    t, m = t.to(dev), m.to(dev)
    pos = torch.arange(PROMPT_LEN - 1, device=dev).unsqueeze(0).repeat(B, 1)
    res = {}
    for rep in range(2):                                   # first pass captures the graph
        model.reset_caches()
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        model.forward_prefix(t[:, :-1], tokens_mask=m, input_pos=pos)
        e1.record()
        model.begin_decode(t[:, -1:], m[:, -1:], torch.tensor([PROMPT_LEN - 1], device=dev))
        model.generate_frames(frames, B, 0, reason_eos=-1, reason_card=REASON_CARD, max_pos=PROMPT_LEN + frames)
        e2.record()
        torch.cuda.synchronize()
        res = {"B": B, "prefill_rows": B * (PROMPT_LEN - 1), "prefill_ms": round(e0.elapsed_time(e1), 2),
               "decode_ms_per_frame": round(e1.elapsed_time(e2) / frames, 3),
               "audio_tokens_per_s": round(8 * B * frames / (e1.elapsed_time(e2) * 1e-3), 1)}
    return res


def cpu_baseline_leg(model, tokens, mask, frames=6):
    """The CPU oracle (fp32 port of the reference algorithm as shipped: full-2048 masked prefill,
    repeat_interleave GQA, lm_head + 8-step local decoder every frame) on this host's cores, same
    weights, same prompt, bounded to a 32-row prefill + `frames` frames."""
    from oracle.lm_oracle import GPTShape, Stage3Oracle, run_decode_loop
    t0 = time.time()
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    shapes = dict(backbone=GPTShape(28, 3072, 24, 8, 8192), understanding=GPTShape(3, 3072, 24, 8, 8192),
                  generation=GPTShape(2, 3072, 24, 8, 8192), decoder=GPTShape(4, 2048, 32, 8, 8192))
    o = Stage3Oracle(sd, shapes, SEM_CARD, REASON_CARD, 8, mode="fp32")
    o.setup_caches(1)
    setup_s = time.time() - t0
    t0 = time.time()
    r = run_decode_loop(o, tokens.cpu(), mask.cpu(), frames, "audio")
    dt = time.time() - t0
    # split prefill / decode by a second, decode-only timing of the last frames
    return {"value": round(8 * frames / dt, 2), "unit": "audio tokens/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{PROMPT_LEN - 1}-row prefill + {frames} greedy frames, fp32, B=1 "
            f"(wall {dt:.1f} s incl. prefill; oracle setup {setup_s:.0f} s excluded)"}, r["samples"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # under torchrun the RCCL path runs even with one rank
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    import uniaudio2_amd  # noqa: F401  (fails loudly without libua2hip.so)

    model = build_model(dev)
    model.setup_caches(1, dtype=torch.bfloat16, max_seq_length=2048, max_rows=64, log_frames=128)
    tokens, mask = make_prompt(dev, seed=1000 + rank)

    def step():
        log = utterance(model, tokens, mask)                    # (FRAMES, 1, 9)
        if use_dist:
            out = [torch.empty_like(log) for _ in range(world)]
            dist.all_gather(out, log.contiguous())              # the path's only exchange
        return log

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        log = step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # decode-only rate (information): frames after the prefill, graph replay
    utterance(model, tokens, mask, frames=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.generate_frames(64, 1, 0, reason_eos=-1, reason_card=REASON_CARD, max_pos=PROMPT_LEN + FRAMES)
    e1.record()
    torch.cuda.synchronize()
    ms_frame = e0.elapsed_time(e1) / 64

    audio_tokens = 8 * FRAMES * a.steps * world
    res = {"metric": "audio tokens/sec (TTS greedy)", "value": round(audio_tokens / dt, 1), "unit": "audio tokens/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "TTS --stage 1, single utterance, greedy (topk=1), bf16, B=1 per GPU: "
                                  f"{PROMPT_LEN}-token prompt + {FRAMES} frames x (8 audio + 1 text) tokens; "
                                  "Llama-3.2-3B backbone + 3L/2L experts + 4L local decoder x8, V_a=12296, random init",
                      "parallelism": f"dp{world} (one utterance per GPU, RCCL all-gather of token tensors)"},
           "decode_ms_per_frame": round(ms_frame, 3), "decode_frames_per_s": round(1e3 / ms_frame, 1)}
    if rank == 0 and world == 1 and not a.no_roofline:
        res["roofline"] = roofline_leg(model)
        # whole-frame view of the same roofline: unique weight bytes a frame must stream (bf16)
        res["frame_hbm_frac_unique_weights"] = round(8.33e9 / (ms_frame * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if rank == 0 and world == 1 and not a.no_roofline:
        res["codec"] = codec_leg(dev)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb, cpu_ids = cpu_baseline_leg(model, tokens, mask)
        res["cpu_baseline"] = cb
        n = cpu_ids.shape[0]
        res["cpu_fp32_vs_gpu_bf16_same_ids_frames"] = int((cpu_ids[:, 0].int() == log[:n, 0].cpu().int()).all(-1).sum())
    if rank == 0 and world == 1 and not a.no_roofline:
        res["batched_decode"] = batched_leg(model, dev)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

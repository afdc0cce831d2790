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
PMC_SWIGLU_TRAFFIC = 100813552              # bytes per launch, profiles/r6_pmc_swiglu.txt (refreshed every round by tools/final_round_run.sh)


SCALAR_CFG = dict(num_bands=1, sample_rate=24000, causal=True, num_samples=2, downsample_factors=[2, 4, 4, 5, 3],
                  downsample_kernel_sizes=[4, 8, 8, 10, 6], upsample_factors=[3, 5, 4, 4, 2], upsample_kernel_sizes=[6, 10, 8, 8, 4],
                  latent_hidden_dim=136, default_kernel_size=7, delay_kernel_size=5, init_channel=32, res_kernel_size=7)   # placeholder widths


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


def utterance(model, tokens, mask, frames=FRAMES, skip_text_head=False):
    """prefill + `frames` frames, all on device; returns the (frames, 1, 9) id log (device)."""
    L = tokens.size(1)
    model.reset_caches()
    pos = torch.arange(L, device=tokens.device).unsqueeze(0)
    model.forward_prefix(tokens[:, :-1], tokens_mask=mask, input_pos=pos[:, :-1])
    model.begin_decode(tokens[:, -1:], mask[:, -1:], torch.tensor([L - 1], device=tokens.device))
    return model.generate_frames(frames, 1, 0, reason_eos=-1, reason_card=REASON_CARD, max_pos=L + frames, skip_text_head=skip_text_head)


def roofline_leg(model):
    """HIP-event timing of the dominant kernel — the fused (RMSNorm) + fc_1/fc_2 + SwiGLU weight-streaming GEMV at
    3072 -> 2x8192, in the form the B = 1 frame runs since round 3: `gemv_kernel<1, 4, 2, 4, true>` (bf16, UA2_PRO_SCALED
    prologue — the operand row RNE_bf16(x (.) w) and the sum-of-squares partials are handed over by the producer of x, the
    row scale is applied to the fp32 sums — SWIGLU epilogue, CPW = 4, multi-round) — over exactly its launches in one frame:
    the 33 layers of the three 3072-d GPTs, each with its own 100.7 MB of weights (so every launch streams cold data).  The
    2048-d local decoder's SwiGLU is a different instantiation and is reported by tools/ubench/gemv_shapes.py, not here.
    ua2_linear_chain_timed brackets the launches with hipEvents on the stream they are issued on."""
    from uniaudio2_amd import ops
    from uniaudio2_amd._lib import EPI_SWIGLU
    PRO_SCALED = 4
    dev = model.projection.weight.device
    args, bytes_total = [], 0
    keep = []

    def add(gpt, reps):
        nonlocal bytes_total
        cfg, p = gpt.config, gpt.plan
        xh = (torch.randn(1, cfg.n_embd, device=dev)).to(torch.bfloat16)
        ssq = torch.rand(1, cfg.n_embd // 16, device=dev) * 16
        y = torch.empty(1, cfg.intermediate_size, device=dev)
        keep.extend([xh, ssq, y])
        for _ in range(reps):
            for l in range(cfg.n_layer):
                args.append(ops.linear(dtype=torch.bfloat16, M=1, N=cfg.intermediate_size, K=cfg.n_embd,
                                       w0=p["fc1"][l], w1=p["fc2"][l], prologue=PRO_SCALED, epilogue=EPI_SWIGLU, x_h=xh, x_ssq=ssq,
                                       eps=cfg.norm_eps, y=y, launch=False))
                # algorithmic bytes: both weight matrices once (bf16) + the operand row (bf16) + its partials (fp32) + y (fp32)
                bytes_total += 2 * cfg.intermediate_size * cfg.n_embd * 2 + cfg.n_embd * 2 + cfg.n_embd // 16 * 4 + cfg.intermediate_size * 4

    add(model.audio_understanding_expert, 1)
    add(model.backbone, 1)
    add(model.audio_generation_expert, 1)
    ops.linear_chain_timed(args, 2)                       # warm
    ms = ops.linear_chain_timed(args, 10)
    per_launch_bytes = bytes_total / len(args)
    achieved = per_launch_bytes / (ms * 1e-3) / 1e9
    return {"kernel": "gemv_kernel<1, 4, 2, 4, true> (scaled-RMSNorm + fc_1/fc_2 + SwiGLU GEMV, 3072 -> 2x8192, bf16)", "bound": "hbm",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            # HBM read bytes per launch from a separate PMC pass on this kernel and shape (profiles/r6_pmc_swiglu.txt:
            # rocprofv3 --pmc FETCH_SIZE, x 1024 x 2 — the gfx950 half-count correction of MI355X_MICROARCH.md §HBM)
            "traffic": PMC_SWIGLU_TRAFFIC, "traffic_source": "profiles/r6_pmc_swiglu.txt (rocprofv3 --pmc FETCH_SIZE pass of round 6 on this kernel and shape, tools/final_round_run.sh; "
                                                              "counters are collected in their own run, not inside bench.py)",
            "launches_per_frame": len(args), "avg_launch_us": round(ms * 1e3, 2),
            "algorithmic_bytes_per_launch": int(per_launch_bytes)}


def scalar_decode_work(cfg, T):
    """Algorithmic bytes / flops of ScalarModel.decode for a (1, latent, T) input (see codec_leg): 1.80 GB and 181.5 GFLOP at the
    placeholder widths, T = 500 — the figures rounds 2-3 measured against."""
    by = fl = wb = 0.0

    def conv(cin, cout, k, tin, tout, res=False):
        nonlocal by, fl, wb
        by += 4.0 * (cin * tin + cout * tout + (cout * tout if res else 0)) + 4.0 * cin * cout * k
        wb += 4.0 * cin * cout * k
        fl += 2.0 * cout * tout * cin * k

    c = cfg["init_channel"] * 2 ** len(cfg["upsample_factors"])
    conv(cfg["latent_hidden_dim"], c, cfg["delay_kernel_size"], T, T)
    for s_, k in zip(cfg["upsample_factors"], cfg["upsample_kernel_sizes"]):
        taps = -(-k // s_)                                   # a transposed conv = s phase filters of ceil(k / s) taps
        by += 4.0 * (c * T + (c // 2) * T * s_) + 4.0 * (s_ * (c // 2)) * c * taps
        wb += 4.0 * (s_ * (c // 2)) * c * taps
        fl += 2.0 * (c // 2) * (T * s_) * c * taps
        c //= 2
        T *= s_
        for _ in range(5):
            if c <= 128:
                by += 4.0 * 3 * c * T + 4.0 * c * c * cfg["res_kernel_size"]
                wb += 4.0 * c * c * cfg["res_kernel_size"]
                fl += 2.0 * c * T * c * cfg["res_kernel_size"]
            else:
                conv(c, c, cfg["res_kernel_size"], T, T)
                conv(c, c, 1, T, T, res=True)
    if cfg["num_samples"] > 1:
        conv(c, c, cfg["default_kernel_size"], T, T * cfg["num_samples"])
        T *= cfg["num_samples"]
    conv(c, cfg["num_bands"], cfg["default_kernel_size"], T, T)
    return {"flop": fl, "bytes": by, "weight_bytes": wb}


def codec_leg(dev, cpu=False):
    """Codec side of the metric ("codec RTF"): the in-scope deterministic stage-2 sub-graph on one 20-s
    window — ScalarModel.decode of a (1, 136, 500) latent -> 480 000 samples — and the RVQ search of a
    10-s clip (125 frames x (1+1+6) levels of 8192 x 32).  Channel widths / strides live in the
    reference's sqcodec_config.yaml, which is not in the repo: placeholder config with hop 960,
    latent 136, init_channel 32."""
    from uniaudio2_amd import ops
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    torch.manual_seed(1)
    sq = ScalarModel(**SCALAR_CFG).to(dev).prepare()
    lat = torch.tanh(torch.randn(1, 136, 500, device=dev))
    # algorithmic work of one decode, a property of the LAYER STACK (not of how this round launches it): per convolution
    # of the reference's decode chain its input + output (+ residual) tensors and its filter once, fp32, and
    # 2 * Cout * Tout * Cin * taps flops — the accounting rounds 2-3 obtained by counting their ua2_conv1d launches (a residual
    # unit of <= 128 channels was one launch there: x, y and the residual read, the k7 filter; its 1 x 1 conv is not counted)
    work = scalar_decode_work(SCALAR_CFG, lat.shape[-1])
    wav = sq.decode(lat)
    launches = {"n": 0}
    saved = {name: getattr(ops, name) for name in ("conv1d", "conv1d_tc", "tc_pack")}
    for name, real in saved.items():
        def counting(*a_, _real=real, **kw_):
            launches["n"] += 1
            return _real(*a_, **kw_)
        setattr(ops, name, counting)
    try:
        sq.decode(lat, use_graph=False)
    finally:
        for name, real in saved.items():
            setattr(ops, name, real)
    work["launches"] = launches["n"]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        wav = sq.decode(lat)
    e1.record(); torch.cuda.synchronize()
    dec_ms = e0.elapsed_time(e1) / 10
    # the same chain with EIGHT windows per launch — what `--codec_batch 8` hands ScalarModel.decode (reason_tokenizer.py
    # detokenize_no_reason_batch: one decode per group of windows): the 512- / 256-channel stages of one window are 1500 / 7500 time
    # steps, grids of 96-200 workgroups; eight windows fill the device.  Algorithmic bytes: activations x 8, filters once.
    NW8 = 8
    lat8 = torch.tanh(torch.randn(NW8, 136, 500, device=dev))
    sq.decode(lat8)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        sq.decode(lat8)
    e1.record(); torch.cuda.synchronize()
    dec8_ms = e0.elapsed_time(e1) / 10
    bytes8 = NW8 * (work["bytes"] - work["weight_bytes"]) + work["weight_bytes"]
    x = torch.randn(125, 32, device=dev)
    emb = torch.randn(6, 8192, 32, device=dev)
    embT = emb.transpose(1, 2).contiguous()
    ops.rvq_encode(x, emb, embT); torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.rvq_encode(x, emb, embT)
    e1.record(); torch.cuda.synchronize()
    rvq_us = e0.elapsed_time(e1) / 10 * 1e3
    res = {"scalar_decode_ms_per_20s_window": round(dec_ms, 3), "scalar_decode_rtf": round(dec_ms / 1e3 / (wav.shape[-1] / 24000.0), 6),
           "scalar_decode_conv_launches": work["launches"], "scalar_decode_gflop": round(work["flop"] / 1e9, 2),
           "scalar_decode_tflops": round(work["flop"] / (dec_ms * 1e-3) / 1e12, 2),
           # the decode-side kernels split every fp32 operand into bf16 hi + lo and issue THREE bf16 MFMAs per product
           # (csrc/ua2_conv.hip): against the dense bf16 peak the issued flops are 3x the useful ones
           "scalar_decode_frac_bf16_mfma_incl_x3": round(3.0 * work["flop"] / (dec_ms * 1e-3) / 2.5e15, 4),
           # algorithmic bytes (every launch's input + output (+ residual) tensors and weights once, fp32) over the time,
           # against the 8 TB/s HBM peak: the roofline that bounds this stack (north_star bar: 0.60)
           "scalar_decode_algorithmic_GBps": round(work["bytes"] / (dec_ms * 1e-3) / 1e9, 1),
           "scalar_decode_frac_hbm": round(work["bytes"] / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "scalar_decode_8_windows_ms_per_window": round(dec8_ms / NW8, 3),
           "scalar_decode_8_windows_frac_hbm": round(bytes8 / (dec8_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "rvq_encode_us_125x6x8192x32": round(rvq_us, 1), "config": "placeholder init_channel=32, hop 960"}
    if cpu:
        res["cpu_baseline"] = codec_cpu_baseline(sq, lat, x, emb, dec_ms, rvq_us, gpu_wav=wav)
        # the north_star's waveform bar (<= 1e-4 RMS against the reference codec on the same input), measured here on the very
        # window the leg times: GPU split-plane decode against the CPU oracle's fp32 decode of the same latent
        for k in ("scalar_decode_rms_vs_cpu_oracle", "scalar_decode_ref_rms", "scalar_decode_rel_rms_vs_cpu_oracle"):
            if k in res["cpu_baseline"]:
                res[k] = res["cpu_baseline"].pop(k)
    return res


def host_cpu_budget():
    """What this process may actually use: logical cores, scheduler affinity, cgroup quota (cpu.max) — `os.cpu_count()` alone
    said 256 on a box where 8 threads were the optimum and 256 threads ran 25x slower."""
    info = {"logical": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        info["affinity"] = info["logical"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    info["cgroup_cpus"] = None if quota is None else round(quota, 2)
    usable = min(info["logical"], info["affinity"])
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.999)))
    info["usable"] = usable
    return info


def codec_cpu_baseline(sq, lat, x, emb, gpu_dec_ms, gpu_rvq_us, gpu_wav=None):
    """the codec half of the metric on the host beside the GPU numbers — `ScalarModel.decode` of the same
    (1, 136, 500) latent through the CPU oracle (oracle/codec_oracle.py, plain PyTorch fp32 convolutions: the reference's own
    arithmetic) and the RVQ search of the same 125 x 6 x 8192 x 32 problem through the C oracle (one thread, the scalar port)."""
    import numpy as np
    from oracle import rvq_oracle
    from oracle.codec_oracle import ScalarOracle
    budget = host_cpu_budget()
    threads = min(budget["usable"], 16)
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in sq.state_dict().items()}
    old = torch.get_num_threads()
    out = {"cores": threads, "kind": "port", "host_cpu_budget": budget}
    try:
        torch.set_num_threads(threads)
        orc = ScalarOracle(sd, SCALAR_CFG)
        with torch.no_grad():
            t0 = time.perf_counter()
            w = orc.decode(lat.detach().cpu())
            dec_s = time.perf_counter() - t0
        out.update({"scalar_decode_s_per_20s_window": round(dec_s, 3), "scalar_decode_rtf": round(dec_s / (w.shape[-1] / 24000.0), 5),
                    "gpu_speedup_scalar_decode": round(dec_s * 1e3 / gpu_dec_ms, 1)})
        if gpu_wav is not None:
            g_, r_ = gpu_wav.detach().cpu().double().reshape(-1), w.double().reshape(-1)
            n_ = min(g_.numel(), r_.numel())
            rms = float((g_[:n_] - r_[:n_]).pow(2).mean().sqrt())
            ref = float(r_[:n_].pow(2).mean().sqrt())
            out.update({"scalar_decode_rms_vs_cpu_oracle": float(f"{rms:.3e}"), "scalar_decode_ref_rms": float(f"{ref:.3e}"),
                        "scalar_decode_rel_rms_vs_cpu_oracle": float(f"{rms / max(ref, 1e-30):.3e}")})
    except Exception as e:  # noqa: BLE001 — an information leg must not take the bench line down
        out["scalar_decode_error"] = repr(e)[:200]
    finally:
        torch.set_num_threads(old)
    try:
        xs, es = np.ascontiguousarray(x.cpu().numpy()), np.ascontiguousarray(emb.cpu().numpy())
        t0 = time.perf_counter()
        rvq_oracle.rvq_encode(xs, es)
        rvq_s = time.perf_counter() - t0
        out.update({"rvq_encode_ms_125x6x8192x32_1thread": round(rvq_s * 1e3, 1), "gpu_speedup_rvq": round(rvq_s * 1e6 / gpu_rvq_us, 1)})
    except Exception as e:  # noqa: BLE001
        out["rvq_error"] = repr(e)[:200]
    out["sample"] = "one 20-s window (1, 136, 500) latent -> 480 000 samples; one 10-s clip's 125 vectors x 6 levels"
    return out


def stage2_leg(dev, steps=10):
    """Codec RTF of the whole stage 2 (`--stage all`'s second half): one 20-s window of semantic codes
    (8, 250) -> RVQ look-ups -> cond_feature_emb -> x2 -> flow-matching DiT at the released shape (32 layers x 1536, 24 heads,
    in 1040 / out 136; classifier-free guidance = batch 2) x `steps` Euler steps (test.sh runs 10) -> SQ-Codec decode ->
    480 000 samples on the host.  Random-init weights, placeholder SQ-Codec widths (the yaml is not in the repo).
    DiT flops per guided step: 2 x 500 rows x 32 layers x 2 x 12 x 1536^2 (attention's own flops excluded)."""
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.AudioDiffusion1D import AudioDiffusion1D
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import RELEASED_CONFIG
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import ReasoningTokenizer
    torch.manual_seed(2)
    model = AudioDiffusion1D(unet_model_config_path=dict(RELEASED_CONFIG), encoder_depth=1)
    with torch.no_grad():
        for _, p_ in model.named_parameters():
            if p_.dim() > 1:
                p_.normal_(0, 0.02)
        for n_, b_ in model.named_buffers():
            if n_.endswith("_codebook.embed"):
                b_.normal_(0, 0.5)
    model = model.to(dev).prepare()
    sq = ScalarModel(**SCALAR_CFG).to(dev).prepare()
    tok = ReasoningTokenizer(sq_codec=sq, model=model, device=dev)
    codes = torch.randint(0, 8192, (8, 250))
    tok.detokenize_no_reason(codes, steps=steps)                   # warm: packs, the recorded solve of this (shape, schedule), the decode graph
    # best of 3: the leg issues ~3000 graph nodes / small launches per window from the host, and on a GPU box whose host cores are
    # shared (cgroup quota) one pass in a few runs 25x slower (1815 ms against 70.5: round-4 evidence run) — a host artefact, not
    # a property of the kernels; every pass is reported
    passes = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        wav = tok.detokenize_no_reason(codes, steps=steps)
        torch.cuda.synchronize()
        passes.append(time.perf_counter() - t0)
    total = min(passes)
    est = model.cfm_wrapper.estimator
    x = torch.randn(2, 500, RELEASED_CONFIG["in_channels"], device=dev)
    est(x, 0.5); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    step_passes = []
    for _ in range(3):
        e0.record()
        for _ in range(5):
            est(x, 0.5)
        e1.record(); torch.cuda.synchronize()
        step_passes.append(e0.elapsed_time(e1) / 5)
    step_ms = min(step_passes)
    flop = 2.0 * 500 * 32 * 2 * 12 * 1536 ** 2
    # encode side of the same model (config 2's "codec encode", everything behind the frozen SSL encoders): one 30-s segment
    # — what a 10-s clip costs since audio2token stopped encoding the segment the reference discards — with synthetic features
    # of the released shapes (Whisper 1024 x 1500 and WavLM 768 x 1500 at 50 Hz, BEST-RQ 1024 x 750 at 25 Hz)
    enc = {}
    try:
        g = torch.Generator(device="cpu").manual_seed(3)
        f = [torch.randn(1, c, t, generator=g).to(dev) for c, t in ((1024, 1500), (768, 1500), (1024, 750), (1024, 750))]
        masks = torch.zeros(3, 1, dtype=torch.bool)
        model.fetch_codes_from_features(*f, film_masks=masks)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(8):
            rc, mc, _ = model.fetch_codes_from_features(*f, film_masks=masks)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t1) / 8 * 1e3
        enc = {"encode_post_ssl_ms_per_30s_segment": round(ms, 2), "encode_post_ssl_clips_per_s": round(1e3 / ms, 1),
               "reason_tokens": list(rc[0].shape), "semantic_tokens": list(mc[0].shape)}
    except Exception as e:  # noqa: BLE001 — an information leg must not take the bench line down
        enc = {"encode_post_ssl_error": repr(e)[:200]}
    # BASELINE config 3's codec-encode half at its batch: 32 clips x 10 s = 32 segments in one call (32 x (125 x 8 +
    # 51 x 8) RVQ searches + the strided down-samplers on (32, 1024, 1500) features), AudioDiffusion1D.py:526-550
    try:
        g = torch.Generator(device="cpu").manual_seed(4)
        f = [torch.randn(32, c, t, generator=g).to(dev) for c, t in ((1024, 1500), (768, 1500), (1024, 750), (1024, 750))]
        masks = torch.zeros(3, 32, dtype=torch.bool)
        model.fetch_codes_from_features(*f, film_masks=masks)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            rc, mc, _ = model.fetch_codes_from_features(*f, film_masks=masks)
        torch.cuda.synchronize()
        ms32 = (time.perf_counter() - t1) / 3 * 1e3
        enc.update({"config3_encode_post_ssl_ms_per_batch32": round(ms32, 2), "config3_encode_post_ssl_clips_per_s": round(32e3 / ms32, 1)})
    except Exception as e:  # noqa: BLE001
        enc["config3_encode_post_ssl_error"] = repr(e)[:200]
    # stage 2 batched over utterances (multi_task_inference.py:540-548's loop as `--codec_batch 8`): window k of 8
    # utterances in one flow-matching solve (2 x 8 x 500 rows per DiT launch: the 256-row-tile order-free GEMM) + one SQ-Codec
    # decode of 8 latents; the solve is one recorded graph.  ms_per_window = the pass / 8.
    bat = {}
    try:
        P = 8
        codes8 = [torch.randint(0, 8192, (8, 250)) for _ in range(P)]
        tok.detokenize_no_reason_batch(codes8, steps=steps, max_batch=P)          # warm: records the (P, steps) solve and the batch decode
        bp = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            wavs = tok.detokenize_no_reason_batch(codes8, steps=steps, max_batch=P)
            torch.cuda.synchronize()
            bp.append(time.perf_counter() - t0)
        x16 = torch.randn(2 * P, 500, RELEASED_CONFIG["in_channels"], device=dev)
        est(x16, 0.5); torch.cuda.synchronize()
        sp = []
        for _ in range(3):
            e0.record()
            for _ in range(3):
                est(x16, 0.5)
            e1.record(); torch.cuda.synchronize()
            sp.append(e0.elapsed_time(e1) / 3)
        med = sorted(bp)[1]
        bat = {"stage2_batched8": {"utterances": P, "ms_per_window_passes": [round(p_ * 1e3 / P, 2) for p_ in bp], "ms_per_window": round(med * 1e3 / P, 2),
                                   "ms_per_window_min": round(min(bp) * 1e3 / P, 2), "rtf": round(med / (P * wavs[0].shape[-1] / 24000.0), 5),
                                   "dit_ms_per_guided_step_8_windows_passes": [round(v, 2) for v in sp], "dit_ms_per_guided_step_per_window": round(sorted(sp)[1] / P, 3),
                                   "dit_tflops": round(P * flop / (sorted(sp)[1] * 1e-3) / 1e12, 1), "dit_frac_bf16_mfma_peak": round(P * flop / (sorted(sp)[1] * 1e-3) / 2.5e15, 4),
                                   "gemm": "order-free 256-row tiles (csrc/ua2_gemm2.hip), sum_order = UA2_SUM_ORDER_FREE"}}
    except Exception as e:  # noqa: BLE001
        bat = {"stage2_batched8": {"error": repr(e)[:300]}}
    return {**enc, **bat, "euler_steps": steps, "window_s": 20.0, "ms_per_window_passes": [round(p_ * 1e3, 1) for p_ in passes],
            "ms_per_window_median": round(sorted(passes)[1] * 1e3, 1), "dit_ms_per_guided_step_median": round(sorted(step_passes)[1], 2),
            "dit_ms_per_guided_step_passes": [round(p_, 2) for p_ in step_passes], "ms_per_window": round(total * 1e3, 1), "rtf": round(total / (wav.shape[-1] / 24000.0), 5),
            "dit_ms_per_guided_step": round(step_ms, 2), "dit_tflops": round(flop / (step_ms * 1e-3) / 1e12, 1),
            "dit_frac_bf16_mfma_peak": round(flop / (step_ms * 1e-3) / 2.5e15, 4)}


# GEMM flops per sequence and frame that the depth decoder's per-id tables replace by gathers (ua2_stage3.hip; 0 with UA2_NO_PROJ_TABLE=1)
TABLE_FLOP = 0.0 if os.environ.get("UA2_NO_PROJ_TABLE") else 7 * 2.0 * 3072 * 2048 * (1 if os.environ.get("UA2_NO_QKV_TABLE") else 2)


def batched_leg(model, dev, B=64, frames=24, max_seq=2048, order_free_rows=0):
    """Information beside the B = 1 headline: one GPU decoding B = 64 sequences together
    (32..33-token prompts, greedy, same kernels; rows bit-identical to their B = 1 runs, tests/test_gpu_configs.py).
    Re-plans the caches for 64 sequences, so it runs last."""
    model.setup_caches(B, dtype=torch.bfloat16, max_seq_length=max_seq, max_rows=B * PROMPT_LEN, log_frames=2 * frames + 8)
    model.set_order_free_rows(order_free_rows)
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
        ms = e1.elapsed_time(e2) / frames
        res = {"B": B, "prefill_rows": B * (PROMPT_LEN - 1), "prefill_ms": round(e0.elapsed_time(e1), 2),
               "decode_ms_per_frame": round(ms, 3), "audio_tokens_per_s": round(8 * B * frames / (e1.elapsed_time(e2) * 1e-3), 1),
               # 11.8 GFLOP per sequence and frame against the dense bf16 MFMA peak — minus what the per-id tables of
               # the depth decoder no longer compute (7 x (projection 3072 -> 2048 + layer 0's q|k|v 2048 -> 3072) = 0.176 GFLOP, round 6)
               "decode_gemm_frac_bf16_mfma_peak": round((11.8e9 - TABLE_FLOP) * B / (ms * 1e-3) / 2.5e15, 4),
               "decode_frac_hbm_streamed_weights": round((11.8e9 - TABLE_FLOP) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    try:   # the same frames with the text head skipped (what the batch generator runs; identical audio ids)
        model.generate_frames(2, B, 0, reason_eos=-1, reason_card=REASON_CARD, skip_text_head=True)
        e1.record()
        model.generate_frames(frames - 4, B, 0, reason_eos=-1, reason_card=REASON_CARD, skip_text_head=True)
        e2.record(); torch.cuda.synchronize()
        res["decode_ms_per_frame_skip_text_head"] = round(e1.elapsed_time(e2) / (frames - 4), 3)
    except Exception as e:  # noqa: BLE001
        res["decode_ms_per_frame_skip_text_head"] = repr(e)[:120]
    return res


def config4_leg(model, dev, world, rank, per_rank=64, seed=0, skip_text_head=True):
    """BASELINE.json config 4 / SURVEY.md §8d: batched TTS, 64 ragged prompts PER GPU (512 over 8 GPUs), prompt lengths
    uniform[24, 48], frames to generate uniform[60, 300] (deterministic stop), sharded longest-first round-robin over the ranks
    (uniaudio2_amd/parallel.py), each rank decoding its shard as ONE continuous batch (sequences retire the frame they finish)
    and the shard ending in the path's only collective — the fixed-shape RCCL all-gather of the token tensors — INSIDE the
    timed region.  Weak scaling: the per-GPU work is fixed (the global prompt list grows with the world size).  Timed like the
    headline: barrier + synchronize on both sides, max over ranks; the first pass (graph captures for every live-batch size
    the retirement schedule visits) is untimed."""
    import numpy as np
    from uniaudio2_amd import parallel
    n_total = per_rank * world
    rs = np.random.RandomState(seed)
    lens = rs.randint(24, 49, size=n_total)
    nfr = rs.randint(60, 301, size=n_total)
    items = []
    for i in range(n_total):
        g = torch.Generator().manual_seed(4000 + i)
        t = torch.zeros(int(lens[i]), 9, dtype=torch.long)
        t[:, -1] = torch.randint(0, 128000, (int(lens[i]),), generator=g)
        m = torch.zeros(int(lens[i]), 9, dtype=torch.bool)
        m[:, -1] = True
        items.append((t.to(dev), m.to(dev), int(nfr[i])))
    model.setup_caches(per_rank, dtype=torch.bfloat16, max_seq_length=512, max_rows=4096, log_frames=320)

    def generate_batch(chunk):
        ids = model.generate_ragged([(t, m) for t, m, _ in chunk], [n for _, _, n in chunk], mode=0, reason_eos=-1, reason_card=REASON_CARD,
                                    skip_text_head=skip_text_head)   # as the product's batch generator runs it (identical audio ids)
        out = []
        for o in ids:                                        # (T, 9) int32 -> the (reason (8, T_r), semantic (8, T_s)) contract
            a = o[:, 1:].t().contiguous()
            k = min(a.shape[1], 50)
            out.append((a[:, :k], a[:, k:]))
        return out

    def fence():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    dt = 0.0
    for rep in range(2):
        fence()
        t0 = time.perf_counter()
        res = parallel.run_sharded_batched(items, nfr.tolist(), generate_batch, per_rank)
        fence()
        dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ok = sum(1 for v in res.values() if not isinstance(v, parallel.Failed))
    frames = int(nfr.sum())
    return {"prompts_total": n_total, "prompts_per_gpu": per_rank, "n_gpus": world, "frames_total": frames, "gathered_ok": ok,
            "seconds": round(dt, 4), "audio_tokens_per_s": round(8 * frames / dt, 1),
            "audio_tokens_per_s_per_gpu": round(8 * frames / dt / world, 1), "max_frames_of_a_sequence": int(nfr.max()),
            "text_head": "skipped on the audio-feedback frames (UA2_FRAME_SKIP_TEXT_HEAD, what evaluation/_generator.py runs: identical audio ids)" if skip_text_head else "computed every frame (the reference's work)",
            "scaling": "weak", "exchange": ("one fixed-shape int32 all-gather of the token tensors per shard (parallel.gather_results: pack -> device -> "
                                            f"dist.all_gather on `{_dist_backend()}` -> parse), inside the timed region, world = {world}" if _dist_backend() else
                                            "none (no process group: a plain `python bench.py` run; under torchrun — any --gpus N, one rank included — "
                                            "parallel.gather_results runs one int32 all-gather per shard)")}


def _dist_backend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else ""


def cpu_baseline_leg(model, tokens, mask, frames_all=20, frames_t1=3):
    """the CPU oracle (fp32 port of the reference algorithm as shipped: full-2048 masked prefill,
    repeat_interleave GQA, lm_head + 8-step local decoder every frame) on this host's cores, same weights, same prompt,
    at two thread settings — T1 (`OMP_NUM_THREADS=1`, what the reference's path.sh:3 imposes) and Tall (the fastest of a
    small sweep up to every core: at 128 threads the round-1 run was SLOWER than 8 threads of the authoring container,
    small GEMVs over-subscribe) — prefill and decode timed separately, decode = median over the per-frame times.
    `value` is the whole-utterance rate the GPU `value` is quoted on (33-token prompt + 74 frames), composed from the
    measured Tall prefill time and median frame time.  Bounded: ~3 frames at T1, ~20 at Tall."""
    import statistics
    import concurrent.futures
    import types
    import torch.nn.functional as TF
    from oracle import lm_oracle
    from oracle.lm_oracle import GPTShape, Stage3Oracle
    # Decode frames are M = 1 products: on the GPU boxes' hosts torch's sgemv ran on ONE core whatever set_num_threads said
    # (round 3: 702.7 ms/frame at T1 vs 688.4 at 16 threads while the prefill scaled 4.9x) — "Tall" was not a multi-thread
    # number.  Same arithmetic (each output row is one fp32 dot product by the same routine), row blocks dealt to a pool of
    # `par["threads"]` workers; many-row products (prefill) keep torch's own threading.
    par = {"threads": 1, "pool": None}

    def par_linear(x, w, bias=None):
        n = par["threads"]
        if n <= 1 or x.numel() != x.shape[-1] or w.shape[0] < 1024 or par["pool"] is None:
            return TF.linear(x, w, bias)
        blocks = w.chunk(n, 0)
        outs = list(par["pool"].map(lambda wb: TF.linear(x, wb), blocks))
        y = torch.cat(outs, dim=-1)
        return y if bias is None else y + bias

    lm_oracle.F = types.SimpleNamespace(linear=par_linear, scaled_dot_product_attention=TF.scaled_dot_product_attention, silu=TF.silu)
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in model.state_dict().items()}
    shapes = dict(backbone=GPTShape(28, 3072, 24, 8, 8192), understanding=GPTShape(3, 3072, 24, 8, 8192),
                  generation=GPTShape(2, 3072, 24, 8, 8192), decoder=GPTShape(4, 2048, 32, 8, 8192))
    o = Stage3Oracle(sd, shapes, SEM_CARD, REASON_CARD, 8, mode="fp32")
    o.setup_caches(1)
    tk, mk = tokens.cpu(), mask.cpu()
    L = tk.size(1)
    pos = torch.arange(0, L).unsqueeze(0)
    budget = host_cpu_budget()
    ncores = budget["usable"]                      # affinity / cgroup-limited, not os.cpu_count()

    def set_threads(t):
        """prefill: torch's intra-op threads; decode GEMVs: `t` pool workers with one torch thread each"""
        if par["pool"] is not None:
            par["pool"].shutdown()
        par["threads"], par["pool"] = t, (concurrent.futures.ThreadPoolExecutor(t) if t > 1 else None)

    def run(threads, frames, prefill=True):
        """-> (prefill seconds or None, per-frame seconds list, (frames, 9) ids)"""
        torch.set_num_threads(threads)
        set_threads(1)
        o.reset_caches()
        t0 = time.perf_counter()
        o.forward_prefix(tk[:, :-1], mk, pos[:, :-1])
        pre = time.perf_counter() - t0
        torch.set_num_threads(1)
        set_threads(threads)
        ct, cm = tk[:, -1:], mk[:, -1:]
        cur, maxp1, per, ids = torch.full((1,), L - 1, dtype=torch.long), L, [], []
        for _ in range(frames):
            t0 = time.perf_counter()
            smp = o.generate_frame(ct, cm, cur, maxp1)
            per.append(time.perf_counter() - t0)
            ids.append(smp)
            audio, text_tok = smp[:, 1:].long(), smp[:, 0:1].long()
            ct = torch.cat([audio, text_tok], dim=-1).unsqueeze(1)
            cm = torch.cat([torch.ones_like(audio).bool(), torch.zeros(1, 1).bool()], dim=1).unsqueeze(1)
            cur, maxp1 = cur + 1, maxp1 + 1
        return pre, per, torch.stack(ids)

    old = torch.get_num_threads()
    try:
        # Tall: pick the thread count by one frame each (no prefill cost: the cache content does not change the timing)
        cands = sorted({t for t in (2, 4, 8, 16, 32, 64, ncores // 2, ncores) if 1 <= t <= ncores})
        o.reset_caches()
        sweep = {}
        for t in cands:
            torch.set_num_threads(1)
            set_threads(t)
            ct, cm = tk[:, -1:], mk[:, -1:]
            o.generate_frame(ct, cm, torch.tensor([L - 1]), L)                       # warm
            t0 = time.perf_counter()
            o.generate_frame(ct, cm, torch.tensor([L - 1]), L)
            sweep[t] = time.perf_counter() - t0
        best = min(sweep, key=sweep.get)
        pre_all, per_all, ids = run(best, frames_all)
        pre_1, per_1, _ = run(1, frames_t1)
        # bf16 contract on the host: 8 free-running frames of the bf16 oracle (what tests/test_gpu_fullsize.py teacher-forces)
        # for the id-agreement count of the bench line
        bf16_ids = None
        try:
            ob = Stage3Oracle(sd, shapes, SEM_CARD, REASON_CARD, 8, mode="bf16")
            ob.setup_caches(1)
            torch.set_num_threads(best)
            set_threads(1)
            ob.forward_prefix(tk[:, :-1], mk, pos[:, :-1])
            torch.set_num_threads(1)
            set_threads(best)
            ct, cm = tk[:, -1:], mk[:, -1:]
            cur, maxp1, got = torch.full((1,), L - 1, dtype=torch.long), L, []
            for _ in range(8):
                smp = ob.generate_frame(ct, cm, cur, maxp1)
                got.append(smp)
                audio, text_tok = smp[:, 1:].long(), smp[:, 0:1].long()
                ct = torch.cat([audio, text_tok], dim=-1).unsqueeze(1)
                cm = torch.cat([torch.ones_like(audio).bool(), torch.zeros(1, 1).bool()], dim=1).unsqueeze(1)
                cur, maxp1 = cur + 1, maxp1 + 1
            bf16_ids = torch.stack(got)[:, 0]
            del ob
        except Exception as e:  # noqa: BLE001 — information only
            bf16_ids = repr(e)[:200]
    finally:
        torch.set_num_threads(old)
        set_threads(1)
        lm_oracle.F = TF
    f_all, f_1 = statistics.median(per_all), statistics.median(per_1)
    whole = 8 * FRAMES / (pre_all + FRAMES * f_all)
    return {"value": round(whole, 2), "unit": "audio tokens/s", "cores": best, "kind": "port",
            "host_cpu_budget": budget,
            "sample": f"fp32 oracle, B=1: {L - 1}-row prefill + {frames_all} greedy frames at Tall={best} threads (host budget: "
                      f"{budget['logical']} logical cores, affinity {budget['affinity']}, cgroup cpu.max {budget['cgroup_cpus']} -> {ncores} usable; "
                      f"sweep {{{', '.join(f'{t}: {v * 1e3:.0f} ms/frame' for t, v in sweep.items())}}}), "
                      f"{frames_t1} frames at T1; value = 8*{FRAMES} tokens / (prefill + {FRAMES} x median frame)",
            "tall": {"threads": best, "prefill_s": round(pre_all, 3), "ms_per_frame_median": round(f_all * 1e3, 1),
                     "decode_audio_tokens_per_s": round(8 / f_all, 2)},
            "t1": {"threads": 1, "prefill_s": round(pre_1, 3), "ms_per_frame_median": round(f_1 * 1e3, 1),
                   "decode_audio_tokens_per_s": round(8 / f_1, 2)},
            "note": "BASELINE.md §3 measured the imported reference itself at 374 ms/frame on 8 threads of the authoring "
                    "container (21 audio tokens/s); round 1 reported 2.3 tokens/s here because it ran the port on all 128 "
                    "threads only (over-subscribed GEMVs) and folded the 2048-slot prefill into 6 frames",
            "decode_threading": "M = 1 products are dealt to a pool of `cores` workers by row blocks (torch's own sgemv ran them on one core: "
                                "round 3's Tall == T1); prefill uses torch's intra-op threads"}, ids[:, 0], bf16_ids


def config3_leg(model, dev, B=32, n_text=15, n_reason=53, n_sem=128, frames=32, order_free_rows=0):
    """SURVEY.md §8d config 3 (ASR batch of 32 x 10-s clips), LLM half: per clip a prompt of 15 text frames + 53 reason
    frames + 128 semantic frames (L = 196; audio ids uniform in the valid card), one ragged prefill of 32 x 195 rows,
    then 32 greedy TEXT frames for all 32 sequences together (on-device text loop, depth decoder skipped —
    asr_task.py:668-673's discarded work, identical text ids).  The prefill is MFMA-bound: its `roofline` counts the
    2*M*N*K flops of the trunk's five Linear layers per layer over the 6240 rows (attention and heads excluded)."""
    g = torch.Generator().manual_seed(303)
    L = n_text + n_reason + n_sem
    prompts = []
    for b in range(B):
        t = torch.zeros(L, 9, dtype=torch.long)
        m = torch.zeros(L, 9, dtype=torch.bool)
        t[:n_text, -1] = torch.randint(0, 128000, (n_text,), generator=g); m[:n_text, -1] = True
        t[n_text:n_text + n_reason, :8] = torch.randint(0, REASON_CARD, (n_reason, 8), generator=g)
        t[n_text + n_reason:, :8] = REASON_CARD + torch.randint(0, SEM_CARD, (n_sem, 8), generator=g)
        m[n_text:, :8] = True
        prompts.append((t.to(dev), m.to(dev)))
    rows = B * (L - 1)
    model.setup_caches(B, dtype=torch.bfloat16, max_seq_length=2048, max_rows=rows, log_frames=frames + 8)
    model.set_order_free_rows(order_free_rows)             # 0 = the default plan: every row keeps the bits of its single-sequence run
    res = {}
    for rep in range(2):                                   # first pass warms / captures
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        model.begin_ragged(prompts)
        e1.record()
        model.generate_frames(frames, B, 1)
        e2.record()
        torch.cuda.synchronize()
        pre_ms, dec_ms = e0.elapsed_time(e1), e1.elapsed_time(e2)
    flop = 0.0
    for gpt in (model.audio_understanding_expert, model.backbone, model.audio_generation_expert):
        c = gpt.config
        per_row = 2.0 * (c.n_embd * (c.n_head + 2 * c.n_query_groups) * c.head_size + c.n_head * c.head_size * c.n_embd +
                         3 * c.n_embd * c.intermediate_size)
        flop += per_row * c.n_layer * rows
    # what the text generators run (evaluation/_generator.py _generate_text): from the second text frame on the understanding /
    # generation experts are not computed (UA2_FRAME_SKIP_AUDIO_EXPERTS: masked-out outputs, caches never read again) — same text ids
    ids_full = model._st["frame_log"][:frames, :B, 0].clone()
    for rep in range(2):
        torch.cuda.synchronize()
        model.begin_ragged(prompts)
        e1.record()
        log = model.generate_frames(frames, B, 1, skip_audio_experts=True)
        e2.record()
        torch.cuda.synchronize()
        dec_skip_ms = e1.elapsed_time(e2)
    same_ids = bool(torch.equal(log[:, :, 0], ids_full))
    tf = flop / (pre_ms * 1e-3) / 1e12
    return {"B": B, "prompt_len": L, "prefill_rows": rows, "prefill_ms": round(pre_ms, 2),
            "text_frames": frames, "decode_ms_per_frame": round(dec_ms / frames, 3),
            "text_tokens_per_s": round(B * frames / (dec_ms * 1e-3), 1),
            "clips_per_s_llm_half": round(B / ((pre_ms + dec_ms) * 1e-3), 2),
            "decode_ms_per_frame_skip_audio_experts": round(dec_skip_ms / frames, 3),
            "clips_per_s_llm_half_skip_audio_experts": round(B / ((pre_ms + dec_skip_ms) * 1e-3), 2),
            "skip_audio_experts_text_ids_identical": same_ids,
            "roofline": {"kernel": ("trunk prefill: prep + order-free 256-row-tile MFMA GEMM (ua2_gemm2.hip; opt-in set_order_free_rows(%d)) x 5 Linear x 33 layers, attention included in the time" % order_free_rows)
                                   if order_free_rows else "trunk prefill: prep + row-invariant 128x128 tiled MFMA GEMM (ua2_gemm.hip) x 5 Linear x 33 layers, attention included in the time",
                         "bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(tf / 2500.0, 4), "traffic": None, "gemm_tflop": round(flop / 1e12, 2)}}


def config5_leg(model, dev, frames=500, prompt_len=35):
    """SURVEY.md §8d config 5 (TTM, the reference's 500-frame cap): 35-token prompt, 500 greedy frames, S grows 35 -> 535
    in the 2048-slot cache; ms/frame over the first and the last 50 frames shows what the growing KV costs."""
    model.setup_caches(1, dtype=torch.bfloat16, max_seq_length=2048, max_rows=64, log_frames=frames + 8)
    g = torch.Generator().manual_seed(505)
    t = torch.zeros(1, prompt_len, 9, dtype=torch.long)
    t[0, :, -1] = torch.randint(0, 128000, (prompt_len,), generator=g)
    m = torch.zeros(1, prompt_len, 9, dtype=torch.bool)
    m[0, :, -1] = True
    t, m = t.to(dev), m.to(dev)
    utterance(model, t, m, frames=4)                      # warm / capture
    torch.cuda.synchronize()
    model.reset_caches()
    pos = torch.arange(prompt_len, device=dev).unsqueeze(0)
    model.forward_prefix(t[:, :-1], tokens_mask=m, input_pos=pos[:, :-1])
    model.begin_decode(t[:, -1:], m[:, -1:], torch.tensor([prompt_len - 1], device=dev))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    model.generate_frames(50, 1, 0, reason_eos=-1, reason_card=REASON_CARD)
    ev[1].record()
    model.generate_frames(frames - 100, 1, 0, reason_eos=-1, reason_card=REASON_CARD)
    ev[2].record()
    model.generate_frames(50, 1, 0, reason_eos=-1, reason_card=REASON_CARD)
    ev[3].record()
    torch.cuda.synchronize()
    total = ev[0].elapsed_time(ev[3])
    return {"frames": frames, "prompt_len": prompt_len, "ms_per_frame_first50": round(ev[0].elapsed_time(ev[1]) / 50, 3),
            "ms_per_frame_last50": round(ev[2].elapsed_time(ev[3]) / 50, 3), "ms_per_frame_mean": round(total / frames, 3),
            "audio_tokens_per_s": round(8 * frames / (total * 1e-3), 1)}


def _respawn_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, RCCL), with the
    rendezvous on 127.0.0.1 as the environment requires, and hand their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-legs", action="store_true", help="skip the codec / batched / config-3 / config-5 information legs")
    ap.add_argument("--config4-leg", action="store_true", help="with --no-legs: still run the config-4 leg (the product's sharded runner + its all-gather)")
    a = ap.parse_args()

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not launched:
        if torch.cuda.device_count() < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        _respawn_under_torchrun(a.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = launched                                          # under torchrun the RCCL path runs even with one rank
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
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
    model.generate_frames(64, 1, 0, reason_eos=-1, reason_card=REASON_CARD)
    e1.record()
    torch.cuda.synchronize()
    ms_frame = e0.elapsed_time(e1) / 64
    # per-frame latency distribution (p50 / p99): 200 frames launched one by one, an event pair around each
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
    utterance(model, tokens, mask, frames=2)
    evs[0].record()
    model.generate_frames(100, 1, 0, reason_eos=-1, reason_card=REASON_CARD, frame_events=evs[1:])
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(100))
    p50, p99 = per[49], per[98]

    # the same utterance with the text head skipped on the audio-feedback frames (UA2_FRAME_SKIP_TEXT_HEAD: identical audio ids,
    # tests/test_gpu_lm.py) — what the product's generators run; `value` above keeps the reference's work (lm_head every frame)
    skip = {}
    try:
        def utt_skip(frames=FRAMES):
            return utterance(model, tokens, mask, frames=frames, skip_text_head=True)
        ref_log = utterance(model, tokens, mask)
        got = utt_skip()
        torch.cuda.synchronize()
        same = bool(torch.equal(ref_log[:, :, 1:], got[:, :, 1:]))
        t1 = time.perf_counter()
        for _ in range(a.steps):
            utt_skip()
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        utt_skip(2)
        e0.record()
        model.generate_frames(64, 1, 0, reason_eos=-1, reason_card=REASON_CARD, skip_text_head=True)
        e1.record(); torch.cuda.synchronize()
        skip = {"value_skip_text_head": round(8 * FRAMES * a.steps / dts, 1), "decode_ms_per_frame_skip_text_head": round(e0.elapsed_time(e1) / 64, 3),
                "skip_text_head_audio_ids_identical": same}
    except Exception as e:  # noqa: BLE001 — information
        skip = {"value_skip_text_head": repr(e)[:200]}
    audio_tokens = 8 * FRAMES * a.steps * world
    res = {"metric": "audio tokens/sec (TTS greedy)", "value": round(audio_tokens / dt, 1), "unit": "audio tokens/s",
           "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 2),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": "TTS --stage all (BASELINE config 2), single utterance, greedy (topk=1), bf16, B=1 per GPU: "
                                  f"{PROMPT_LEN}-token prompt + {FRAMES} frames x (8 audio + 1 text) tokens; `value` = the stage-1 "
                                  "audio-token rate (multi_task_inference.py:486-525), stage 2 (codes -> waveform, :527-548) = "
                                  "top-level `codec_rtf`, `stage_all_ms_per_utterance` = both; "
                                  "Llama-3.2-3B backbone + 3L/2L experts + 4L local decoder x8, V_a=12296, random init",
                      "parallelism": (f"dp{world} (one utterance per GPU, RCCL all-gather of token tensors inside the timed region)" if use_dist else
                                      "dp1, no collective (a single rank outside torchrun; `torchrun --nproc-per-node=1 bench.py --gpus 1` runs the RCCL path)")},
           **skip,
           "decode_ms_per_frame": round(ms_frame, 3), "decode_frames_per_s": round(1e3 / ms_frame, 1),
           "decode_ms_per_frame_p50": round(p50, 3), "decode_ms_per_frame_p99": round(p99, 3)}
    solo = rank == 0 and world == 1
    if solo and not a.no_legs:
        # the contract on which ids are IDENTICAL to the reference's (UA2_F32: the reference ships fp32, multi_task_inference.py:181-183)
        try:
            model.setup_caches(1, dtype=torch.float32, max_seq_length=2048, max_rows=64, log_frames=128)
            utterance(model, tokens, mask)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                utterance(model, tokens, mask)
            torch.cuda.synchronize()
            res["fp32_contract_audio_tokens_per_s"] = round(8 * FRAMES * 2 / (time.perf_counter() - t0), 1)
        except Exception as e:  # noqa: BLE001
            res["fp32_contract_audio_tokens_per_s"] = repr(e)[:200]
        model.setup_caches(1, dtype=torch.bfloat16, max_seq_length=2048, max_rows=64, log_frames=128)
    if solo and not a.no_roofline:
        res["roofline"] = roofline_leg(model)
        # whole-frame view of the same roofline: unique weight bytes a frame must stream (bf16)
        res["frame_hbm_frac_unique_weights"] = round(8.33e9 / (ms_frame * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if solo and not a.no_cpu_baseline:
        cb, cpu_ids, bf16_ids = cpu_baseline_leg(model, tokens, mask)
        res["cpu_baseline"] = cb
        n = cpu_ids.shape[0]
        res["cpu_fp32_vs_gpu_bf16_same_ids_frames"] = int((cpu_ids.int() == log[:n, 0].cpu().int()).all(-1).sum())
        if torch.is_tensor(bf16_ids):
            # free-running: after the first differing id the two runs see different inputs, so this is a lower bound on agreement
            eq = (bf16_ids.int() == log[:bf16_ids.shape[0], 0].cpu().int())
            first = int((~eq.all(-1)).nonzero()[0]) if not bool(eq.all()) else int(eq.shape[0])
            res["bf16_free_running_ids_equal_bf16_oracle_first8frames"] = {"equal": int(eq.sum()), "of": int(eq.numel()), "frames_before_first_difference": first}
        else:
            res["bf16_free_running_ids_equal_bf16_oracle_first8frames"] = {"error": bf16_ids}
        res["parity_notes"] = ("bf16 ids are asserted teacher-forced against the bf16 oracle (tests/test_gpu_fullsize.py); the live "
                               "codec's ResidualVQ is restated from vector_quantize_pytorch==1.27.15's published algorithm and is "
                               "parity-UNPINNED against the package itself (absent here)")
    if (not a.no_legs or a.config4_leg) and (world > 1 or solo):
        # the named multi-GPU config (BASELINE.json configs[3]): every rank takes part (collective inside)
        res["config4_batched_tts"] = config4_leg(model, dev, world, rank)
    if solo and not a.no_legs:
        res["codec"] = codec_leg(dev, cpu=not a.no_cpu_baseline)
        res["codec"]["stage2_codes_to_wav"] = stage2_leg(dev)
        s2 = res["codec"]["stage2_codes_to_wav"]
        res["codec_rtf"] = s2.get("rtf")                         # stage 2 of `--stage all`: wall / audio seconds of one 20-s window
        res["codec_rtf_scalar_decode_only"] = res["codec"].get("scalar_decode_rtf")
        if s2.get("ms_per_window") is not None:                 # a 74-frame utterance decodes as one window (<= 250 semantic frames)
            res["stage_all_ms_per_utterance"] = round(dt / a.steps * 1e3 + s2["ms_per_window"], 2)
        res["config5_ttm_500_frames"] = config5_leg(model, dev)
        res["batched_decode"] = batched_leg(model, dev)
        res["batched_decode_256"] = batched_leg(model, dev, B=256, frames=12, max_seq=128)
        try:   # where the decode GEMMs stop being an HBM problem: 1024 live sequences (serving regime; the tiled MFMA GEMM takes every Linear)
            res["batched_decode_1024"] = batched_leg(model, dev, B=1024, frames=6, max_seq=64)
        except Exception as e:  # noqa: BLE001 — information leg
            res["batched_decode_1024"] = {"error": repr(e)[:200]}
        try:   # the same frame with the trunk's launches opted into the order-free GEMM (set_order_free_rows: rows >= 1024)
            res["batched_decode_1024_order_free"] = batched_leg(model, dev, B=1024, frames=6, max_seq=64, order_free_rows=1024)
        except Exception as e:  # noqa: BLE001
            res["batched_decode_1024_order_free"] = {"error": repr(e)[:200]}
        res["config3_asr_batch32"] = config3_leg(model, dev)
        try:   # ... and config 3's prefill (6240 rows) with it: what the opt-in buys, beside the default plan's line above
            res["config3_asr_batch32_order_free"] = config3_leg(model, dev, order_free_rows=2048)
        except Exception as e:  # noqa: BLE001
            res["config3_asr_batch32_order_free"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""GPU parity of the codec's neural stages (§8a a15, a18) and the BASELINE config-1 plumbing.

Oracles: oracle/codec_model_oracle.py + oracle/codec_oracle.py, both pinned on outputs of the reference's own code
(tests/test_oracle_codec_model.py, tests/test_oracle_codec.py); the goldens themselves (tests/golden/codec_model_toy.npz)
are compared directly where they apply.  Unpinned pieces (diffusers DiT, vector_quantize_pytorch RVQ, torchaudio
resampler) are compared with the oracle's restatement of the published algorithms."""
import json
import os

import zlib

import numpy as np
import pytest
import torch

from codec_model_stub import CFG, StubEstimator, fetch_inputs, infer_inputs, module_state_dict, think_inputs
from weights import seeded_tensor

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(__file__), "golden")


def _gold():
    return np.load(os.path.join(HERE, "codec_model_toy.npz")), json.load(open(os.path.join(HERE, "codec_model_toy.json")))


def _toy_model(meta, dit=False, vq_seed=None):
    """AudioDiffusion1D mirror at the goldens' toy sizes with the goldens' weights (audio_thinking.* seed 301, layers 302)."""
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.AudioDiffusion1D import AudioDiffusion1D
    import dit_toy
    c = CFG
    m = AudioDiffusion1D(whisper_fea_dim=c["Cw"], wavlm_fea_dim=c["Cl"], codec_dim=c["D"], encoder_depth=c["depth"],
                         unet_model_config_path=dict(num_attention_heads=dit_toy.CFG["heads"], attention_head_dim=dit_toy.CFG["head_dim"],
                                                     in_channels=dit_toy.CFG["in_channels"], out_channels=dit_toy.CFG["out_channels"],
                                                     num_layers=dit_toy.CFG["layers"]) if dit else None)
    m.sq_codec_latent = c["latent"]
    sd = m.state_dict()
    new = {"audio_thinking." + k: v for k, v in module_state_dict({k: tuple(s) for k, s in meta["think_keys"]}, 301).items()}
    new.update(module_state_dict({k: tuple(s) for k, s in meta["fetch_keys"]}, 302))
    if dit:
        new.update({"cfm_wrapper.estimator." + k: v for k, v in dit_toy.state_dict(5).items()})
    rest = {k: tuple(v.shape) for k, v in sd.items() if k not in new}
    new.update(module_state_dict(rest, vq_seed or 304))                  # RVQ codebooks / projections, zero_cond_embedding1
    for k in rest:
        if k.endswith("_codebook.embed"):
            lvl = int(k.split("layers.")[1].split(".")[0])
            new[k] = seeded_tensor(rest[k], zlib.crc32(k.encode()) % 100000, std=0.7 ** lvl)   # stable across processes (str hash is salted)
    m.load_state_dict(new)
    return m.cuda().prepare(), new


def _close(got, ref, tol, name=""):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = float(np.abs(got - ref).max())
    assert err < tol * max(1.0, float(np.abs(ref).max())), (name, err)


def test_thinking_encoder_vs_reference_golden():
    """AudioThinking encoder (strided conv, merge projection, cls interleave, 2 transformer blocks with q/k LayerNorm, partial
    rotary, sigmoid-GLU, LayerScale, cls extraction) on the exact-fp32 kernels vs the reference's own output."""
    d, meta = _gold()
    m, _ = _toy_model(meta)
    w, mu = think_inputs()
    q = m.encode_reasoning_query(w.cuda(), mu.cuda()).cpu().numpy()
    _close(q, d["think_query"], 1e-4, "think_query")


def test_fetch_codes_pipeline_vs_reference_golden_and_oracle():
    """fetch_codes_from_features with the golden's FiLM masks: every tensor that enters an RVQ equals the reference's
    (1e-4); then, with real RVQs, codes against the oracle's ResidualVQ restatement on the same features: identical wherever
    the oracle's nearest / runner-up distance gap is above fp32 noise (the projections run on different fp32 summation
    orders), and those must be nearly all frames."""
    from oracle import rvq_oracle
    from oracle.codec_model_oracle import ResidualVQOracle, sub
    d, meta = _gold()
    m, sd = _toy_model(meta)
    f = fetch_inputs()
    masks = torch.from_numpy(d["fetch_film_masks"])
    real_vq = m.audio_thinking.reasoning_vq
    # the golden was produced with identity quantisers (vector_quantize_pytorch is absent where the reference ran), so the
    # FiLM conditioning of the three branches comes from the UN-quantised query tokens: mirror that for the comparison
    class IdentityVQ(torch.nn.Module):
        def forward(self, x):
            return x, torch.zeros(x.shape[0], x.shape[1], 8, dtype=torch.long, device=x.device), None

    m.audio_thinking.reasoning_vq = IdentityVQ()
    r = m.fetch_codes_from_features(f["whisper"].cuda(), f["wavlm"].cuda(), f["bestrq_acoustic"].cuda(), f["bestrq_semantic"].cuda(),
                                    film_masks=masks, return_intermediates=True)
    for name, key in (("reason_query", "fetch_reason_query"), ("pre_vq_phone", "fetch_pre_vq_phone"), ("pre_vq_semantic", "fetch_pre_vq_semantic"),
                      ("pre_vq_acoustic", "fetch_pre_vq_acoustic")):
        _close(r[name].cpu().numpy(), d[key], 1e-4, name)
    assert r["merge_codes"].shape == (CFG["B"], 15, 8) and r["merge_codes"].dtype == torch.int64
    m.audio_thinking.reasoning_vq = real_vq
    rc, mc, mf = m.fetch_codes_from_features(f["whisper"].cuda(), f["wavlm"].cuda(), f["bestrq_acoustic"].cuda(), f["bestrq_semantic"].cuda(), film_masks=masks)
    assert rc[0].shape == (CFG["B"], 6, 8) and mc[0].shape == (CFG["B"], 15, 8) and mf[0].shape == (CFG["B"], 15, CFG["D"])
    assert int(mc[0].max()) < 8192 and int(rc[0].max()) < 4096 and int(mc[0].min()) >= 0
    checked = total = 0
    for key, feat, cols in (("vq_pronunciation_semantic.", d["fetch_pre_vq_phone"], slice(0, 1)), ("vq_structure_semantic.", d["fetch_pre_vq_semantic"], slice(1, 2)),
                            ("vq_acoustic.", d["fetch_pre_vq_acoustic"], slice(2, 8))):
        o = ResidualVQOracle(sub(sd, key))
        x = torch.from_numpy(feat)
        _, o_codes, h = o(x)
        _, _, margin = rvq_oracle.rvq_encode(np.ascontiguousarray(h.numpy()), o.emb.numpy(), want_margin=True)
        got = getattr(m, key[:-1])(x.cuda())[1].cpu()
        safe = torch.from_numpy((margin > 1e-4).all(1)).view(o_codes.shape[:2])
        assert torch.equal(got[safe], o_codes[safe]), key
        checked += int(safe.sum()); total += safe.numel()
    assert checked >= 0.95 * total, (checked, total)


@pytest.mark.parametrize("tag", ["infer_first", "infer_other"])
def test_inference_codes_and_euler_vs_reference_golden(tag):
    """inference_codes + BASECFM.solve_euler on the device (look-up sum, cond_feature_emb GEMM, x2 nearest gather, masks,
    zero_cond rows, in-context blend, guidance, Euler update) with the stand-in estimator of the golden vs the reference's own
    output."""
    d, meta = _gold()
    m, _ = _toy_model(meta, dit=True)
    i = infer_inputs()
    cfe = module_state_dict({k: tuple(s) for k, s in meta["cfe_keys"]}, 303)
    with torch.no_grad():
        m.cond_feature_emb.weight.copy_(cfe["weight"]); m.cond_feature_emb.bias.copy_(cfe["bias"])
        m.zero_cond_embedding1.copy_(i["zero_cond"])
    m.prepare()

    class Table(torch.nn.Module):
        def __init__(self, t):
            super().__init__()
            self.t = t.cuda()

        def get_output_from_indices(self, idx):
            return sum(self.t[l][idx[..., l]] for l in range(idx.shape[-1]))

    m.vq_pronunciation_semantic, m.vq_structure_semantic, m.vq_acoustic = Table(i["tab_phone"]), Table(i["tab_sem"]), Table(i["tab_ac"])
    est = StubEstimator("cuda")
    true_lat, n_inc = (i["first_latent"], 0) if tag == "infer_first" else (i["true_latent"], i["incontext"])
    lat = m.inference_codes([i["codes"].cuda()], None, true_lat.cuda(), i["latent_length"], n_inc, additional_feats=[], guidance_scale=1.5,
                            num_steps=CFG["steps"], scenario="other_seg", noise=i["noise"].cuda(),
                            estimator=lambda x, t: est(x, timestep=torch.full((2,), t, device="cuda")).sample)
    _close(lat.cpu().numpy(), d[tag], 2e-4, tag)


@pytest.mark.parametrize("bmt", [8, 16])
def test_dit_order_free_plan_vs_oracle_at_640_rows(bmt, monkeypatch):
    """VERDICT r5 #2a: the DiT's DEFAULT bf16 plan (sum_order = FREE -> csrc/ua2_gemm2.hip, attention with bf16 q / softmax weights)
    compared with the ORACLE directly — not with another HIP kernel — at a row count that reaches the order-free kernel
    (2 x 320 = 640 rows >= its 256-row entry bar; the toy widths give too few tiles for the launcher's cost rule, so the tile
    form is pinned with UA2_GEMM2_BMT: 128-row tiles and 256-row tiles both).  The launch counter proves the kernel ran; the
    same input through the row-invariant plan gives the A/B: both plans sit at the same distance from the fp32 oracle."""
    from oracle.codec_model_oracle import dit_forward
    from uniaudio2_amd._lib import lib
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import Transformer1DModel
    import dit_toy
    c = dit_toy.CFG
    sd = dit_toy.state_dict(5)
    T = 320
    x = seeded_tensor((2, T, c["in_channels"]), 19, std=1.0)
    ref = dit_forward(sd, x, torch.tensor([0.6, 0.6]), c["heads"], c["head_dim"]).numpy()
    rel = lambda a: float(np.sqrt(np.mean((a - ref) ** 2)) / max(1e-6, np.sqrt(np.mean(ref ** 2))))

    def run(order):
        m = Transformer1DModel(num_attention_heads=c["heads"], attention_head_dim=c["head_dim"], in_channels=c["in_channels"],
                               out_channels=c["out_channels"], num_layers=c["layers"])
        m.load_state_dict(sd)
        m.sum_order = order
        m = m.cuda().prepare(torch.bfloat16)
        n0 = lib.ua2_debug_kernel_launches(b"gemm2")
        y = m(x.cuda(), 0.6, use_graph=False).cpu().numpy()
        return y, lib.ua2_debug_kernel_launches(b"gemm2") - n0

    monkeypatch.delenv("UA2_DIT_SUM_ORDER", raising=False)
    monkeypatch.setenv("UA2_GEMM2_BMT", str(bmt))
    lib.ua2_debug_refresh_env()
    try:
        free, n_free = run(1)
        inv, n_inv = run(0)
    finally:
        monkeypatch.delenv("UA2_GEMM2_BMT")
        lib.ua2_debug_refresh_env()
    # every GEMM of the two blocks (q|k|v, to_out, ff.net.0, ff.net.2) and proj_in's Linear reach the kernel (the conv taps have
    # K = 304, not a multiple of 32, and proj_out has 24 columns: those stay on ua2_gemm.hip)
    assert n_free >= 4 * c["layers"] + 1, n_free
    assert n_inv == 0, n_inv
    e_free, e_inv = rel(free), rel(inv)
    print(f"DiT bf16 at 2 x {T} rows, {16 * bmt}-row tiles: order-free plan vs fp32 oracle {e_free:.3e} ({n_free} gemm2 launches), "
          f"row-invariant plan {e_inv:.3e}, plans apart {rel(free - inv + ref):.3e}")
    assert e_free < 6e-2, e_free                       # the bar of test_dit_forward_vs_oracle's bf16 case
    assert e_free < 1.5 * e_inv + 2e-3, (e_free, e_inv)   # no worse than the invariant plan's own bf16 noise


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-3), (torch.bfloat16, 6e-2)])
def test_dit_forward_vs_oracle(dtype, tol):
    """Transformer1DModel mirror vs the oracle's restatement (both PARITY UNPINNED vs diffusers — the package is absent):
    adaLN-single modulation, fused q|k|v with bias, dense attention, gated residuals, tanh-GELU feed-forward, 3-tap
    ProjectLayers, sinusoidal position table, final modulation."""
    from oracle.codec_model_oracle import dit_forward
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import Transformer1DModel
    import dit_toy
    c = dit_toy.CFG
    m = Transformer1DModel(num_attention_heads=c["heads"], attention_head_dim=c["head_dim"], in_channels=c["in_channels"],
                           out_channels=c["out_channels"], num_layers=c["layers"])
    sd = dit_toy.state_dict(5)
    m.load_state_dict(sd)
    m = m.cuda().prepare(dtype)
    x = seeded_tensor((2, 37, c["in_channels"]), 9, std=1.0)
    ref = dit_forward(sd, x, torch.tensor([0.35, 0.35]), c["heads"], c["head_dim"]).numpy()
    got = m(x.cuda(), 0.35).cpu().numpy()
    assert got.shape == ref.shape
    err = np.sqrt(np.mean((got - ref) ** 2)) / max(1e-6, np.sqrt(np.mean(ref ** 2)))
    print(f"DiT {dtype}: relative rms error {err:.3e}")
    assert err < tol, err
    if dtype == torch.bfloat16:
        # attention -> O-projection and GELU -> down-projection hand their operand over in fragment order (no prep launch):
        # the same bits as the route through the consumer's prep launch
        from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import BasicTransformerBlock
        BasicTransformerBlock.packed_handoff = False
        try:
            plain = m(x.cuda(), 0.35, use_graph=False).cpu().numpy()
        finally:
            BasicTransformerBlock.packed_handoff = True
        assert np.array_equal(m(x.cuda(), 0.35, use_graph=False).cpu().numpy(), plain)


def test_stage_all_decode_runs_at_real_dit_size():
    """`--stage all`'s second half at the released DiT's shape (models/model_config.json: 32 layers x 1536, 24 heads x 64,
    in 1040, out 136): one 20-s window, 2 Euler steps, guided (batch 2) -> latent (1, 500, 136) -> ScalarModel.decode ->
    480 000 samples, finite.  Times the stage (information)."""
    import time
    from test_gpu_codec import BENCH_SCALAR_CFG
    from make_golden_codec import codec_state_dict
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.AudioDiffusion1D import AudioDiffusion1D
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import ReasoningTokenizer
    torch.manual_seed(0)
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import RELEASED_CONFIG
    model = AudioDiffusion1D(unet_model_config_path=dict(RELEASED_CONFIG), encoder_depth=1, device="cuda")
    with torch.no_grad():
        for n_, p_ in model.named_parameters():
            if p_.dim() > 1:
                p_.normal_(0, 0.02)
        for n_, b_ in model.named_buffers():
            if n_.endswith("_codebook.embed"):
                b_.normal_(0, 0.5)
    model = model.cuda().prepare()
    sq = ScalarModel(**BENCH_SCALAR_CFG)
    sq.load_state_dict(codec_state_dict({k: tuple(v.shape) for k, v in sq.state_dict().items()}, 77))
    sq = sq.cuda().prepare()
    tok = ReasoningTokenizer(sq_codec=sq, model=model, device="cuda")
    codes = torch.randint(0, 8192, (8, 250))
    wav = tok.detokenize_no_reason(codes, steps=2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wav = tok.detokenize_no_reason(codes, steps=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert wav.shape == (1, 480000) and wav.device.type == "cpu" and torch.isfinite(wav).all()
    print(f"stage-2 of one 20-s window, 2 Euler steps: {dt * 1e3:.1f} ms -> {dt * 1e3 / 2:.1f} ms per guided step + decode")


def _stage2_tokenizer(dit_cfg, sum_order):
    """ReasoningTokenizer over a seeded AudioDiffusion1D (DiT of `dit_cfg`) + the bench-size ScalarModel."""
    from test_gpu_codec import BENCH_SCALAR_CFG
    from make_golden_codec import codec_state_dict
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.AudioDiffusion1D import AudioDiffusion1D
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import Transformer1DModel
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import ReasoningTokenizer
    torch.manual_seed(0)
    model = AudioDiffusion1D(unet_model_config_path=dict(dit_cfg), encoder_depth=1, device="cuda")
    with torch.no_grad():
        for _, p_ in model.named_parameters():
            if p_.dim() > 1:
                p_.normal_(0, 0.02)
        for n_, b_ in model.named_buffers():
            if n_.endswith("_codebook.embed"):
                b_.normal_(0, 0.5)
    saved = Transformer1DModel.sum_order
    Transformer1DModel.sum_order = sum_order
    try:
        model = model.cuda().prepare()
    finally:
        Transformer1DModel.sum_order = saved
    sq = ScalarModel(**BENCH_SCALAR_CFG)
    sq.load_state_dict(codec_state_dict({k: tuple(v.shape) for k, v in sq.state_dict().items()}, 77))
    return ReasoningTokenizer(sq_codec=sq.cuda().prepare(), model=model, device="cuda")


def test_batched_stage2_equals_one_by_one_under_the_invariant_contract(monkeypatch):
    """detokenize_no_reason_batch (window k of several utterances in one DiT solve + one SQ-Codec decode, the whole solve one
    recorded graph) against a loop over detokenize_no_reason, same seed: with the row-invariant GEMM contract and no K slabs every
    kernel of the path gives a row the same bits whatever shares its launch, so the waves must be EQUAL — which proves the
    batching logic (window grid per utterance, in-context hand-over, the draw order of both generators, the estimator-row order
    of the guided step, the flat ProjectLayer) exactly.  Utterances of 1, 2 and 3 windows, batches of 2 (so groups change from
    window to window)."""
    monkeypatch.setenv("UA2_GEMM_NO_KSPLIT", "1")
    L = 136
    dit = dict(num_attention_heads=4, attention_head_dim=64, in_channels=2 * L + 768, out_channels=L, num_layers=2)
    tok = _stage2_tokenizer(dit, sum_order=0)
    g = torch.Generator().manual_seed(11)
    codes = [torch.randint(0, 8192, (8, T), generator=g) for T in (437, 100, 250, 600)]
    torch.manual_seed(123)
    single = [tok.detokenize_no_reason(c, steps=3) for c in codes]
    torch.manual_seed(123)
    batch = tok.detokenize_no_reason_batch(codes, steps=3, max_batch=2)
    torch.manual_seed(123)
    batch4 = tok.detokenize_no_reason_batch(codes, steps=3, max_batch=8)
    for c, a, b, b4 in zip(codes, single, batch, batch4):
        assert a.shape == b.shape == (1, int(c.shape[-1] / 12.5 * 24000)) and torch.isfinite(a).all() and a.abs().max() > 0
        assert torch.equal(a, b), (c.shape, float((a - b).abs().max()))
        assert torch.equal(a, b4), (c.shape, float((a - b4).abs().max()))
    # the graph of the solve replays: a second seeded pass gives the same waves
    torch.manual_seed(123)
    again = tok.detokenize_no_reason_batch(codes, steps=3, max_batch=2)
    assert all(torch.equal(a, b) for a, b in zip(batch, again))
    # ... and equals the eager (un-recorded) solve
    monkeypatch.setenv("UA2_EULER_NO_GRAPH", "1")
    torch.manual_seed(123)
    eager = tok.detokenize_no_reason_batch(codes, steps=3, max_batch=2)
    assert all(torch.equal(a, b) for a, b in zip(batch, eager))


def test_batched_stage2_at_the_released_dit_size_order_free():
    """The default plan at the released DiT's shape: order-free GEMMs (2 x P x 500 rows on the 256-row tiles for P = 4, the
    invariant kernels for P = 1).  Same seed: the batched waves agree with the one-by-one ones to the DiT's own bf16 noise
    (another summation order can flip bf16 roundings of a GEMM operand; DESIGN.md §2: 5.6e-3 on the velocity field against the
    fp32 oracle, bar 2e-2)."""
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import RELEASED_CONFIG
    tok = _stage2_tokenizer(RELEASED_CONFIG, sum_order=1)
    g = torch.Generator().manual_seed(12)
    codes = [torch.randint(0, 8192, (8, T), generator=g) for T in (250, 250, 437, 120)]
    torch.manual_seed(5)
    single = [tok.detokenize_no_reason(c, steps=2) for c in codes]
    torch.manual_seed(5)
    batch = tok.detokenize_no_reason_batch(codes, steps=2, max_batch=4)
    for a, b in zip(single, batch):
        assert a.shape == b.shape and torch.isfinite(b).all()
        rel = float(((a - b).double().pow(2).mean().sqrt()) / a.double().pow(2).mean().sqrt().clamp_min(1e-9))
        print(f"batched vs one-by-one wave, released DiT size: relative rms {rel:.2e}")
        assert rel < 2e-2, rel


def test_config1_codec_plumbing_p225(tmp_path):
    """BASELINE.json config 1: samples/p225_002.wav (fixture: its 86 848 samples at 22 050 Hz) -> load ->
    resample to 24 kHz (94 529 samples) -> ScalarModel.encode -> latent (1, 136, 99); synthetic features (1, 50, 768) seed 0
    -> RVQ 1 + 1 + 6 levels of 8192 x 32 (seed 1) -> codes (8, 50) -> `*_semantic.pt` round trip -> look-up ->
    stand-in latent -> ScalarModel.decode.  GPU vs the CPU oracles: resampler 1e-5, latent 1e-5 rms, codes equal (near-tie
    frames enumerated: none allowed above the fp32-noise margin), wav 1e-4 rms."""
    from scipy.io import wavfile
    from oracle import rvq_oracle
    from oracle.codec_model_oracle import ResidualVQOracle
    from oracle.codec_oracle import resample as resample_oracle
    from test_gpu_codec import _bench_scalar_model
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.residual_vq import ResidualVQ
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film import reason_tokenizer as rt
    fx = np.load(os.path.join(HERE, "p225_002.npz"))
    path = str(tmp_path / "p225_002.wav")
    wavfile.write(path, int(fx["sample_rate"]), fx["samples"])
    audio, sr = rt.load_wav(path)
    assert sr == 22050 and audio.shape == (1, 86848) and np.array_equal(audio.numpy()[0], fx["samples"])
    got24 = rt.resample(audio.cuda(), sr, 24000).cpu()
    ref24 = resample_oracle(audio, sr, 24000)
    assert got24.shape == ref24.shape == (1, 94529)
    assert float((got24 - ref24).abs().max()) < 1e-5
    sq, so = _bench_scalar_model()
    lat = sq.encode(ref24.view(1, 1, -1).cuda()).cpu()
    lat_ref = so.encode(ref24.view(1, 1, -1))
    assert lat.shape == lat_ref.shape == (1, 136, 99)          # 94 529 samples / hop 960, causal padding: 99 latent frames
    assert float(((lat - lat_ref) ** 2).mean().sqrt()) < 1e-5
    # synthetic features -> the three RVQs (phone 1, semantic 1, acoustic 6 levels; AudioDiffusion1D.py:256-264)
    feats = torch.randn(1, 50, 768, generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    codes, near_ties, vqs = [], [], []
    for nq in (1, 1, 6):
        vq = ResidualVQ(dim=768, codebook_size=8192, num_quantizers=nq, codebook_dim=32)
        sd = {k: (torch.randn(v.shape, generator=g) * (768 ** -0.5 if "project_in.weight" in k else 0.3 if "bias" not in k else 0.01)) for k, v in vq.state_dict().items()}
        for l in range(nq):
            sd[f"layers.{l}._codebook.embed"] = torch.randn(1, 8192, 32, generator=g) * 0.6 ** l
        vq.load_state_dict(sd)
        vq = vq.cuda()
        vqs.append(vq)
        o = ResidualVQOracle(sd)
        _, o_codes, h = o(feats)
        _, _, margin = rvq_oracle.rvq_encode(np.ascontiguousarray(h.numpy()), o.emb.numpy(), want_margin=True)
        got = vq(feats.cuda())[1].cpu()
        bad = (got != o_codes).any(-1).view(-1).nonzero().view(-1).tolist()
        for t in bad:                                                  # a differing frame must be an fp32 near-tie of the oracle
            assert float(margin[t].min()) < 1e-4, (nq, t, margin[t])
        near_ties += bad
        codes.append(o_codes[0].transpose(0, 1))                        # (nq, 50)
    print("config 1: RVQ frames differing from the oracle (near-ties):", near_ties)
    assert near_ties == []
    rec = torch.cat(codes, 0)                                           # (8, 50) int64: phone | semantic | acoustic x 6
    assert rec.shape == (8, 50)
    torch.save(rec.cpu(), str(tmp_path / "p225_002_semantic.pt"))       # the encode side writes int64
    back = torch.load(str(tmp_path / "p225_002_semantic.pt"), map_location="cpu")
    assert back.dtype == torch.int64 and torch.equal(back, rec)
    # look-up sum -> stand-in latent (fixed projection 768 -> 136, x2 nearest, tanh) -> decode
    tok = rt.ReasoningTokenizer(sq_codec=sq, vq_phone=vqs[0], vq_semantic=vqs[1], vq_acoustic=vqs[2], device="cuda")
    cond = tok.codes_to_condition(back.unsqueeze(0).cuda()).cpu()
    os_ = [ResidualVQOracle({k: v.cpu() for k, v in vq.state_dict().items()}) for vq in vqs]
    cond_ref = sum(o.get_output_from_indices(c.transpose(1, 2)) for o, c in zip(os_, (back[None, 0:1], back[None, 1:2], back[None, 2:])))
    assert float((cond - cond_ref).abs().max()) < 1e-5 * max(1.0, float(cond_ref.abs().max()))
    P = seeded_tensor((768, 136), 5, std=768 ** -0.5)
    latent = torch.tanh(cond_ref @ P).repeat_interleave(2, dim=1).transpose(1, 2).contiguous()      # (1, 136, 100)
    wav = sq.decode(latent.cuda()).cpu()
    wav_ref = so.decode(latent)
    assert wav.shape == wav_ref.shape == (1, 1, 96000)
    rms = float(((wav - wav_ref) ** 2).mean().sqrt())
    assert rms < 1e-4 * max(1.0, float((wav_ref ** 2).mean().sqrt())), rms


def test_audio2token_skips_discarded_segments_with_identical_tokens():
    """SURVEY.md §8f rank-2 waste removal on the encode side: reason_tokenizer.py:98-128 encodes every 30-s segment of the
    self-concatenated clip and then slices the tokens to the clip's own length.  The mirror computes only the segments whose
    tokens survive, with the FiLM draws of each chunk made for the reference's full batch (AudioDiffusion1D.py:435 draws per
    batch).  Same seed -> the same tokens as the everything-computed route, with fewer rows through the model."""
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.reason_tokenizer import ReasoningTokenizer, segment_plan
    d, meta = _gold()
    m, _ = _toy_model(meta)
    c = CFG
    rows_seen = []

    def ssl_features(audio, spectrograms):
        """Deterministic stand-in for the frozen encoders: features are a function of the row's own samples only."""
        B = audio.shape[0]
        rows_seen.append(B)
        out = {k: [] for k in ("whisper", "wavlm", "bestrq_acoustic", "bestrq_semantic")}
        for b in range(B):
            seed = int(audio[b].double().abs().sum().item() * 1000) % 100000
            out["whisper"].append(seeded_tensor((c["Cw"], 2 * c["T25"]), seed + 1, std=1.0))
            out["wavlm"].append(seeded_tensor((c["Cl"], 2 * c["T25"]), seed + 2, std=1.0))
            out["bestrq_acoustic"].append(seeded_tensor((c["Cb"], c["T25"]), seed + 3, std=1.0))
            out["bestrq_semantic"].append(seeded_tensor((c["Cb"], c["T25"]), seed + 4, std=1.0))
        return {k: torch.stack(v).cuda() for k, v in out.items()}

    m.ssl_features = ssl_features
    tok = ReasoningTokenizer(model=m, device="cuda")
    wav = seeded_tensor((1, 26000), 77, std=0.1)                       # ~1.08 s: 14 rec / 6 reason tokens kept, one toy segment yields 15 / 6
    plan = segment_plan(wav.shape[-1])
    assert plan["n_segments"] == 2
    results = {}
    for skip in (True, False):
        tok.skip_discarded_segments = skip
        rows_seen.clear()
        torch.manual_seed(5); torch.cuda.manual_seed(5)
        reason, rec = tok.audio2token(wav, 24000)
        results[skip] = (reason.cpu(), rec.cpu(), sum(rows_seen))
    assert results[True][2] == 1 and results[False][2] == 2            # rows encoded: 1 instead of 2
    assert torch.equal(results[True][0], results[False][0]) and torch.equal(results[True][1], results[False][1])
    assert results[True][1].shape == (1, 8, plan["output_len"]) and results[True][0].shape[-1] == min(plan["output_len_reason"], 6)


def test_dit_layernorm_handover_equals_the_prep_route_to_bf16_noise():
    """Round 6: at one released-width window (2 x 500 rows, D = 1536) the DiT's o-projection and FF2 run as K slabs whose combine also
    builds the next GEMM's LayerNorm-ed operand (BasicTransformerBlock.ln_handover; transformer_1d_flow.py / attention.py:311-405).
    Against the same model with the consumers' own LayerNorm prep launches: the velocity field agrees to the DiT's bf16 noise, two
    launches per block fewer, and the hand-over really ran (gemm2 launch count)."""
    from uniaudio2_amd._lib import lib
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import BasicTransformerBlock, Transformer1DModel
    from codec_model_stub import module_state_dict
    import dit_toy
    cfg = dict(heads=24, head_dim=64, layers=3, in_channels=2 * 136 + 768, out_channels=136)
    m = Transformer1DModel(num_attention_heads=24, attention_head_dim=64, in_channels=cfg["in_channels"], out_channels=136, num_layers=3)
    m.load_state_dict(dit_toy.state_dict(7, cfg))
    x = seeded_tensor((2, 500, cfg["in_channels"]), 23, std=1.0).cuda()
    outs, counts = [], []
    for on in (True, False):
        BasicTransformerBlock.ln_handover = on
        try:
            mm = m.cuda().prepare(torch.bfloat16)
            n0 = lib.ua2_debug_kernel_launches(b"gemm2")
            outs.append(mm(x, 0.4, use_graph=False).float().cpu().numpy())
            counts.append(lib.ua2_debug_kernel_launches(b"gemm2") - n0)
        finally:
            BasicTransformerBlock.ln_handover = True
    a, b = outs
    rel = float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    print(f"DiT 2 x 500 rows, LayerNorm hand-over vs prep launches: relative rms {rel:.3e}; gemm2 launches {counts}")
    assert rel < 1e-2, rel
    assert counts[0] == counts[1] + 3, counts             # the o-projection (48 tiles: ua2_gemm.hip without the hand-over) joins as K slabs, once per block

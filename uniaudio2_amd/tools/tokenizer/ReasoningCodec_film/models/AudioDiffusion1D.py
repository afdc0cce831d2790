"""The live codec's neural model, host side: SSL features -> three RVQ code groups (encode), codes -> SQ-Codec latent
through the flow-matching DiT (decode).

Mirror of the reference's tools/tokenizer/ReasoningCodec_film/models/AudioDiffusion1D.py for its inference methods —
`fetch_codes_batch` (:493-551), `inference_codes` (:554-624), `BASECFM.solve_euler` (:89-129) and the helpers they call
(`encode_reasoning_part` :372-390, `time_film` :428-438, `set_masking` / `extract_mask_positions` :458-486) — with the same
attribute names, hence the same state-dict keys for everything in scope (`d_conv_*`, `cond_fusion_layer_*`, `time_film_*`,
`reason_adaptor`, `cond_feature_emb`, `zero_cond_embedding1`, `vq_*`, `audio_thinking.*`, `cfm_wrapper.estimator.*`).

Out of scope: the three frozen SSL encoders (Whisper-medium, WavLM, BEST-RQ — third-party models on
un-vendored packages).  `fetch_codes_batch` therefore takes them as an injected callable `ssl_features(input_audios,
spectrograms) -> dict(whisper (B, Cw, T50), wavlm (B, Cl, T50), bestrq_acoustic / bestrq_semantic (B, 1024, T25))`;
`fetch_codes_from_features` is everything after them.  The reasoning-text LLM of AudioThinking is likewise not built.

Randomness is explicit: the reference draws `torch.rand(B,1,1) < 0.2` inside time_film at inference time (:435)
and `randn` latents inside inference_codes (:655); here both are arguments (defaults draw from torch's generator exactly
where the reference does, so a seeded run consumes the stream in the same order).

All arithmetic runs in libua2hip.so: strided k = s convolutions (ua2_conv1d), Linear layers with bias / FiLM / gated
residuals fused (ua2_linear), the encoder's transformer blocks (modules/transformer.py), RVQ search / look-up
(ua2_rvq_*), nearest-neighbour interpolation as a row gather, the DiT (transformer_1d_flow.py) and the Euler update
(ua2_ew_fma).
"""
import os
from typing import Callable, Dict, List, Optional

import torch
import torch.nn as nn

from ..... import ops
from ....._lib import EPI_STORE
from ..modules.transformer import TransformerBlock
from ._dense import DenseKV, PackedLinear, nearest_indices
from .residual_vq import ResidualVQ
from .transformer_1d_flow import Transformer1DModel


class _StridedConv:
    """nn.Conv1d(C, C, kernel_size=s, stride=s) (AudioDiffusion1D.py:188,244-251) -> one ua2_conv1d launch."""

    def __init__(self, conv):
        self.w, self.K = ops.pack_conv_weight(conv.weight.detach().float())
        self.bias = conv.bias.detach().float().contiguous()
        self.cout, self.s = conv.out_channels, conv.stride[0]

    def __call__(self, x):
        T = (x.shape[-1] - self.K) // self.s + 1
        return ops.conv1d(x.float().contiguous(), self.w, self.K, self.cout, stride=self.s, Tout=T, bias=self.bias)


def _rows(x_bct):
    """(B, C, T) -> [B*T, C] rows (data movement only)."""
    return x_bct.transpose(1, 2).reshape(-1, x_bct.shape[1]).contiguous()


class BASECFM(nn.Module):
    """AudioDiffusion1D.py:57-129: Euler solver of the flow ODE with classifier-free guidance."""

    def __init__(self, estimator):
        super().__init__()
        self.sigma_min = 1e-4
        self.estimator = estimator
        self._graphs, self._pool, self._graphs_gen = {}, None, None

    def _euler_steps(self, x, noise, inc, n, ts, mu, guidance_scale, est):
        """The guided Euler loop proper (:99-127) for P utterances at once: rows [0, P) of the estimator batch are the
        unconditional halves, [P, 2P) the conditional ones (P = 1: the reference's two rows)."""
        P, _, L = x.shape
        zeros = torch.zeros_like(mu)
        t, dt = ts[0], ts[1] - ts[0]
        for step in range(1, len(ts)):
            if n > 0:                                                    # :104 in-context frames follow the known latent
                blend = ops.ew_fma(noise[:, :n].contiguous(), alpha=1 - (1 - self.sigma_min) * t)
                x[:, :n] = ops.ew_fma(inc[:, :n].contiguous(), c=blend, alpha=t)
            inp = torch.cat([torch.cat([x, x], 0), torch.cat([inc, inc], 0), torch.cat([zeros, mu], 0)], 2)      # :107-112
            d = est(inp, t).float().contiguous()                         # (2P, T, L): [unconditional | conditional]
            g = ops.ew_fma(d[P:].reshape(-1), alpha=guidance_scale)      # u + s (c - u) = s c + (1 - s) u   :115-116
            g = ops.ew_fma(d[:P].reshape(-1), c=g, alpha=1.0 - guidance_scale)
            x = ops.ew_fma(g, c=x.reshape(-1), alpha=dt).view(P, -1, L)  # :123
            t = t + dt
            if step < len(ts) - 1:
                dt = ts[step + 1] - t
        return x

    @torch.inference_mode()
    def solve_euler(self, x, incontext_x, incontext_length, t_span, mu, added_cond_kwargs=None, guidance_scale=1.5, estimator=None,
                    use_graph=None):
        """x (P, T, L) noise, incontext_x (P, T, L), mu (P, T, D), t_span host tensor of times; P = 1 in the reference (its guided
        step builds a 2-entry timestep for a 2-row batch, :113), P utterances here share one timestep per step, which is what that
        line means for each of them.  `estimator(x_cat, t)` defaults to the DiT; tests inject a stand-in to pin the solver against
        the reference's own solve_euler.
        use_graph (default: on with the DiT): the whole solve — every step's in-context blend, concatenation, DiT forward,
        guidance and Euler update — is captured ONCE per (P, T, in-context length, schedule) into a HIP graph and replayed per
        window (the times of the schedule are host constants, so they become kernel arguments of the recording).  Before, a
        window was 10 graph replays + ~30 small launches per step issued from Python: one pass in three ran 1.5x longer on a
        busy host (round-4 bench: 60.4 / 60.8 / 92.8 ms)."""
        if guidance_scale <= 1.0:
            raise NotImplementedError("the un-guided branch of the reference concatenates on the wrong axis (SURVEY A.10) and is unreachable: "
                                      "every caller passes guidance_scale = 1.5")
        ts = [float(v) for v in t_span]
        n = int(incontext_length)
        x = x.float().contiguous().clone()
        inc = incontext_x.float().contiguous()
        mu = mu.float().contiguous()
        if use_graph is None:
            use_graph = estimator is None and os.environ.get("UA2_EULER_NO_GRAPH") is None
        if estimator is not None or not use_graph or torch.cuda.is_current_stream_capturing():
            est = estimator or (lambda inp, t: self.estimator(inp, t))
            return self._euler_steps(x, x.clone(), inc, n, ts, mu, guidance_scale, est)
        key = (tuple(x.shape), tuple(mu.shape), n, tuple(ts), float(guidance_scale))
        dit = self.estimator
        if not getattr(dit, "_ready", True):
            dit.prepare()
        gen = getattr(dit, "_plan_gen", 0)
        if gen != self._graphs_gen:                                       # the DiT was re-prepared (dtype, UA2_DIT_SUM_ORDER): every recording
            self._graphs.clear()                                          # points at weights / K-V plans / embeddings that no longer exist
            self._graphs_gen = gen
        g = self._graphs.pop(key, None)                                   # re-inserted below: the dict's order is the LRU order
        if g is None:
            P, T, _ = x.shape
            dev = x.device
            dit.ensure_plan(2 * P, T, dev)
            for t in ts:
                dit.step_embedding(t, dev)
            x_in, inc_in, mu_in = torch.empty_like(x), torch.empty_like(inc), torch.empty_like(mu)
            x_in.copy_(x); inc_in.copy_(inc); mu_in.copy_(mu)
            est = lambda inp, t: dit(inp, t, use_graph=False)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                                 # warm-up outside capture: one-time kernel attributes, scratch, allocator pools
                self._euler_steps(x_in.clone(), x_in.clone(), inc_in, n, ts[:2], mu_in, guidance_scale, est)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self._pool):
                y = self._euler_steps(x_in.clone(), x_in.clone(), inc_in, n, ts, mu_in, guidance_scale, lambda inp, t: dit(inp, t))
            if self._pool is None:
                self._pool = graph.pool()                                 # one memory pool for every recorded shape
            while len(self._graphs) >= 18:                                # P = 1 .. codec_batch utterances x {first, later} windows; one shared pool
                self._graphs.pop(next(iter(self._graphs)))                # least recently used first
            g = (graph, x_in, inc_in, mu_in, y)
        self._graphs[key] = g
        graph, x_in, inc_in, mu_in, y = g
        x_in.copy_(x); inc_in.copy_(inc); mu_in.copy_(mu)
        graph.replay()
        return y.clone()


class AudioThinking(nn.Module):
    """AudioDiffusion1D.py:168-188 (the reasoning-text LLM it can also host is not part of the token path)."""

    def __init__(self, dim, interval, encoder_depth, whisper_fea_dim, **unused):
        super().__init__()
        self.cls_token = nn.Parameter(torch.randn(1, dim))
        self.interval, self.whisper_fea_dim = interval, whisper_fea_dim
        self.encoder_transformers = nn.Sequential(*[
            TransformerBlock(dim, dim_heads=128, causal=False, power_normalized=True, layer_scale=True, add_rope=True,
                             attn_kwargs={"qk_norm": True}, ff_kwargs={"mult": 4, "no_bias": False}) for _ in range(encoder_depth)])
        self.semantic_merge_proj = nn.Linear(whisper_fea_dim + 1024, dim)
        self.reasoning_vq = ResidualVQ(dim=dim, codebook_size=4096, codebook_dim=64, num_quantizers=8)
        self.down_sampling_layer_whisper = nn.Conv1d(whisper_fea_dim, whisper_fea_dim, kernel_size=2, stride=2, padding=0, bias=True)


class AudioDiffusion1D(nn.Module):
    def __init__(self, num_channels=None, unet_model_config_path=None, whisper_fea_dim=1024, wavlm_fea_dim=768, codec_dim=768,
                 encoder_depth=5, use_detokenizer=True, ssl_features: Optional[Callable] = None, device=None, **unused):
        super().__init__()
        self.max_t_len, self.sample_rate, self.sq_codec_latent = 30 * 50, 24000, 136
        self.whisper_fea_dim, self.wavlm_fea_dim, self.codec_dim = whisper_fea_dim, wavlm_fea_dim, codec_dim
        self.ssl_features = ssl_features
        D = codec_dim
        self.d_conv_whisper = nn.Conv1d(whisper_fea_dim, whisper_fea_dim, kernel_size=4, stride=4)
        self.d_conv_wavlm = nn.Conv1d(wavlm_fea_dim, wavlm_fea_dim, kernel_size=4, stride=4)
        self.d_conv_embedding_semantic = nn.Conv1d(1024, 1024, kernel_size=2, stride=2)
        self.d_conv_embedding_acoustic = nn.Conv1d(1024, 1024, kernel_size=2, stride=2)
        self.vq_acoustic = ResidualVQ(dim=D, codebook_size=8192, codebook_dim=32, num_quantizers=6)
        self.vq_structure_semantic = ResidualVQ(dim=D, codebook_size=8192, codebook_dim=32, num_quantizers=1)
        self.vq_pronunciation_semantic = ResidualVQ(dim=D, codebook_size=8192, codebook_dim=32, num_quantizers=1)
        self.cond_fusion_layer_semantic = nn.Linear(1024, D)
        self.cond_fusion_layer_acoustic = nn.Linear(1024 + whisper_fea_dim, D)
        self.cond_fusion_layer_phone = nn.Linear(wavlm_fea_dim, D)
        self.time_film_phone, self.time_film_semantic, self.time_film_acoustic = (nn.Linear(D, 2 * D) for _ in range(3))
        self.gamma = 0.1
        self.reason_adaptor = nn.Linear(D, D)
        self.cond_feature_emb = nn.Linear(D, D)
        self.zero_cond_embedding1 = nn.Parameter(torch.randn(D))
        if use_detokenizer and unet_model_config_path is not None:
            self.cfm_wrapper = BASECFM(Transformer1DModel.from_config(unet_model_config_path))
        self.audio_thinking = AudioThinking(dim=D, interval=5, encoder_depth=encoder_depth, whisper_fea_dim=whisper_fea_dim)
        self._p = None

    def init_device_dtype(self, device, dtype):
        self.device, self.dtype = device, dtype

    # ---- plan --------------------------------------------------------------------------------------------------
    def prepare(self, encode_dtype=torch.float32, dit_dtype=torch.bfloat16):
        """Pack every filter for the device.  The encode side defaults to the exact-fp32 kernels: its outputs are integer
        codes, and the CPU-run reference this is compared with computes them in fp32."""
        dev = self.cond_feature_emb.weight.device
        if dev.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback); move the model to cuda")
        lin = lambda m: PackedLinear(m.weight, m.bias, encode_dtype)
        at = self.audio_thinking
        p = dict(dtype=encode_dtype,
                 d_conv_whisper=_StridedConv(self.d_conv_whisper), d_conv_wavlm=_StridedConv(self.d_conv_wavlm),
                 d_conv_embedding_semantic=_StridedConv(self.d_conv_embedding_semantic),
                 d_conv_embedding_acoustic=_StridedConv(self.d_conv_embedding_acoustic),
                 down_whisper=_StridedConv(at.down_sampling_layer_whisper), merge=lin(at.semantic_merge_proj),
                 cls=at.cls_token.detach().float().contiguous(),
                 fusion_phone=lin(self.cond_fusion_layer_phone), fusion_semantic=lin(self.cond_fusion_layer_semantic),
                 fusion_acoustic=lin(self.cond_fusion_layer_acoustic), film_phone=lin(self.time_film_phone),
                 film_semantic=lin(self.time_film_semantic), film_acoustic=lin(self.time_film_acoustic),
                 reason_adaptor=lin(self.reason_adaptor), cond_feature_emb=lin(self.cond_feature_emb),
                 zero_cond=self.zero_cond_embedding1.detach().float().contiguous())
        for blk in at.encoder_transformers:
            blk.prepare(encode_dtype)
        if hasattr(self, "cfm_wrapper"):
            self.cfm_wrapper.estimator.prepare(dit_dtype)
        self._p = p
        return self

    def _plan(self):
        return self._p or self.prepare()._p

    # ---- encode side -----------------------------------------------------------------------------------------------
    def set_masking(self, x_rows, B, T):
        """AudioDiffusion1D.py:458-476 as one row gather: after every `interval` frames one cls token (the last row of the
        source table).  x_rows [B*T, D] -> [B*(T + T//interval), D]."""
        p, iv = self._plan(), self.audio_thinking.interval
        n = T // iv
        src = torch.cat([x_rows, p["cls"]], 0)                                   # row B*T = the cls token
        per = torch.arange(n * iv).view(n, iv)
        per = torch.cat([per, torch.full((n, 1), -1)], 1).reshape(-1)            # -1 marks a cls slot
        idx = torch.cat([torch.where(per >= 0, per + b * T, torch.full_like(per, B * T)) for b in range(B)])
        return ops.gather_rows(src, idx.to(torch.int32).to(x_rows.device)), n * (iv + 1)

    def extract_mask_positions(self, x_rows, B, T_new):
        """:478-486: the rows of the cls tokens."""
        iv = self.audio_thinking.interval
        n = (T_new - T_new // (iv + 1)) // iv
        idx = torch.tensor([b * T_new + (i + 1) * (iv + 1) - 1 for b in range(B) for i in range(n)], dtype=torch.int32)
        return ops.gather_rows(x_rows, idx.to(x_rows.device)), n

    @torch.inference_mode()
    def encode_reasoning_query(self, whisper_embeds, muencoder_embeds):
        """AudioDiffusion1D.py:372-387 up to reasoning_vq: (B, Cw, T50), (B, 1024, T25) -> query tokens (B, T25 // 5, D)."""
        p = self._plan()
        B = whisper_embeds.shape[0]
        w = p["down_whisper"](whisper_embeds)                                    # (B, Cw, T25)
        n = min(w.shape[-1], muencoder_embeds.shape[-1])
        cat = torch.cat([w[:, :, :n], muencoder_embeds[:, :, :n].float()], dim=1)
        x = p["merge"](_rows(cat))
        x, Tn = self.set_masking(x, B, n)
        blocks = self.audio_thinking.encoder_transformers
        kv = DenseKV(B, Tn, blocks[0].self_attn.num_heads, blocks[0].dim_heads, p["dtype"], x.device)
        cos, sin = blocks[0].rope_tables(Tn, x.device)
        for blk in blocks:
            blk.run(x, kv, cos, sin)
        q, nq = self.extract_mask_positions(x, B, Tn)
        return q.view(B, nq, -1)

    def encode_reasoning_part(self, whisper_embeds, muencoder_embeds):
        """:372-390 -> (quantized, indices, commitment loss)."""
        return self.audio_thinking.reasoning_vq(self.encode_reasoning_query(whisper_embeds, muencoder_embeds))

    def time_film(self, cond_rows, feat_rows, key, B, mask=None):
        """:428-438; `mask` (B,) bool is the reference's torch.rand(B,1,1) < 0.2 draw (drawn here, in the same place, when None)."""
        p = self._plan()
        if mask is None:
            mask = (torch.rand(B, 1, 1, device=cond_rows.device) < 0.2).view(B)
        params = p[key](cond_rows)
        return ops.time_film(params, feat_rows, mask.to(device=feat_rows.device, dtype=torch.uint8).contiguous(), feat_rows.shape[0] // B,
                             self.gamma)

    @torch.inference_mode()
    def fetch_codes_from_features(self, whisper, wavlm, bestrq_acoustic, bestrq_semantic, film_masks=None, return_intermediates=False):
        """AudioDiffusion1D.py:511-551: everything of fetch_codes_batch after the SSL encoders.  film_masks (3, B) bool in the
        reference's call order (phone, semantic, acoustic) or None to draw them.  Returns ([reasoning_codes], [merge_codes],
        [merge_features]) as the reference does: codes (B, T, 8) int64 with columns [phone, semantic, acoustic x 6]."""
        p = self._plan()
        B = whisper.shape[0]
        whisper, wavlm = whisper.float().contiguous(), wavlm.float().contiguous()
        whisper_rec = p["d_conv_whisper"](whisper)
        wavlm_f = p["d_conv_wavlm"](wavlm)
        sem_rec = p["d_conv_embedding_semantic"](bestrq_semantic.float().contiguous())
        ac = p["d_conv_embedding_acoustic"](bestrq_acoustic.float().contiguous())
        query = self.encode_reasoning_query(whisper, bestrq_semantic)
        q_reason, reason_codes, _ = self.audio_thinking.reasoning_vq(query)
        rf = p["reason_adaptor"](q_reason.reshape(-1, q_reason.shape[-1]).contiguous())
        Tq = q_reason.shape[1]
        up = nearest_indices(Tq, 2.5, rf.device)                                  # :522
        T = up.numel()
        rf = ops.gather_rows(rf, torch.cat([up + b * Tq for b in range(B)]))
        m = film_masks if film_masks is not None else [None] * 3
        phone = self.time_film(rf, p["fusion_phone"](_rows(wavlm_f)), "film_phone", B, m[0])
        q_phone, c_phone, _ = self.vq_pronunciation_semantic(phone.view(B, -1, self.codec_dim))
        sem = self.time_film(rf, p["fusion_semantic"](_rows(sem_rec)), "film_semantic", B, m[1])
        q_sem, c_sem, _ = self.vq_structure_semantic(sem.view(B, -1, self.codec_dim))
        n = min(ac.shape[-1], whisper_rec.shape[-1])                              # :538
        acf = p["fusion_acoustic"](_rows(torch.cat([ac[:, :, :n], whisper_rec[:, :, :n]], dim=1)))
        acf = self.time_film(rf, acf, "film_acoustic", B, m[2])
        q_ac, c_ac, _ = self.vq_acoustic(acf.view(B, -1, self.codec_dim))
        qsum = ops.ew_fma(q_phone.reshape(-1), c=ops.ew_fma(q_sem.reshape(-1), c=q_ac.reshape(-1)))
        merge = p["cond_feature_emb"](qsum.view(-1, self.codec_dim)).view(B, -1, self.codec_dim)
        codes = torch.cat([c_phone, c_sem, c_ac], dim=-1)                         # :549 order phone | semantic | acoustic
        if return_intermediates:
            return dict(reason_query=query, reason_codes=reason_codes, pre_vq_phone=phone.view(B, -1, self.codec_dim),
                        pre_vq_semantic=sem.view(B, -1, self.codec_dim), pre_vq_acoustic=acf.view(B, -1, self.codec_dim),
                        merge_features=merge, merge_codes=codes)
        return [reason_codes], [codes], [merge]

    @torch.inference_mode()
    def fetch_codes_batch(self, input_audios, spectrograms, additional_feats=None, return_reasoning_text=False, film_masks=None):
        """:493-551.  The frozen SSL encoders are the injected `ssl_features` callable (see the module docstring).  film_masks
        (3, B) bool: the three FiLM draws of this call made by the caller (ReasoningTokenizer.audio2token draws them for the
        reference's whole chunk and computes only the rows whose tokens are kept)."""
        if return_reasoning_text:
            raise NotImplementedError("the reasoning-text LLM of AudioThinking is not part of the token path and is not built")
        if self.ssl_features is None:
            raise NotImplementedError("tokenising audio needs the frozen Whisper / WavLM / BEST-RQ encoders (out of scope, SURVEY.md §2.1); "
                                      "construct AudioDiffusion1D(ssl_features=...) with a callable that returns their features, or call "
                                      "fetch_codes_from_features")
        f = self.ssl_features(input_audios, spectrograms)
        return self.fetch_codes_from_features(f["whisper"], f["wavlm"], f["bestrq_acoustic"], f["bestrq_semantic"], film_masks=film_masks)

    # ---- decode side -----------------------------------------------------------------------------------------------
    def prepare_latents(self, batch_size, num_frames, dtype, device):
        return torch.randn(batch_size, num_frames, self.sq_codec_latent, device=device, dtype=torch.float32)      # :651-656

    @torch.inference_mode()
    def codes_to_condition(self, codes):
        """:563-590 without reasoning codes: (B, 8, T) -> (B, 2T, D): sum of the three look-ups, cond_feature_emb, x2 nearest."""
        p = self._plan()
        B, _, T = codes.shape
        parts = ((self.vq_pronunciation_semantic, codes[:, 0:1]), (self.vq_structure_semantic, codes[:, 1:2]), (self.vq_acoustic, codes[:, 2:]))
        q = None
        for vq, c in parts:
            r = vq.get_output_from_indices(c.transpose(1, 2).contiguous()).reshape(-1)
            q = r if q is None else ops.ew_fma(q, c=r)
        m = p["cond_feature_emb"](q.view(B * T, self.codec_dim))
        up = nearest_indices(T, 2, m.device)
        return ops.gather_rows(m, torch.cat([up + b * T for b in range(B)])).view(B, up.numel(), self.codec_dim)

    @torch.inference_mode()
    def inference_codes(self, codes, spk_embeds, true_latents, latent_length, incontext_length, additional_feats=None, guidance_scale=2,
                        num_steps=20, disable_progress=True, scenario="start_seg", noise=None, estimator=None):
        """:554-624.  codes = [rec_codes (1, 8, T)] (token2audio_no_reason's form); returns latents (1, 2T, 136)."""
        if len(codes) != 1:
            raise NotImplementedError("reasoning-code conditioning (feature_combine, :440-456) is used by token2audio only; "
                                      "the CLI decodes with token2audio_no_reason (multi_task_inference.py:545-548)")
        p = self._plan()
        merge = self.codes_to_condition(codes[0].to(p["zero_cond"].device))
        B, T, D = merge.shape
        dev = merge.device
        lat = noise if noise is not None else self.prepare_latents(B, T, torch.float32, dev)
        masks = torch.zeros(B, T, dtype=torch.int64, device=dev)
        masks[:, 0:latent_length] = 2
        if scenario == "other_seg":
            masks[:, 0:incontext_length] = 1
        keep = (masks > 0.5).view(-1)                                             # :606-607: frames past latent_length get the learnt "no condition" vector
        rows = torch.where(keep, torch.arange(B * T, device=dev), torch.full((B * T,), B * T, device=dev)).to(torch.int32)
        merge = ops.gather_rows(torch.cat([merge.view(B * T, D), p["zero_cond"].view(1, D)], 0), rows).view(B, T, D)
        inc_rows = ((masks > 0.5) & (masks < 1.5))
        n_inc = int(inc_rows.sum(-1)[0])
        inc = torch.zeros_like(lat)
        inc[:, :n_inc] = true_latents.to(dev).float()[:, :n_inc]                  # :609 (mask multiply = copy of the in-context frames)
        t_span = torch.linspace(0, 1, num_steps + 1)
        out = self.cfm_wrapper.solve_euler(lat.float(), inc, n_inc, t_span, merge, None, guidance_scale, estimator=estimator)
        out[:, :n_inc] = inc[:, :n_inc]                                           # :623
        return out

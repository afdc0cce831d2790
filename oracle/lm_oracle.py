"""CPU ORACLE (test infrastructure, NOT product code) for the audio-token decode loop.

A plain-PyTorch fp32 CPU restatement of the reference algorithm *as shipped*; every
function cites the reference file:line it follows (paths relative to /root/reference).
It is pinned against golden vectors produced by running the reference itself in the
authoring container (tests/golden/make_golden_lm.py -> tests/golden/lm_toy_fp32.npz;
checked by tests/test_oracle_lm.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
The product (uniaudio2_amd/) never does and has no CPU fallback.

Two arithmetic contracts are restated:
  mode="fp32"  the reference as shipped (multi_task_inference.py:181-183 builds the LM in
               fp32 and never casts it).
  mode="bf16"  the product's reduced-precision contract: weights and
               embedding tables rounded to bf16, every GEMM input rounded to bf16, K/V rounded
               to bf16 when written to the cache, everything else (residual stream, norms,
               RoPE, softmax, logits) and all accumulation in fp32.  This is NOT
               `model.to(bfloat16)` of the reference (which also rounds every activation).
               RMSNorm + Linear pairs (norm_1 -> qkv, norm_2 -> fc_1 / fc_2, the depth decoder's
               ln_f -> audio_head) are evaluated in the product's "scaled" form (round 3,
               include/ua2hip.h UA2_PRO_SCALED):  rstd * ( RNE_bf16(x * w) W^T )  with
               rstd = rsqrt(mean(x^2) + eps) in fp32 — the same function as
               RNE_bf16((x * rstd) * w) W^T of lit_model.py:883-890 + the Linear, with the bf16
               rounding taken before the row scale instead of after it.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class GPTShape:
    """The subset of llm_models/config.py:Config the Llama-3.2 entries use (config.py:804-899)."""
    n_layer: int
    n_embd: int
    n_head: int
    n_query_groups: int
    intermediate_size: int
    padded_vocab_size: int = 128256
    norm_eps: float = 1e-5                     # config.py:38
    rope_base: int = 500000
    rope_adjustments: Optional[dict] = field(default_factory=lambda: dict(
        factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_seq_len=8192))

    @property
    def head_size(self):
        return self.n_embd // self.n_head      # config.py:105-107


def build_rope_cache(seq_len: int, n_elem: int, base: int, extra: Optional[dict]):
    """llm_models/lit_model.py:634-706 (Llama-3 frequency smoothing :662-676, halves duplicated :684)."""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2).float() / n_elem))
    if extra is not None:
        factor = extra["factor"]
        if "original_max_seq_len" in extra:
            wavelen = 2 * torch.pi / theta
            ratio = extra["original_max_seq_len"] / wavelen
            smooth = (ratio - extra["low_freq_factor"]) / (extra["high_freq_factor"] - extra["low_freq_factor"])
            smooth = torch.clamp(smooth, min=0.0, max=1.0)
            theta = (1 - smooth) * (theta / factor) + smooth * theta
        else:
            theta = theta / factor
    seq_idx = torch.arange(seq_len) / 1
    idx_theta = torch.outer(seq_idx, theta).repeat(1, 2)
    return torch.cos(idx_theta), torch.sin(idx_theta)


def apply_rope(x, cos, sin):
    """llm_models/lit_model.py:778-807 — half-split rotate; x (B, nh, T, hs), cos/sin (B, T, hs)."""
    h = x.size(-1) // 2
    rotated = torch.cat((-x[..., h:], x[..., :h]), dim=-1)
    cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    return (x * cos) + (rotated * sin)


def rmsnorm(x, w, eps):
    """llm_models/lit_model.py:883-890."""
    x = x.float()
    norm_x = torch.mean(x * x, dim=-1, keepdim=True)
    return (x * torch.rsqrt(norm_x + eps)) * w.float()


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def norm_linear(x, w, eps, mats, mode, scaled=True):
    """RMSNorm(x) followed by one or more Linear layers (lit_model.py:883-890 + :424 / :591-592).  fp32: as the reference
    evaluates it.  bf16: the product's scaled form (module docstring) on decode frames; `scaled=False` = the form its prefill
    passes keep (and plans for more than 64 sequences): the normalised row rounded to bf16, RNE_bf16((x * rstd) * w) W^T."""
    if mode != "bf16":
        xn = rmsnorm(x, w, eps)
        return [F.linear(xn, m) for m in mats]
    if not scaled:
        a = _bf16(rmsnorm(x, w, eps))
        return [F.linear(a, m) for m in mats]
    x = x.float()
    rstd = torch.rsqrt(torch.mean(x * x, dim=-1, keepdim=True) + eps)
    a = _bf16(x * w.float())
    return [F.linear(a, m) * rstd for m in mats]


class GPTOracle:
    """llm_models/lit_model.py:22-275 GPT with embeddings in / hidden out (:180), linear KV cache (:814-860)."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, shape: GPTShape, mode: str, max_seq: int):
        self.s, self.mode, self.max_seq = shape, mode, max_seq
        q = _bf16 if mode == "bf16" else (lambda t: t)
        self.qa = q
        g = lambda k: sd[prefix + k].float()
        self.layers = []
        for i in range(shape.n_layer):
            p = f"transformer.h.{i}."
            self.layers.append(dict(
                norm_1=g(p + "norm_1.weight"), qkv=q(g(p + "attn.qkv.weight")), proj=q(g(p + "attn.proj.weight")),
                norm_2=g(p + "norm_2.weight"), fc_1=q(g(p + "mlp.fc_1.weight")), fc_2=q(g(p + "mlp.fc_2.weight")),
                mlp_proj=q(g(p + "mlp.proj.weight"))))
        self.ln_f = g("transformer.ln_f.weight")
        self.cos, self.sin = build_rope_cache(max_seq, shape.head_size, shape.rope_base, shape.rope_adjustments)
        # lit_model.py:863-866 build_mask_cache
        self.mask_cache = torch.tril(torch.ones(max_seq, max_seq, dtype=torch.bool))
        self.k = self.v = None

    def set_kv_cache(self, batch):          # lit_model.py:224-254, 564-581
        s = self.s
        shp = (batch, s.n_query_groups, self.max_seq, s.head_size)
        self.k = [torch.zeros(shp) for _ in range(s.n_layer)]
        self.v = [torch.zeros(shp) for _ in range(s.n_layer)]

    def reset_kv_cache(self):               # lit_model.py:256-263
        for t in self.k + self.v:
            t.zero_()

    def forward(self, x, input_pos, maxp1=None, final_norm=True, scaled=True):
        """x (B, T, C) fp32; input_pos (B, T) long (per-sequence positions).  lit_model.py:83-180.
        final_norm=False returns the stream BEFORE ln_f (the caller folds ln_f into the Linear that follows)."""
        s = self.s
        B, T, C = x.shape
        cos, sin = self.cos[input_pos], self.sin[input_pos]             # (B, T, hs)   :129-130
        mask = self.mask_cache[input_pos]                               # (B, T, max_seq) :137
        L = self.max_seq if maxp1 is None else maxp1                    # :141-145
        mask = mask[..., :L].unsqueeze(1)
        nh, ng, hs = s.n_head, s.n_query_groups, s.head_size
        bidx = torch.arange(B).unsqueeze(1).expand(B, T)
        for li, W in enumerate(self.layers):                            # Block.forward :337-349
            qkv, = norm_linear(x, W["norm_1"], s.norm_eps, [W["qkv"]], self.mode, scaled)     # :341, :424
            q, k, v = qkv.split((nh * hs, ng * hs, ng * hs), dim=-1)    # :431
            q = q.view(B, T, nh, hs).transpose(1, 2)
            k = k.view(B, T, ng, hs).transpose(1, 2)
            v = v.view(B, T, ng, hs).transpose(1, 2)
            q, k = apply_rope(q, cos, sin), apply_rope(k, cos, sin)     # :458-461
            k, v = self.qa(k), self.qa(v)                               # bf16 contract: cache holds bf16
            self.k[li][bidx, :, input_pos] = k.transpose(1, 2)          # KVCache.forward :831-856
            self.v[li][bidx, :, input_pos] = v.transpose(1, 2)
            kk, vv = self.k[li][:B, :, :L], self.v[li][:B, :, :L]       # :468-471
            if ng != nh:                                                # :478-481
                kk = kk.repeat_interleave(nh // ng, dim=1)
                vv = vv.repeat_interleave(nh // ng, dim=1)
            y = F.scaled_dot_product_attention(q, kk, vv, attn_mask=mask, dropout_p=0.0,
                                               scale=1.0 / math.sqrt(hs))       # :529-531
            y = y.transpose(1, 2).reshape(B, T, nh * hs)
            x = F.linear(self.qa(y), W["proj"]) + x                     # :511, :345
            g1, g2 = norm_linear(x, W["norm_2"], s.norm_eps, [W["fc_1"], W["fc_2"]], self.mode, scaled)
            h = F.silu(g1) * g2                                         # LLaMAMLP :591-595
            x = F.linear(self.qa(h), W["mlp_proj"]) + x
        return rmsnorm(x, self.ln_f, s.norm_eps) if final_norm else x   # :164


class Stage3Oracle:
    """llm_models/model_new.py:334-687 Model_stage3 inference methods."""

    def __init__(self, sd, shapes: Dict[str, GPTShape], audio_semantic_vocab_size, audio_reason_vocab_size,
                 audio_num_codebooks=8, mode="fp32", max_seq=2048):
        assert mode in ("fp32", "bf16")
        self.mode = mode
        q = _bf16 if mode == "bf16" else (lambda t: t)
        self.qa = q
        self.ncb = audio_num_codebooks
        self.va = audio_semantic_vocab_size + audio_reason_vocab_size
        # model_new.py:560-565: 2048-slot caches for the three 3072-d GPTs, 8 slots for the decoder
        self.backbone = GPTOracle(sd, "backbone.", shapes["backbone"], mode, max_seq)
        self.und = GPTOracle(sd, "audio_understanding_expert.", shapes["understanding"], mode, max_seq)
        self.gen = GPTOracle(sd, "audio_generation_expert.", shapes["generation"], mode, max_seq)
        self.decoder = GPTOracle(sd, "decoder.", shapes["decoder"], mode, audio_num_codebooks)
        self.wte = q(sd["backbone.transformer.wte.weight"].float())
        self.lm_head = q(sd["backbone.lm_head.weight"].float())
        self.audio_embeddings = q(sd["audio_embeddings.weight"].float())
        self.projection = q(sd["projection.weight"].float())
        self.audio_head = q(sd["audio_head"].float())                   # (ncb, D_dec, V_a)  :349
        self.last_text_logits = None
        self.last_audio_logits = None

    def setup_caches(self, max_batch_size):                             # :554-565
        for g in (self.backbone, self.und, self.gen, self.decoder):
            g.set_kv_cache(max_batch_size)

    def reset_caches(self):                                             # :647-651
        for g in (self.backbone, self.und, self.gen, self.decoder):
            g.reset_kv_cache()

    def _embed_audio_tokens(self, tokens):                              # :665-673
        off = self.va * torch.arange(self.ncb)
        return self.audio_embeddings[(tokens[:, :, :-1] + off)]

    def _trunk(self, tokens, mask, input_pos, maxp1, scaled=True):
        """Shared by forward_prefix (:471-497) and generate_frame (:594-613).
        tokens (B,S,9) long; mask (B,S,9) bool; input_pos (B,S)."""
        a_step = mask[:, :, 0].unsqueeze(-1).float()
        t_step = mask[:, :, -1].unsqueeze(-1).float()
        a_in = (self._embed_audio_tokens(tokens) * mask[:, :, :-1].unsqueeze(-1).float()).sum(dim=2)
        h_a = self.und.forward(a_in, input_pos, maxp1, scaled=scaled)
        text = self.wte[tokens[:, :, -1]]
        h = self.backbone.forward(h_a * a_step + text * t_step, input_pos, maxp1, scaled=scaled)
        h_g = self.gen.forward(h * a_step, input_pos, maxp1, scaled=scaled)
        return h_g * a_step + h * t_step

    @torch.inference_mode()
    def forward_prefix(self, tokens, tokens_mask, input_pos):
        """model_new.py:456-507.  tokens (B,S,9); tokens_mask (B,S+1,9) as the caller passes it
        (evaluation/tts_task.py:244); input_pos (B,S).  No input_pos_maxp1 => attends all slots
        under the mask.  The discarded lm_head / local-decoder work (:498-506) is not restated."""
        return self._trunk(tokens, tokens_mask[:, :-1], input_pos, None, scaled=False)     # the product's prefill passes keep the unscaled form

    @torch.inference_mode()
    def generate_frame(self, tokens, tokens_mask, input_pos, input_pos_maxp1=None, forbid_prefix=0, cfg_scale=1.0, scaled=True):
        """model_new.py:568-645 with topk=1, temperature=1 (greedy).  tokens (B,1,9); input_pos (B,) or (1,).
        Returns (B, 9) int32 [text, a0..a7].  Tie-break = lowest index (the reference resolves exact
        ties with the RNG, :141-143; the golden vectors record that no tie occurred).
        cfg_scale > 1 with B > 1 (:618-622, 634-637): row 0 is the conditional prompt, rows 1.. the unconditional one;
        the samplers see l[1:] + (l[0:1] - l[1:]) * cfg_scale and every row continues from that sample.
        scaled (bf16 mode only): the form of the RMSNorm + Linear pairs of a decode frame — True: the product's plans for <= 64
        sequences (row scale applied to the fp32 sums), False: its plans for more than 64 sequences (the prefill form:
        normalised row rounded to bf16) — csrc/ua2_stage3.hip, DESIGN.md §2."""
        B = tokens.size(0)
        cfg = cfg_scale > 1.0 and B > 1
        mix = (lambda l: l[1:] + (l[0:1] - l[1:]) * cfg_scale) if cfg else (lambda l: l)
        rep = (lambda t: t.repeat(2, 1)) if cfg else (lambda t: t)
        pos = input_pos.view(-1, 1).expand(B, 1) if input_pos.numel() in (1, B) else input_pos
        h_final = self._trunk(tokens, tokens_mask, pos, input_pos_maxp1, scaled=scaled)
        last_h = h_final[:, -1, :]
        text_logits = F.linear(self.qa(last_h), self.lm_head)          # :617
        text_logits = mix(text_logits)
        self.last_text_logits = text_logits
        out = [rep(text_logits.argmax(-1, keepdim=True))]
        curr_h = last_h.unsqueeze(1)
        alog = []
        for i in range(self.ncb):                                      # :630-641
            d_in = F.linear(self.qa(curr_h), self.projection)
            d_x = self.decoder.forward(d_in, torch.full((B, 1), i, dtype=torch.long), None, final_norm=False, scaled=scaled)
            lg, = norm_linear(d_x[:, -1, :], self.decoder.ln_f, self.decoder.s.norm_eps, [self.audio_head[i].t()], self.mode, scaled)   # ln_f :164 + :632
            lg = mix(lg)
            alog.append(lg)
            lg2 = lg.clone()
            if forbid_prefix > 0:
                lg2[:, :forbid_prefix] = float("-inf")                  # :168-170
            tok = rep(lg2.argmax(-1, keepdim=True))
            out.append(tok)
            curr_h = self.audio_embeddings[tok + i * self.va]           # :662-663
        self.last_audio_logits = torch.stack(alog, dim=1)               # (B, 8, V_a)
        return torch.cat(out, dim=1).to(torch.int32)


def shapes_from_configs(cfgs: Dict[str, dict]):
    """cfgs: registry-name -> kwargs (tests/golden/toy_configs.py or the real sizes)."""
    pick = lambda d: GPTShape(**{k: d[k] for k in ("n_layer", "n_embd", "n_head", "n_query_groups",
                                                   "intermediate_size", "padded_vocab_size")})
    return dict(backbone=pick(cfgs["Llama-3.2-3B"]), understanding=pick(cfgs["Llama-3.2-Understanding"]),
                generation=pick(cfgs["Llama-3.2-Generation"]),
                decoder=pick(cfgs.get("Llama-3.2-300M") or cfgs["Llama-3.2-4Layer"]))


def run_decode_loop(model, tokens, mask, frames, feedback, forbid_switch=None, reason_card=0, collect_logits=False,
                    cfg_scale=1.0, scaled=True, teacher=None):
    """The generators' loop (evaluation/tts_task.py:244-282 "audio" feedback;
    evaluation/asr_task.py:658-682 "text" feedback) at fixed length (no EOS exit)."""
    B, L, _ = tokens.shape
    model.reset_caches()
    pos = torch.arange(0, L).unsqueeze(0).repeat(B, 1)
    model.forward_prefix(tokens[:, :-1], mask, pos[:, :-1])
    curr_pos = torch.full((B,), L - 1, dtype=torch.long)
    maxp1 = L
    ct, cm = tokens[:, -1:], mask[:, -1:]
    forbid = 0
    samples, tl, al = [], [], []
    for f in range(frames):
        if forbid_switch is not None and f == forbid_switch:
            forbid = reason_card
        kw = {"cfg_scale": cfg_scale} if cfg_scale != 1.0 else {}
        if not scaled:
            kw["scaled"] = False
        s = model.generate_frame(ct, cm, curr_pos, maxp1, forbid_prefix=forbid, **kw)
        samples.append(s)
        if teacher is not None:                  # teacher forcing: continue from the given (frames, B, 9) samples, not from our own
            s = teacher[f]
        if collect_logits:
            tl.append(model.last_text_logits.clone()); al.append(model.last_audio_logits.clone())
        text_tok, audio = s[:, 0:1].long(), s[:, 1:].long()
        if feedback == "audio":
            ct = torch.cat([audio, text_tok], dim=-1).unsqueeze(1)
            cm = torch.cat([torch.ones_like(audio).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
        else:
            ct = torch.cat([torch.zeros_like(audio), text_tok], dim=-1).unsqueeze(1)
            cm = torch.cat([torch.zeros_like(audio).bool(), torch.ones(B, 1).bool()], dim=1).unsqueeze(1)
        curr_pos = curr_pos + 1
        maxp1 += 1
    out = dict(samples=torch.stack(samples))
    if collect_logits:
        out.update(text_logits=torch.stack(tl), audio_logits=torch.stack(al))
    return out

"""CPU ORACLE (test infrastructure, NOT product code) for the neural stages of the live codec (ReasoningCodec_film):
the AudioThinking encoder + FiLM + strided convs that turn SSL features into the three RVQ inputs,
and the code -> latent stage: conditioning assembly, flow-matching DiT, guided Euler ODE.

Plain PyTorch fp32 on the CPU, functional over reference state dicts (same keys); every function cites the reference
file:line it follows (paths relative to /root/reference/tools/tokenizer/ReasoningCodec_film/).

Pinning.  `encode_reasoning_query`, `fetch_codes_from_features`, `inference_codes` and `solve_euler` are checked
against outputs of the reference's OWN code (tests/golden/codec_model_toy.npz, made by
tests/golden/make_golden_codec_model.py which imports and runs models/AudioDiffusion1D.py and modules/transformer.py;
tests/test_oracle_codec_model.py).  PARITY UNPINNED (packages absent from the authoring container, nothing vendored):
  * `dit_forward` restates diffusers' BasicTransformerBlock / Attention / FeedForward / TimestepEmbedding /
    SinusoidalPositionalEmbedding as models/transformer_1d_flow.py and models/attention.py use them (diffusers>=0.25);
  * `ResidualVQOracle` restates vector_quantize_pytorch==1.27.15's eval-mode ResidualVQ (see rvq_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import rvq_oracle


# ---- helpers ------------------------------------------------------------------------------------------------------

def wn_weight(sd, prefix):
    """Effective weight of a `torch.nn.utils.parametrizations.weight_norm` Linear (modules/transformer.py:218,331-343,302):
    original0 = g [out, 1], original1 = v [out, in]; w = g * v / ||v||_row.  Plain `weight` otherwise."""
    k0 = prefix + "parametrizations.weight.original0"
    if k0 in sd:
        return torch._weight_norm(sd[prefix + "parametrizations.weight.original1"].float(), sd[k0].float(), 0)
    return sd[prefix + "weight"].float()


def linear(sd, prefix, x):
    return F.linear(x, wn_weight(sd, prefix), sd.get(prefix + "bias"))


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


# ---- AudioThinking encoder (models/AudioDiffusion1D.py:372-390, 458-486; modules/transformer.py:645-783) ------------

def rotary_freqs(inv_freq, T):
    """modules/transformer.py:120-134 RotaryEmbedding.forward_from_seq_len: freqs = cat(outer(t, inv_freq) x 2)."""
    f = torch.einsum("i,j->ij", torch.arange(T).float(), inv_freq.float())
    return torch.cat((f, f), dim=-1)


def apply_rotary(t, freqs):
    """modules/transformer.py:146-170: partial rotary on the first rot_dim dims, rotate_half = cat(-x2, x1)."""
    rot = freqs.shape[-1]
    a, rest = t[..., :rot], t[..., rot:]
    x1, x2 = a[..., :rot // 2], a[..., rot // 2:]
    a = a * freqs.cos() + torch.cat((-x2, x1), dim=-1) * freqs.sin()
    return torch.cat((a, rest), dim=-1)


def thinking_block(sd, x, dim_heads=128):
    """One TransformerBlock as AudioThinking builds it (AudioDiffusion1D.py:177-179): power_normalized -> no pre / ff
    norms (modules/transformer.py:672-674), weight-normed to_qkv / to_out / GLU proj / linear_out, qk_norm = LayerNorm over
    the head dim (:347-350, 452-455), rotary on max(dim_heads // 2, 32) dims (:740, 457-478), non-causal softmax
    attention (:563-588), sigmoid-GLU feed-forward (:208-243, 262-266), LayerScale on both branches (:690, 708), residuals
    (:773-781)."""
    B, T, D = x.shape
    h = D // dim_heads
    qkv = linear(sd, "self_attn.to_qkv.", x)
    q, k, v = (t.view(B, T, h, dim_heads).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
    q = F.layer_norm(q, (dim_heads,), sd["self_attn.q_norm.weight"], sd["self_attn.q_norm.bias"], 1e-5)
    k = F.layer_norm(k, (dim_heads,), sd["self_attn.k_norm.weight"], sd["self_attn.k_norm.bias"], 1e-5)
    freqs = rotary_freqs(sd["rope.inv_freq"], T)
    q, k = apply_rotary(q, freqs), apply_rotary(k, freqs)
    att = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * dim_heads ** -0.5, dim=-1)
    o = torch.einsum("bhij,bhjd->bhid", att, v).transpose(1, 2).reshape(B, T, D)
    x = x + linear(sd, "self_attn.to_out.", o) * sd["self_attn_scale.scale"]
    a, gate = linear(sd, "ff.ff.0.proj.", x).chunk(2, dim=-1)
    x = x + linear(sd, "ff.ff.2.", a * torch.sigmoid(gate)) * sd["ff_scale.scale"]
    return x


def set_masking(x, cls_token, interval):
    """AudioDiffusion1D.py:458-476: one cls token after every `interval` frames."""
    B, T, D = x.shape
    n = T // interval
    xr = x.reshape(B, n, interval, D)
    cls = cls_token.view(1, 1, 1, D).expand(B, n, 1, D)
    return torch.cat([xr, cls], dim=2).reshape(B, -1, D)


def extract_mask_positions(x, interval):
    """AudioDiffusion1D.py:478-486."""
    new_T = x.shape[1]
    original_T = new_T - new_T // (interval + 1)
    idx = [(i + 1) * (interval + 1) - 1 for i in range(original_T // interval)]
    return x[:, idx, :]


@torch.no_grad()
def encode_reasoning_query(sd, whisper_embeds, muencoder_embeds, interval=5, dim_heads=128):
    """AudioDiffusion1D.py:372-387 up to (not including) reasoning_vq.  sd = the `audio_thinking.` sub-dict;
    whisper_embeds (B, Cw, T50), muencoder_embeds (B, 1024, T25) -> query tokens (B, T25 // 5, D)."""
    w = F.conv1d(whisper_embeds, sd["down_sampling_layer_whisper.weight"], sd["down_sampling_layer_whisper.bias"], stride=2)
    w, m = w.transpose(1, 2), muencoder_embeds.transpose(1, 2)
    n = min(w.shape[1], m.shape[1])
    x = linear(sd, "semantic_merge_proj.", torch.cat((w[:, :n], m[:, :n]), dim=-1))
    x = set_masking(x, sd["cls_token"], interval)
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("encoder_transformers."))
    for i in range(depth):
        x = thinking_block(sub(sd, f"encoder_transformers.{i}."), x, dim_heads)
    return extract_mask_positions(x, interval)


class ResidualVQOracle:
    """Eval-mode `vector_quantize_pytorch.ResidualVQ` as AudioDiffusion1D.py:183-187,256-264 builds it (PARITY UNPINNED,
    see the header): project_in -> per level nearest codeword by squared L2 on the residual (lowest index on ties) ->
    sum -> project_out.  The search itself is oracle/rvq_oracle.c."""

    def __init__(self, sd):
        self.w_in, self.b_in = sd.get("project_in.weight"), sd.get("project_in.bias")
        self.w_out, self.b_out = sd.get("project_out.weight"), sd.get("project_out.bias")
        L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("layers."))
        self.emb = torch.cat([sd[f"layers.{l}._codebook.embed"].float() for l in range(L)], 0).contiguous()     # [L, C, D]

    def __call__(self, x):
        B, T, _ = x.shape
        h = x.reshape(B * T, -1).float()
        if self.w_in is not None:
            h = F.linear(h, self.w_in.float(), self.b_in.float())
        codes, q = rvq_oracle.rvq_encode(np.ascontiguousarray(h.numpy()), self.emb.numpy())
        return self._out(torch.from_numpy(q)).view(B, T, -1), torch.from_numpy(codes.astype(np.int64)).view(B, T, -1), h

    def _out(self, q):
        return F.linear(q, self.w_out.float(), self.b_out.float()) if self.w_out is not None else q

    def get_output_from_indices(self, idx):
        B, T, L = idx.shape
        q = rvq_oracle.rvq_decode(np.ascontiguousarray(idx.reshape(B * T, L).numpy().astype(np.int32)), self.emb.numpy())
        return self._out(torch.from_numpy(q)).view(B, T, -1)


def time_film(sd, prefix, cond_seq, feats, mask, gamma=0.1):
    """AudioDiffusion1D.py:428-438; `mask` (B,) bool = the reference's `torch.rand(B, 1, 1) < 0.2` draw, passed in."""
    dg, beta = linear(sd, prefix, cond_seq).chunk(2, dim=-1)
    g = 1.0 + gamma * dg.tanh()
    m = mask.view(-1, 1, 1).float()
    return (g * (1 - m) + 1.0 * m) * feats + beta * (1 - m)


@torch.no_grad()
def fetch_codes_from_features(sd, whisper, wavlm, bestrq_acoustic, bestrq_semantic, film_masks, vqs=None):
    """AudioDiffusion1D.py:493-551 from the SSL features on.  sd = the AudioDiffusion1D state dict (in-scope keys);
    film_masks (3, B) bool in call order (phone, semantic, acoustic); vqs = dict(reason, phone, semantic, acoustic) of
    callables x -> (quantized, codes, ...) or None for identity quantisers (what the golden generator uses).
    Returns the intermediate tensors the goldens hold plus the codes."""
    ident = lambda x: (x, torch.zeros(x.shape[0], x.shape[1], 1, dtype=torch.long))
    vqs = vqs or {}
    conv = lambda name, x, s: F.conv1d(x, sd[name + ".weight"], sd[name + ".bias"], stride=s)
    whisper_rec = conv("d_conv_whisper", whisper, 4)                                   # :514
    wavlm_f = conv("d_conv_wavlm", wavlm, 4)                                           # :515
    sem_rec = conv("d_conv_embedding_semantic", bestrq_semantic, 2)                    # :516
    ac = conv("d_conv_embedding_acoustic", bestrq_acoustic, 2)                         # :517
    query = encode_reasoning_query(sub(sd, "audio_thinking."), whisper, bestrq_semantic)
    q_reason, reason_codes = (vqs.get("reason") or ident)(query)[:2]                   # :388
    rf = linear(sd, "reason_adaptor.", q_reason)                                       # :521
    rf = F.interpolate(rf.permute(0, 2, 1), scale_factor=2.5, mode="nearest").permute(0, 2, 1)     # :522
    phone = linear(sd, "cond_fusion_layer_phone.", wavlm_f.transpose(1, 2))            # :525
    phone = time_film(sd, "time_film_phone.", rf, phone, film_masks[0])                # :527
    q_phone, c_phone = (vqs.get("phone") or ident)(phone)[:2]                          # :528
    sem = linear(sd, "cond_fusion_layer_semantic.", sem_rec.transpose(1, 2))           # :531
    sem = time_film(sd, "time_film_semantic.", rf, sem, film_masks[1])                 # :533
    q_sem, c_sem = (vqs.get("semantic") or ident)(sem)[:2]                             # :534
    n = min(ac.shape[-1], whisper_rec.shape[-1])                                       # :538
    acf = torch.cat([ac[:, :, :n], whisper_rec[:, :, :n]], dim=1)
    acf = linear(sd, "cond_fusion_layer_acoustic.", acf.transpose(1, 2))               # :540
    acf = time_film(sd, "time_film_acoustic.", rf, acf, film_masks[2])                 # :542
    q_ac, c_ac = (vqs.get("acoustic") or ident)(acf)[:2]                               # :543
    merge = linear(sd, "cond_feature_emb.", q_phone + q_sem + q_ac)                    # :544-546
    return dict(reason_query=query, reason_codes=reason_codes, pre_vq_phone=phone, pre_vq_semantic=sem, pre_vq_acoustic=acf,
                merge_features=merge, merge_codes=torch.cat([c_phone, c_sem, c_ac], dim=-1))      # :549 order phone, semantic, acoustic


# ---- code -> latent: conditioning, Euler ODE (AudioDiffusion1D.py:89-129, 554-624) ---------------------------------

@torch.no_grad()
def solve_euler(estimator, x, incontext_x, incontext_length, t_span, mu, guidance_scale, sigma_min=1e-4):
    """AudioDiffusion1D.py:89-129 (guided branch :105-116; the un-guided branch concatenates on the wrong axis, SURVEY
    A.10, and is unreachable: callers pass guidance_scale = 1.5).  estimator(x_cat (2B, T, 2L + D), timestep (2,)) -> (2B, T, L)."""
    assert guidance_scale > 1.0
    t, dt = t_span[0], t_span[1] - t_span[0]
    noise = x.clone()
    for step in range(1, len(t_span)):
        x[:, 0:incontext_length, :] = (1 - (1 - sigma_min) * t) * noise[:, 0:incontext_length, :] + t * incontext_x[:, 0:incontext_length, :]
        inp = torch.cat([torch.cat([x, x], 0), torch.cat([incontext_x, incontext_x], 0), torch.cat([torch.zeros_like(mu), mu], 0)], 2)
        d = estimator(inp, t.unsqueeze(-1).repeat(2))
        d_uncond, d_cond = d.chunk(2, 0)
        x = x + dt * (d_uncond + guidance_scale * (d_cond - d_uncond))
        t = t + dt
        if step < len(t_span) - 1:
            dt = t_span[step + 1] - t
    return x


@torch.no_grad()
def codes_to_condition(lookups, cfe_w, cfe_b, codes):
    """AudioDiffusion1D.py:563-590 (no reasoning codes: token2audio_no_reason passes one code tensor): split [0:1] phone,
    [1:2] semantic, [2:] acoustic; sum of the three look-ups; cond_feature_emb; x2 nearest up-sampling along time."""
    parts = (codes[:, 0:1, :], codes[:, 1:2, :], codes[:, 2:, :])
    q = sum(lk(p.transpose(1, 2)) for lk, p in zip(lookups, parts))
    m = F.linear(q, cfe_w, cfe_b)
    return F.interpolate(m.permute(0, 2, 1), scale_factor=2, mode="nearest").permute(0, 2, 1)


@torch.no_grad()
def inference_codes(lookups, cfe_w, cfe_b, zero_cond, estimator, codes, true_latents, latent_length, incontext_length, noise,
                    guidance_scale=1.5, num_steps=20):
    """AudioDiffusion1D.py:554-624 with scenario='other_seg' (the only one token2audio_no_reason uses, :271,280) and no
    speaker embedding; `noise` = the prepare_latents draw (:655), passed in."""
    merge = codes_to_condition(lookups, cfe_w, cfe_b, codes)
    B, T, _ = merge.shape
    masks = torch.zeros(B, T, dtype=torch.int64)
    masks[:, 0:latent_length] = 2
    masks[:, 0:incontext_length] = 1                                                              # :603-604
    merge = (masks > 0.5).unsqueeze(-1) * merge + (masks < 0.5).unsqueeze(-1) * zero_cond.unsqueeze(0)   # :606-607
    inc = true_latents * ((masks > 0.5) * (masks < 1.5)).unsqueeze(-1).float()                    # :609
    n_inc = int(((masks > 0.5) * (masks < 1.5)).sum(-1)[0])                                       # :610
    t_span = torch.linspace(0, 1, num_steps + 1)
    lat = solve_euler(estimator, noise.clone(), inc, n_inc, t_span, merge, guidance_scale)
    lat[:, 0:n_inc, :] = inc[:, 0:n_inc, :]                                                       # :623
    return lat


# ---- flow-matching DiT (models/transformer_1d_flow.py:162-386, models/attention.py:97-420) — PARITY UNPINNED ----------

def timestep_embedding(t, dim=512, max_period=10000, scale=1000):
    """transformer_1d_flow.py:57-72."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t.float()[:, None] * freqs[None] * scale
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def sinusoidal_pos_embed(T, dim):
    """diffusers SinusoidalPositionalEmbedding as transformer_1d_flow.py:232 uses it: pe[:, 0::2] = sin, pe[:, 1::2] = cos."""
    pos = torch.arange(T).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, dim, 2).float() * (-math.log(10000.0) / dim))
    pe = torch.zeros(T, dim)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def project_layer(sd, p, x, k=3):
    """transformer_1d_flow.py:19-33 ProjectLayer: Conv1d(k, padding k // 2) over time, * k^-0.5, Linear."""
    y = F.conv1d(x.transpose(1, 2), sd[p + "ffn_1.weight"], sd[p + "ffn_1.bias"], padding=k // 2).transpose(1, 2)
    return F.linear(y * k ** -0.5, sd[p + "ffn_2.weight"], sd[p + "ffn_2.bias"])


@torch.no_grad()
def dit_forward(sd: Dict[str, torch.Tensor], hidden, timestep, heads: int, head_dim: int, eps: float = 1e-6):
    """Transformer1DModel.forward (transformer_1d_flow.py:279-386) with norm_type='ada_norm_single',
    activation 'gelu-approximate', attention_bias=True, no cross attention (models/model_config.json).
    hidden (B, T, in_channels), timestep (B,) in [0, 1] -> (B, T, out_channels)."""
    sd = {k: v.float() for k, v in sd.items()}
    B, T, _ = hidden.shape
    D = heads * head_dim
    h = project_layer(sd, "proj_in.", hidden.float()) + sinusoidal_pos_embed(T, D)[None]                   # :332-338
    emb = timestep_embedding(timestep)                                                                    # adaln_single :347
    emb = F.linear(F.silu(F.linear(emb, sd["adaln_single.emb.timestep_embedder.linear_1.weight"], sd["adaln_single.emb.timestep_embedder.linear_1.bias"])),
                   sd["adaln_single.emb.timestep_embedder.linear_2.weight"], sd["adaln_single.emb.timestep_embedder.linear_2.bias"])
    ts = F.linear(F.silu(emb), sd["adaln_single.linear.weight"], sd["adaln_single.linear.bias"])           # (B, 6D)  :113
    n_layers = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("transformer_blocks."))
    for i in range(n_layers):
        p = f"transformer_blocks.{i}."
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = (sd[p + "scale_shift_table"][None] + ts.reshape(B, 6, -1)).chunk(6, dim=1)   # attention.py:308-310
        n = F.layer_norm(h, (D,), None, None, eps) * (1 + sc_a) + sh_a                                     # :311-319
        q, k, v = (F.linear(n, sd[p + f"attn1.to_{c}.weight"], sd[p + f"attn1.to_{c}.bias"]).view(B, T, heads, head_dim).transpose(1, 2)
                   for c in "qkv")
        att = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * head_dim ** -0.5, dim=-1)
        o = torch.einsum("bhij,bhjd->bhid", att, v).transpose(1, 2).reshape(B, T, D)
        h = g_a * F.linear(o, sd[p + "attn1.to_out.0.weight"], sd[p + "attn1.to_out.0.bias"]) + h           # :345-349
        n = F.layer_norm(h, (D,), None, None, eps) * (1 + sc_m) + sh_m                                     # :388-390
        f = F.gelu(F.linear(n, sd[p + "ff.net.0.proj.weight"], sd[p + "ff.net.0.proj.bias"]), approximate="tanh")
        h = g_m * F.linear(f, sd[p + "ff.net.2.weight"], sd[p + "ff.net.2.bias"]) + h                      # :401-405
    shift, scale = (sd["scale_shift_table"][None] + emb[:, None]).chunk(2, dim=1)                          # transformer_1d_flow.py:378
    h = F.layer_norm(h, (D,), None, None, 1e-6) * (1 + scale) + shift                                      # :379-381
    return project_layer(sd, "proj_out.", h)                                                              # :384

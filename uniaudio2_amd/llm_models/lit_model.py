"""Transformer core of the decode loop, host side.

Mirror of the reference's llm_models/lit_model.py for the classes on the hot path
(GPT :22-275, Block :278-349, CausalSelfAttention :352-581, LLaMAMLP :582-595, KVCache
:814-860, RMSNorm :869-890, build_rope_cache :634-706): same class and parameter names, hence the
same state-dict keys (`transformer.h.{i}.attn.qkv.weight` with fused [q;k;v] rows, ...), same
`forward(x, input_pos, input_pos_maxp1)` contract (embeddings in, hidden after ln_f out, :180).

The modules only *hold* parameters.  The arithmetic runs in libua2hip.so: `GPT.set_kv_cache`
packs the weights into MFMA-fragment order, allocates the paged KV pools and publishes a
`GptDesc` for the frame executor; `GPT.forward` drives the same kernels op by op (used by the
parity tests and for stand-alone GPTs).  No torch math on the data path.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import ops
from .._lib import (EPI_QKV_ROPE, EPI_RESIDUAL, EPI_SWIGLU, PRO_CAST, PRO_NORM, UA2_PAGE,
                    GptDesc, vp)
from .config import Config


def build_rope_cache(seq_len: int, n_elem: int, base: int = 10000, extra_config: Optional[dict] = None):
    """cos/sin tables [seq_len, n_elem/2] (the reference stores both halves, which are equal;
    lit_model.py:684).  Llama-3 frequency smoothing as lit_model.py:662-676.  Built on the host
    in fp32 with the same torch ops so the table bits match the reference's."""
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2).float() / n_elem))
    if extra_config is not None:
        factor = extra_config["factor"]
        if "original_max_seq_len" in extra_config:
            ratio = extra_config["original_max_seq_len"] / (2 * torch.pi / theta)
            smooth = torch.clamp((ratio - extra_config["low_freq_factor"]) /
                                 (extra_config["high_freq_factor"] - extra_config["low_freq_factor"]), 0.0, 1.0)
            theta = (1 - smooth) * (theta / factor) + smooth * theta
        else:
            theta = theta / factor
    idx_theta = torch.outer(torch.arange(seq_len) / 1, theta)
    return torch.cos(idx_theta), torch.sin(idx_theta)


class RMSNorm(nn.Module):
    def __init__(self, size: int, eps: float = 1e-5, device=None):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(size, device=device))
        self.eps = eps


class LLaMAMLP(nn.Module):
    def __init__(self, config: Config, device=None):
        super().__init__()
        self.fc_1 = nn.Linear(config.n_embd, config.intermediate_size, bias=False, device=device)
        self.fc_2 = nn.Linear(config.n_embd, config.intermediate_size, bias=False, device=device)
        self.proj = nn.Linear(config.intermediate_size, config.n_embd, bias=False, device=device)


class CausalSelfAttention(nn.Module):
    def __init__(self, config: Config, block_idx: int, device=None):
        super().__init__()
        self.qkv = nn.Linear(config.n_embd, (config.n_head + 2 * config.n_query_groups) * config.head_size,
                             bias=False, device=device)
        self.proj = nn.Linear(config.head_size * config.n_head, config.n_embd, bias=False, device=device)
        self.block_idx = block_idx


class Block(nn.Module):
    def __init__(self, config: Config, block_idx: int, device=None):
        super().__init__()
        self.norm_1 = RMSNorm(config.n_embd, eps=config.norm_eps, device=device)
        self.attn = CausalSelfAttention(config, block_idx, device=device)
        self.norm_2 = RMSNorm(config.n_embd, eps=config.norm_eps, device=device)
        self.mlp = LLaMAMLP(config, device=device)


class KVCache:
    """Paged K/V pools of one GPT: per layer [n_pages, n_kv, 64, head_size] (K and V), plus the
    page table [max_batch, max_pages].  Replaces the reference's dense (B, n_kv, max_seq, hs)
    buffers (lit_model.py:814-860); positions are still absolute `input_pos`."""

    def __init__(self, config: Config, n_layer, batch_size, max_seq_length, dtype, device):
        self.max_pages = (max_seq_length + UA2_PAGE - 1) // UA2_PAGE
        n_pages = batch_size * self.max_pages
        shape = (n_pages, config.n_query_groups, UA2_PAGE, config.head_size)
        self.k = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(n_layer)]
        self.v = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(n_layer)]
        # static assignment: sequence b owns pages [b*max_pages, (b+1)*max_pages); the kernels only
        # ever see the table, so a free-list allocator can replace this without touching them
        self.page_table = torch.arange(n_pages, dtype=torch.int32, device=device).view(batch_size, self.max_pages)

    def zero_(self):
        for t in self.k + self.v:
            t.zero_()


def _ptr_array(tensors):
    arr = (vp * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr


class GPT(nn.Module):
    def __init__(self, config: Config, device=None, with_embeddings: bool = True):
        super().__init__()
        self.config = config
        if with_embeddings:
            self.lm_head = nn.Linear(config.n_embd, config.padded_vocab_size, bias=False, device=device)
            wte = nn.Embedding(config.padded_vocab_size, config.n_embd, device=device)
        else:                                   # model_new.py:111-115 _prepare_transformer
            self.lm_head = nn.Identity()
            wte = nn.Identity()
        self.transformer = nn.ModuleDict(dict(
            wte=wte,
            h=nn.ModuleList(Block(config, i, device=device) for i in range(config.n_layer)),
            ln_f=RMSNorm(config.n_embd, eps=config.norm_eps, device=device)))
        self.max_seq_length = config.block_size
        self.kv_cache: Optional[KVCache] = None
        self.plan = None

    @classmethod
    def from_name(cls, name: str, **kwargs):
        return cls(Config.from_name(name, **kwargs))

    # ---- device plan -----------------------------------------------------------------------
    def set_kv_cache(self, batch_size: int, max_seq_length: Optional[int] = None, rope_cache_length=None,
                     device=None, dtype=None):
        """lit_model.py:224-254.  Also (re)packs the weights for `dtype` (torch.float32 or bfloat16)."""
        cfg = self.config
        p0 = self.transformer.h[0].attn.qkv.weight
        device = device or p0.device
        dtype = dtype or p0.dtype
        if device.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback); move the model to cuda")
        max_seq_length = max_seq_length or self.max_seq_length
        self.kv_cache = KVCache(cfg, cfg.n_layer, batch_size, max_seq_length, dtype, device)
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        pk = lambda lin: ops.pack_linear(lin.weight.detach().to(device), dtype)
        plan = dict(dtype=dtype, device=device, batch=batch_size, max_seq=max_seq_length)
        plan["qkv"] = [ops.pack_linear(b.attn.qkv.weight.detach().to(device), dtype, rope_head_size=cfg.head_size)
                       for b in self.transformer.h]
        plan["proj"] = [pk(b.attn.proj) for b in self.transformer.h]
        plan["fc1"] = [pk(b.mlp.fc_1) for b in self.transformer.h]
        plan["fc2"] = [pk(b.mlp.fc_2) for b in self.transformer.h]
        plan["mlp_proj"] = [pk(b.mlp.proj) for b in self.transformer.h]
        plan["norm1"] = [f32(b.norm_1.weight) for b in self.transformer.h]
        plan["norm2"] = [f32(b.norm_2.weight) for b in self.transformer.h]
        plan["ln_f"] = f32(self.transformer.ln_f.weight)
        cos, sin = build_rope_cache(max_seq_length, cfg.rope_n_elem, cfg.rope_base, cfg.rope_adjustments)
        plan["cos"], plan["sin"] = cos.to(device).contiguous(), sin.to(device).contiguous()
        # scratch for the op-by-op forward
        self.plan = plan
        return self

    def reset_kv_cache(self, zero_memory: bool = True):
        """lit_model.py:256-263.  Slots are always overwritten before they become visible (the
        attention length is row_pos+1), so zeroing is not needed for correctness; kept for parity
        of observable state."""
        if self.kv_cache is not None and zero_memory:
            self.kv_cache.zero_()

    def clear_kv_cache(self):
        self.kv_cache = None
        self.plan = None

    def desc(self) -> GptDesc:
        cfg, p, kv = self.config, self.plan, self.kv_cache
        d = GptDesc()
        d.n_layer, d.n_embd, d.n_head, d.n_kv = cfg.n_layer, cfg.n_embd, cfg.n_head, cfg.n_query_groups
        d.head_size, d.inter, d.eps = cfg.head_size, cfg.intermediate_size, cfg.norm_eps
        keep = {}
        for f, k in (("qkv", "qkv"), ("proj", "proj"), ("fc1", "fc1"), ("fc2", "fc2"), ("mlp_proj", "mlp_proj"),
                     ("norm1", "norm1"), ("norm2", "norm2")):
            keep[f] = _ptr_array(p[k])
            setattr(d, f, keep[f])
        keep["k"], keep["v"] = _ptr_array(kv.k), _ptr_array(kv.v)
        d.k_pool, d.v_pool = keep["k"], keep["v"]
        d.ln_f, d.rope_cos, d.rope_sin = p["ln_f"].data_ptr(), p["cos"].data_ptr(), p["sin"].data_ptr()
        d.page_table, d.max_pages = kv.page_table.data_ptr(), kv.max_pages
        d._keep = keep      # keep the ctypes arrays alive as long as the descriptor
        return d

    # ---- op-by-op forward (same kernels as the frame executor) ------------------------------
    def forward(self, x: torch.Tensor, input_pos: Optional[torch.Tensor] = None,
                input_pos_maxp1: Optional[int] = None) -> torch.Tensor:
        """x (B, T, C) fp32 on device; input_pos (T,) or (B, T) absolute positions.  Returns the
        hidden state after ln_f (lit_model.py:164,180).  `input_pos_maxp1` only bounds how much
        of the cache the reference reads (:141-145, 468-471); here the bound is exact per row."""
        if self.plan is None:
            raise TypeError("You need to call `gpt.set_kv_cache()`")
        cfg, p, kv = self.config, self.plan, self.kv_cache
        B, T, Cc = x.shape
        dev = x.device
        if input_pos is None:
            input_pos = torch.arange(T, device=dev)
        if input_pos.dim() > 2:
            raise ValueError(f"input_pos must have 1 or 2 dimensions, input_pos.shape = {input_pos.shape}")
        if input_pos.shape[-1] != T:
            raise ValueError(f"input_pos.shape[-1] = {input_pos.shape[-1]} != {T} = idx.shape[1], must be the same")
        pos = (input_pos.unsqueeze(0).expand(B, T) if input_pos.dim() == 1 else input_pos)
        row_pos = pos.reshape(-1).to(torch.int32).contiguous()
        row_seq = torch.arange(B, device=dev, dtype=torch.int32).unsqueeze(1).expand(B, T).reshape(-1).contiguous()
        R = B * T
        xs = x.reshape(R, Cc).to(torch.float32).contiguous().clone()
        self._layers(xs, R, row_pos, row_seq)
        out = ops.rmsnorm_blend(xs, p["ln_f"], cfg.norm_eps)
        return out.view(B, T, Cc)

    def _layers(self, xs, R, row_pos, row_seq):
        cfg, p, kv = self.config, self.plan, self.kv_cache
        dt, dev = p["dtype"], xs.device
        nh, ng, hs, Cc, I = cfg.n_head, cfg.n_query_groups, cfg.head_size, cfg.n_embd, cfg.intermediate_size
        q = torch.empty(R, nh * hs, dtype=torch.float32, device=dev)
        act = torch.empty(R, I, dtype=torch.float32, device=dev)
        ya = torch.empty(R, nh * hs, dtype=torch.float32, device=dev)
        for l in range(cfg.n_layer):
            geom = ops.kv_geom(kv.k[l], kv.v[l], kv.page_table, nh, ng, hs)
            ops.linear(dtype=dt, M=R, N=(nh + 2 * ng) * hs, K=Cc, w0=p["qkv"][l], prologue=PRO_NORM,
                       epilogue=EPI_QKV_ROPE, x=xs, norm_w=p["norm1"][l], eps=cfg.norm_eps, row_pos=row_pos,
                       row_seq=row_seq, rope_cos=p["cos"], rope_sin=p["sin"], q_out=q, kv=geom)
            ops.attn(dtype=dt, R=R, q=q, row_pos=row_pos, row_seq=row_seq, kv=geom, y=ya)
            ops.linear(dtype=dt, M=R, N=Cc, K=nh * hs, w0=p["proj"][l], prologue=PRO_CAST, epilogue=EPI_RESIDUAL,
                       x=ya, y=xs, resid=xs)
            ops.linear(dtype=dt, M=R, N=I, K=Cc, w0=p["fc1"][l], w1=p["fc2"][l], prologue=PRO_NORM,
                       epilogue=EPI_SWIGLU, x=xs, norm_w=p["norm2"][l], eps=cfg.norm_eps, y=act)
            ops.linear(dtype=dt, M=R, N=Cc, K=I, w0=p["mlp_proj"][l], prologue=PRO_CAST, epilogue=EPI_RESIDUAL,
                       x=act, y=xs, resid=xs)

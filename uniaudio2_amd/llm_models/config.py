"""Model-shape registry for the hot path: the five Llama-3.2 entries the reference's
Model_stage3 instantiates (reference llm_models/config.py:804-899, looked up by
Config.from_name :136-154).  The reference's 3.1k-line model zoo (other families, MoE,
Gemma, sliding window...) is out of scope.
"""
from dataclasses import dataclass, field
from typing import Any, Optional

_LLAMA3_ROPE = dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_seq_len=8192)


@dataclass
class Config:
    name: str = ""
    block_size: int = 131072
    vocab_size: int = 128000
    padded_vocab_size: int = 128256
    n_layer: int = 28
    n_embd: int = 3072
    n_head: int = 24
    n_query_groups: Optional[int] = 8
    head_size: Optional[int] = None
    intermediate_size: int = 8192
    rotary_percentage: float = 1.0
    parallel_residual: bool = False
    bias: bool = False
    norm_class_name: str = "RMSNorm"
    mlp_class_name: str = "LLaMAMLP"
    norm_eps: float = 1e-5                      # reference config.py:38
    rope_base: int = 500000
    rope_adjustments: Optional[dict] = field(default_factory=lambda: dict(_LLAMA3_ROPE))
    hf_config: dict = field(default_factory=dict)

    def __post_init__(self):
        if self.head_size is None:
            assert self.n_embd % self.n_head == 0
            self.head_size = self.n_embd // self.n_head
        if self.n_query_groups is None:
            self.n_query_groups = self.n_head
        assert self.n_head % self.n_query_groups == 0
        self.rope_n_elem = int(self.rotary_percentage * self.head_size)
        unsupported = []
        if self.rotary_percentage != 1.0: unsupported.append("rotary_percentage != 1")
        if self.parallel_residual: unsupported.append("parallel_residual")
        if self.bias: unsupported.append("bias")
        if self.norm_class_name != "RMSNorm": unsupported.append(self.norm_class_name)
        if self.mlp_class_name != "LLaMAMLP": unsupported.append(self.mlp_class_name)
        if unsupported:
            raise NotImplementedError(f"config {self.name!r}: outside the hot path: {', '.join(unsupported)}")

    @classmethod
    def from_name(cls, name: str, **kwargs: Any) -> "Config":
        key = name.split("/")[-1]
        if key not in name_to_config:
            raise ValueError(f"{name!r} is not a supported config name")
        conf = dict(name_to_config[key])
        conf.update(kwargs)
        return cls(**conf)


def _llama(name, n_layer, n_embd, n_head):
    return dict(name=name, hf_config=dict(org="meta-llama", name=name), n_layer=n_layer, n_embd=n_embd,
                n_head=n_head, n_query_groups=8, intermediate_size=8192)


configs = []
for _kind in ("", "-Instruct"):
    configs += [
        _llama("Llama-3.2-300M" + _kind, 4, 2048, 32),
        _llama("Llama-3.2-Understanding" + _kind, 3, 3072, 24),
        _llama("Llama-3.2-Generation" + _kind, 2, 3072, 24),
        _llama("Llama-3.2-4Layer" + _kind, 4, 2048, 32),
        _llama("Llama-3.2-3B" + _kind, 28, 3072, 24),
    ]
name_to_config = {c["name"]: c for c in configs}

"""Llama-3.2 BPE wrapper — host-side mirror of the reference's tools/tokenizer/Text2ID/text_tokenizer.py
(TextTokenizer :12, tokenize :149-166, decode :76).  Pure host code (HF `tokenizers`); out of the
kernel scope but part of the CLI contract: BOS and EOS are always added.

`tokenizer.json` is not shipped with the reference (.MISSING_LARGE_BLOBS), so synthetic runs pass
pre-tokenised ids instead (`IdPassthroughTokenizer`)."""
import json
from pathlib import Path
from typing import Union

import torch


class TextTokenizer:
    def __init__(self, checkpoint_dir: Union[Path, str], max_length=-1) -> None:
        checkpoint_dir = Path(checkpoint_dir)
        if not checkpoint_dir.exists():
            raise NotADirectoryError(f"The checkpoint directory does not exist: {str(checkpoint_dir)}")
        vocabulary_path = checkpoint_dir / "tokenizer.json"
        if not vocabulary_path.is_file():
            raise FileNotFoundError(f"No tokenizer.json in {str(checkpoint_dir)}")
        from tokenizers import Tokenizer as HFTokenizer
        self.model = HFTokenizer.from_file(str(vocabulary_path))
        self.backend = "huggingface"
        self.bos_id, self.eos_id = 128000, 128001
        cfg = checkpoint_dir / "tokenizer_config.json"
        if cfg.is_file():
            with open(cfg, encoding="utf-8") as fp:
                config = json.load(fp)
            for name in ("bos", "eos"):
                tok = config.get(f"{name}_token")
                if isinstance(tok, dict):
                    tok = tok.get("content")
                if tok is not None and self.model.token_to_id(tok) is not None:
                    setattr(self, f"{name}_id", self.model.token_to_id(tok))
        self.pad_id, self.epad_id = 128004, 128005
        self.use_bos = self.use_eos = True
        self.max_length = max_length

    @property
    def is_discrete(self):
        return True

    def tokenize(self, text):
        ids = self.model.encode(text).ids
        if self.use_bos and (not ids or ids[0] != self.bos_id):
            ids = [self.bos_id] + ids
        if self.use_eos and (not ids or ids[-1] != self.eos_id):
            ids = ids + [self.eos_id]
        if self.max_length > 0:
            ids = ids[:self.max_length]
        return ids

    def decode(self, tensor: torch.Tensor) -> str:
        tokens = [tensor.item()] if tensor.ndim == 0 else tensor.tolist()
        return self.model.decode(tokens)


class IdPassthroughTokenizer:
    """Stand-in when no tokenizer.json exists (synthetic runs): `tokenize` accepts a string of
    space-separated ids, `decode` prints ids."""
    bos_id, eos_id = 128000, 128001

    def tokenize(self, text):
        ids = [int(t) for t in str(text).split()]
        if not ids or ids[0] != self.bos_id:
            ids = [self.bos_id] + ids
        if ids[-1] != self.eos_id:
            ids = ids + [self.eos_id]
        return ids

    def decode(self, tensor):
        tokens = [tensor.item()] if tensor.ndim == 0 else tensor.tolist()
        return " ".join(str(int(t)) for t in tokens)


def load_text_tokenizer(path):
    if path is None or str(path) == "ids":
        return IdPassthroughTokenizer()
    return TextTokenizer(path)

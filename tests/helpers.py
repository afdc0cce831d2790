"""Shared builders for the parity tests (toy-sized models with seeded weights)."""
import json
import os

import numpy as np
import torch

from toy_configs import TOY_LM, TOY_MODEL_ARGS
from weights import checksum, seeded_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden_lm():
    d = np.load(os.path.join(GOLDEN, "lm_toy_fp32.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "lm_toy_fp32.json")))
    return d, meta


def toy_state_dict(meta):
    shapes = {k: tuple(s) for k, s in meta["keys"]}
    sd = seeded_state_dict(shapes, meta["seed"])
    assert np.allclose(checksum(sd), meta["checksum"], rtol=1e-12), "seeded weight generator drifted"
    return sd


class toy_registry:
    """Context manager: the same shrink the golden generator applies to the reference registry, undone on exit so that
    tests building the real sizes in the same process (bench.build_model) never see toy entries."""

    def __enter__(self):
        import copy
        from uniaudio2_amd.llm_models import config as cfg
        self.cfg, self.saved = cfg, copy.deepcopy(cfg.name_to_config)
        for name, kw in TOY_LM.items():
            keep = {k: v for k, v in kw.items() if k in cfg.Config.__dataclass_fields__}
            for key in (name, name + "-Instruct"):
                cfg.name_to_config[key].update(keep)
        return self

    def __exit__(self, *exc):
        self.cfg.name_to_config.clear()
        self.cfg.name_to_config.update(self.saved)
        return False


def build_toy_module():
    """Model_stage3 at the toy sizes, on CPU, registry restored."""
    from uniaudio2_amd.llm_models.model_new import Model_stage3, ModelArgs
    with toy_registry():
        return Model_stage3(ModelArgs(**TOY_MODEL_ARGS))


def build_product_model(sd, dtype, batch=1, device="cuda", **setup_kw):
    m = build_toy_module()
    m.load_state_dict(sd, strict=True)
    m = m.to(device)
    m.setup_caches(batch, dtype=dtype, **setup_kw)
    return m


def build_oracle(sd, mode, batch=1):
    from oracle.lm_oracle import Stage3Oracle, shapes_from_configs
    o = Stage3Oracle(sd, shapes_from_configs(TOY_LM), TOY_MODEL_ARGS["audio_semantic_vocab_size"],
                     TOY_MODEL_ARGS["audio_reason_vocab_size"], TOY_MODEL_ARGS["audio_num_codebooks"], mode=mode)
    o.setup_caches(batch)
    return o


def product_decode_loop(model, tokens, mask, frames, feedback, forbid_switch=None, reason_card=0, fast=False,
                        collect_logits=False, cfg_scale=1.0):
    """Same protocol as oracle.lm_oracle.run_decode_loop, through the product's reference-compatible API
    (fast=False: forward_prefix + generate_frame per frame) or its on-device loop (fast=True)."""
    dev = next(model.parameters()).device
    tokens, mask = tokens.to(dev), mask.to(dev)
    B, L, _ = tokens.shape
    model.reset_caches()
    pos = torch.arange(0, L, device=dev).unsqueeze(0).repeat(B, 1)
    model.forward_prefix(tokens[:, :-1], labels=tokens[:, 1:, :-1], tokens_mask=mask, loss_mask=mask,
                         input_pos=pos[:, :-1])
    curr_pos = torch.tensor([L - 1], device=dev, dtype=torch.int64)
    maxp1 = L
    ct, cm = tokens[:, -1:], mask[:, -1:]
    if fast:
        assert forbid_switch is None
        model.set_cfg(cfg_scale)
        model.begin_decode(ct, cm, curr_pos)
        mode = (2 if cfg_scale > 1.0 else 0) if feedback == "audio" else 1
        log = model.generate_frames(frames, B, mode, reason_eos=-1, reason_card=reason_card)
        model.set_cfg(1.0)
        return dict(samples=log.cpu())
    forbid = 0
    samples, tl, al = [], [], []
    for f in range(frames):
        if forbid_switch is not None and f == forbid_switch:
            forbid = reason_card
        s = model.generate_frame(ct, cm, input_pos=curr_pos, input_pos_maxp1=maxp1, temperature=1.0, topk=1,
                                 forbid_prefix=forbid, cfg_scale=cfg_scale)
        samples.append(s.cpu())
        if collect_logits:
            tl.append(model.buffer("text_logits", B).cpu().clone())
            al.append(model.buffer("audio_logits", B).cpu().clone())
        text_tok, audio = s[:, 0:1].long(), s[:, 1:].long()
        if feedback == "audio":
            ct = torch.cat([audio, text_tok], dim=-1).unsqueeze(1)
            cm = torch.cat([torch.ones_like(audio).bool(), torch.zeros(B, 1, device=dev).bool()], dim=1).unsqueeze(1)
        else:
            ct = torch.cat([torch.zeros_like(audio), text_tok], dim=-1).unsqueeze(1)
            cm = torch.cat([torch.zeros_like(audio).bool(), torch.ones(B, 1, device=dev).bool()], dim=1).unsqueeze(1)
        curr_pos = curr_pos + 1
        maxp1 += 1
    out = dict(samples=torch.stack(samples))
    if collect_logits:
        out.update(text_logits=torch.stack(tl), audio_logits=torch.stack(al))
    return out

#!/usr/bin/env python3
"""Generate golden vectors for the audio-token decode loop by RUNNING THE REFERENCE.

Container-only tool: needs /root/reference (read-only) on PYTHONPATH.  It never travels to
the GPU box; only its output ``tests/golden/lm_*.npz`` does.  Recipe follows SURVEY.md
Appendix B: the reference's `llm_models.model_new.Model_stage3` over `llm_models.lit_model.GPT`
is imported with container-only stubs for the un-installed `litgpt` / `torchtune` packages,
where `litgpt.model` is aliased to the reference's own vendored classes
(llm_models/lit_model.py:582-595 LLaMAMLP, :869-890 RMSNorm), so every arithmetic op that
runs is the reference's.

What is recorded (fp32, CPU, torch.manual_seed-free: greedy topk=1):
  * the reference state-dict key/shape list (the checkpoint layout the product must accept),
  * prompts, per-frame sampled ids (F, 9) int32 from `generate_frame`, driven by the same
    feedback protocol as evaluation/tts_task.py:244-282 and evaluation/asr_task.py:658-682,
  * the text / audio logits that fed each argmax (teacher-forcing checks, top-2 margins),
  * tie flags (topk=1 keeps every tied maximum and lets the RNG choose; model_new.py:141-187).

Usage:  PYTHONPATH=/root/reference python tests/golden/make_golden_lm.py
"""
import json
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("NO_TORCH_COMPILE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

from toy_configs import TOY_LM, TOY_MODEL_ARGS
from weights import checksum, seeded_state_dict


def import_reference():
    import llm_models.config as cfg
    litgpt = types.ModuleType("litgpt"); litgpt.__path__ = []
    lc = types.ModuleType("litgpt.config"); lc.Config = cfg.Config
    ls = types.ModuleType("litgpt.scripts"); ls.__path__ = []
    cv = types.ModuleType("litgpt.scripts.convert_hf_checkpoint"); cv.qkv_reassemble = lambda w, c: w
    sys.modules.update({"litgpt": litgpt, "litgpt.config": lc, "litgpt.scripts": ls,
                        "litgpt.scripts.convert_hf_checkpoint": cv,
                        "torchtune": types.ModuleType("torchtune")})
    import llm_models.lit_model as lm
    sys.modules["litgpt.model"] = lm
    litgpt.model = lm
    import llm_models.model_new as mn
    return cfg, lm, mn


def shrink_registry(cfg):
    for name, kw in TOY_LM.items():
        for c in cfg.configs:
            if c["name"] == name:
                c.update(kw)
        if name in cfg.name_to_config:
            cfg.name_to_config[name].update(kw)


def build_model(mn, seed):
    args = mn.ModelArgs(**TOY_MODEL_ARGS)
    model = mn.Model_stage3(args)
    with torch.no_grad():
        model.audio_head.zero_()        # torch.empty in the reference (model_new.py:349)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed)
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model, shapes, sd


class LogitTap:
    """Wraps the reference samplers to record the logits they receive."""

    def __init__(self, mn):
        self.mn = mn
        self.text, self.audio, self.ties = [], [], []
        self._st, self._sa = mn.sample_topk, mn.audio_sample_topk

    def __enter__(self):
        def st(logits, topk, temperature):
            self.text.append(logits.detach().float().clone())
            self.ties.append(int((logits == logits.max(-1, keepdim=True)[0]).sum(-1).max()) > 1)
            return self._st(logits, topk, temperature)

        def sa(logits, topk, temperature, forbid_prefix=0):
            self.audio.append(logits.detach().float().clone())
            l2 = logits.clone()
            if forbid_prefix > 0:
                l2[..., :forbid_prefix] = float("-inf")
            self.ties.append(int((l2 == l2.max(-1, keepdim=True)[0]).sum(-1).max()) > 1)
            return self._sa(logits, topk, temperature, forbid_prefix)

        self.mn.sample_topk, self.mn.audio_sample_topk = st, sa
        return self

    def __exit__(self, *a):
        self.mn.sample_topk, self.mn.audio_sample_topk = self._st, self._sa


@torch.inference_mode()
def run_loop(mn, model, tokens, mask, frames, feedback, forbid_switch=None, reason_card=40, cfg_scale=1.0):
    """tokens (B, L, 9) long, mask (B, L, 9) bool.  feedback in {"audio", "text"}.

    Protocol = evaluation/tts_task.py:244-282 ("audio": feed back the 8 sampled audio ids,
    audio-only mask) / evaluation/asr_task.py:658-682 ("text": zeros for audio, text-only
    mask), without the EOS exits so that the run length is fixed.
    """
    B, L, _ = tokens.shape
    model.setup_caches(B)            # Generator.__init__, evaluation/tts_task.py:64-67
    model.reset_caches()
    pos = torch.arange(0, L).unsqueeze(0).long().repeat(B, 1)
    model.forward_prefix(tokens[:, :-1], labels=tokens[:, 1:, :-1], tokens_mask=mask,
                         loss_mask=mask, input_pos=pos[:, :-1])
    curr_pos = torch.tensor([L - 1], dtype=torch.int64)
    maxp1 = L
    curr_tokens, curr_mask = tokens[:, -1:], mask[:, -1:]
    samples, forbids = [], []
    forbid = 0
    with LogitTap(mn) as tap:
        for f in range(frames):
            if forbid_switch is not None and f == forbid_switch:
                forbid = reason_card
            s = model.generate_frame(curr_tokens, curr_mask, input_pos=curr_pos, input_pos_maxp1=maxp1,
                                     temperature=1.0, topk=1, forbid_prefix=forbid, cfg_scale=cfg_scale)
            samples.append(s.clone())
            forbids.append(forbid)
            text_tok, audio = s[:, 0:1].long(), s[:, 1:].long()
            if feedback == "audio":
                curr_tokens = torch.cat([audio, text_tok], dim=-1).unsqueeze(1)
                curr_mask = torch.cat([torch.ones_like(audio).bool(), torch.zeros(B, 1).bool()], dim=1).unsqueeze(1)
            else:
                curr_tokens = torch.cat([torch.zeros_like(audio), text_tok], dim=-1).unsqueeze(1)
                curr_mask = torch.cat([torch.zeros_like(audio).bool(), torch.ones(B, 1).bool()], dim=1).unsqueeze(1)
            curr_pos = curr_pos + 1
            maxp1 += 1
    Bl = 1 if cfg_scale > 1.0 and B > 1 else B                # with guidance the samplers see one mixed row (model_new.py:618-622)
    text_logits = torch.stack(tap.text)                      # (F, Bl, Vt)
    audio_logits = torch.stack(tap.audio).view(frames, 8, Bl, -1).permute(0, 2, 1, 3)  # (F, Bl, 8, Va)
    return dict(samples=torch.stack(samples).int().numpy(),          # (F, B, 9)
                forbid=np.asarray(forbids, dtype=np.int32),
                text_logits=text_logits.numpy(), audio_logits=audio_logits.contiguous().numpy(),
                ties=np.asarray(tap.ties, dtype=np.bool_))


def text_prompt(g, L, vt):
    t = torch.zeros(L, 9, dtype=torch.long)
    t[:, -1] = torch.randint(0, vt, (L,), generator=g)
    m = torch.zeros(L, 9, dtype=torch.bool); m[:, -1] = True
    return t, m


def audio_prompt(g, n_text, n_audio, vt, va):
    tt, tm = text_prompt(g, n_text, vt)
    at = torch.zeros(n_audio, 9, dtype=torch.long)
    at[:, :-1] = torch.randint(0, va, (n_audio, 8), generator=g)
    am = torch.zeros(n_audio, 9, dtype=torch.bool); am[:, :-1] = True
    # text tail so the last prompt frame is a text step, as in prepare_asr_task + task prompt order
    t2, m2 = text_prompt(g, 2, vt)
    return torch.cat([tt, at, t2]), torch.cat([tm, am, m2])


def main():
    cfg, lm, mn = import_reference()
    shrink_registry(cfg)
    torch.set_num_threads(4)
    seed = 7
    model, shapes, sd = build_model(mn, seed)
    vt, va = 500, TOY_MODEL_ARGS["audio_semantic_vocab_size"] + TOY_MODEL_ARGS["audio_reason_vocab_size"]
    g = torch.Generator().manual_seed(1234)
    out = {}
    meta = dict(seed=seed, keys=[[k, list(s)] for k, s in shapes.items()], checksum=checksum(sd),
                torch=torch.__version__, model_args=TOY_MODEL_ARGS)

    # case 1: TTS-style, B=1, text prompt of 12, 24 frames, forbid_prefix switches at frame 9
    t, m = text_prompt(g, 12, vt)
    r = run_loop(mn, model, t[None], m[None], 24, "audio", forbid_switch=9)
    out.update({f"tts1_{k}": v for k, v in r.items()}); out["tts1_tokens"] = t.numpy(); out["tts1_mask"] = m.numpy()

    # case 2: ASR-style, B=1, prompt with audio frames, 10 text frames
    t, m = audio_prompt(g, 5, 9, vt, va)
    r = run_loop(mn, model, t[None], m[None], 10, "text")
    out.update({f"asr1_{k}": v for k, v in r.items()}); out["asr1_tokens"] = t.numpy(); out["asr1_mask"] = m.numpy()

    # case 3: TTS-style, B=2 (aligned lengths, reference constraint - SURVEY Appendix A.17)
    t0, m0 = text_prompt(g, 9, vt); t1, m1 = text_prompt(g, 9, vt)
    t, m = torch.stack([t0, t1]), torch.stack([m0, m1])
    r = run_loop(mn, model, t, m, 12, "audio", forbid_switch=5)
    out.update({f"tts2_{k}": v for k, v in r.items()}); out["tts2_tokens"] = t.numpy(); out["tts2_mask"] = m.numpy()

    # case 4: classifier-free guidance pair (model_new.py:618-622, 634-637): row 0 conditional, row 1 unconditional
    # (a shorter real prompt left-padded by the caller in the reference; here simply a different prompt of the same
    # length), cfg_scale 1.5, both rows continue from the guided sample
    t0, m0 = text_prompt(g, 10, vt); t1, m1 = text_prompt(g, 10, vt)
    tc, mc = torch.stack([t0, t1]), torch.stack([m0, m1])
    rc = run_loop(mn, model, tc, mc, 10, "audio", forbid_switch=4, cfg_scale=1.5)
    out.update({f"cfg2_{k}": v for k, v in rc.items()}); out["cfg2_tokens"] = tc.numpy(); out["cfg2_mask"] = mc.numpy()

    # determinism: rerun case 3
    r2 = run_loop(mn, model, t, m, 12, "audio", forbid_switch=5)
    assert (r2["samples"] == r["samples"]).all()
    meta["any_ties"] = bool(any(out[k].any() for k in out if k.endswith("_ties")))

    np.savez_compressed(os.path.join(HERE, "lm_toy_fp32.npz"), **out)
    with open(os.path.join(HERE, "lm_toy_fp32.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote lm_toy_fp32.npz", {k: v.shape for k, v in out.items()}, "ties:", meta["any_ties"])


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden vectors for the codec's neural glue, produced by RUNNING THE REFERENCE's own code
(tools/tokenizer/ReasoningCodec_film/models/AudioDiffusion1D.py and modules/transformer.py) at toy sizes:

  think_*   AudioDiffusion1D.encode_reasoning_part (:372-390): down_sampling_layer_whisper, concat, semantic_merge_proj,
            set_masking (:458-476), 2 x TransformerBlock (modules/transformer.py:645-783: power-normalised weight-norm
            Linear layers, q/k LayerNorm, partial rotary, sigmoid-GLU feed-forward, LayerScale), extract_mask_positions
            (:478-486) -> the query tokens that enter reasoning_vq
  fetch_*   AudioDiffusion1D.fetch_codes_batch (:493-551) from the SSL features on: the four strided convs, the three
            cond_fusion linears, reason_adaptor + x2.5 nearest interpolation, time_film (:428-438) with its torch.rand mask,
            the features that enter the three RVQs, their sum through cond_feature_emb
  infer_*   AudioDiffusion1D.inference_codes (:554-624) + BASECFM.solve_euler (:89-129): code split, sum of the three
            look-ups, cond_feature_emb, x2 nearest up-sampling, latent masks, zero_cond_embedding1, in-context latents,
            the guided Euler loop, the final in-context overwrite

What is NOT the reference here (third-party packages absent from the container): the frozen SSL encoders
(their outputs are the seeded inputs), `vector_quantize_pytorch.ResidualVQ` (stand-ins defined below: an identity
quantiser for think_/fetch_, a table look-up for infer_) and the diffusers-based DiT estimator (stand-in: StubEstimator
in codec_model_stub.py, shared with the tests).  Everything between them is the reference's code, executed.

Container-only (needs /root/reference).  Usage: python tests/golden/make_golden_codec_model.py
"""
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

os.environ.setdefault("NO_TORCH_COMPILE", "1")
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch
import torch.nn as nn

from codec_model_stub import (CFG, StubEstimator, fetch_inputs, infer_inputs, module_state_dict, think_inputs)
from weights import seeded_tensor

P = "tools.tokenizer.ReasoningCodec_film."
STUBBED = ["torchaudio", "diffusers", "diffusers.utils", "diffusers.utils.torch_utils", "whisper", "whisper.audio", "peft",
           "vector_quantize_pytorch", "soft_moe_pytorch", "transformers", P + "models.processor", P + "models.transformer_1d_flow",
           P + "modules.our_MERT_BESTRQ.test", P + "models.PretrainedModel", P + "models.modeling_whisper", P + "models.vocos",
           P + "models.model_utils"]


class IdentityVQ(nn.Module):
    """Stand-in for ResidualVQ in think_/fetch_: returns its input as `quantized`, zero codes; records the input."""

    def __init__(self, *a, num_quantizers=1, **kw):
        super().__init__()
        self.nq = num_quantizers
        self.seen = None

    def forward(self, x):
        self.seen = x.detach().clone()
        return x, torch.zeros(x.shape[0], x.shape[1], self.nq, dtype=torch.long), torch.zeros(1, self.nq)


class TableVQ(nn.Module):
    """Stand-in for ResidualVQ.get_output_from_indices in infer_: sum of per-level table rows (tables from seeds)."""

    def __init__(self, tables):
        super().__init__()
        self.tables = tables                               # [L, C, D]

    def get_output_from_indices(self, idx):                # (B, T, L) -> (B, T, D)
        return sum(self.tables[l][idx[..., l]] for l in range(idx.shape[-1]))


def main():
    for name in STUBBED:
        sys.modules[name] = MagicMock()
    ad = importlib.import_module(P + "models.AudioDiffusion1D")
    ad.ResidualVQ = IdentityVQ
    torch.set_num_threads(4)
    out, meta = {}, {}
    c = CFG

    # ---- AudioThinking encoder ------------------------------------------------------------------------------------
    think = ad.AudioThinking(dim=c["D"], interval=5, encoder_depth=c["depth"], whisper_fea_dim=c["Cw"], llm_path=None, prompt_path=None)
    shapes = {k: tuple(v.shape) for k, v in think.state_dict().items()}
    meta["think_keys"] = [[k, list(s)] for k, s in shapes.items()]
    think.load_state_dict(module_state_dict(shapes, 301))
    think.eval()
    fake = types.SimpleNamespace(audio_thinking=think)
    fake.set_masking = types.MethodType(ad.AudioDiffusion1D.set_masking, fake)
    fake.extract_mask_positions = types.MethodType(ad.AudioDiffusion1D.extract_mask_positions, fake)
    whisper, bestrq_sem = think_inputs()
    with torch.no_grad():
        q, _, _ = ad.AudioDiffusion1D.encode_reasoning_part(fake, whisper.clone(), bestrq_sem.clone())
    out["think_query"] = q.numpy()

    # ---- fetch_codes_batch from the SSL features on ----------------------------------------------------------------
    f = fetch_inputs()
    mods = dict(
        d_conv_whisper=nn.Conv1d(c["Cw"], c["Cw"], 4, stride=4), d_conv_wavlm=nn.Conv1d(c["Cl"], c["Cl"], 4, stride=4),
        d_conv_embedding_semantic=nn.Conv1d(c["Cb"], c["Cb"], 2, stride=2), d_conv_embedding_acoustic=nn.Conv1d(c["Cb"], c["Cb"], 2, stride=2),
        cond_fusion_layer_semantic=nn.Linear(c["Cb"], c["D"]), cond_fusion_layer_acoustic=nn.Linear(c["Cb"] + c["Cw"], c["D"]),
        cond_fusion_layer_phone=nn.Linear(c["Cl"], c["D"]), time_film_phone=nn.Linear(c["D"], 2 * c["D"]),
        time_film_semantic=nn.Linear(c["D"], 2 * c["D"]), time_film_acoustic=nn.Linear(c["D"], 2 * c["D"]),
        reason_adaptor=nn.Linear(c["D"], c["D"]), cond_feature_emb=nn.Linear(c["D"], c["D"]))
    holder = nn.ModuleDict(mods)
    shapes = {k: tuple(v.shape) for k, v in holder.state_dict().items()}
    meta["fetch_keys"] = [[k, list(s)] for k, s in shapes.items()]
    holder.load_state_dict(module_state_dict(shapes, 302))
    fake = types.SimpleNamespace(audio_thinking=think, gamma=0.1, **{k: holder[k] for k in mods})
    fake.vq_pronunciation_semantic, fake.vq_structure_semantic, fake.vq_acoustic = IdentityVQ(), IdentityVQ(), IdentityVQ(num_quantizers=6)
    think.reasoning_vq = IdentityVQ(num_quantizers=8)
    fake.pretrained_model = types.SimpleNamespace(eval=lambda: None, extract_continous_embeds_multiple=lambda x: (f["bestrq_acoustic"].clone(), f["bestrq_semantic"].clone()))
    fake.wavlm_encoder = fake.whisper_encoder = types.SimpleNamespace(eval=lambda: None)
    fake.get_whisper_feature = lambda mels, n, ls: f["whisper"].clone()
    fake.get_wavlm_feature = lambda wav, ls: f["wavlm"].clone()
    for name in ("set_masking", "extract_mask_positions", "time_film", "encode_reasoning_part"):
        setattr(fake, name, types.MethodType(getattr(ad.AudioDiffusion1D, name), fake))
    B = f["whisper"].shape[0]
    torch.manual_seed(c["film_seed"])
    with torch.no_grad():
        rc, mc, mf = ad.AudioDiffusion1D.fetch_codes_batch(fake, torch.zeros(B, 1, 8), None)
    torch.manual_seed(c["film_seed"])                     # replay: the only RNG consumers are the three time_film draws, in order
    masks = [(torch.rand(B, 1, 1) < 0.2) for _ in range(3)]
    out["fetch_film_masks"] = torch.stack(masks).view(3, B).numpy()
    out["fetch_reason_query"] = think.reasoning_vq.seen.numpy()
    out["fetch_pre_vq_phone"] = fake.vq_pronunciation_semantic.seen.numpy()
    out["fetch_pre_vq_semantic"] = fake.vq_structure_semantic.seen.numpy()
    out["fetch_pre_vq_acoustic"] = fake.vq_acoustic.seen.numpy()
    out["fetch_merge_features"] = mf[0].numpy()
    assert mc[0].shape[-1] == 8 and rc[0].shape[-1] == 8
    print("film masks", out["fetch_film_masks"].astype(int).tolist())

    # ---- inference_codes + solve_euler -----------------------------------------------------------------------------
    i = infer_inputs()
    cfe = nn.Linear(c["D"], c["D"])
    shapes = {k: tuple(v.shape) for k, v in cfe.state_dict().items()}
    cfe.load_state_dict(module_state_dict(shapes, 303))
    est = StubEstimator()
    fake = types.SimpleNamespace(device=torch.device("cpu"), dtype=torch.float32, max_t_len=30 * 50, sq_codec_latent=c["latent"],
                                 cond_feature_emb=cfe, zero_cond_embedding1=i["zero_cond"],
                                 vq_pronunciation_semantic=TableVQ(i["tab_phone"]), vq_structure_semantic=TableVQ(i["tab_sem"]),
                                 vq_acoustic=TableVQ(i["tab_ac"]), cfm_wrapper=ad.BASECFM(est))
    fake.prepare_latents = lambda bsz, nf, dtype, device: i["noise"].clone()
    ad.tqdm = lambda it: it
    with torch.no_grad():
        for tag, kw in (("infer_first", dict(true_latents=i["first_latent"].clone(), incontext_length=0)),
                        ("infer_other", dict(true_latents=i["true_latent"].clone(), incontext_length=i["incontext"]))):
            lat = ad.AudioDiffusion1D.inference_codes(fake, [i["codes"]], None, kw["true_latents"], i["latent_length"], kw["incontext_length"],
                                                      additional_feats=[], guidance_scale=1.5, num_steps=c["steps"], disable_progress=True,
                                                      scenario="other_seg")
            out[tag] = lat.numpy()
    meta["cfe_keys"] = [[k, list(s)] for k, s in shapes.items()]
    np.savez_compressed(os.path.join(HERE, "codec_model_toy.npz"), **out)
    import json
    json.dump(meta, open(os.path.join(HERE, "codec_model_toy.json"), "w"))
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

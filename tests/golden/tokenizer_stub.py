"""Deterministic stand-ins shared by tests/golden/make_golden_tokenizer.py (which runs the REFERENCE's
ReasoningTokenizer.token2audio_no_reason / audio2token on them) and by the tests (which run the oracle and the product on
the same stand-ins and never touch the reference).  They replace the neural pieces around the host logic under test —
`model.inference_codes`, `SQCodec.decode`, `model.fetch_codes_batch`, the Whisper front end — with cheap closed-form
functions that (i) record every argument the host logic passes and (ii) make every argument matter in the result, so a
wrong window, a wrong in-context slice or a different random draw changes the output.  Test data plumbing only."""
import math

import numpy as np
import torch

LATENT = 136
T_CASES = (1, 249, 250, 251, 437, 1000)          # one window short, exact, one over, two windows, six
SEED = 1234
WAVE_STRIDE = 487                                # the fixture keeps every 487th sample + float64 checksums
CLIP_CASES = ((24000, 6), (240000, 6), (720000, 2), (1128000, 2), (2300000, 2))   # (samples at 24 kHz, batch_size)


def make_codes(T):
    """(1, 8, T) int64, every entry distinct: the value encodes (level, time)."""
    t = torch.arange(T, dtype=torch.int64)
    return torch.stack([(t * 8 + l) % 8192 for l in range(8)]).unsqueeze(0)


class StubModel:
    """inference_codes: lat = sin(codes) pattern + 0.05 * true_latents, in-context frames copied through (as the real one
    does at the end of the Euler loop, AudioDiffusion1D.py:620-623).  Every call is recorded."""

    sq_codec_latent = LATENT
    vq_pronunciation_semantic = vq_structure_semantic = vq_acoustic = None
    cfm_wrapper = object()              # the product checks that a flow-matching stage is present before decoding

    def __init__(self):
        self.calls = []

    def inference_codes(self, codes_input, spk_embeds, true_latents, latent_length, incontext_length, additional_feats=None,
                        guidance_scale=1.5, num_steps=20, disable_progress=True, scenario="other_seg", **kw):
        codes = codes_input[-1]
        assert spk_embeds is None and additional_feats == [] and guidance_scale == 1.5 and scenario == "other_seg"
        self.calls.append(dict(codes=codes.detach().cpu().clone(), incontext=int(incontext_length), latent_length=int(latent_length),
                               true=true_latents.detach().cpu().float().clone(), steps=int(num_steps)))
        B = codes.shape[0]
        t = torch.arange(latent_length, device=codes.device)
        f = torch.arange(LATENT, device=codes.device)
        c = codes[:, f % 8][:, :, t // 2].permute(0, 2, 1).float()          # (B, L, 136): level f % 8, code frame t // 2
        lat = torch.sin(0.01 * c + 0.1 * f.float()[None, None, :]) + 0.05 * true_latents.float()
        if incontext_length > 0:
            lat[:, :incontext_length] = true_latents[:, :incontext_length].float()
        return lat


class StubCodec:
    """decode (B, 136, T) -> (B, 1, T * 960): sample j of frame t = 0.5 * x[j % 136, t] + 0.001 * cos(j)."""

    def decode(self, x):
        B, C, T = x.shape
        j = torch.arange(960, device=x.device)
        w = 0.5 * x[:, j % C, :].permute(0, 2, 1) + 0.001 * torch.cos(j.float())[None, None, :]     # (B, T, 960)
        return w.reshape(B, 1, T * 960)


def wave_digest(w):
    w = w.detach().cpu().double().reshape(-1).numpy()
    return np.array([w.sum(), np.abs(w).sum(), (w * w).sum(), float(len(w))])


def make_clip(n, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(1, n, generator=g) - 0.5)


def codes_from_segment(seg, masks):
    """seg (b, 1, S) -> (reason (b, 150, 8), rec (b, 375, 8)) int64; masks = three (b,) bool tensors (the time_film draws)."""
    b = seg.shape[0]
    m = (masks[0].long() + 2 * masks[1].long() + 4 * masks[2].long()).view(b, 1, 1)
    x = seg[:, 0, :]
    lv = torch.arange(8, device=seg.device)

    def take(n, stride):
        idx = (torch.arange(n, device=seg.device)[:, None] * stride + lv[None, :] * 7)        # (n, 8)
        return (x[:, idx].abs() * 100000.0).floor().long() % 8192 + m
    return take(150, 4800), take(375, 1920)


class StubEncoderModel(StubModel):
    """fetch_codes_batch as the reference calls it (reason_tokenizer.py:117-118): draws the three time_film masks itself
    (AudioDiffusion1D.py:435: torch.rand(B, 1, 1) per call, phone -> semantic -> acoustic), unless the caller hands them over
    (`film_masks`, the product's waste-removal path)."""

    def __init__(self):
        super().__init__()
        self.fetch_calls = []

    def fetch_codes_batch(self, audio, mels, additional_feats=None, return_reasoning_text=False, film_masks=None):
        b = audio.shape[0]
        if film_masks is None:
            film_masks = [(torch.rand(b, 1, 1, device=audio.device) < 0.2).view(b) for _ in range(3)]
        self.fetch_calls.append(dict(rows=b, masks=[m.cpu().clone() for m in film_masks]))
        reason, rec = codes_from_segment(audio, film_masks)
        return [reason], [rec], None

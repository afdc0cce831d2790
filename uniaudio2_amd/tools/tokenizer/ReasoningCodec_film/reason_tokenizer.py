"""ReasoningCodec tokenizer, host side: the tensor contract, the windowing and the in-scope part of decode.

Mirror of the reference's tools/tokenizer/ReasoningCodec_film/reason_tokenizer.py `ReasoningTokenizer`
for what is on the hot path:
  * `detokenize_no_reason(rec_codec (8,T), ...) -> wave (1, N) float32 CPU` (:399-404) through
    `token2audio_no_reason` (:229-306): tile / pad the codes to 20-s windows (250 codes, hop 186,
    overlap 64), per window: RVQ lookup of the three code groups (AudioDiffusion1D.py:570-583, in
    scope) -> latent generator (flow-matching DiT + Euler ODE: OUT OF SCOPE this round, SURVEY §8f #1,
    injected through `latent_fn`) -> `SQCodec.decode` (ScalarModel, in scope) -> linear cross-fade of
    the 25 % overlap in float64 on the host, crop to T / 12.5 * 24000 samples.
  * `tokenize(tensor)` passes tensors through (:387-388); tokenising a wav path needs the frozen
    Whisper / WavLM / BEST-RQ encoders (out of scope) and raises.
"""
import math

import numpy as np
import torch


def window_plan(rec_codes_len, duration=20, rec_frame_rate=12.5, sample_rate=24000, sq_codec_hz=25):
    """Index arithmetic of token2audio_no_reason (:239-262, 267, 293-297) — pure function."""
    min_samples = int(duration * rec_frame_rate)
    hop_samples = min_samples // 4 * 3
    ovlp_samples = min_samples - hop_samples
    target_len = int(rec_codes_len / 12.5 * sample_rate)
    n = rec_codes_len
    if n < min_samples:
        n = min_samples                                             # self-concatenated then cut (:251-254)
    if (n - ovlp_samples) % hop_samples > 0:
        n = math.ceil((n - ovlp_samples) / float(hop_samples)) * hop_samples + ovlp_samples     # (:256-260)
    starts = list(range(0, n - hop_samples, hop_samples))
    wav_min = int(duration * sample_rate)
    wav_hop = wav_min // 4 * 3
    return dict(tiled_len=n, min_codes=min_samples, hop_codes=hop_samples, ovlp_codes=ovlp_samples, starts=starts,
                ovlp_frames=ovlp_samples // 2, latent_length=int(duration * sq_codec_hz), target_len=target_len,
                wav_window=wav_min, wav_hop=wav_hop, wav_ovlp=wav_min - wav_hop)


def tile_codes(rec_codec, tiled_len):
    """Self-concatenate along time until `tiled_len`, then cut (:251-260)."""
    while rec_codec.shape[-1] < tiled_len:
        rec_codec = torch.cat([rec_codec, rec_codec], -1)
    return rec_codec[..., :tiled_len]


def crossfade_concat(segments, wav_window, wav_ovlp, target_len):
    """Linear cross-fade of consecutive windows, float64 ramp on the host, as :293-305."""
    output = None
    for cur in segments:
        cur = cur[:, 0:wav_window].detach().cpu()
        if output is None:
            output = cur
        else:
            ov = torch.from_numpy(np.linspace(0, 1, wav_ovlp)[None, :])
            ov = torch.cat([ov, 1 - ov], -1)
            output[:, -wav_ovlp:] = output[:, -wav_ovlp:] * ov[:, -wav_ovlp:] + cur[:, 0:wav_ovlp] * ov[:, 0:wav_ovlp]
            output = torch.cat([output, cur[:, wav_ovlp:]], -1)
    return output[:, 0:target_len]


class ReasoningTokenizer:
    """`sq_codec`: a prepared ScalarModel; `vq_*`: ResidualVQ mirrors of the three code groups
    [phone (1 level), semantic (1), acoustic (6)] (AudioDiffusion1D.py:256-264); `latent_fn(cond, steps)`:
    (B, 500, 768) conditioning -> (B, 500, 136) SQ-Codec latent — the DiT stage this build does not contain."""

    def __init__(self, sq_codec=None, vq_phone=None, vq_semantic=None, vq_acoustic=None, latent_fn=None, device="cuda",
                 train_config=None, model_path=None, music_ssl_folder=None):
        self.device = torch.device(device)
        self.sample_rate = 24000
        self.rec_frame_rate, self.reason_frame_rate, self.sq_codec_hz = 12.5, 5, 25        # :31-33
        self.SQCodec, self.latent_fn = sq_codec, latent_fn
        self.vq = (vq_phone, vq_semantic, vq_acoustic)
        if train_config is not None or model_path is not None:
            raise NotImplementedError("loading the released codec checkpoint needs the un-vendored DiT / SSL stack "
                                      "(diffusers, fairseq, whisper): out of scope this round, SURVEY.md §8f")

    @property
    def is_discrete(self):
        return True

    def tokenize(self, wav, return_reasoning_text=False, task_name="asr", min_duration=30):
        if isinstance(wav, torch.Tensor):
            return wav                                              # :387-388
        raise NotImplementedError("tokenising audio needs the frozen Whisper / WavLM / BEST-RQ encoders "
                                  "(out of scope, SURVEY.md §2.1 row 15); pass --reason_pt/--semantic_pt instead")

    def codes_to_condition(self, codes):
        """codes (B, 8, T) -> (B, T, 768): sum of the three RVQ lookups (AudioDiffusion1D.py:570-583)."""
        groups = (codes[:, 0:1], codes[:, 1:2], codes[:, 2:])
        out = None
        for vq, c in zip(self.vq, groups):
            q = vq.get_output_from_indices(c.transpose(1, 2).contiguous())
            out = q if out is None else out + q
        return out

    @torch.no_grad()
    def token2audio_no_reason(self, rec_codec, return_reasoning_text=False, duration=20, guidance_scale=1.5, num_steps=20,
                              disable_progress=False):
        if self.latent_fn is None:
            raise NotImplementedError("the flow-matching DiT that turns code conditioning into SQ-Codec latents is out of "
                                      "scope this round (SURVEY.md §8f #1); supply latent_fn for synthetic runs")
        rec_codec = rec_codec.to(self.device)
        plan = window_plan(rec_codec.shape[-1], duration, self.rec_frame_rate, self.sample_rate, self.sq_codec_hz)
        rec_codec = tile_codes(rec_codec, plan["tiled_len"])
        segments = []
        for s0 in plan["starts"]:
            cond = self.codes_to_condition(rec_codec[:, :, s0:s0 + plan["min_codes"]])
            latent = self.latent_fn(cond, num_steps).float()                       # (B, latent_length, 136)
            wav = self.SQCodec.decode(latent.transpose(1, 2).contiguous()).squeeze(0)   # (1, N) per :295
            segments.append(wav)
        return crossfade_concat(segments, plan["wav_window"], plan["wav_ovlp"], plan["target_len"])

    def detokenize_no_reason(self, rec_codec, return_reasoning_text=False, min_duration=30, steps=50, guidance_scale=1.5,
                             disable_progress=False):
        """rec_codec (8, T) -> wave (1, N) float32 on the CPU (:399-404)."""
        return self.token2audio_no_reason(rec_codec.unsqueeze(0), return_reasoning_text, guidance_scale=guidance_scale,
                                          num_steps=steps, disable_progress=disable_progress)

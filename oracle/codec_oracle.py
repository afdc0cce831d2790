"""CPU ORACLE (test infrastructure, NOT product code) for the live codec's waveform auto-encoder.

A functional plain-PyTorch (fp32, CPU) restatement of the reference's
tools/tokenizer/ReasoningCodec_film/models/scalar24k.py `ScalarModel.encode` / `.decode` (:392-407) driven by a
reference state dict (same keys: `encoder.N...weight_g / weight_v / bias`, `...activation.weight`), plus the codec's
windowing arithmetic of tools/tokenizer/ReasoningCodec_film/reason_tokenizer.py:229-306 (index lists only).
Every function cites the reference lines it follows (paths relative to /root/reference).

Pinned against outputs of the reference itself: tests/golden/codec_toy.npz (made by tests/golden/make_golden_codec.py,
which imports and runs the reference's ScalarModel); checked by tests/test_oracle_codec.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.  The product
(uniaudio2_amd/) never does and has no CPU fallback.
"""
from typing import Dict

import torch
import torch.nn.functional as F


def _w(sd, prefix):
    """Effective filter of a conv: weight-norm folded exactly as torch.nn.utils.weight_norm does at forward time
    (w = g * v / ||v||, norm over all dims but 0), or the plain weight (PreProcessor / PostProcessor convs,
    scalar24k.py:119,132 have no weight_norm)."""
    if prefix + "weight_g" in sd:
        return torch._weight_norm(sd[prefix + "weight_v"].float(), sd[prefix + "weight_g"].float(), 0)
    return sd[prefix + "weight"].float()


def conv1d(sd, prefix, x, causal, dilation=1, stride=1):
    """scalar24k.py:36-74 Conv1d: causal = left zero-pad dilation*(k-1) (:50-52,70-72), else symmetric
    get_padding (:17-18,54)."""
    w = _w(sd, prefix)
    k = w.shape[-1]
    b = sd.get(prefix + "bias")
    if causal:
        x = F.pad(x, (dilation * (k - 1), 0))
        return F.conv1d(x, w, b, stride=stride, dilation=dilation)
    return F.conv1d(x, w, b, stride=stride, dilation=dilation, padding=int((k * dilation - dilation) / 2))


def conv_transpose1d(sd, prefix, x, causal, stride):
    """scalar24k.py:76-112 ConvTranspose1d: causal needs k == 2*stride and trims the last `stride` samples (:108-111);
    otherwise padding (k - stride) // 2 (:90-91)."""
    w = _w(sd, prefix)
    k = w.shape[-1]
    b = sd.get(prefix + "bias")
    if causal:
        assert k == 2 * stride
        return F.conv_transpose1d(x, w, b, stride=stride)[:, :, :-stride]
    return F.conv_transpose1d(x, w, b, stride=stride, padding=(k - stride) // 2)


def prelu(sd, key, x):
    return F.prelu(x, sd[key].float())


def residual_unit(sd, p, x, dilation, causal):
    """scalar24k.py:143-151."""
    y = prelu(sd, p + "activation1.weight", conv1d(sd, p + "conv1.", x, causal, dilation=dilation))
    y = prelu(sd, p + "activation2.weight", conv1d(sd, p + "conv2.", y, causal))
    return y + x


class ScalarOracle:
    """cfg = the ScalarModel constructor arguments (scalar24k.py:306-309)."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: dict):
        self.sd = {k: v.detach().float().cpu() for k, v in sd.items()}
        self.cfg = cfg
        self.causal = cfg["causal"]
        self.ns = cfg["num_samples"]
        self.down, self.up = list(cfg["downsample_factors"]), list(cfg["upsample_factors"])
        # scalar24k.py:206 `activation=nn.PReLU()` is a default argument, i.e. ONE module shared by every
        # DownsampleLayer: the state dict lists it once per layer, load_state_dict copies them in order into the same
        # tensor, so the value in effect is the last key's (real checkpoints hold identical copies).
        shared = [k for k in self.sd if k.endswith("down_conv.activation.weight")]
        self.down_act = shared[-1] if shared else None

    @torch.no_grad()
    def encode(self, x):
        """scalar24k.py:392-401: conv -> [PreProcessor] -> ResEncoderBlocks -> conv -> tanh; returns the un-rounded
        latent (`emb`, :398,401)."""
        sd, c = self.sd, self.causal
        x = x.float()
        i = 0
        x = conv1d(sd, f"encoder.{i}.", x, c); i += 1
        if self.ns > 1:                                                     # PreProcessor :114-123
            x = prelu(sd, f"encoder.{i}.activation.weight", conv1d(sd, f"encoder.{i}.conv.", x, c))
            x = F.avg_pool1d(x, self.ns); i += 1
        for f in self.down:                                                 # ResEncoderBlock :154-173
            for j, d in enumerate((1, 3, 5, 7, 9)):
                x = residual_unit(sd, f"encoder.{i}.convs.{j}.", x, d, c)
            p = f"encoder.{i}.down_conv."                                   # DownsampleLayer :199-229
            x = prelu(sd, self.down_act, conv1d(sd, p + "layer.", x, c, stride=f)); i += 1
        return torch.tanh(conv1d(sd, f"encoder.{i}.", x, c))

    @torch.no_grad()
    def decode(self, x):
        """scalar24k.py:403-407: round(9x)/9 (:285-290) -> look-ahead conv (non-causal, :354-358) -> ResDecoderBlocks
        (:176-196) -> [PostProcessor :126-140: repeat each step num_samples times, conv, PReLU] -> conv."""
        sd, c = self.sd, self.causal
        x = torch.round(9 * x.float()) / 9
        i = 0
        x = conv1d(sd, f"decoder.{i}.", x, False); i += 1
        for f in self.up:
            x = conv_transpose1d(sd, f"decoder.{i}.up_conv.layer.", x, c, f)   # UpsampleLayer, activation=None (:180)
            for j, d in enumerate((1, 3, 5, 7, 9)):
                x = residual_unit(sd, f"decoder.{i}.convs.{j}.", x, d, c)
            i += 1
        if self.ns > 1:
            x = x.repeat_interleave(self.ns, dim=-1)                        # == transpose/repeat/view/transpose :135-138
            x = prelu(sd, f"decoder.{i}.activation.weight", conv1d(sd, f"decoder.{i}.conv.", x, c)); i += 1
        return conv1d(sd, f"decoder.{i}.", x, c)


# ---- windowing arithmetic of token2audio_no_reason (reason_tokenizer.py:229-306) ------------------------------------

def window_indices(rec_codes_len: int, duration: int = 20, rec_frame_rate: float = 12.5, sample_rate: int = 24000):
    """Index lists only, no NN: which code indices of the ORIGINAL (8, T) tensor each window reads, and the waveform
    bookkeeping.  Follows reason_tokenizer.py line by line: min/hop/overlap in codes (:239-241), target length (:250),
    self-concatenation up to one window (:251-254) and up to a whole number of hops (:256-260), the window loop
    `range(0, len - hop, hop)` (:267), the waveform-domain window / hop / overlap (:289-291)."""
    import math
    min_samples = int(duration * rec_frame_rate)
    hop_samples = min_samples // 4 * 3
    ovlp_samples = min_samples - hop_samples
    idx = list(range(rec_codes_len))
    target_len = int(rec_codes_len / 12.5 * sample_rate)
    if len(idx) < min_samples:
        while len(idx) < min_samples:
            idx = idx + idx
        idx = idx[:min_samples]
    if (len(idx) - ovlp_samples) % hop_samples > 0:
        len_codes = math.ceil((len(idx) - ovlp_samples) / float(hop_samples)) * hop_samples + ovlp_samples
        while len(idx) < len_codes:
            idx = idx + idx
        idx = idx[:len_codes]
    windows = [idx[s:s + min_samples] for s in range(0, len(idx) - hop_samples, hop_samples)]
    wav_min = int(duration * sample_rate)
    wav_hop = wav_min // 4 * 3
    return dict(windows=windows, ovlp_frames=(ovlp_samples // 2), target_len=target_len, wav_window=wav_min,
                wav_ovlp=wav_min - wav_hop)


def crossfade(segments, wav_window: int, wav_ovlp: int, target_len: int):
    """reason_tokenizer.py:292-305: float64 linear ramp over the overlap, running output kept on the CPU."""
    import numpy as np
    output = None
    for cur in segments:
        cur = cur[:, 0:wav_window].detach().cpu()
        if output is None:
            output = cur
        else:
            ov_win = torch.from_numpy(np.linspace(0, 1, wav_ovlp)[None, :])
            ov_win = torch.cat([ov_win, 1 - ov_win], -1)
            output[:, -wav_ovlp:] = output[:, -wav_ovlp:] * ov_win[:, -wav_ovlp:] + cur[:, 0:wav_ovlp] * ov_win[:, 0:wav_ovlp]
            output = torch.cat([output, cur[:, wav_ovlp:]], -1)
    return output[:, 0:target_len]


def token2audio_no_reason(rec_codec, inference_codes, decode, duration: int = 20, num_steps: int = 20, latent_dim: int = 136):
    """reason_tokenizer.py:229-306 around two callables (`inference_codes` = AudioDiffusion1D.inference_codes, `decode` =
    SQCodec.decode): rec_codec (B, 8, T) -> waveform (B, N) fp32 on the CPU.  PINNED on tests/golden/tokenizer_host.npz, which
    the reference's own method produced on the same stand-ins (windows, in-context chain, randn draw order, cross-fade, crop).
    Draws: `first_latent` (B, 500, 136) before anything else (:235), then per later window the (B, 500 - 32, 136) tail of
    `true_latent` (:282) — both from the CPU generator."""
    B, _, T = rec_codec.shape
    plan = window_indices(T, duration)
    latent_length = int(duration * 25)
    first_latent = torch.randn(B, latent_length, latent_dim)
    latents = []
    for i, idx in enumerate(plan["windows"]):
        window = rec_codec[:, :, idx]
        if i == 0:
            lat = inference_codes([window], None, first_latent, latent_length, 0, additional_feats=[], guidance_scale=1.5,
                                  num_steps=num_steps, disable_progress=True, scenario="other_seg")
        else:
            true = latents[-1][:, -plan["ovlp_frames"]:, :]
            pad = torch.randn(true.shape[0], latent_length - true.shape[1], true.shape[-1])
            lat = inference_codes([window], None, torch.cat([true, pad], 1), latent_length, true.shape[1], additional_feats=[],
                                  guidance_scale=1.5, num_steps=num_steps, disable_progress=True, scenario="other_seg")
        latents.append(lat)
    latents = [l.float() for l in latents]
    segments = [decode(l.transpose(1, 2)).squeeze(0) for l in latents]
    return crossfade(segments, plan["wav_window"], plan["wav_ovlp"], plan["target_len"])


def audio2token(orig_samples, fetch_codes_batch, mel_fn=None, sample_rate: int = 24000, min_duration: int = 30, batch_size: int = 6,
                rec_frame_rate: float = 12.5, reason_frame_rate: float = 5):
    """reason_tokenizer.py:86-129 around `fetch_codes_batch` (AudioDiffusion1D.fetch_codes_batch): (1, N) samples at 24 kHz ->
    (reason (1, 8, T_r), rec (1, 8, T_s)).  PINNED on tests/golden/tokenizer_host.npz.  The clip is self-concatenated until one
    segment (30 s + 240 samples) fits (:101-102), doubled once more (:105), cut into int_max_len segments (:104-106), encoded
    batch_size segments at a time (:110-122) — EVERY segment, as the reference does — and the token grid cropped to
    int(dur * 12.5) + 1 / int(dur * 5) + 1 (:125-128)."""
    audios = orig_samples if orig_samples.ndim == 2 else orig_samples.squeeze(0)
    orig_length = audios.shape[-1]
    min_samples = int(min_duration * sample_rate)
    output_len = int(orig_length / float(sample_rate) * rec_frame_rate) + 1
    output_len_reason = int(orig_length / float(sample_rate) * reason_frame_rate) + 1
    while audios.shape[-1] < min_samples + 240:
        audios = torch.cat([audios, audios], -1)
    int_max_len = audios.shape[-1] // min_samples + 1
    audios = torch.cat([audios, audios], -1)
    audios = audios[:, :int(int_max_len * (min_samples + 240))]
    audio_input = audios.reshape(1, -1, min_samples + 240).permute(1, 0, 2).reshape(-1, 1, min_samples + 240)
    reason_list, rec_list = [], []
    for i in range(0, audio_input.shape[0], batch_size):
        chunk = audio_input[i:i + batch_size]
        mels = mel_fn(chunk[:, 0, :]) if mel_fn is not None else None
        reasoning_codes, rec_codes, _ = fetch_codes_batch(chunk, mels, additional_feats=[], return_reasoning_text=False)
        reason_list.append(torch.cat(reasoning_codes, 1))
        rec_list.append(torch.cat(rec_codes, 1))
    reason = torch.cat(reason_list, 0).reshape(-1, 8).unsqueeze(0)
    rec = torch.cat(rec_list, 0).reshape(-1, 8).unsqueeze(0)
    return reason[:, :output_len_reason, :].transpose(1, 2), rec[:, :output_len, :].transpose(1, 2)


# ---- torchaudio.functional.resample (reason_tokenizer.py:383-385) — PARITY UNPINNED: torchaudio is not installed ---------

def resample(waveform, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Restatement of torchaudio's published sinc_interp_hann algorithm (`_get_sinc_resample_kernel` +
    `_apply_sinc_resample_kernel`) with its defaults, as the reference calls it: float64 index grid, clamp to the filter
    width, Hann window cos^2, sinc, scale; then a strided conv1d of the (width, width + orig) padded signal.
    waveform (C, L) fp32 -> (C, ceil(new * L / orig))."""
    import math
    if int(orig_freq) == int(new_freq):
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t = t * base
    t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = (kernels * window * (base / o)).to(torch.float32)
    C, L = waveform.shape
    x = F.pad(waveform.float(), (width, width + o))
    y = F.conv1d(x[:, None], kernels, stride=o)                 # (C, n, frames)
    y = y.transpose(1, 2).reshape(C, -1)
    return y[..., :math.ceil(n * L / o)]

"""ReasoningCodec tokenizer, host side: audio <-> the (8, T) token tensors of the `*_reason.pt` / `*_semantic.pt` contract.

Mirror of the reference's tools/tokenizer/ReasoningCodec_film/reason_tokenizer.py `ReasoningTokenizer`
(SURVEY.md §8a rows a15, a17-a19, §8f #1, #3):
  * `detokenize_no_reason(rec_codec (8,T), ...) -> wave (1, N) float32 CPU` (:399-404) through
    `token2audio_no_reason` (:229-306): tile / pad the codes to 20-s windows (250 codes, hop 186,
    overlap 64); per window `model.inference_codes` (RVQ look-ups -> cond_feature_emb -> x2 nearest ->
    flow-matching DiT + guided Euler ODE, the previous window's last 32 latent frames as in-context
    frames) -> `SQCodec.decode` (ScalarModel) -> linear cross-fade of the 25 % overlap in float64 on the
    host, crop to T / 12.5 * 24000 samples.
  * `tokenize(wav path) -> (reason (8, T_r), rec (8, T_s))` (:377-387): load, down-mix, resample to 24 kHz
    (torchaudio's windowed-sinc algorithm restated as one MFMA GEMM, `resample`), then `audio2token`
    (:86-129): tile the clip to 30-s segments (+240 samples), `model.fetch_codes_batch` per batch of 6
    segments, crop to int(dur * 12.5) + 1 / int(dur * 5) + 1 tokens.  The frozen SSL encoders inside
    fetch_codes_batch are an injected callable (models/AudioDiffusion1D.py); `tokenize(tensor)` passes
    tensors through (:387-388).
"""
import math

import numpy as np
import torch

from .... import ops


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


def resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Filter bank of `torchaudio.functional.resample` (sinc_interp_hann, the defaults reason_tokenizer.py:383-385 uses):
    [new/gcd, 2*width + orig/gcd] fp32 + width.  torchaudio is not installed here (parity with the package UNPINNED,
    SURVEY.md §8d config 1); this follows its published algorithm: float64 index grid, clamp to the filter width,
    Hann window cos^2, sinc, scale by base_freq / orig."""
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base / o
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t)
    kernels = kernels * window * scale
    return kernels.to(torch.float32).view(n, -1), width, o, n


def resample(wav, orig_freq, new_freq):
    """wav (C, L) fp32 on the device -> (C, ceil(new * L / orig)).  The polyphase FIR is one exact-fp32 GEMM: frames of
    2*width + orig samples every `orig` samples (a strided view of the padded signal) x the [new, taps] filter bank."""
    if int(orig_freq) == int(new_freq):
        return wav
    kern, width, o, n = resample_kernel(orig_freq, new_freq)
    C_, L = wav.shape
    taps = kern.shape[1]
    kp = (taps + 3) // 4 * 4                                          # ua2_linear wants K % 4 == 0: zero taps
    w = torch.zeros(n, kp)
    w[:, :taps] = kern
    wp = ops.pack_linear(w.to(wav.device), torch.float32)
    outs = []
    for c in range(C_):
        x = torch.nn.functional.pad(wav[c].float(), (width, width + o))
        frames = x.unfold(0, taps, o)                                  # [n_frames, taps] strided view (data movement only)
        fr = torch.zeros(frames.shape[0], kp, dtype=torch.float32, device=wav.device)
        fr[:, :taps] = frames
        y = torch.empty(fr.shape[0], n, dtype=torch.float32, device=wav.device)
        ws = ops.linear_workspace(torch.float32, fr.shape[0], kp, wav.device)
        ops.linear(dtype=torch.float32, M=fr.shape[0], N=n, K=kp, w0=wp, x=fr, y=y, workspace=ws)
        outs.append(y.reshape(-1)[:math.ceil(n * L / o)])
    return torch.stack(outs)


def load_wav(path):
    """(channels, samples) fp32 in [-1, 1] + sample rate, as torchaudio.load normalises (int PCM / 2^(bits-1))."""
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    x = np.asarray(data)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(2 ** (8 * x.dtype.itemsize - 1))
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim == 1:
        x = x[None]
    else:
        x = x.T
    return torch.from_numpy(np.ascontiguousarray(x)), int(sr)


def segment_plan(orig_length, sample_rate=24000, min_duration=30, rec_frame_rate=12.5, reason_frame_rate=5):
    """Index arithmetic of audio2token (:98-109, 125-128) — pure function: how often the clip is self-concatenated, how many
    (min_samples + 240)-sample segments are cut, how many tokens are kept."""
    min_samples = int(min_duration * sample_rate)
    output_len = int(orig_length / float(sample_rate) * rec_frame_rate) + 1
    output_len_reason = int(orig_length / float(sample_rate) * reason_frame_rate) + 1
    n = orig_length
    while n < min_samples + 240:
        n *= 2
    int_max_len = n // min_samples + 1
    n *= 2
    seg = min_samples + 240
    total = int(int_max_len * seg)
    if n < total:
        raise ValueError("audio2token: the doubled clip is shorter than the segment grid (the reference's reshape fails here too)")
    return dict(segment=seg, n_segments=int_max_len, total=total, output_len=output_len, output_len_reason=output_len_reason)


class ReasoningTokenizer:
    """`model`: a prepared models.AudioDiffusion1D.AudioDiffusion1D (RVQs, AudioThinking encoder, DiT); `sq_codec`: a
    prepared ScalarModel.  For runs without a DiT, `latent_fn(cond (B, 500, 768), steps) -> (B, 500, 136)` may stand in for
    the flow-matching stage, with `vq_*` given directly (round-1 interface, kept for the sub-graph tests)."""

    def __init__(self, sq_codec=None, vq_phone=None, vq_semantic=None, vq_acoustic=None, latent_fn=None, device="cuda",
                 train_config=None, model_path=None, music_ssl_folder=None, model=None, feature_extractor=None):
        self.device = torch.device(device)
        self.sample_rate = 24000
        self.rec_frame_rate, self.reason_frame_rate, self.sq_codec_hz = 12.5, 5, 25        # :31-33
        self.SQCodec, self.latent_fn, self.model = sq_codec, latent_fn, model
        self.feature_extractor = feature_extractor          # Whisper log-mel front end (:67-72), out of scope: injected or None
        self.skip_discarded_segments = True                 # audio2token: do not encode segments whose tokens the reference slices away
        self.vq = (vq_phone, vq_semantic, vq_acoustic)
        if model is not None:
            self.vq = (model.vq_pronunciation_semantic, model.vq_structure_semantic, model.vq_acoustic)
        if train_config is not None:
            self._load_released(train_config, model_path)

    def _load_released(self, train_config, model_path):
        """The reference's constructor (:22-66): yaml -> SQ-Codec (sq_config yaml `generator.config`, sq_resume['codec_model'],
        scalar24k.py:423-437) + AudioDiffusion1D (transformer_diffusion_config json; model_path['model'], 'module.' prefixes
        stripped, strict=False).  The frozen SSL encoders / Whisper front end named in the yaml are not loaded (out of scope):
        decoding works, `tokenize(path)` needs `model.ssl_features`.  Widths of the encode side are read off the checkpoint."""
        import yaml
        from .models.AudioDiffusion1D import AudioDiffusion1D
        from .models.scalar24k import ScalarModel
        with open(train_config, "r", encoding="utf-8") as f:
            ta = yaml.safe_load(f)
        with open(ta["sq_config"], "r", encoding="utf-8") as f:
            sq_cfg = yaml.safe_load(f)["generator"]["config"]
        sq = ScalarModel(**sq_cfg)
        sq.load_state_dict(torch.load(ta["sq_resume"], map_location="cpu")["codec_model"])
        self.SQCodec = sq.to(self.device).prepare()
        sd = {}
        if model_path is not None:
            sd = torch.load(model_path, map_location="cpu")["model"]
            sd = {(k.split("module.")[-1] if k.startswith("module.") else k): v for k, v in sd.items()}
        dims = dict(whisper_fea_dim=sd["d_conv_whisper.weight"].shape[0], wavlm_fea_dim=sd["d_conv_wavlm.weight"].shape[0],
                    codec_dim=sd["cond_feature_emb.weight"].shape[0],
                    encoder_depth=1 + max(int(k.split(".")[2]) for k in sd if k.startswith("audio_thinking.encoder_transformers."))) if sd else {}
        model = AudioDiffusion1D(num_channels=ta.get("num_channels"), unet_model_config_path=ta["transformer_diffusion_config"], **dims)
        mine = model.state_dict()
        model.load_state_dict({k: v for k, v in sd.items() if k in mine}, strict=False)       # SSL / LLM keys of the checkpoint are not ours
        if sd:
            # strict=False hides name mismatches: a parameter of THIS module tree that the checkpoint does not name would silently
            # keep its random initialisation and stage 2 would write garbage waveforms without an error
            missing = sorted(set(mine) - set(sd))
            if missing:
                raise RuntimeError(f"{model_path}: {len(missing)} parameter(s) of the codec model are not in the checkpoint "
                                   f"(first: {missing[:5]}) — a key-name mismatch between this module tree and the released one")
        self.model = model.to(self.device)
        self.model.sq_codec_latent = sq_cfg["latent_hidden_dim"]
        self.model.prepare()
        self.model.init_device_dtype(self.device, torch.float32)
        self.vq = (model.vq_pronunciation_semantic, model.vq_structure_semantic, model.vq_acoustic)

    @property
    def is_discrete(self):
        return True

    # ---- audio -> tokens ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def audio2token(self, orig_samples, sr, return_reasoning_text=False, task_name="speech_reasoning", min_duration=30, batch_size=6):
        """:86-129.  orig_samples (1, N) at 24 kHz -> (reason (1, 8, T_r), rec (1, 8, T_s)) int64."""
        if self.model is None:
            raise NotImplementedError("audio2token needs the codec model (AudioDiffusion1D) — pass model=")
        audios = orig_samples.to(self.device)
        audios = audios.squeeze(0) if audios.ndim == 3 else audios
        plan = segment_plan(audios.shape[-1], self.sample_rate, min_duration, self.rec_frame_rate, self.reason_frame_rate)
        while audios.shape[-1] < plan["segment"]:
            audios = torch.cat([audios, audios], -1)
        audios = torch.cat([audios, audios], -1)[:, :plan["total"]]
        audio_input = audios.reshape(1, -1, plan["segment"]).permute(1, 0, 2).reshape(-1, 1, plan["segment"])
        # Waste removal, outputs identical: the reference encodes every segment of the doubled clip and
        # then keeps output_len / output_len_reason tokens (:125-128) — for a 10-s clip the whole second 30-s segment is
        # discarded.  Segments are batch rows (independent of each other), so only the rows whose tokens survive the slice are
        # computed.  The one cross-row coupling is the RNG: time_film draws torch.rand(B, 1, 1) per call (AudioDiffusion1D.py:435)
        # — the three draws of a chunk are made here for the reference's full chunk, in its order, and sliced per row.
        need_rec, need_reason = plan["output_len"], plan["output_len_reason"]
        have_rec = have_reason = 0
        per_rec = per_reason = None                               # tokens per segment, known after the first computed row
        reason_list, rec_list = [], []
        for i in range(0, audio_input.shape[0], batch_size):
            chunk = audio_input[i:i + batch_size]
            Bc = chunk.shape[0]
            if self.skip_discarded_segments and have_rec >= need_rec and have_reason >= need_reason:
                break
            masks = [(torch.rand(Bc, 1, 1, device=self.device) < 0.2).view(Bc) for _ in range(3)]
            r = 0
            while r < Bc:
                if not self.skip_discarded_segments:
                    k = Bc
                elif have_rec >= need_rec and have_reason >= need_reason:
                    break
                elif per_rec is None:
                    k = 1
                else:
                    k = max(-(-(need_rec - have_rec) // per_rec), -(-(need_reason - have_reason) // per_reason), 1)
                k = min(k, Bc - r)
                rows = chunk[r:r + k]
                mels = self.feature_extractor(rows[:, 0, :]) if self.feature_extractor is not None else None
                reasoning_codes, rec_codes, _ = self.model.fetch_codes_batch(rows, mels, additional_feats=[], return_reasoning_text=return_reasoning_text,
                                                                            film_masks=[m[r:r + k] for m in masks])
                rc, mc = torch.cat(reasoning_codes, 1), torch.cat(rec_codes, 1)
                reason_list.append(rc)
                rec_list.append(mc)
                per_reason, per_rec = rc.shape[1], mc.shape[1]
                have_reason += k * per_reason
                have_rec += k * per_rec
                r += k
        reason = torch.cat(reason_list, 0).reshape(-1, 8).unsqueeze(0)
        rec = torch.cat(rec_list, 0).reshape(-1, 8).unsqueeze(0)
        return reason[:, :plan["output_len_reason"], :].transpose(1, 2), rec[:, :plan["output_len"], :].transpose(1, 2)

    def tokenize(self, wav, return_reasoning_text=False, task_name="asr", min_duration=30):
        """:377-390: wav path -> (reason (8, T_r), rec (8, T_s)); tensors pass through."""
        if isinstance(wav, torch.Tensor):
            return wav                                              # :387-388
        if isinstance(wav, str):
            audio, fs = load_wav(wav)
            if audio.shape[0] == 2:
                audio = audio.mean(0, keepdim=True)
            audio = audio.to(self.device)
            if fs != self.sample_rate:
                audio = resample(audio, fs, self.sample_rate)
            reason, rec = self.audio2token(audio, self.sample_rate, return_reasoning_text, task_name=task_name)
            return reason.squeeze(0), rec.squeeze(0)
        raise NotImplementedError

    # ---- tokens -> audio ---------------------------------------------------------------------------------------------
    def codes_to_condition(self, codes):
        """codes (B, 8, T) -> (B, T, 768): sum of the three RVQ look-ups (AudioDiffusion1D.py:570-583) — the conditioning BEFORE
        cond_feature_emb and the x2 up-sampling (those live in AudioDiffusion1D.codes_to_condition); what `latent_fn` stand-ins
        of the sub-graph tests receive."""
        groups = (codes[:, 0:1], codes[:, 1:2], codes[:, 2:])
        out = None
        for vq, c in zip(self.vq, groups):
            q = vq.get_output_from_indices(c.transpose(1, 2).contiguous())
            out = q if out is None else out + q
        return out

    @torch.no_grad()
    def token2audio_no_reason(self, rec_codec, return_reasoning_text=False, duration=20, guidance_scale=1.5, num_steps=20,
                              disable_progress=False):
        if self.latent_fn is None and (self.model is None or not hasattr(self.model, "cfm_wrapper")):
            raise NotImplementedError("decoding needs the codec model with its flow-matching DiT (AudioDiffusion1D(unet_model_config_path=...)) "
                                      "or a latent_fn stand-in")
        rec_codec = rec_codec.to(self.device)
        B = rec_codec.shape[0]
        plan = window_plan(rec_codec.shape[-1], duration, self.rec_frame_rate, self.sample_rate, self.sq_codec_hz)
        rec_codec = tile_codes(rec_codec, plan["tiled_len"])
        L = plan["latent_length"]
        latents = []
        for i, s0 in enumerate(plan["starts"]):
            window = rec_codec[:, :, s0:s0 + plan["min_codes"]]
            if self.latent_fn is not None:
                lat = self.latent_fn(self.codes_to_condition(window), num_steps).float()
            elif i == 0:                                             # :271-276: random "true" latent, no in-context frames
                # drawn from the CPU generator and moved, as the reference does (:235 `torch.randn(...).to(self.device)`): the
                # device generator is consumed only by prepare_latents, in the reference's order
                first = torch.randn(B, L, self.model.sq_codec_latent).to(self.device)
                lat = self.model.inference_codes([window], None, first, L, 0, additional_feats=[], guidance_scale=1.5, num_steps=num_steps,
                                                 scenario="other_seg")
            else:                                                    # :277-284: the previous window's tail as in-context frames
                true = latents[-1][:, -plan["ovlp_frames"]:, :]
                pad = torch.randn(B, L - true.shape[1], true.shape[-1]).to(self.device)       # :282, CPU generator as well
                lat = self.model.inference_codes([window], None, torch.cat([true, pad], 1), L, true.shape[1], additional_feats=[],
                                                 guidance_scale=1.5, num_steps=num_steps, scenario="other_seg")
            latents.append(lat.float())
        segments = [self.SQCodec.decode(lat.transpose(1, 2).contiguous()).squeeze(0) for lat in latents]      # (1, N) per :295
        return crossfade_concat(segments, plan["wav_window"], plan["wav_ovlp"], plan["target_len"])

    @torch.no_grad()
    def detokenize_no_reason_batch(self, rec_codecs, steps=50, guidance_scale=1.5, max_batch=8, duration=20):
        """list of rec_codec (8, T_u) -> list of waves (1, N_u) float32 on the CPU: `detokenize_no_reason` for several utterances
        with window k of up to `max_batch` of them going through ONE flow-matching solve (2 x P x 500 rows per DiT step instead of
        1000: the regime the many-row GEMM is built for) and ONE SQ-Codec decode.  Windows of one utterance stay sequential — window
        k + 1 takes the last 32 latent frames of window k as in-context frames (reason_tokenizer.py:277-283) — so the batch is over
        utterances ("shard by utterance, not by window").
        Randomness: every `torch.randn` of the one-by-one path (:235 / :282 on the CPU generator, AudioDiffusion1D.py:655 on the
        device generator) is drawn HERE, utterance by utterance and window by window — the order a loop over
        `detokenize_no_reason` consumes the two generators in — and handed to the windows, so a seeded batch run produces the
        utterances' one-by-one noise.  With the row-invariant GEMM contract (Transformer1DModel.sum_order = 0, no K slabs) the
        waves are then bit-identical to the one-by-one ones (tests/test_gpu_codec_model.py); with the default order-free DiT they
        agree to the DiT's own bf16 noise.
        `guidance_scale` is accepted and NOT used, exactly as in the reference: its token2audio_no_reason passes the literal 1.5 to
        inference_codes whatever the caller asked for (reason_tokenizer.py:274,282) — the one-by-one path here does the same, and the
        batch path must produce the one-by-one waves."""
        if self.latent_fn is not None or self.model is None or not hasattr(self.model, "cfm_wrapper"):
            return [self.detokenize_no_reason(c, steps=steps, guidance_scale=guidance_scale) for c in rec_codecs]
        dev = self.device
        plans = [window_plan(c.shape[-1], duration, self.rec_frame_rate, self.sample_rate, self.sq_codec_hz) for c in rec_codecs]
        tiled = [tile_codes(c.unsqueeze(0).to(dev), p["tiled_len"]) for c, p in zip(rec_codecs, plans)]
        L, ov, Cl = plans[0]["latent_length"], plans[0]["ovlp_frames"], self.model.sq_codec_latent
        nwin = [len(p["starts"]) for p in plans]
        cpu_noise, dev_noise = [], []
        for p, w in zip(plans, nwin):                                 # the one-by-one consumption order of both generators: utterance by
            cn, dn = [], []                                           # utterance, window by window, :235 / :282 first, then :655
            for i in range(w):
                cn.append(torch.randn(1, L if i == 0 else L - ov, Cl))
                dn.append(self.model.prepare_latents(1, L, torch.float32, dev))
            cpu_noise.append(cn); dev_noise.append(dn)
        latents = [[] for _ in rec_codecs]
        segments = [[] for _ in rec_codecs]
        for i in range(max(nwin)):
            active = [u for u, w in enumerate(nwin) if w > i]
            for g0 in range(0, len(active), max(1, max_batch)):
                grp = active[g0:g0 + max(1, max_batch)]
                s0 = plans[grp[0]]["starts"][i]                        # i * hop: the same for every utterance
                window = torch.cat([tiled[u][:, :, s0:s0 + plans[u]["min_codes"]] for u in grp], 0)
                noise = torch.cat([dev_noise[u][i] for u in grp], 0)
                if i == 0:
                    true, n_inc = torch.cat([cpu_noise[u][0] for u in grp], 0).to(dev), 0
                else:
                    true = torch.cat([torch.cat([latents[u][-1][:, -ov:, :], cpu_noise[u][i].to(dev)], 1) for u in grp], 0)
                    n_inc = ov
                lat = self.model.inference_codes([window], None, true, L, n_inc, additional_feats=[], guidance_scale=1.5, num_steps=steps,
                                                 scenario="other_seg", noise=noise).float()
                wav = self.SQCodec.decode(lat.transpose(1, 2).contiguous())          # (P, 1, N)
                for k, u in enumerate(grp):
                    latents[u].append(lat[k:k + 1])
                    segments[u].append(wav[k])
        return [crossfade_concat(seg, p["wav_window"], p["wav_ovlp"], p["target_len"]) for seg, p in zip(segments, plans)]

    def detokenize_no_reason(self, rec_codec, return_reasoning_text=False, min_duration=30, steps=50, guidance_scale=1.5,
                             disable_progress=False):
        """rec_codec (8, T) -> wave (1, N) float32 on the CPU (:399-404)."""
        return self.token2audio_no_reason(rec_codec.unsqueeze(0), return_reasoning_text, guidance_scale=guidance_scale,
                                          num_steps=steps, disable_progress=disable_progress)

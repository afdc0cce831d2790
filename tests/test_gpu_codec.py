"""GPU parity of the codec conv stacks against outputs of the reference itself (golden, fp32 CPU):
waveforms / latents within 1e-4 RMS (the north_star tolerance), in fact ~1e-6."""
import json
import os

import numpy as np
import pytest
import torch

from make_golden_codec import SCALAR_CFG, SEANET_CFG, codec_state_dict
from weights import seeded_tensor

pytestmark = pytest.mark.gpu


def _rms(a, b):
    return float(np.sqrt(np.mean((a - b) ** 2)))


def test_scalar_model_encode_decode_vs_reference(golden_dir):
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    d = np.load(os.path.join(golden_dir, "codec_toy.npz"))
    m = ScalarModel(**SCALAR_CFG)
    m.load_state_dict(codec_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 21))
    m = m.cuda().prepare()
    wav = seeded_tensor((2, 1, 16 * 130 + 5), 31, std=0.3).cuda()
    lat = m.encode(wav)
    assert lat.shape == d["scalar_latent"].shape
    assert _rms(lat.cpu().numpy(), d["scalar_latent"]) < 1e-5
    # decode the REFERENCE's latent (the round(9x)/9 snap makes decode discontinuous in its input)
    rec = m.decode(torch.from_numpy(d["scalar_latent"]).cuda())
    assert rec.shape == d["scalar_wav"].shape
    assert _rms(rec.cpu().numpy(), d["scalar_wav"]) < 1e-4 * max(1.0, float(np.abs(d["scalar_wav"]).max()))


def test_seanet_encoder_decoder_vs_reference(golden_dir):
    from uniaudio2_amd.tools.tokenizer.MimiCodec.model.modules.seanet import SEANetDecoder, SEANetEncoder
    d = np.load(os.path.join(golden_dir, "codec_toy.npz"))
    enc, dec = SEANetEncoder(**SEANET_CFG), SEANetDecoder(**SEANET_CFG)
    for mod, seed in ((enc, 41), (dec, 42)):
        mod.load_state_dict(codec_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()}, seed))
    enc, dec = enc.cuda(), dec.cuda()
    wav = seeded_tensor((2, 1, 8 * 77 + 3), 32, std=0.3).cuda()
    z = enc(wav)
    assert z.shape == d["seanet_latent"].shape and _rms(z.cpu().numpy(), d["seanet_latent"]) < 1e-5
    y = dec(torch.from_numpy(d["seanet_latent"]).cuda())
    assert y.shape == d["seanet_wav"].shape and _rms(y.cpu().numpy(), d["seanet_wav"]) < 1e-4

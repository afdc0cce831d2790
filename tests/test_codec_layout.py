"""State-dict layout of the codec mirrors == the reference's (no GPU): a reference checkpoint loads key for key."""
import json
import os

from make_golden_codec import SCALAR_CFG, SEANET_CFG


def test_codec_state_dict_keys_match_reference(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "codec_toy.json")))
    from uniaudio2_amd.tools.tokenizer.MimiCodec.model.modules.seanet import SEANetDecoder, SEANetEncoder
    from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.scalar24k import ScalarModel
    for key, mod in (("scalar_keys", ScalarModel(**SCALAR_CFG)), ("seanet_enc_keys", SEANetEncoder(**SEANET_CFG)),
                     ("seanet_dec_keys", SEANetDecoder(**SEANET_CFG))):
        mine = {k: list(v.shape) for k, v in mod.state_dict().items()}
        assert mine == {k: s for k, s in meta[key]}, key

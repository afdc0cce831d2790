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


def test_mimicodec_mirror_has_the_reference_checkpoint_layout():
    """The MimiCodec assembly mirror (encode / decode) exposes exactly the state-dict keys and shapes the reference
    model had when tests/golden/make_golden_mimi.py ran it (recorded in mimi_toy.json)."""
    import json
    import os
    from uniaudio2_amd.tools.tokenizer.MimiCodec.model.models.MimiCodec import MimiCodec
    meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mimi_toy.json")))
    m = MimiCodec(**meta["config"])
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    want = {k: s for k, s in meta["keys"]}
    assert got == want, (sorted(set(got) ^ set(want))[:10], [k for k in got if k in want and got[k] != want[k]][:10])

"""Toy flow-matching DiT (Transformer1DModel, models/model_config.json shrunk): key names and shapes as diffusers' modules
register them under models/transformer_1d_flow.py.  Test data plumbing."""
from collections import OrderedDict

from codec_model_stub import module_state_dict

CFG = dict(heads=4, head_dim=64, layers=2, in_channels=2 * 24 + 256, out_channels=24)


def shapes(cfg=CFG):
    D, Cin, Cout = cfg["heads"] * cfg["head_dim"], cfg["in_channels"], cfg["out_channels"]
    s = OrderedDict()
    s["proj_in.ffn_1.weight"], s["proj_in.ffn_1.bias"] = (D, Cin, 3), (D,)
    s["proj_in.ffn_2.weight"], s["proj_in.ffn_2.bias"] = (D, D), (D,)
    for i in range(cfg["layers"]):
        p = f"transformer_blocks.{i}."
        s[p + "scale_shift_table"] = (6, D)
        for c in "qkv":
            s[p + f"attn1.to_{c}.weight"], s[p + f"attn1.to_{c}.bias"] = (D, D), (D,)
        s[p + "attn1.to_out.0.weight"], s[p + "attn1.to_out.0.bias"] = (D, D), (D,)
        s[p + "ff.net.0.proj.weight"], s[p + "ff.net.0.proj.bias"] = (4 * D, D), (4 * D,)
        s[p + "ff.net.2.weight"], s[p + "ff.net.2.bias"] = (D, 4 * D), (D,)
    s["scale_shift_table"] = (2, D)
    s["proj_out.ffn_1.weight"], s["proj_out.ffn_1.bias"] = (Cout, D, 3), (Cout,)
    s["proj_out.ffn_2.weight"], s["proj_out.ffn_2.bias"] = (Cout, Cout), (Cout,)
    s["adaln_single.emb.timestep_embedder.linear_1.weight"], s["adaln_single.emb.timestep_embedder.linear_1.bias"] = (D, 512), (D,)
    s["adaln_single.emb.timestep_embedder.linear_2.weight"], s["adaln_single.emb.timestep_embedder.linear_2.bias"] = (D, D), (D,)
    s["adaln_single.linear.weight"], s["adaln_single.linear.bias"] = (6 * D, D), (6 * D,)
    return s


def state_dict(seed, cfg=CFG):
    return module_state_dict(shapes(cfg), seed)

"""One-window DiT step (2 x 500 rows, released shape, random init) replayed N times: run under `rocprofv3 --kernel-trace --stats` and
summarise with tools/rocpd_stats.py --by-grid to see every launch of a layer.  Usage: python tools/ubench/dit_step_profile.py [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.AudioDiffusion1D import AudioDiffusion1D
from uniaudio2_amd.tools.tokenizer.ReasoningCodec_film.models.transformer_1d_flow import RELEASED_CONFIG
dev = torch.device("cuda")
torch.manual_seed(2)
model = AudioDiffusion1D(unet_model_config_path=dict(RELEASED_CONFIG), encoder_depth=1)
with torch.no_grad():
    for _, p_ in model.named_parameters():
        if p_.dim() > 1:
            p_.normal_(0, 0.02)
model = model.to(dev).prepare()
est = model.cfm_wrapper.estimator
x = torch.randn(2, 500, RELEASED_CONFIG["in_channels"], device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
est(x, 0.5); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    est(x, 0.5)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms per guided step (graph replay)")

import sys, time, torch
sys.path.insert(0, '.')
import bench
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
for i in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est(x, 0.5)
    torch.cuda.synchronize()
    print(i, f"{(time.perf_counter() - t0) * 1e3:.2f} ms", "graphs", len(est._graphs), flush=True)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    est(x, 0.5, use_graph=False)
    torch.cuda.synchronize()
    print("eager", i, f"{(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)

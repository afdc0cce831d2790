"""The exact-fp32 plan's B = 1 frame (the contract whose ids are identical to the reference's): a few utterances for
`rocprofv3 --kernel-trace --stats`.  Usage: python tools/ubench/fp32_frame_profile.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
model.setup_caches(1, dtype=torch.float32, max_seq_length=2048, max_rows=64, log_frames=128)
t, m = bench.make_prompt(dev, 1000)
bench.utterance(model, t, m)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    bench.utterance(model, t, m)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"fp32 plan: {dt * 1e3:.1f} ms per utterance = {8 * bench.FRAMES / dt:.1f} audio tokens/s")

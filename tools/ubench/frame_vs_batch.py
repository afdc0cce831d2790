"""Decode frame time against the number of live sequences (what config 4's retirement schedule walks through): bench.batched_leg at
B = 1 ... 64.  Usage on the GPU box: python tools/ubench/frame_vs_batch.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
model = bench.build_model(dev)
for B in [int(v) for v in sys.argv[1:]] or [1, 4, 8, 16, 17, 24, 32, 33, 48, 64]:
    r = bench.batched_leg(model, dev, B=B, frames=16, max_seq=256)
    print(f"B={B:3d}  frame {r['decode_ms_per_frame']:.3f} ms  skip-text {r['decode_ms_per_frame_skip_text_head']}  per row {1e3 * r['decode_ms_per_frame_skip_text_head'] / B:.1f} us", flush=True)

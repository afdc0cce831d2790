"""Profile driver (rocprofv3 --kernel-trace --stats -- python tools/ubench/prof_legs.py LEG): one information leg of bench.py
per run so that the kernel-stats table is that leg's alone.  LEG in {config3, batched, config5, codec, frame}."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

leg = sys.argv[1] if len(sys.argv) > 1 else "config3"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
if leg == "codec":
    print(bench.codec_leg(dev))
    sys.exit(0)
if leg == "stage2":
    print(bench.stage2_leg(dev))
    sys.exit(0)
model = bench.build_model(dev)
if os.environ.get("UA2_FORCE_LINEAR_MODE"):      # A/B: 4 = weights-stationary forms wherever they exist, 5 = the tiled GEMM everywhere (same bits)
    from uniaudio2_amd._lib import lib
    lib.ua2_debug_force_general_linear(int(os.environ["UA2_FORCE_LINEAR_MODE"]))
ofr = int(os.environ.get("UA2_ORDER_FREE_ROWS", "0"))
if leg == "config3":
    print(bench.config3_leg(model, dev, order_free_rows=ofr))
elif leg == "batched":
    print(bench.batched_leg(model, dev))
elif leg == "batched256":
    print(bench.batched_leg(model, dev, B=256, frames=12, max_seq=128))
elif leg == "batched1024":
    print(bench.batched_leg(model, dev, B=1024, frames=6, max_seq=64, order_free_rows=ofr))
elif leg == "batched_both":
    print(bench.batched_leg(model, dev))
    print(bench.batched_leg(model, dev, B=256, frames=12, max_seq=128))
elif leg == "config5":
    print(bench.config5_leg(model, dev, frames=120))
else:
    model.setup_caches(1, dtype=torch.bfloat16, max_seq_length=2048, max_rows=64, log_frames=128)
    t, m = bench.make_prompt(dev, 1000)
    for _ in range(3):
        bench.utterance(model, t, m)
    torch.cuda.synchronize()
    print("ok")

#!/usr/bin/env python3
"""Host <-> device copy rates for the buffers that cross the boundary (PCIe-inclusive note): the config-3 feature batch
(32 clips x the four SSL feature maps), one 20-s waveform, a frame log.  python tools/ubench/pcie.py"""
import time
import torch

dev = torch.device("cuda:0")


def rate(t_host, to_dev, n=5):
    d = torch.empty_like(t_host, device=dev) if to_dev else t_host.to(dev)
    h = t_host if to_dev else torch.empty_like(t_host)
    (d.copy_(h) if to_dev else h.copy_(d)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        (d.copy_(h) if to_dev else h.copy_(d))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt * 1e3, t_host.numel() * t_host.element_size() / dt / 1e9


feat = torch.randn(32 * (1024 * 1500 + 768 * 1500 + 2 * 1024 * 750))          # what bench.py's config-3 encode leg is handed (fp32)
for name, t in (("config-3 features, 32 clips (pageable)", feat), ("config-3 features, 32 clips (pinned)", feat.pin_memory()),
                ("20-s waveform (480 000 fp32, pageable)", torch.randn(480000)), ("frame log 74 x 9 int32", torch.zeros(74 * 9, dtype=torch.int32))):
    for to_dev in (True, False):
        ms, gbs = rate(t, to_dev)
        print(f"{name:45s} {'H2D' if to_dev else 'D2H'}: {ms:8.3f} ms  {gbs:6.2f} GB/s")

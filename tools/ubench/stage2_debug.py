"""Bisect helper for the batched stage 2 (toy DiT of tests/test_gpu_codec_model.py): the test's scenario step by step with a
device synchronisation and a marker after every stage.  Env UA2_EULER_NO_GRAPH / UA2_CODEC_NO_GRAPH switch the two recorded graphs off."""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
from test_gpu_codec_model import _stage2_tokenizer

os.environ["UA2_GEMM_NO_KSPLIT"] = "1"
L = 136
dit = dict(num_attention_heads=4, attention_head_dim=64, in_channels=2 * L + 768, out_channels=L, num_layers=2)
tok = _stage2_tokenizer(dit, sum_order=int(os.environ.get("ORDER", "0")))
g = torch.Generator().manual_seed(11)
codes = [torch.randint(0, 8192, (8, T), generator=g) for T in (437, 100, 250, 600)]


def mark(s):
    torch.cuda.synchronize()
    print("OK", s, flush=True)


orig_inf, orig_dec = tok.model.inference_codes, tok.SQCodec.decode


def inf(*a, **k):
    r = orig_inf(*a, **k)
    mark(f"inference_codes P={a[0][0].shape[0]} n_inc={a[4]}")
    return r


def dec(x, *a, **k):
    r = orig_dec(x, *a, **k)
    mark(f"decode B={x.shape[0]}")
    return r


tok.model.inference_codes, tok.SQCodec.decode = inf, dec
if os.environ.get("SINGLE_FIRST", "1") == "1":
    torch.manual_seed(123)
    single = [tok.detokenize_no_reason(c, steps=3) for c in codes]
    mark("single loop")
torch.manual_seed(123)
batch = tok.detokenize_no_reason_batch(codes, steps=3, max_batch=2)
mark("batch of 2")
torch.manual_seed(123)
batch4 = tok.detokenize_no_reason_batch(codes, steps=3, max_batch=8)
mark("batch of 4")
if os.environ.get("SINGLE_FIRST", "1") == "1":
    for a, b, c in zip(single, batch, batch4):
        print("equal:", torch.equal(a, b), torch.equal(a, c), float((a - b).abs().max()), float((a - c).abs().max()))

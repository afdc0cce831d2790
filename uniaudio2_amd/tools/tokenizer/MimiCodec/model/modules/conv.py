"""Causal / asymmetric-padding conv wrappers of the SEANet family, host side.

Mirror of the reference's llm_modules/conv.py == tools/tokenizer/MimiCodec/model/modules/conv.py
(StreamingConv1d :168-254, StreamingConvTranspose1d :265-329, NormConv1d :111-132,
NormConvTranspose1d :135-158; same attribute tree, so the same state-dict keys `conv.conv.weight`,
`convtr.convtr.weight`).  Scope: what the Mimi instance uses (MimiCodec.py:47-50, 67-68) — norm "none",
pad_mode "constant" (SEANet) or "replicate" (the down-sampler), groups 1 or channel-wise (the up-sampler);
whole-sequence forward and the causal streaming mode (`with module.streaming(batch):`, conv.py:245-252, 306-329:
the fixed left padding is added once, trims are skipped, streaming.py's state machines do the rest).
Arithmetic: ua2_conv1d / ua2_dwconv1d.
"""
import math

import torch.nn as nn
import torch.nn.functional as F

from ......_lib import ACT_NONE
from .streaming import RawStreamingConv1d, RawStreamingConvTranspose1d, StreamingModule


def get_extra_padding_for_conv1d(length: int, kernel_size: int, stride: int, padding_total: int = 0) -> int:
    """conv.py:50-58: right padding so that the last window is full."""
    n_frames = (length - kernel_size + padding_total) / stride + 1
    ideal_length = (math.ceil(n_frames) - 1) * stride + (kernel_size - padding_total)
    return ideal_length - length


class NormConv1d(nn.Module):
    def __init__(self, *args, causal=False, norm="none", **kwargs):
        super().__init__()
        if norm != "none":
            raise NotImplementedError("only norm='none' (the Mimi configuration) is on the hot path")
        self.conv = RawStreamingConv1d(*args, **kwargs)


class NormConvTranspose1d(nn.Module):
    def __init__(self, *args, causal=False, norm="none", **kwargs):
        super().__init__()
        if norm != "none":
            raise NotImplementedError("only norm='none' (the Mimi configuration) is on the hot path")
        self.convtr = RawStreamingConvTranspose1d(*args, **kwargs)


class StreamingConv1d(StreamingModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True, causal=False,
                 norm="none", norm_kwargs=None, pad_mode="constant"):
        super().__init__()
        if pad_mode not in ("constant", "replicate"):
            raise NotImplementedError("pad_mode 'constant' / 'replicate' (the Mimi configuration) are on the hot path")
        self.conv = NormConv1d(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups, bias=bias,
                               causal=causal, norm=norm)
        self.causal, self.pad_mode = causal, pad_mode

    def _init_streaming_state(self, batch_size: int):
        assert self.causal, "streaming is only supported for causal convs"
        c = self.conv.conv
        return {"padding_to_add": (c.kernel_size[0] - 1) * c.dilation[0] + 1 - c.stride[0]}

    def prepare(self):
        self.conv.conv._weights()

    def forward(self, x, pre_act=ACT_NONE, residual=None):
        """conv.py:232-254; `pre_act` fuses the activation SEANet applies before the conv, `residual` its skip add."""
        c = self.conv.conv
        k, s, d = c.kernel_size[0], c.stride[0], c.dilation[0]
        k_eff = (k - 1) * d + 1
        padding_total = k_eff - s
        state = self._streaming_state
        if state is not None:                                             # conv.py:245-252
            if state["padding_to_add"] > 0 and x.shape[-1] > 0:
                x = F.pad(x, (state["padding_to_add"], 0), mode=self.pad_mode)
                state["padding_to_add"] = 0
            return c.run(x, pre_act=pre_act, residual=residual)
        T = x.shape[-1]
        extra = get_extra_padding_for_conv1d(T, k_eff, s, padding_total)
        if self.causal:
            pad_l, pad_r = padding_total, extra
        else:
            pad_r = padding_total // 2
            pad_l = padding_total - pad_r
            pad_r += extra
        tout = (T + pad_l + pad_r - k_eff) // s + 1
        if self.pad_mode == "replicate" and (pad_l or pad_r):     # conv.py:80-94 pad1d: edge values, then a plain conv
            x = F.pad(x, (pad_l, pad_r), mode="replicate")
            pad_l = 0
        return c.run(x, pad_left=pad_l, tout=tout, pre_act=pre_act, residual=residual)


class StreamingConvTranspose1d(StreamingModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, groups=1, bias=True, causal=False, norm="none",
                 trim_right_ratio=1.0, norm_kwargs=None):
        super().__init__()
        self.convtr = NormConvTranspose1d(in_channels, out_channels, kernel_size, stride, groups=groups, bias=bias, causal=causal,
                                          norm=norm)
        self.causal, self.trim_right_ratio = causal, trim_right_ratio
        assert self.causal or self.trim_right_ratio == 1.0, "`trim_right_ratio` != 1.0 only makes sense for causal convolutions"

    def _init_streaming_state(self, batch_size: int):
        assert self.causal, "streaming is only supported for causal convtrs"
        return {}

    def prepare(self):
        self.convtr.convtr._weights()

    def forward(self, x, pre_act=ACT_NONE):
        """conv.py:306-329: trim the k - s fixed padding (all of it on the right for causal, trim_right_ratio = 1);
        nothing is trimmed in streaming mode (the held-back tail plays that role)."""
        c = self.convtr.convtr
        if self.is_streaming:
            return c.run(x, pre_act=pre_act)
        padding_total = c.kernel_size[0] - c.stride[0]
        if self.causal:
            pad_r = math.ceil(padding_total * self.trim_right_ratio)
            pad_l = padding_total - pad_r
        else:
            pad_r = padding_total // 2
            pad_l = padding_total - pad_r
        return c.run(x, trim_left=pad_l, trim_right=pad_r, pre_act=pre_act)

"""Causal / asymmetric-padding conv wrappers of the SEANet family, host side.

Mirror of the reference's llm_modules/conv.py == tools/tokenizer/MimiCodec/model/modules/conv.py
(StreamingConv1d :168-254, StreamingConvTranspose1d :265-329, NormConv1d :111-132,
NormConvTranspose1d :135-158; same attribute tree, so the same state-dict keys `conv.conv.weight`,
`convtr.convtr.weight`).  Scope: what the Mimi instance uses (MimiCodec.py:47-50, 67-68) — norm "none",
pad_mode "constant" (SEANet) or "replicate" (the down-sampler), groups 1 or channel-wise (the up-sampler);
the non-streaming (whole-sequence) forward.  Arithmetic: ua2_conv1d / ua2_dwconv1d.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...... import ops
from ......_lib import ACT_ELU, ACT_NONE


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
        self.conv = nn.Conv1d(*args, **kwargs)


class NormConvTranspose1d(nn.Module):
    def __init__(self, *args, causal=False, norm="none", **kwargs):
        super().__init__()
        if norm != "none":
            raise NotImplementedError("only norm='none' (the Mimi configuration) is on the hot path")
        self.convtr = nn.ConvTranspose1d(*args, **kwargs)


class StreamingConv1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True, causal=False,
                 norm="none", norm_kwargs=None, pad_mode="constant"):
        super().__init__()
        if pad_mode not in ("constant", "replicate"):
            raise NotImplementedError("pad_mode 'constant' / 'replicate' (the Mimi configuration) are on the hot path")
        if groups not in (1, in_channels) or (groups != 1 and in_channels != out_channels):
            raise NotImplementedError("groups must be 1 or channel-wise (in == out == groups)")
        self.conv = NormConv1d(in_channels, out_channels, kernel_size, stride, dilation=dilation, groups=groups, bias=bias,
                               causal=causal, norm=norm)
        self.causal, self.pad_mode = causal, pad_mode
        self._w = None

    def prepare(self):
        c = self.conv.conv
        if c.groups == 1:
            self._w, self._k = ops.pack_conv_weight(c.weight.detach().float())
        else:
            self._w, self._k = c.weight.detach().float().reshape(c.out_channels, -1).contiguous(), c.kernel_size[0]
        self._bias = c.bias.detach().float().contiguous() if c.bias is not None else None

    def forward(self, x, pre_act=ACT_NONE, residual=None):
        """conv.py:232-254 (non-streaming branch); `pre_act` fuses the activation SEANet applies before the conv."""
        if self._w is None:
            self.prepare()
        c = self.conv.conv
        k, s, d = c.kernel_size[0], c.stride[0], c.dilation[0]
        k_eff = (k - 1) * d + 1
        padding_total = k_eff - s
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
        if c.groups != 1:
            assert pre_act == ACT_NONE and residual is None
            return ops.dwconv1d(x, self._w, stride=s, dilation=d, pad_left=pad_l, Tout=tout, bias=self._bias)
        return ops.conv1d(x, self._w, k, c.out_channels, stride=s, dilation=d, pad_left=pad_l, Tout=tout, bias=self._bias,
                          pre_act=pre_act, residual=residual)


class StreamingConvTranspose1d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, groups=1, bias=True, causal=False, norm="none",
                 trim_right_ratio=1.0, norm_kwargs=None):
        super().__init__()
        if groups not in (1, in_channels) or (groups != 1 and in_channels != out_channels):
            raise NotImplementedError("groups must be 1 or channel-wise (in == out == groups)")
        self.convtr = NormConvTranspose1d(in_channels, out_channels, kernel_size, stride, groups=groups, bias=bias, causal=causal,
                                          norm=norm)
        self.causal, self.trim_right_ratio = causal, trim_right_ratio
        assert self.causal or self.trim_right_ratio == 1.0, "`trim_right_ratio` != 1.0 only makes sense for causal convolutions"
        self._w = None

    def prepare(self):
        c = self.convtr.convtr
        if c.groups == 1:
            self._w, self._m = ops.pack_convtr_weight(c.weight.detach().float(), c.stride[0])
        else:
            self._w, self._m = c.weight.detach().float().reshape(c.in_channels, -1).contiguous(), None
        self._bias = c.bias.detach().float().contiguous() if c.bias is not None else None

    def forward(self, x, pre_act=ACT_NONE):
        """conv.py:306-329: trim the k - s fixed padding (all of it on the right for causal, trim_right_ratio = 1)."""
        if self._w is None:
            self.prepare()
        c = self.convtr.convtr
        k, s = c.kernel_size[0], c.stride[0]
        padding_total = k - s
        if self.causal:
            pad_r = math.ceil(padding_total * self.trim_right_ratio)
            pad_l = padding_total - pad_r
        else:
            pad_r = padding_total // 2
            pad_l = padding_total - pad_r
        full = (x.shape[-1] - 1) * s + k
        if c.groups != 1:
            assert pre_act == ACT_NONE
            return ops.dwconv1d(x, self._w, stride=s, pad_left=pad_l, Tout=full - pad_l - pad_r, bias=self._bias, transposed=True)
        return ops.conv1d(x, self._w, self._m, c.out_channels, pad_left=self._m - 1, Tout=full - pad_l - pad_r,
                          bias=self._bias, pre_act=pre_act, out_phases=s, out_trim_left=pad_l)

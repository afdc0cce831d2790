"""Integer-factor resamplers of the Mimi latent, host side.

Mirror of tools/tokenizer/MimiCodec/model/modules/resample.py: ConvDownsample1d :13-62 (kernel 2*stride, "replicate"
padding, learnt full or channel-wise weights, or the fixed 1/(2*stride) box filter applied per channel) and
ConvTrUpsample1d :65-119 (transposed kernel 2*stride; the non-learnt form divides by the response to ones :112-118).
Same attribute tree (`conv.conv.conv.weight`, `convtr.convtr.convtr.weight`).  Arithmetic: ua2_conv1d (full) /
ua2_dwconv1d (channel-wise and the fixed one-channel filters).
"""
import torch
import torch.nn as nn

from .conv import StreamingConv1d, StreamingConvTranspose1d


class ConvDownsample1d(nn.Module):
    def __init__(self, stride, dimension=None, causal=False, learnt=False, channel_wise=False):
        super().__init__()
        self.learnt, self.channel_wise = learnt, channel_wise
        groups = 1
        if learnt:
            assert dimension is not None, "Dimension required for learnt convolutions."
            in_channels = out_channels = dimension
            if channel_wise:
                groups = dimension
        else:
            in_channels = out_channels = 1
        self.conv = StreamingConv1d(in_channels, out_channels, kernel_size=2 * stride, stride=stride, causal=causal, groups=groups,
                                    bias=False, pad_mode="replicate")
        if not learnt:
            w = self.conv.conv.conv.weight
            w.requires_grad_(False)
            w.data.fill_(1.0 / (2 * stride))

    def forward(self, x: torch.Tensor):
        b, c, t = x.shape
        if not self.learnt:
            x = x.reshape(b * c, 1, t)
        y = self.conv(x)
        if not self.learnt:
            y = y.reshape(b, c, -1)
        return y


class ConvTrUpsample1d(nn.Module):
    def __init__(self, stride, dimension=None, causal=False, learnt=False, channel_wise=False):
        super().__init__()
        self.learnt, self.channel_wise = learnt, channel_wise
        groups = 1
        if learnt:
            assert dimension is not None, "Dimension required for learnt convolutions."
            in_channels = out_channels = dimension
            if channel_wise:
                groups = dimension
        else:
            in_channels = out_channels = 1
        self.convtr = StreamingConvTranspose1d(in_channels, out_channels, kernel_size=2 * stride, stride=stride, causal=causal,
                                               groups=groups, bias=False)
        if not learnt:
            w = self.convtr.convtr.convtr.weight
            w.requires_grad_(False)
            w.data.fill_(1.0)

    def forward(self, x: torch.Tensor):
        b, c, t = x.shape
        if not self.learnt:
            x = x.reshape(b * c, 1, t)
        y = self.convtr(x)
        if not self.learnt:
            y = y / self.convtr(torch.ones_like(x[:1]))          # resample.py:112-118
            y = y.reshape(b, c, -1)
        return y

"""SEANet encoder / decoder (the conv stack north_star names), host side.

Mirror of the reference's llm_modules/seanet.py == tools/tokenizer/MimiCodec/model/modules/seanet.py
(SEANetResnetBlock :21-94, SEANetEncoder :97-241, SEANetDecoder :244-395): same `model` Sequential
layout, so the same state-dict keys (`model.3.block.1.conv.conv.weight`, ...).  Every ELU -> conv pair
(conv or transposed conv) is one ua2_conv1d launch (ELU fused as the kernel's input activation); the
resblock's skip add is fused into its second conv.
"""
import typing as tp

import numpy as np
import torch
import torch.nn as nn

from ......_lib import ACT_ELU, ACT_NONE
from .conv import StreamingConv1d, StreamingConvTranspose1d
from .streaming import StreamingModule


class SEANetResnetBlock(StreamingModule):
    def __init__(self, dim, kernel_sizes=(3, 1), dilations=(1, 1), activation="ELU", activation_params=None, norm="none",
                 norm_params=None, causal=False, pad_mode="constant", compress=2, true_skip=True):
        super().__init__()
        assert len(kernel_sizes) == len(dilations)
        if activation != "ELU" or not true_skip:
            raise NotImplementedError("only activation='ELU', true_skip=True (the Mimi configuration) are on the hot path")
        hidden = dim // compress
        block = []
        for i, (k, d) in enumerate(zip(kernel_sizes, dilations)):
            in_chs = dim if i == 0 else hidden
            out_chs = dim if i == len(kernel_sizes) - 1 else hidden
            block += [nn.ELU(alpha=1.0), StreamingConv1d(in_chs, out_chs, kernel_size=k, dilation=d, norm=norm, causal=causal, pad_mode=pad_mode)]
        self.block = nn.Sequential(*block)
        self.shortcut = nn.Identity()

    def forward(self, x):
        convs = [m for m in self.block if isinstance(m, StreamingConv1d)]
        h = x
        for i, c in enumerate(convs):                                   # seanet.py:92-94: x + block(x)
            h = c(h, pre_act=ACT_ELU, residual=x if i == len(convs) - 1 else None)
        return h


def _run_sequential(model, x):
    """Walks the reference's Sequential, fusing each ELU into the conv that follows it."""
    pending_elu = False
    for m in model:
        if isinstance(m, nn.ELU):
            pending_elu = True
        elif isinstance(m, StreamingConv1d):
            x = m(x, pre_act=ACT_ELU if pending_elu else ACT_NONE)
            pending_elu = False
        elif isinstance(m, StreamingConvTranspose1d):
            x = m(x, pre_act=ACT_ELU if pending_elu else ACT_NONE)
            pending_elu = False
        elif isinstance(m, SEANetResnetBlock):
            assert not pending_elu
            x = m(x)
        else:
            raise NotImplementedError(type(m))
    assert not pending_elu
    return x


class SEANetEncoder(StreamingModule):
    def __init__(self, channels=1, dimension=128, n_filters=32, n_residual_layers=3, ratios=(8, 5, 4, 2), activation="ELU",
                 activation_params=None, norm="none", norm_params=None, kernel_size=7, last_kernel_size=7,
                 residual_kernel_size=3, dilation_base=2, causal=False, pad_mode="constant", true_skip=True, compress=2,
                 disable_norm_outer_blocks=0, mask_fn=None, mask_position=None):
        super().__init__()
        self.channels, self.dimension, self.n_filters = channels, dimension, n_filters
        self.ratios = list(reversed(ratios))
        self.hop_length = int(np.prod(self.ratios))
        mult = 1
        model: tp.List[nn.Module] = [StreamingConv1d(channels, mult * n_filters, kernel_size, norm=norm, causal=causal, pad_mode=pad_mode)]
        for ratio in self.ratios:
            for j in range(n_residual_layers):
                model += [SEANetResnetBlock(mult * n_filters, kernel_sizes=[residual_kernel_size, 1], dilations=[dilation_base ** j, 1],
                                            norm=norm, activation=activation, causal=causal, pad_mode=pad_mode, compress=compress,
                                            true_skip=true_skip)]
            model += [nn.ELU(alpha=1.0), StreamingConv1d(mult * n_filters, mult * n_filters * 2, kernel_size=ratio * 2, stride=ratio,
                                                         norm=norm, causal=causal, pad_mode=pad_mode)]
            mult *= 2
        model += [nn.ELU(alpha=1.0), StreamingConv1d(mult * n_filters, dimension, last_kernel_size, norm=norm, causal=causal, pad_mode=pad_mode)]
        self.model = nn.Sequential(*model)

    @torch.inference_mode()
    def forward(self, x):
        return _run_sequential(self.model, x.float().contiguous())


class SEANetDecoder(StreamingModule):
    def __init__(self, channels=1, dimension=128, n_filters=32, n_residual_layers=3, ratios=(8, 5, 4, 2), activation="ELU",
                 activation_params=None, final_activation=None, final_activation_params=None, norm="none", norm_params=None,
                 kernel_size=7, last_kernel_size=7, residual_kernel_size=3, dilation_base=2, causal=False, pad_mode="constant",
                 true_skip=True, compress=2, disable_norm_outer_blocks=0, trim_right_ratio=1.0):
        super().__init__()
        if final_activation is not None:
            raise NotImplementedError("final_activation is not used by the Mimi configuration")
        self.dimension, self.channels, self.n_filters, self.ratios = dimension, channels, n_filters, list(ratios)
        self.hop_length = int(np.prod(self.ratios))
        mult = int(2 ** len(self.ratios))
        model: tp.List[nn.Module] = [StreamingConv1d(dimension, mult * n_filters, kernel_size, norm=norm, causal=causal, pad_mode=pad_mode)]
        for ratio in self.ratios:
            model += [nn.ELU(alpha=1.0), StreamingConvTranspose1d(mult * n_filters, mult * n_filters // 2, kernel_size=ratio * 2, stride=ratio,
                                                                  norm=norm, causal=causal, trim_right_ratio=trim_right_ratio)]
            for j in range(n_residual_layers):
                model += [SEANetResnetBlock(mult * n_filters // 2, kernel_sizes=[residual_kernel_size, 1], dilations=[dilation_base ** j, 1],
                                            activation=activation, norm=norm, causal=causal, pad_mode=pad_mode, compress=compress,
                                            true_skip=true_skip)]
            mult //= 2
        model += [nn.ELU(alpha=1.0), StreamingConv1d(n_filters, channels, last_kernel_size, norm=norm, causal=causal, pad_mode=pad_mode)]
        self.model = nn.Sequential(*model)

    @torch.inference_mode()
    def forward(self, z):
        return _run_sequential(self.model, z.float().contiguous())

"""Waveform conv auto-encoder of the live codec (SQ-Codec `ScalarModel`), host side.

Mirror of the reference's tools/tokenizer/ReasoningCodec_film/models/scalar24k.py: same class names,
constructor arguments and module tree, hence the same state-dict keys (`encoder.N.convs.M.conv1.weight_g`
...), same `encode(x) -> latent`, `decode(latent) -> wav` (:392-407).  The modules only hold
parameters; `prepare()` folds the weight-norm reparametrisation (w = g * v / ||v||, exactly
torch._weight_norm) and packs every filter for ua2_conv1d; encode/decode then run one fused HIP
launch per convolution (bias, PReLU, residual add, tanh, round(9x)/9, repeat-upsampling and the
transposed conv's interleave live inside the kernel).  No torch math on the data path.
"""
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from ..... import ops
from ....._lib import ACT_NONE, ACT_PRELU, ACT_ROUND9, ACT_TANH


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


class Conv1d(nn.Conv1d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, bias=True, padding=None, causal=False):
        self.causal = causal
        self.left_padding = 0
        if padding is None:
            if causal:
                padding = 0
                self.left_padding = dilation * (kernel_size - 1)
            else:
                padding = get_padding(kernel_size, dilation)
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation, bias=bias)


class ConvTranspose1d(nn.ConvTranspose1d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, bias=True, padding=None, causal=False):
        if padding is None:
            padding = 0 if causal else (kernel_size - stride) // 2
        if causal:
            assert padding == 0, "padding is not allowed in causal ConvTranspose1d."
            assert kernel_size == 2 * stride, "kernel_size must be equal to 2*stride is not allowed in causal ConvTranspose1d."
        super().__init__(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.causal = causal


def _folded(conv):
    """Effective weight of a (possibly weight-normed) conv module, fp32."""
    if hasattr(conv, "weight_g"):
        return torch._weight_norm(conv.weight_v.detach().float(), conv.weight_g.detach().float(), 0)
    return conv.weight.detach().float()


class _ConvOp:
    """One packed ua2_conv1d launch.  `fast` (default: the class attribute ScalarModel.prepare sets while it builds the
    decoder) selects the bf16 x 3 form of the kernel; the exact-fp32 form otherwise."""
    default_fast = False

    def __init__(self, conv, post_act=ACT_NONE, alpha=None, pre_act=ACT_NONE, in_repeat=1, fast=None):
        fast = _ConvOp.default_fast if fast is None else fast
        dev = conv.bias.device if conv.bias is not None else _folded(conv).device
        w = _folded(conv).to(dev)
        self.bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
        self.post_act, self.pre_act, self.in_repeat = post_act, pre_act, in_repeat
        self.alpha = alpha.detach().float().contiguous() if alpha is not None else None
        if isinstance(conv, nn.ConvTranspose1d):
            self.transposed, self.stride = True, conv.stride[0]
            self.cin = w.shape[0]
            self.cout, self.kfull = w.shape[1], w.shape[2]
            rows, self.K = ops.convtr_phase_rows(w, self.stride)
            self.w, self.w_lo = ops.pack_conv_weight_x3(rows) if fast else (ops.pack_conv_weight(rows)[0], None)
            self.trim = 0 if conv.causal else conv.padding[0]
            self.causal = conv.causal
        else:
            self.transposed, self.stride, self.dil = False, conv.stride[0], conv.dilation[0]
            self.cin = w.shape[1]
            self.cout, self.kfull = w.shape[0], w.shape[2]
            self.K = w.shape[2]
            self.w, self.w_lo = ops.pack_conv_weight_x3(w) if fast else (ops.pack_conv_weight(w)[0], None)
            self.pad_l = conv.left_padding if conv.causal else conv.padding[0]
            self.pad_r = 0 if conv.causal else conv.padding[0]

    def tc_ok(self):
        """This op can run on time-major split planes (ua2_conv1d_tc): bf16 x 3 weights, stride 1, no input pre-activation,
        PReLU / no activation, whole channel groups in."""
        cin = self.cin
        return (self.w_lo is not None and (self.transposed or self.stride == 1) and self.pre_act == ACT_NONE and
                self.post_act in (ACT_NONE, ACT_PRELU) and cin % 32 == 0)

    def run_tc(self, x, residual=None, out_f32=False, fused2=None, variant=0):
        """x: ops.TC.  The decode side's form of __call__ (same geometry, planes in / planes or the fp32 waveform out)."""
        T = x.shape[-1] * self.in_repeat
        if self.transposed:
            full = (T - 1) * self.stride + self.kfull
            tout = full - self.stride if self.causal else full - 2 * self.trim
            return ops.conv1d_tc(x, self.w, self.w_lo, self.K, self.cout, pad_left=self.K - 1, Tout=tout, bias=self.bias,
                                 post_act=self.post_act, post_alpha=self.alpha, out_phases=self.stride, out_trim_left=self.trim,
                                 residual=residual, out_f32=out_f32, variant=variant)
        tout = (T + self.pad_l + self.pad_r - self.dil * (self.kfull - 1) - 1) // self.stride + 1
        return ops.conv1d_tc(x, self.w, self.w_lo, self.K, self.cout, dilation=self.dil, pad_left=self.pad_l, Tout=tout,
                             bias=self.bias, post_act=self.post_act, post_alpha=self.alpha, in_repeat=self.in_repeat,
                             residual=residual, out_f32=out_f32, fused2=fused2, variant=variant)

    def __call__(self, x, residual=None):
        if isinstance(x, ops.TC):
            return self.run_tc(x, residual=residual)
        T = x.shape[-1] * self.in_repeat
        if self.transposed:
            full = (T - 1) * self.stride + self.kfull
            tout = full - self.stride if self.causal else full - 2 * self.trim
            return ops.conv1d(x, self.w, self.K, self.cout, pad_left=self.K - 1, Tout=tout, bias=self.bias,
                              post_act=self.post_act, post_alpha=self.alpha, out_phases=self.stride,
                              out_trim_left=self.trim, residual=residual, w_lo=self.w_lo)
        tout = (T + self.pad_l + self.pad_r - self.dil * (self.kfull - 1) - 1) // self.stride + 1
        return ops.conv1d(x, self.w, self.K, self.cout, stride=self.stride, dilation=self.dil, pad_left=self.pad_l,
                          Tout=tout, bias=self.bias, pre_act=self.pre_act, post_act=self.post_act, post_alpha=self.alpha,
                          residual=residual, in_repeat=self.in_repeat, w_lo=self.w_lo)


class PreProcessor(nn.Module):
    def __init__(self, n_in, n_out, num_samples, kernel_size=7, causal=False):
        super().__init__()
        self.num_samples = num_samples
        self.conv = Conv1d(n_in, n_out, kernel_size=kernel_size, causal=causal)
        self.activation = nn.PReLU()

    def prepare(self):
        self._op = _ConvOp(self.conv, ACT_PRELU, self.activation.weight)

    def run(self, x):
        return ops.avgpool1d(self._op(x), self.num_samples)          # scalar24k.py:120-121


class PostProcessor(nn.Module):
    def __init__(self, n_in, n_out, num_samples, kernel_size=7, causal=False):
        super().__init__()
        self.num_samples = num_samples
        self.conv = Conv1d(n_in, n_out, kernel_size=kernel_size, causal=causal)
        self.activation = nn.PReLU()

    def prepare(self):
        self._op = _ConvOp(self.conv, ACT_PRELU, self.activation.weight, in_repeat=self.num_samples)   # :136-140

    def run(self, x):
        return self._op(x)


class ResidualUnit(nn.Module):
    def __init__(self, n_in, n_out, dilation, res_kernel_size=7, causal=False):
        super().__init__()
        self.conv1 = weight_norm(Conv1d(n_in, n_out, kernel_size=res_kernel_size, dilation=dilation, causal=causal))
        self.conv2 = weight_norm(Conv1d(n_in, n_out, kernel_size=1, causal=causal))
        self.activation1 = nn.PReLU()
        self.activation2 = nn.PReLU()

    def prepare(self):
        self._op1 = _ConvOp(self.conv1, ACT_PRELU, self.activation1.weight)
        self._op2 = _ConvOp(self.conv2, ACT_PRELU, self.activation2.weight)
        # bf16 x 3 form with up to 128 channels: both convs, both PReLUs and the residual add in ONE launch (the 1 x 1 conv's
        # reduction over channels closes inside the workgroup that holds them), h never leaves the chip
        c = self.conv1.out_channels
        self._fused = None
        self._fused_tc = None
        if self._op1.w_lo is not None and self.conv1.in_channels == c and c in (32, 64, 128) and self.conv2.kernel_size[0] == 1:
            self._fused = (self._op2.w, self._op2.w_lo, self._op2.bias, self._op2.alpha)
            # the split-plane kernels reduce the 1 x 1 conv in their own K order (ops.tc_w2_order)
            self._fused_tc = (*ops.pack_conv_weight_x3(ops.tc_w2_order(_folded(self.conv2))), self._op2.bias, self._op2.alpha)

    def run(self, x):
        if isinstance(x, ops.TC):
            if self._fused_tc is not None:
                return self._op1.run_tc(x, fused2=self._fused_tc)     # x read once; h and the residual never leave the chip
            return self._op2.run_tc(self._op1.run_tc(x), residual=x)
        if self._fused is not None:
            o = self._op1
            return ops.conv1d(x, o.w, o.K, o.cout, dilation=o.dil, pad_left=o.pad_l, Tout=x.shape[-1], bias=o.bias, post_act=o.post_act,
                              post_alpha=o.alpha, residual=x, w_lo=o.w_lo, fused2=self._fused)
        return self._op2(self._op1(x), residual=x)                    # :148-151


class DownsampleLayer(nn.Module):
    # `activation=nn.PReLU()` is evaluated once, as in the reference (scalar24k.py:203): every
    # DownsampleLayer built with the default shares ONE PReLU parameter (several state-dict keys, one tensor)
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, causal=False, activation=nn.PReLU(), use_weight_norm=True):
        super().__init__()
        self.activation = activation
        self.layer = Conv1d(in_channels, out_channels, kernel_size, stride=stride, causal=causal)
        if use_weight_norm:
            self.layer = weight_norm(self.layer)

    def prepare(self):
        self._op = _ConvOp(self.layer, ACT_PRELU, self.activation.weight)

    def run(self, x):
        return self._op(x)


class UpsampleLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, causal=False, activation=None, use_weight_norm=True):
        super().__init__()
        self.activation = activation
        self.layer = ConvTranspose1d(in_channels, out_channels, kernel_size, stride=stride, causal=causal)
        if use_weight_norm:
            self.layer = weight_norm(self.layer)

    def prepare(self):
        if self.activation is not None:
            self._op = _ConvOp(self.layer, ACT_PRELU, self.activation.weight)
        else:
            self._op = _ConvOp(self.layer)

    def run(self, x):
        return self._op(x)


class ResEncoderBlock(nn.Module):
    def __init__(self, n_in, n_out, stride, down_kernel_size, res_kernel_size=7, causal=False):
        super().__init__()
        self.convs = nn.ModuleList([ResidualUnit(n_in if i == 0 else n_out // 2, n_out // 2, dilation=d,
                                                 res_kernel_size=res_kernel_size, causal=causal)
                                    for i, d in enumerate((1, 3, 5, 7, 9))])
        self.down_conv = DownsampleLayer(n_in, n_out, down_kernel_size, stride=stride, causal=causal)

    def prepare(self):
        for c in self.convs:
            c.prepare()
        self.down_conv.prepare()

    def run(self, x):
        for c in self.convs:
            x = c.run(x)
        return self.down_conv.run(x)


class ResDecoderBlock(nn.Module):
    def __init__(self, n_in, n_out, stride, up_kernel_size, res_kernel_size=7, causal=False):
        super().__init__()
        self.up_conv = UpsampleLayer(n_in, n_out, kernel_size=up_kernel_size, stride=stride, causal=causal, activation=None)
        self.convs = nn.ModuleList([ResidualUnit(n_out, n_out, dilation=d, res_kernel_size=res_kernel_size, causal=causal)
                                    for d in (1, 3, 5, 7, 9)])

    def prepare(self):
        self.up_conv.prepare()
        for c in self.convs:
            c.prepare()

    def run(self, x):
        x = self.up_conv.run(x)
        for c in self.convs:
            x = c.run(x)
        return x


class _PlainConv(nn.Module):
    """Holder so that `weight_norm(Conv1d(...))` entries of encoder / decoder can be prepared and run uniformly."""


class ScalarModel(nn.Module):
    def __init__(self, num_bands, sample_rate, causal, num_samples, downsample_factors, downsample_kernel_sizes,
                 upsample_factors, upsample_kernel_sizes, latent_hidden_dim, default_kernel_size, delay_kernel_size,
                 init_channel, res_kernel_size):
        super().__init__()
        enc, dec = [], []
        enc.append(weight_norm(Conv1d(num_bands, init_channel, kernel_size=default_kernel_size, causal=causal)))
        if num_samples > 1:
            enc.append(PreProcessor(init_channel, init_channel, num_samples, kernel_size=default_kernel_size, causal=causal))
        for i, f in enumerate(downsample_factors):
            enc.append(ResEncoderBlock(init_channel * int(np.power(2, i)), init_channel * int(np.power(2, i + 1)), f,
                                       downsample_kernel_sizes[i], res_kernel_size, causal=causal))
        enc.append(weight_norm(Conv1d(init_channel * int(np.power(2, len(downsample_factors))), latent_hidden_dim,
                                      kernel_size=default_kernel_size, causal=causal)))
        dec.append(weight_norm(Conv1d(latent_hidden_dim, init_channel * int(np.power(2, len(upsample_factors))),
                                      kernel_size=delay_kernel_size)))
        for i, f in enumerate(upsample_factors):
            dec.append(ResDecoderBlock(init_channel * int(np.power(2, len(upsample_factors) - i)),
                                       init_channel * int(np.power(2, len(upsample_factors) - i - 1)), f,
                                       upsample_kernel_sizes[i], res_kernel_size, causal=causal))
        if num_samples > 1:
            dec.append(PostProcessor(init_channel, init_channel, num_samples, kernel_size=default_kernel_size, causal=causal))
        dec.append(weight_norm(Conv1d(init_channel, num_bands, kernel_size=default_kernel_size, causal=causal)))
        self.encoder = nn.ModuleList(enc)
        self.decoder = nn.ModuleList(dec)
        self._ready = False

    def prepare(self, fast_decode=True):
        """Fold weight-norm and pack every filter (call after load_state_dict / .to(device)).
        fast_decode: the decoder's convolutions run the bf16 x 3 form of ua2_conv1d (16-bit split operands on the bf16 MFMA,
        fp32 accumulation: end-to-end error ~1e-5 relative, well inside the 1e-4 RMS waveform bound, 3x faster than the
        exact-fp32 matrix pipe); the encoder always runs the exact form — its latents feed integer decisions (round(9x),
        RVQ) and are compared at 1e-5."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("uniaudio2_amd runs on a ROCm device only (no CPU fallback); move the model to cuda")
        self._enc_ops, self._dec_ops = [], []
        _ConvOp.default_fast = False
        for i, layer in enumerate(self.encoder):
            if isinstance(layer, nn.Conv1d):
                last = i == len(self.encoder) - 1
                self._enc_ops.append(_ConvOp(layer, ACT_TANH if last else ACT_NONE))            # tanh: scalar24k.py:397
            else:
                layer.prepare(); self._enc_ops.append(layer.run)
        _ConvOp.default_fast = bool(fast_decode)
        try:
            for i, layer in enumerate(self.decoder):
                if isinstance(layer, nn.Conv1d):
                    self._dec_ops.append(_ConvOp(layer, pre_act=ACT_ROUND9 if i == 0 else ACT_NONE))  # round9: :404
                else:
                    layer.prepare(); self._dec_ops.append(layer.run)
        finally:
            _ConvOp.default_fast = False
        self._dec_tc = bool(fast_decode) and self._decoder_tc_ok()
        self._graphs, self._graph_pool = {}, None
        self._ready = True
        return self

    def _decoder_tc_ok(self):
        """Every decoder conv behind the first one can run on split planes (channel counts in whole groups of 32, PReLU-only
        epilogues); the last conv writes the fp32 waveform."""
        convs = []
        n_dec = len(list(self.decoder))
        for i, layer in enumerate(list(self.decoder)[1:], start=1):
            if isinstance(layer, nn.Conv1d):
                if i < n_dec - 1:                       # a plain conv in the middle of the stack: it runs on planes too, same requirements
                    convs.append(self._dec_ops[i])
                continue
            if isinstance(layer, ResDecoderBlock):
                convs.append(layer.up_conv._op)
                for u in layer.convs:
                    convs += [u._op1, u._op2]
            elif isinstance(layer, PostProcessor):
                convs.append(layer._op)
            else:
                return False
        last = self._dec_ops[-1]
        return isinstance(last, _ConvOp) and last.tc_ok() and all(c.tc_ok() and c.cout % 32 == 0 for c in convs)

    @torch.inference_mode()
    def encode(self, x):
        """x (B, num_bands, N) fp32 -> latent (B, latent_hidden_dim, N / hop), tanh-squashed, not rounded (:392-401)."""
        if not self._ready:
            self.prepare()
        x = x.float().contiguous()
        for op in self._enc_ops:
            x = op(x)
        return x

    def _decode_impl(self, x):
        ops_ = self._dec_ops
        x = ops_[0](x)                                                # latent (fp32 [C][T], snapped to the 1/9 grid) -> widest layer
        if self._dec_tc:
            # round 4: from here to the waveform the activations travel as hi / lo bf16 planes [B][T][C] (ops.TC): every layer's
            # window goes L2 -> LDS by LDS-DMA and every epilogue writes the next layer's matrix-pipe operand
            x = ops.tc_pack(x)
            for op in ops_[1:-1]:
                x = op(x)
            return ops_[-1].run_tc(x, out_f32=True)
        for op in ops_[1:]:
            x = op(x)
        return x

    @torch.inference_mode()
    def decode(self, x, use_graph=None):
        """latent -> wav; the latent is snapped to the 1/9 grid first (:403-407).
        use_graph (default: on for the split-plane path): the ~44 launches of a decode are captured once per (input shape, device)
        into a HIP graph — an LRU of 12 shapes on one shared memory pool — and replayed — stage 2 decodes window after window of one shape (reason_tokenizer.py:277-290), and issued one by
        one from Python the chain is ~0.7 ms of host time against ~1.1 ms of kernels: the next kernel speed-up would have been
        host-bound."""
        if not self._ready:
            self.prepare()
        x = x.float().contiguous()
        if use_graph is None:
            use_graph = self._dec_tc and os.environ.get("UA2_CODEC_NO_GRAPH") is None
        if not use_graph or torch.cuda.is_current_stream_capturing():
            return self._decode_impl(x)
        key = (tuple(x.shape), x.device.index)
        g = self._graphs.pop(key, None)                               # re-inserted below: the dict's order is the LRU order
        if g is None:
            while len(self._graphs) >= 12:                            # least recently used shape goes; all graphs share one memory pool
                self._graphs.pop(next(iter(self._graphs)))
            x_in = torch.empty_like(x)
            x_in.copy_(x)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                             # warm-up outside capture: one-time kernel attributes, allocator pools
                self._decode_impl(x_in)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, pool=self._graph_pool):
                y = self._decode_impl(x_in)
            if self._graph_pool is None:
                self._graph_pool = graph.pool()
            g = (graph, x_in, y)
        self._graphs[key] = g
        graph, x_in, y = g
        x_in.copy_(x)
        graph.replay()
        return y.clone()

"""Streaming (chunk-by-chunk) execution of the conv stack, host side.

Mirror of the reference's llm_modules/streaming.py == tools/tokenizer/MimiCodec/model/modules/streaming.py:
StreamingModule :32-170 (a per-module state created by `streaming(batch)` / `streaming_forever`, propagated to every
streaming child, dropped on exit), RawStreamingConv1d :205-244 (keeps the input tail that the next frames still need),
RawStreamingConvTranspose1d :255-303 (keeps the output tail that the next inputs still add to).  The state machines
are host logic on device tensors (slices, concatenation, one overlap add); every convolution is ua2_conv1d.  The
reference's own self-test (streaming.py:306-358: chunked == whole sequence over a grid of kernel sizes, strides,
lengths and chunk sizes) is ported in tests/test_gpu_conv.py.
"""
import math
from contextlib import contextmanager

import torch
import torch.nn as nn

from ...... import ops
from ......_lib import ACT_NONE


class StreamingModule(nn.Module):
    """streaming.py:32-170, the part inference uses."""

    def __init__(self):
        super().__init__()
        self._streaming_state = None
        self._streaming_propagate = True

    @property
    def is_streaming(self):
        return self._streaming_state is not None

    def set_streaming_propagate(self, streaming_propagate: bool):
        self._streaming_propagate = streaming_propagate

    def _apply_named_streaming(self, fn):
        def handle(prefix, module, recurse=True):
            propagate = True
            if isinstance(module, StreamingModule):
                if module._streaming_propagate:
                    fn(prefix, module)
                else:
                    propagate = False
            if recurse and propagate:
                for name, child in module.named_children():
                    handle(prefix + "." + name, child)

        handle("", self, recurse=False)
        for name, child in self.named_children():
            handle(name, child)

    def _init_streaming_state(self, batch_size: int):
        return {}

    def _start_streaming(self, batch_size: int):
        def start(name, module):
            module._streaming_state = module._init_streaming_state(batch_size)
        self._apply_named_streaming(start)

    def _stop_streaming(self):
        def stop(name, module):
            module._streaming_state = None
        self._apply_named_streaming(stop)

    def streaming_forever(self, batch_size: int):
        self._start_streaming(batch_size)

    @contextmanager
    def streaming(self, batch_size: int):
        self._start_streaming(batch_size)
        try:
            yield
        finally:
            self._stop_streaming()

    def reset_streaming(self):
        def reset(name, module):
            if module._streaming_state is None:
                raise ValueError(f"Trying to reset streaming, but {name} wasn't streaming.")
            module._streaming_state = module._init_streaming_state(0)
        self._apply_named_streaming(reset)


class RawStreamingConv1d(nn.Conv1d, StreamingModule):
    """nn.Conv1d without padding; in streaming mode only complete frames are produced and the unconsumed input tail is
    kept for the next call (streaming.py:205-244)."""

    def __init__(self, *args, **kwargs):
        nn.Conv1d.__init__(self, *args, **kwargs)
        self._streaming_state = None
        self._streaming_propagate = True
        assert self.padding[0] == 0, "Padding should be handled outside."
        assert self.stride[0] <= self.kernel_size[0], "stride must be less than kernel_size."
        if self.groups not in (1, self.in_channels) or (self.groups != 1 and self.in_channels != self.out_channels):
            raise NotImplementedError("groups must be 1 or channel-wise (in == out == groups)")
        self._packed = None

    def _init_streaming_state(self, batch_size: int):
        return {"previous": None}

    def _weights(self):
        if self._packed is None:
            if self.groups == 1:
                w, _ = ops.pack_conv_weight(self.weight.detach().float())
            else:
                w = self.weight.detach().float().reshape(self.out_channels, -1).contiguous()
            self._packed = (w, self.bias.detach().float().contiguous() if self.bias is not None else None)
        return self._packed

    def _conv(self, x, pad_left, tout, pre_act=ACT_NONE, residual=None):
        w, b = self._weights()
        k, s, d = self.kernel_size[0], self.stride[0], self.dilation[0]
        x = x.float().contiguous()
        if self.groups != 1:
            assert pre_act == ACT_NONE and residual is None
            return ops.dwconv1d(x, w, stride=s, dilation=d, pad_left=pad_left, Tout=tout, bias=b)
        return ops.conv1d(x, w, k, self.out_channels, stride=s, dilation=d, pad_left=pad_left, Tout=tout, bias=b, pre_act=pre_act,
                          residual=residual)

    def run(self, x, pad_left=0, tout=None, pre_act=ACT_NONE, residual=None):
        """Whole-sequence mode: conv of x with `pad_left` implicit zeros and `tout` outputs.  Streaming mode (pad_left must
        be 0): prepend the kept tail, emit the complete frames, keep the new tail."""
        s = self.stride[0]
        kernel = (self.kernel_size[0] - 1) * self.dilation[0] + 1
        state = self._streaming_state
        if state is None:
            if tout is None:
                tout = (x.shape[-1] + pad_left - kernel) // s + 1
            return self._conv(x, pad_left, tout, pre_act, residual)
        assert pad_left == 0
        if state["previous"] is not None:
            x = torch.cat([state["previous"], x], dim=-1)
        B, _, T = x.shape
        num_frames = max(0, int(math.floor((T - kernel) / s) + 1))
        state["previous"] = x[..., num_frames * s:]
        if num_frames == 0:
            return torch.empty(B, self.out_channels, 0, device=x.device, dtype=torch.float32)
        return self._conv(x[..., :(num_frames - 1) * s + kernel], 0, num_frames, pre_act, residual)

    def forward(self, input):
        return self.run(input)


class RawStreamingConvTranspose1d(nn.ConvTranspose1d, StreamingModule):
    """nn.ConvTranspose1d without padding; in streaming mode the last kernel - stride outputs are held back until the
    next inputs have added their share (streaming.py:255-303)."""

    def __init__(self, *args, **kwargs):
        nn.ConvTranspose1d.__init__(self, *args, **kwargs)
        self._streaming_state = None
        self._streaming_propagate = True
        assert self.padding[0] == 0, "Padding should be handled outside."
        assert self.dilation[0] == 1, "No dilation for now"
        assert self.stride[0] <= self.kernel_size[0], "stride must be less than kernel_size."
        assert self.output_padding[0] == 0, "Output padding not supported."
        if self.groups not in (1, self.in_channels) or (self.groups != 1 and self.in_channels != self.out_channels):
            raise NotImplementedError("groups must be 1 or channel-wise (in == out == groups)")
        self._packed = None

    def _init_streaming_state(self, batch_size: int):
        return {"partial": None}

    def _weights(self):
        if self._packed is None:
            if self.groups == 1:
                w, m = ops.pack_convtr_weight(self.weight.detach().float(), self.stride[0])
            else:
                w, m = self.weight.detach().float().reshape(self.in_channels, -1).contiguous(), None
            self._packed = (w, m, self.bias.detach().float().contiguous() if self.bias is not None else None)
        return self._packed

    def _convtr(self, x, trim_left, tout, bias, pre_act=ACT_NONE):
        w, m, _ = self._weights()
        x = x.float().contiguous()
        if self.groups != 1:
            assert pre_act == ACT_NONE
            return ops.dwconv1d(x, w, stride=self.stride[0], pad_left=trim_left, Tout=tout, bias=bias, transposed=True)
        return ops.conv1d(x, w, m, self.out_channels, pad_left=m - 1, Tout=tout, bias=bias, pre_act=pre_act,
                          out_phases=self.stride[0], out_trim_left=trim_left)

    def run(self, x, trim_left=0, trim_right=0, pre_act=ACT_NONE):
        """Whole-sequence mode: the (T - 1) * stride + kernel outputs minus the trims.  Streaming mode (no trims): overlap-add
        with the held-back tail, emit everything that is final, hold back the new tail."""
        B, _, T = x.shape
        s, k = self.stride[0], self.kernel_size[0]
        bias = self._weights()[2]
        state = self._streaming_state
        if state is None:
            return self._convtr(x, trim_left, (T - 1) * s + k - trim_left - trim_right, bias, pre_act)
        assert trim_left == 0 and trim_right == 0
        if T == 0:
            return torch.empty(B, self.out_channels, 0, device=x.device, dtype=torch.float32)
        out = self._convtr(x, 0, (T - 1) * s + k, None, pre_act)           # bias added once, on what is emitted
        partial = state["partial"]
        if partial is not None:
            out[..., :partial.shape[-1]] += partial
        invalid = k - s
        OT = out.shape[-1]
        state["partial"] = out[..., OT - invalid:].clone()
        out = out[..., :OT - invalid]
        if bias is not None:
            out = out + bias[:, None]
        return out.contiguous()

    def forward(self, x):
        return self.run(x)

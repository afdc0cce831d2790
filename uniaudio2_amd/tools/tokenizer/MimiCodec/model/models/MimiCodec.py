"""MimiCodec assembly (inference: encode / decode), host side.

Mirror of tools/tokenizer/MimiCodec/model/models/MimiCodec.py:25-117 — same constructor arguments, same sub-module
names (encoder, decoder, downsample, upsample, semantic_mapping_layer, encoder_transformer, decoder_transformer,
quantizer) and therefore the same checkpoint keys:
  encode :92-100  wav (B,1,T) -> SEANetEncoder -> ProjectedTransformer -> ConvDownsample1d -> SplitRVQ codes (B,K,T')
  decode :102-109 codes -> SplitRVQ.decode -> ConvTrUpsample1d (channel-wise) -> ProjectedTransformer -> SEANetDecoder
forward :73-90 (training: random quantisation mask, semantic distillation) is out of scope.
`with model.streaming(batch):` runs encode / decode chunk by chunk (causal convs keep their tails, the transformers
their position and cache), equal to the whole-sequence result.
Every arithmetic op runs in libua2hip.so (ua2_conv1d / ua2_dwconv1d / ua2_linear / ua2_attn / ua2_rvq_*).
"""
import json

import torch
import torch.nn as nn

from ..modules import transformer as Stransformer
from ..modules.resample import ConvDownsample1d, ConvTrUpsample1d
from ..modules.seanet import SEANetDecoder, SEANetEncoder
from ..modules.streaming import StreamingModule
from ..quantization.vq import SplitResidualVectorQuantizer


class Semantic_linear_pool(nn.Module):
    """MimiCodec.py:15-23; only forward() (training) calls it — kept for checkpoint compatibility."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.ln_layer = nn.Linear(in_channels, out_channels)
        self.pl = nn.AvgPool1d(kernel_size=8, stride=4)


class MimiCodec(StreamingModule):
    def __init__(self, sample_rate=24000, n_filters=64, encoder_rates=[4, 5, 6, 8], compress=2, causal=True, latent_dim=512,
                 codebook_size=4096, codebook_dim=32, rvq_layers=8, num_heads=8, num_layers=8, layer_scale=0.01, context=250,
                 dim_feedforward=2048, semantic_feature_dim=1024, target_frame_rate=12.5):
        super().__init__()
        self.sample_rate = sample_rate
        seanet = dict(channels=1, dimension=latent_dim, causal=causal, n_filters=n_filters, n_residual_layers=1, activation="ELU",
                      compress=compress, dilation_base=2, disable_norm_outer_blocks=0, kernel_size=7, residual_kernel_size=3,
                      last_kernel_size=3, norm="none", pad_mode="constant", ratios=encoder_rates, true_skip=True)
        quant = dict(dimension=codebook_dim, n_q=rvq_layers, bins=codebook_size, input_dimension=latent_dim,
                     output_dimension=latent_dim)
        tr = dict(d_model=latent_dim, num_heads=num_heads, num_layers=num_layers, causal=causal, layer_scale=layer_scale,
                  context=context, conv_layout=True, max_period=10000, gating="none", norm="layer_norm",
                  positional_embedding="rope", dim_feedforward=2048, input_dimension=latent_dim, output_dimensions=[latent_dim])
        self.encoder = SEANetEncoder(**seanet)
        self.decoder = SEANetDecoder(**seanet)
        self.hop_length = encoder_rates[0] * encoder_rates[1] * encoder_rates[2] * encoder_rates[3]
        self.encoder_frame_rate = 24000 / self.hop_length
        self.target_frame_rate = target_frame_rate
        self.learnt = True
        stride = int(self.encoder_frame_rate / self.target_frame_rate)
        self.downsample = ConvDownsample1d(stride, dimension=latent_dim, learnt=self.learnt, causal=causal)
        self.upsample = ConvTrUpsample1d(stride, dimension=latent_dim, learnt=self.learnt, causal=causal, channel_wise=True)
        self.semantic_mapping_layer = Semantic_linear_pool(semantic_feature_dim, latent_dim)
        self.encoder_transformer = Stransformer.ProjectedTransformer(**tr)
        self.decoder_transformer = Stransformer.ProjectedTransformer(**tr)
        self.quantizer = SplitResidualVectorQuantizer(**quant)

    def forward(self, audio_data, semantic_features):
        raise NotImplementedError("MimiCodec.forward is the training pass (random quantisation mask, semantic distillation); "
                                  "use encode / decode")

    @torch.inference_mode()
    def encode(self, audio_data: torch.Tensor) -> torch.Tensor:
        z = self.encoder(audio_data)
        z = self.encoder_transformer(z)[0]
        z = self.downsample(z)
        return self.quantizer.encode(z)

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        z_q = self.quantizer.decode(codes)
        z_q = self.upsample(z_q)
        z_q = self.decoder_transformer(z_q)[0]
        return self.decoder(z_q)

    @classmethod
    def from_config(cls, config_path):
        with open(config_path, "r") as f:
            return cls(**json.load(f))

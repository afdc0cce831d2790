"""Toy (small) shapes for golden-vector generation and the parity tests.

Same structure as the named configs (Llama-3.2-3B backbone, 3-layer
understanding expert, 2-layer generation expert, 4-layer 2048-d local decoder), shrunk so
the reference finishes in seconds on CPU and the fixtures stay small.  Keys are the
reference registry names (llm_models/config.py:804-899).
"""

ROPE_ADJ = dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_seq_len=8192)

_COMMON = dict(
    block_size=2048, vocab_size=500, padded_vocab_size=512,
    rotary_percentage=1.0, parallel_residual=False, bias=False,
    norm_class_name="RMSNorm", mlp_class_name="LLaMAMLP",
    rope_base=500000, rope_adjustments=ROPE_ADJ,
)

TOY_LM = {
    # backbone: 256-d, 4 heads x 64, 2 kv groups
    "Llama-3.2-3B": dict(_COMMON, n_layer=3, n_embd=256, n_head=4, n_query_groups=2, intermediate_size=512),
    "Llama-3.2-Understanding": dict(_COMMON, n_layer=2, n_embd=256, n_head=4, n_query_groups=2, intermediate_size=512),
    "Llama-3.2-Generation": dict(_COMMON, n_layer=2, n_embd=256, n_head=4, n_query_groups=2, intermediate_size=512),
    # local decoder: 128-d, 4 heads x 32, 2 kv groups
    "Llama-3.2-300M": dict(_COMMON, n_layer=2, n_embd=128, n_head=4, n_query_groups=2, intermediate_size=256),
}

TOY_MODEL_ARGS = dict(
    llm_name="Llama-3.2-3B",
    decoder_name="Llama-3.2-300M",
    llm_pretrained_model="",
    audio_embeddings_path="",
    audio_understanding_expert_path="",
    audio_semantic_vocab_size=70,
    audio_reason_vocab_size=40,     # V_a = 110: deliberately not a multiple of 16
    audio_num_codebooks=8,
)

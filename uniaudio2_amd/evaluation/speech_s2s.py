"""Mirror of the reference's evaluation/speech_s2s.py `Generator`: the prompt is the condition sequence of the source
utterance's reason / semantic tokens (get_condition_seq :233-281, identical to audio_understanding.py's), the loop is
the shared audio loop (:283-347; never classifier-free guided) in _generator.py."""
import torch

from ._generator import GeneratorBase

# llm_utils/task_definition.py:217-222 (speech_s2s_format): the last two keys are the targets
S2S_KEYS = ["reason_seq_1", "semantic_seq_1", "reason_seq_2", "semantic_seq_2"]
S2S_TYPES = ["audio", "audio", "audio", "audio"]


class Generator(GeneratorBase):
    @torch.inference_mode()
    def generate_audio(self, task_prompt, task_name, d=None, keys=None, types=None, temperature: float = 0.9,
                       topk: int = 200, cfg_scale=1.0):
        """d: {"reason_seq_1": (8, T_r), "semantic_seq_1": (8, T_s), ...}; keys / types: the condition part of the
        task format.  Returns (reason (8, T_r'), semantic (8, T_s')) int32."""
        tokens, mask = self.get_condition_seq(d, keys, types, task_prompt)
        was_cfg, self.is_cfg = self.is_cfg, False            # the reference's s2s loop has batch size 1 (:301-304)
        try:
            return self._generate_audio_tokens(tokens, mask, topk=topk, temperature=temperature)
        finally:
            self.is_cfg = was_cfg

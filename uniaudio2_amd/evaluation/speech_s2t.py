"""Mirror of the reference's evaluation/speech_s2t.py `Generator` (task "speech_s2t", multi_task_inference.py:213-218):
audio_understanding's condition sequence with two differences the reference's file has — audio codes may arrive as (8, T) or
(T, 8) (:290-294) and a prompt of 1500 frames or more is refused with (-1, -1) (:351-352) — and a `generate_answer` that
returns (text, 1) (:381: the text and `curr_tokens.shape[1]` of the last fed-back frame, which is 1)."""
import torch

from ._generator import GeneratorBase

MAX_PROMPT_FRAMES = 1500          # speech_s2t.py:351


class Generator(GeneratorBase):
    def get_condition_seq(self, d, keys, types, task_prompt_data):
        """speech_s2t.py:274-326: as audio_understanding's, audio entries in either orientation."""
        fixed = dict(d)
        for key, tp in zip(keys, types):
            if tp != "text":
                x = d[key].long()
                if x.dim() == 2 and x.shape[0] == 8 and x.shape[1] != 8:
                    x = x.transpose(0, 1)             # :291-293, verbatim: an (8, T != 8) input becomes (T, 8); an (8, 8) one is taken as (T, 8)
                fixed[key] = x.transpose(0, 1)        # the shared builder takes (8, T) and transposes back
        return super().get_condition_seq(fixed, keys, types, task_prompt_data)

    @torch.inference_mode()
    def generate_answer(self, task_prompt, task_name, d=None, keys=None, types=None, temperature: float = 0.9,
                        topk: int = 200, cfg_scale=1.0):
        tokens, mask = self.get_condition_seq(d, keys, types, task_prompt)
        if tokens.shape[0] >= MAX_PROMPT_FRAMES:
            return -1, -1
        return self._generate_text(tokens, mask, topk=topk, temperature=temperature), 1

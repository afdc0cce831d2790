"""Mirror of the reference's evaluation/audio_understanding.py `Generator` (task "audio_understanding",
multi_task_inference.py:207-212): the condition sequence of `get_condition_seq` (:233-282: task prompt | question text with its
special tokens | reason bos..eos | semantic bos..eos + offset, in the order of `keys`) and the text loop `generate_answer`
(:284-339) — both in _generator.py."""
import torch

from ._generator import GeneratorBase


class Generator(GeneratorBase):
    @torch.inference_mode()
    def generate_answer(self, task_prompt, task_name, d=None, keys=None, types=None, temperature: float = 0.9,
                        topk: int = 200, cfg_scale=1.0) -> str:
        """d[key]: text ids (text) or (8, T) audio codes (audio); returns the answer text (audio_understanding.py:284-339)."""
        tokens, mask = self.get_condition_seq(d, keys, types, task_prompt)
        return self._generate_text(tokens, mask, topk=topk, temperature=temperature)

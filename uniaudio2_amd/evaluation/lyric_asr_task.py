"""Mirror of the reference's evaluation/lyric_asr_task.py `Generator` (task "lyric_recognition",
multi_task_inference.py:195-200): the ASR prompt layout under another name (`prepare_lyric_asr_task` :175-200 is
`prepare_asr_task` line for line: task prompt | reason bos..eos | semantic bos..eos + offset) and the greedy text loop
(`generate_lyric_asr` :202-254) — both in _generator.py."""
import torch

from .asr_task import Generator as _AsrGenerator


class Generator(_AsrGenerator):
    def prepare_lyric_asr_task(self, task_prompt, this_reason_data, this_semantic_data):
        """lyric_asr_task.py:175-200."""
        return self.prepare_asr_task(task_prompt, this_reason_data, this_semantic_data)

    @torch.inference_mode()
    def generate_lyric_asr(self, task_prompt, task_name, text_token=None, semantic_token=None, reason_token=None,
                           temperature: float = 0.9, topk: int = 200, cfg_scale=1.0) -> str:
        """reason_token (T_r, 8), semantic_token (T_s, 8) long -> lyric text (lyric_asr_task.py:202-254)."""
        tokens, mask = self.prepare_lyric_asr_task(task_prompt, reason_token, semantic_token)
        return self._generate_text(tokens, mask, topk=topk, temperature=temperature)

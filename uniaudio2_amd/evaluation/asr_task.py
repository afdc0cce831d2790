"""Mirror of the reference's evaluation/asr_task.py `Generator` (greedy `generate_asr`).  Not mirrored: the beam
search (:438-, calls a method Model_stage3 lacks — SURVEY Appendix A.9) and the n-gram variant (:329-405: it prefills
with the training forward `self._model(...)`, :356, which never writes the KV cache, so it decodes without its prompt)."""
import torch

from ._generator import GeneratorBase


class Generator(GeneratorBase):
    @torch.inference_mode()
    def generate_asr(self, task_prompt, task_name, text_token=None, semantic_token=None, reason_token=None,
                     temperature: float = 0.9, topk: int = 200, cfg_scale=1.0) -> str:
        """reason_token (T_r, 8), semantic_token (T_s, 8) long -> transcription text (asr_task.py:630-688)."""
        tokens, mask = self.prepare_asr_task(task_prompt, reason_token, semantic_token)
        return self._generate_text(tokens, mask, topk=topk, temperature=temperature)

    # audio_music_caption_task.py uses the same prompt layout and loop under another name
    generate_audio_caption = generate_asr

    @torch.inference_mode()
    def generate_answer(self, task_prompt, task_name, d=None, keys=None, types=None, temperature: float = 0.9,
                        topk: int = 200, cfg_scale=1.0) -> str:
        """audio_understanding.py:284-339."""
        tokens, mask = self.get_condition_seq(d, keys, types, task_prompt)
        return self._generate_text(tokens, mask, topk=topk, temperature=temperature)

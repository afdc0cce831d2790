"""Mirror of the reference's evaluation/audio_music_caption_task.py `Generator` (tasks "audio_caption" / "music_caption",
multi_task_inference.py:201-206): `prepare_audio_caption_task` (:175-200, the ASR prompt layout) and the greedy text loop
`generate_audio_caption` (:202-257) — both in _generator.py."""
import torch

from .asr_task import Generator as _AsrGenerator


class Generator(_AsrGenerator):
    def prepare_audio_caption_task(self, task_prompt, this_reason_data, this_semantic_data):
        """audio_music_caption_task.py:175-200."""
        return self.prepare_asr_task(task_prompt, this_reason_data, this_semantic_data)

    @torch.inference_mode()
    def generate_audio_caption(self, task_prompt, task_name, text_token=None, semantic_token=None, reason_token=None,
                               temperature: float = 0.9, topk: int = 200, cfg_scale=1.0) -> str:
        """reason_token (T_r, 8), semantic_token (T_s, 8) long -> caption text (audio_music_caption_task.py:202-257)."""
        tokens, mask = self.prepare_audio_caption_task(task_prompt, reason_token, semantic_token)
        return self._generate_text(tokens, mask, topk=topk, temperature=temperature)

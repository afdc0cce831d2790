"""Mirror of the reference's evaluation/audiogen_task.py `Generator` (prompt key 'caption_seq'); the shared
implementation lives in _generator.py."""
import torch

from ._generator import GeneratorBase


class Generator(GeneratorBase):
    def prepare_AudioGen_task(self, task_prompt, text_seq):
        return self._prepare_text_conditioned(task_prompt, text_seq, 'caption_seq')

    def prepare_AudioGen_task_for_cfg(self, task_prompt, text_seq):
        return self._prepare_text_conditioned(task_prompt, text_seq, 'caption_seq', cfg=True)

    @torch.inference_mode()
    def generate_audio(self, task_prompt, task_name, text_token=None, semantic_token=None, reason_token=None,
            temperature: float = 0.9, topk: int = 200, cfg_scale=1.0):
        """Returns (reason (8, T_r), semantic (8, T_s)) int32.  `cfg_scale` is accepted and ignored, as in the
        reference (it is never forwarded to generate_frame; with is_cfg the unconditional row is computed
        and discarded)."""
        tokens, mask = self.prepare_AudioGen_task(task_prompt, text_token)
        cfg_t = cfg_m = None
        if self.is_cfg:
            cfg_t, cfg_m = self.prepare_AudioGen_task_for_cfg(task_prompt, text_token)
        return self._generate_audio_tokens(tokens, mask, cfg_t, cfg_m, topk=topk, temperature=temperature)

"""Mirror of the reference's evaluation/tts_task.py `Generator` (prompt key 'transcription_seq'); the shared
implementation lives in _generator.py."""
import torch

from ._generator import GeneratorBase


class Generator(GeneratorBase):
    def prepare_tts_task(self, task_prompt, text_seq):
        return self._prepare_text_conditioned(task_prompt, text_seq, 'transcription_seq')

    def prepare_tts_task_for_cfg(self, task_prompt, text_seq):
        return self._prepare_text_conditioned(task_prompt, text_seq, 'transcription_seq', cfg=True)

    @torch.inference_mode()
    def generate_tts(self, task_prompt, task_name, text_token=None, semantic_token=None, reason_token=None,
            temperature: float = 0.9, topk: int = 200, cfg_scale=1.0):
        """Returns (reason (8, T_r), semantic (8, T_s)) int32.  `cfg_scale` is accepted and ignored, as in the
        reference (it is never forwarded to generate_frame; with is_cfg the unconditional row is computed
        and discarded)."""
        tokens, mask = self.prepare_tts_task(task_prompt, text_token)
        cfg_t = cfg_m = None
        if self.is_cfg:
            cfg_t, cfg_m = self.prepare_tts_task_for_cfg(task_prompt, text_token)
        return self._generate_audio_tokens(tokens, mask, cfg_t, cfg_m, topk=topk, temperature=temperature)

    @torch.inference_mode()
    def generate_tts_batch(self, task_prompt, task_name, text_tokens, temperature: float = 0.9, topk: int = 200, cfg_scale=1.0):
        """Not in the reference: `generate_tts` for a list of texts decoded together on one GPU (continuous batching);
        returns [(reason (8, T_r), semantic (8, T_s)), ...] in input order, each equal to its own generate_tts result
        under greedy decoding."""
        prompts = [self.prepare_tts_task(task_prompt, t) for t in text_tokens]
        cfg = [self.prepare_tts_task_for_cfg(task_prompt, t) for t in text_tokens] if self.is_cfg else None
        return self._generate_audio_tokens_batch(prompts, topk=topk, temperature=temperature, cfg_prompts=cfg)

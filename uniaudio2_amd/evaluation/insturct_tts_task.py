"""Mirror of the reference's evaluation/insturct_tts_task.py `Generator` (file name as in the reference): prompt =
task prompt + <caption>..</caption> + <transcription>..</transcription> text frames (:201-219), the CFG twin replaces
every id by text_pad (:171-198); the loop is the shared audio loop (:221-296) in _generator.py."""
import torch

from ._generator import GeneratorBase


class Generator(GeneratorBase):
    def _prepare(self, task_prompt, caption_seq, text_seq, cfg=False):
        caption_seq = self.add_special_token('caption_seq', caption_seq)
        text_seq = self.add_special_token('transcription_seq', text_seq)
        if cfg:
            task_prompt = torch.ones_like(task_prompt) * self.text_pad_token
            caption_seq = torch.ones_like(caption_seq) * self.text_pad_token
            text_seq = torch.ones_like(text_seq) * self.text_pad_token
        blocks = [self._text_block(t) for t in (task_prompt, caption_seq, text_seq)]
        return torch.cat([b[0] for b in blocks], dim=0), torch.cat([b[1] for b in blocks], dim=0)

    def prepare_instruct_tts_task(self, task_prompt, caption_seq, text_seq):
        return self._prepare(task_prompt, caption_seq, text_seq)

    def prepare_instruct_tts_task_for_cfg(self, task_prompt, caption_seq, text_seq):
        return self._prepare(task_prompt, caption_seq, text_seq, cfg=True)

    @torch.inference_mode()
    def generate_instruct_tts(self, task_prompt, task_name, text_token=None, caption=None, semantic_token=None,
                              reason_token=None, temperature: float = 0.9, topk: int = 200, cfg_scale=1.0):
        """Returns (reason (8, T_r), semantic (8, T_s)) int32.  `cfg_scale` is accepted and ignored, as in the
        reference (never forwarded to generate_frame, :255-256)."""
        tokens, mask = self.prepare_instruct_tts_task(task_prompt, caption, text_token)
        cfg_t = cfg_m = None
        if self.is_cfg:
            cfg_t, cfg_m = self.prepare_instruct_tts_task_for_cfg(task_prompt, caption, text_token)
        return self._generate_audio_tokens(tokens, mask, cfg_t, cfg_m, topk=topk, temperature=temperature)

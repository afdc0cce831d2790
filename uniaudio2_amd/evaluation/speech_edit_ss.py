"""Mirror of the reference's evaluation/speech_edit_ss.py `Generator` (speech editing / source separation by tokens): the
condition sequence of `get_condition_seq` (:229-278, audio_understanding's builder) and the audio loop `generate_audio`
(:280-343, speech_s2s's: batch size 1, never guided) — both in _generator.py."""
from .speech_s2s import Generator as _S2sGenerator


class Generator(_S2sGenerator):
    pass

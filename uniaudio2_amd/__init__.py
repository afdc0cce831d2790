"""uniaudio2_amd — MI355X-native audio-token generation hot path of UniAudio 2.0.

Host side mirrors the reference's Python interface for this path (same module/class/method
names under llm_models/, evaluation/, tools/tokenizer/); the compute is hand-written gfx950
HIP behind the C ABI in include/ua2hip.h (libua2hip.so).  Importing the package loads the
shared library and fails loudly if it is missing: there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when libua2hip.so is absent)

__version__ = "0.1.0"

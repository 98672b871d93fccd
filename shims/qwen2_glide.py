"""Drop-in for the reference's ``longspec/test/qwen2_glide.py`` (``from qwen2_glide import Qwen2Glide``,
``inference_qwq.py:1``); see ``shims/llama_glide.py``."""
from longspec_amd.qwen2_glide import *            # noqa: F401,F403
from longspec_amd.qwen2_glide import Qwen2Glide  # noqa: F401

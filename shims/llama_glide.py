"""Drop-in for the reference's ``longspec/test/llama_glide.py``: same import line
(``from llama_glide import LlamaGlide``, ``inference_long-bench.py:1``), MI355X implementation.
Put this directory ahead of the reference's on ``sys.path`` -- or use ``python -m longspec_amd.harness <script>``,
which needs no path games (a script's own directory otherwise wins over PYTHONPATH)."""
from longspec_amd.llama_glide import *            # noqa: F401,F403
from longspec_amd.llama_glide import LlamaGlide, LlamaGlideDecoderLayer, GlideAttention  # noqa: F401

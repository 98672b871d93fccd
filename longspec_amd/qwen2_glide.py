"""Host mirror of ``longspec/test/qwen2_glide.py`` (``Qwen2Glide``), on the HIP operator layer.

Same round as ``LlamaGlide`` (the two reference files differ on the decode path in exactly the points
listed here; everything else is shared code in ``longspec_amd.llama_glide``):

* the draft's self-attention cache is allocated ``q_len + max_len`` rows at prefill, without the
  Llama twin's extra 128 (qwen2_glide.py:225-226 vs llama_glide.py:218-219),
* ``vanilla_generate`` and ``tree_spec_generate`` stop on the ``eos_id`` ARGUMENT
  (qwen2_glide.py:580,949) while ``spec_generate`` keeps ``config.eos_token_id`` (:735),
* ``tree_spec_generate`` starts from a ZERO ``output_ids`` buffer (qwen2_glide.py:766), not an
  eos-filled one, so its stop test only fires on a generated eos.
"""
from __future__ import annotations

from .llama_glide import GlideAttention, LlamaGlide, LlamaGlideDecoderLayer
from .qwen2 import Qwen2ForCausalLM, Qwen2Model


class Qwen2GlideAttention(GlideAttention):
    """``GlideAttention`` of the Qwen2 twin (qwen2_glide.py:26-388)."""
    CACHE_PAD = 0


class Qwen2GlideDecoderLayer(LlamaGlideDecoderLayer):
    """``Qwen2GlideDecoderLayer`` (qwen2_glide.py:391-472)."""
    ATTENTION_CLS = Qwen2GlideAttention


class Qwen2Glide(LlamaGlide):
    """``Qwen2Glide(config, target_model_path, glide_path=None)`` (qwen2_glide.py:475-): same
    methods, arguments and return tuples as ``LlamaGlide``."""
    MODEL_CLS = Qwen2Model
    GLIDE_LAYER_CLS = Qwen2GlideDecoderLayer

    def _stop_id(self, eos_id, loop: str):
        if loop == "spec":
            return getattr(self.config, "eos_token_id", None)      # qwen2_glide.py:735
        return eos_id                                              # :580, :949

    def _tree_output_fill(self, eos_id):
        return 0                                                   # :766


__all__ = ["Qwen2Glide", "Qwen2GlideDecoderLayer", "Qwen2GlideAttention", "Qwen2ForCausalLM"]

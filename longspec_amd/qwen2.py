"""Host mirror of the reference's Qwen2 target (``longspec/test/qwen2.py``), on the HIP operator layer.

The decode algorithm is the Llama twin's (``longspec/test/llama.py``) line for line -- prefill
(:336-372), decoding (:442-491), tree_decoding (:493-531) and tree_part_fwd (:533-560) call the same
operators in the same order with the same fp16 rounding points -- so the classes below reuse
``longspec_amd.llama`` and state only what differs:

* q/k/v projections carry a bias unconditionally (qwen2.py:277-279; Llama reads ``attention_bias``),
* ``softmax_scale`` is the same hard-coded ``1/sqrt(128)`` (qwen2.py:292), head_dim 128,
* GQA group 7 (Qwen2.5-7B: 28 q heads / 4 kv heads) or 5 (QwQ-32B: 40/8) -- handled by the kernels'
  row packing (rows x group per KV head), not by the host.

Module and parameter names match the HF Qwen2 checkpoint layout (``model.layers.N.self_attn.q_proj.bias`` ...),
so ``checkpoint.load_target_checkpoint`` reads a Qwen2 directory unchanged.
"""
from __future__ import annotations

from .llama import (LlamaAttention, LlamaDecoderLayer, LlamaForCausalLM, LlamaMLP, LlamaModel, LlamaRMSNorm,
                    LlamaRotaryEmbedding)


class Qwen2RMSNorm(LlamaRMSNorm):
    """``Qwen2RMSNorm`` (qwen2.py:73-91) -- the same fp32-variance / dtype-weight norm (K8)."""


class Qwen2RotaryEmbedding(LlamaRotaryEmbedding):
    """``Qwen2RotaryEmbedding`` (qwen2.py:94-178): fp32 inv_freq x positions -> cos/sin in the model dtype (K9)."""


class Qwen2MLP(LlamaMLP):
    """``Qwen2MLP`` (qwen2.py:218-230): down(act(gate(x)) * up(x)), no biases."""


class Qwen2Attention(LlamaAttention):
    """``Qwen2Attention`` (qwen2.py:245-560)."""
    QKV_BIAS = True


class Qwen2DecoderLayer(LlamaDecoderLayer):
    """``Qwen2DecoderLayer`` (qwen2.py:616-668)."""
    ATTENTION_CLS = Qwen2Attention


class Qwen2Model(LlamaModel):
    """``Qwen2Model`` (qwen2.py:694-801)."""
    LAYER_CLS = Qwen2DecoderLayer


class Qwen2ForCausalLM(LlamaForCausalLM):
    """``Qwen2ForCausalLM`` (qwen2.py:804-): ``set_max_gen_len`` :822-824."""
    MODEL_CLS = Qwen2Model

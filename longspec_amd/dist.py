"""Sequence-sharded prefix KV over the GPUs of one node.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI; ``gloo`` in the
CPU tests).  Weights, the draft's own (constant-size) KV cache and all non-attention compute
are replicated: every rank runs the same round on the same tokens.  The long prefix KV of every
target layer is split by sequence:

    rank r < W-1 : rows [r*Ls, (r+1)*Ls)                  (fixed)
    rank W-1     : rows [(W-1)*Ls, ...)  + everything generated afterwards (the growing tail)

Per attention call each rank streams only its slice and produces one normalised partial
``(o fp32 [R,H,128], lse [H,R])`` = 1.22 MB at R=74, H=32.  The partials are exchanged with ONE
all-gather (each of the 7 xGMI links carries 1/7 of the traffic in a single step -- a ring
all-reduce would be 14 dependent per-link steps for this latency-bound message) and merged by
every rank in rank order, so the result is bit-identical on all ranks and independent of timing.
The reference has no counterpart (``device_map="auto"`` only, ``llama_glide.py:474``); the merge
is the N-way form of its 2-way ``o_p*sigmoid(lse_p-lse_t) + o_t*(1-sigmoid)`` (``llama.py:385-387,420``).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class KVShard:
    def __init__(self, rank: int, world: int, shard_rows: int, group=None):
        self.rank, self.world, self.Ls, self.group = rank, world, int(shard_rows), group
        self.start = rank * self.Ls
        self.is_tail = rank == world - 1
        self._send = {}
        self._recv = {}
        self._pass_len = {}
        self.prefill_ctx = None          # (first global row, local rows, prompt rows) while a sharded prefill runs

    # ---- lengths ---------------------------------------------------------------------------------
    def local_len(self, global_len: torch.Tensor) -> torch.Tensor:
        """Valid prefix rows this rank attends: clamp(global - start, 0, Ls) (unbounded on the tail rank)."""
        l = (global_len.to(torch.int32) - self.start).clamp_(min=0)
        return l if self.is_tail else l.clamp_(max=self.Ls)

    def begin_pass(self):
        """A model pass starts: the length tensors do not change until it ends, so every layer of the pass shares
        one ``local_len`` result (3 tiny kernels otherwise, per layer)."""
        self._pass_len = {}

    def pass_len(self, global_len: torch.Tensor) -> torch.Tensor:
        hit = self._pass_len.get(global_len.data_ptr())
        if hit is None or hit[0] is not global_len:
            hit = (global_len, self.local_len(global_len))       # holds the key tensor: its address cannot be reused
            self._pass_len[global_len.data_ptr()] = hit
        return hit[1]

    def local_hint(self, global_hint: Optional[int]) -> Optional[int]:
        if global_hint is None:
            return None
        h = max(int(global_hint) - self.start, 0)
        return h if self.is_tail else min(h, self.Ls)

    # ---- the one collective of the data path ---------------------------------------------------
    def buffers(self, n_floats: int, device):
        key = (n_floats, str(device))
        if key not in self._send:
            self._send[key] = torch.empty(n_floats, dtype=torch.float32, device=device)
            self._recv[key] = torch.empty((self.world, n_floats), dtype=torch.float32, device=device)
        return self._send[key], self._recv[key]

    def gather_rows(self, rows: torch.Tensor) -> torch.Tensor:
        """All-gather of one equally-shaped tensor per rank along a new leading dimension (sharded prefill: every rank's
        K or V rows of a layer)."""
        rows = rows.contiguous()
        out = rows.new_empty((self.world,) + tuple(rows.shape))
        dist.all_gather_into_tensor(out.view(-1), rows.view(-1), group=self.group)
        return out

    def broadcast_from_tail(self, t: torch.Tensor) -> torch.Tensor:
        dist.broadcast(t, src=self.world - 1 if self.group is None else dist.get_global_rank(self.group, self.world - 1),
                       group=self.group)
        return t

    def exchange(self, send: torch.Tensor, recv: torch.Tensor) -> torch.Tensor:
        dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
        return recv

    def attend(self, call) -> torch.Tensor:
        """partial -> all-gather -> merge, for one ``ShardedAttnCall``-like object."""
        send, recv = self.buffers(call.record_floats, call.device)
        call.partial(send)
        return call.finish(self.exchange(send, recv))


def shard_model_kv(model, shard: KVShard, prompt_len: int, draft_too: bool = False):
    """After a (replicated) prefill: keep only this rank's slice of every target layer's KV and
    switch the attention modules to the sharded path.  ``prompt_len`` = rows valid after prefill."""
    for layer in model.model.layers:
        attn = layer.self_attn
        K, V = attn.K_Cache, attn.V_Cache
        tail_rows = K.shape[1] - prompt_len
        lo = shard.start
        hi = prompt_len if shard.is_tail else min(prompt_len, lo + shard.Ls)
        n = max(hi - lo, 0)
        rows = (n if shard.is_tail else shard.Ls) + tail_rows
        k2 = K.new_zeros((K.shape[0], rows, K.shape[2], K.shape[3]))
        v2 = V.new_zeros((V.shape[0], rows, V.shape[2], V.shape[3]))
        k2[:, :n] = K[:, lo:hi]
        v2[:, :n] = V[:, lo:hi]
        attn.K_Cache, attn.V_Cache = k2, v2
        attn.shard = shard
    model.glide.cross_attn.shard = shard

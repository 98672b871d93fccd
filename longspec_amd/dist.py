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
On GPUs the all-gather is ``PeerExchange``: every rank stores its record into IPC-mapped mailboxes in
its peers' HBM and raises a flag (``csrc/xgmi.hip``) -- two kernel launches, so that the decode round
stays capturable in a HIP graph; the ``torch.distributed`` collective remains for the CPU tests, for
records larger than a mailbox slot (prefill) and as the fallback when the self-check fails.
The reference has no counterpart (``device_map="auto"`` only, ``llama_glide.py:474``); the merge
is the N-way form of its 2-way ``o_p*sigmoid(lse_p-lse_t) + o_t*(1-sigmoid)`` (``llama.py:385-387,420``).
"""
from __future__ import annotations

from typing import Optional

import ctypes as C
import os
import sys

import torch
import torch.distributed as dist


class PeerExchange:
    """All-gather of one fp32 record per rank through IPC-mapped mailboxes (``include/longspec_hip.h``: ``ls_xchg_*``).
    Construction is collective over ``group``: the 64-byte IPC handles travel by ``torch.distributed``."""

    def __init__(self, rank: int, world: int, cap_floats: int, device, group=None):
        from . import _C
        self._C, self.lib = _C, _C.load()
        self.rank, self.world, self.group, self.device = rank, world, group, device
        self.cap = (int(cap_floats) + 3) // 4 * 4
        self._x = C.c_void_p()
        on = device if dist.get_backend(group) == "nccl" else "cpu"
        self._on = on

        # Construction is a SEQUENCE of collectives with local steps in between that can fail on one rank only (out of memory
        # in ls_xchg_create, hipIpcOpenMemHandle in ls_xchg_connect ...).  A rank that raised out of the sequence on its own
        # would leave its peers blocked in the next collective, so every local step reports a status, the ranks agree on it
        # (all-reduce MIN) BEFORE the next collective, and on any rank's failure all of them close and raise together.
        def step(what, fn):
            err = None
            try:
                fn()
            except Exception as e:                  # noqa: BLE001
                err = e
            self._agree(err is None, what, err)

        def create():
            with torch.cuda.device(device):
                _C.check(self.lib.ls_xchg_create(rank, world, self.cap, C.byref(self._x)), "ls_xchg_create")
                self._h = (C.c_ubyte * 64)()
                _C.check(self.lib.ls_xchg_handle(self._x, self._h), "ls_xchg_handle")

        step("create", create)
        mine = torch.tensor(list(bytes(self._h)), dtype=torch.uint8, device=on)
        every = torch.empty(world * 64, dtype=torch.uint8, device=on)
        dist.all_gather_into_tensor(every, mine, group=group)

        def connect():
            with torch.cuda.device(device):
                _C.check(self.lib.ls_xchg_connect(self._x, bytes(every.cpu().tolist())), "ls_xchg_connect")

        step("connect", connect)
        # Ranks that share one GPU (the 2-process tests on a 1-GPU box) keep each other off the CUs while they poll: a
        # wait can last whole scheduling quanta there.  Give those a long fuse.
        ident = [None] * world
        dist.all_gather_object(ident, (os.uname().nodename, str(getattr(torch.cuda.get_device_properties(device), "uuid", device))),
                               group=group)
        self.shared_gpu = len(set(ident)) < world

        def fuse():
            with torch.cuda.device(device):        # separate GPUs: 5 s covers a peer that is capturing a graph meanwhile
                _C.check(self.lib.ls_xchg_set_timeout(self._x, 60.0 if self.shared_gpu else 5.0), "ls_xchg_set_timeout")

        step("set_timeout", fuse)                  # (its all-reduce is also the barrier: nobody pushes before every mailbox is mapped)

    def _agree(self, ok: bool, what: str, err=None):
        """All ranks learn whether `what` worked EVERYWHERE; if not, every rank closes its half and raises."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._on)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if not int(flag.item()):
            self.close()
            raise RuntimeError(f"peer exchange: {what} failed on " + (f"this rank ({err})" if not ok else "another rank"))

    def all_gather(self, send: torch.Tensor, recv: torch.Tensor) -> torch.Tensor:
        """recv [world, stride] <- every rank's ``send`` (fp32, numel a multiple of 4, <= cap).  Current stream, no host sync."""
        assert send.dtype == recv.dtype == torch.float32 and send.is_contiguous() and recv.stride(1) == 1
        self._C.check(self.lib.ls_xchg_all_gather(self._x, send.data_ptr(), send.numel(), recv.data_ptr(), recv.stride(0),
                                                  torch.cuda.current_stream(send.device).cuda_stream), "ls_xchg_all_gather")
        return recv

    def status(self):
        """(exchanges completed, timed_out) -- synchronises the device."""
        epoch, bad = C.c_uint64(0), C.c_int(0)
        with torch.cuda.device(self.device):
            self._C.check(self.lib.ls_xchg_status(self._x, C.byref(epoch), C.byref(bad)), "ls_xchg_status")
        return int(epoch.value) - 1, bool(bad.value)

    def self_check(self, rounds: int = 3, n: int = 4096) -> bool:
        """A few exchanges of rank-dependent patterns compared with the library collective; every rank gets the same verdict."""
        ok = True
        n = min(n, self.cap)
        for i in range(rounds):
            send = (torch.arange(n, dtype=torch.float32, device=self.device) * (self.rank + 1) + i).contiguous()
            got = torch.zeros((self.world, n), dtype=torch.float32, device=self.device)
            ref = torch.zeros_like(got)
            self.all_gather(send, got)
            dist.all_gather_into_tensor(ref.view(-1), send, group=self.group)
            ok = ok and bool(torch.equal(got, ref))
        ok = ok and not self.status()[1]
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._on)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()))

    def close(self):
        if self._x:
            with torch.cuda.device(self.device):
                self.lib.ls_xchg_destroy(self._x)
            self._x = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KVShard:
    def __init__(self, rank: int, world: int, shard_rows: int, group=None, vocab_parallel: bool = True):
        self.rank, self.world, self.Ls, self.group = rank, world, int(shard_rows), group
        # the lm_head sharded by vocabulary (round 3): 1.05 GB of the 23 GB a Llama-3 round streams is the lm_head, six times --
        # the one replicated item that splits without any change of arithmetic (see `head_select`)
        self.vocab_parallel = bool(vocab_parallel)
        self._head = {}
        self.start = rank * self.Ls
        self.is_tail = rank == world - 1
        self._send = {}
        self._recv = {}
        self._pass_len = {}
        self.prefill_ctx = None          # (first global row, local rows, prompt rows) while a sharded prefill runs
        self.peer: Optional[PeerExchange] = None     # set by enable_peer_exchange: the graph-capturable exchange
        self.peer_tried = False

    # ---- the graph-capturable exchange ----------------------------------------------------------
    def enable_peer_exchange(self, cap_floats: int, device) -> bool:
        """Collective: map the peers' mailboxes and verify a few exchanges against the library collective.  On any
        failure the shard stays on the ``torch.distributed`` all-gather (and says so on stderr) -- never silently."""
        if self.peer is not None and self.peer.cap >= cap_floats:
            return True
        self.peer_tried = True
        if self.peer is not None:
            self.peer.close()
            self.peer = None
        # every way out of this block is taken by ALL ranks together: PeerExchange() agrees on each of its local steps and
        # raises everywhere or nowhere, self_check() returns one all-reduced verdict -- so `self.peer` (and with it the
        # path KVShard.attend takes) is the same on every rank
        try:
            peer = PeerExchange(self.rank, self.world, cap_floats, device, self.group)
        except RuntimeError as e:                   # IPC not available on this system (agreed by all ranks)
            print(f"[longspec_amd.dist] rank {self.rank}: peer-store exchange unavailable ({e}); "
                  "using the torch.distributed all-gather", file=sys.stderr, flush=True)
            return False
        if peer.self_check():
            self.peer = peer
        else:
            peer.close()
            print(f"[longspec_amd.dist] rank {self.rank}: peer-store exchange failed its self-check; "
                  "using the torch.distributed all-gather", file=sys.stderr, flush=True)
        return self.peer is not None

    def raise_if_exchange_failed(self):
        """Reads the peer exchange's timeout latch (one device synchronisation).  Once a wait has given up, every later
        attention call merged whatever was in the mailboxes: the generation is void and says so."""
        if self.peer is None:
            return
        done, timed_out = self.peer.status()
        if timed_out:
            raise RuntimeError(f"rank {self.rank}: a peer-exchange wait timed out (after {done} exchanges): a peer is slow or gone; "
                               "the tokens generated since then are invalid")

    @property
    def graph_safe(self) -> bool:
        """A decode round under this shard is pure kernel launches."""
        return self.peer is not None

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
        n_floats = (n_floats + 3) // 4 * 4          # the peer exchange moves 16-byte units
        key = (n_floats, str(device))
        if key not in self._send:
            self._send[key] = torch.zeros(n_floats, dtype=torch.float32, device=device)
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
        if self.peer is not None and send.numel() <= self.peer.cap:
            return self.peer.all_gather(send, recv)
        dist.all_gather_into_tensor(recv.view(-1), send, group=self.group)
        return recv

    def attend(self, call) -> torch.Tensor:
        """partial -> all-gather -> merge, for one ``ShardedAttnCall``-like object."""
        if self.peer is not None and call.record_floats <= self.peer.cap and hasattr(call, "attend_peer"):
            return call.attend_peer(self.peer._x)
        send, recv = self.buffers(call.record_floats, call.device)
        call.partial(send)
        xt = getattr(call, "xchg_timing", None)
        if xt is not None:                        # (includes the local reduction's tail: the collective is not a kernel of ours)
            xt[0].record()
        out = call.finish(self.exchange(send, recv))
        if xt is not None:
            xt[1].record()
        return out


    # ---- the lm_head, sharded by vocabulary ------------------------------------------------------
    VOCAB_CHUNK = 8192          # = ops.TOPK_CHUNK: the unit stage 1 of the fused log-softmax / top-k works in

    def vocab_slice(self, V: int):
        """(chunk slots per rank, first column, end column) of this rank: whole 8192-column chunks, dealt in rank order -- so
        that the ranks' chunk records, concatenated in rank order, are the one-GPU kernel's records in its own chunk order."""
        nchunks = (V + self.VOCAB_CHUNK - 1) // self.VOCAB_CHUNK
        ncl = (nchunks + self.world - 1) // self.world
        lo = min(V, self.rank * ncl * self.VOCAB_CHUNK)
        hi = min(V, (self.rank + 1) * ncl * self.VOCAB_CHUNK)
        return ncl, lo, hi

    def head_select(self, lm_head, hidden: torch.Tensor, ops, k: int = 1, history: Optional[torch.Tensor] = None, argmax: bool = False):
        """``ops.logprob_topk(lm_head(hidden), history, k)`` or ``ops.argmax_rows(lm_head(hidden))`` with the lm_head's rows
        (vocabulary entries) split over the ranks: every rank multiplies by ITS slice of the weight only.
        On the GPU the ranks exchange the per-chunk records of stage 1 (max, sum of exp, k candidates: a few KB, through the
        same mailboxes as the attention records) and every rank runs stage 2 on all of them: each logit is produced by one
        rank with the arithmetic of the one-GPU launch, chunk records do not depend on other chunks, stage 2 sees them in the
        same order -- values and indices are bit-identical for any number of ranks.  Operator sets without the two-stage form
        (the CPU oracle of the tests) gather the logits themselves."""
        V, Hd = lm_head.out_features, lm_head.in_features
        x = hidden.reshape(-1, Hd)
        rows = x.shape[0]
        ncl, lo, hi = self.vocab_slice(V)
        key = (lm_head.weight.data_ptr(), lm_head.weight._version, lo, hi)
        if self._head.get("key") != key:
            w = lm_head.weight[lo:hi]
            self._head = {"key": key, "w": w, "packed": ops.pack_weight(w) if hi > lo and hasattr(ops, "topk_stage1") else None}
        kk = 1 if argmax else k
        nchunks = (V + self.VOCAB_CHUNK - 1) // self.VOCAB_CHUNK
        # stage 2's candidate budget (csrc/topk.hip::stage2; the same test ops.logprob_topk makes before it falls back to the
        # library): counted on the REAL chunks -- the slots past the vocabulary (world * ncl - nchunks of them, all at the end
        # of the rank-ordered records) are dropped before stage 2, so a W-rank call has exactly the one-GPU call's budget
        fits = V % 8 == 0 and (argmax or (kk <= 64 and rows <= 128 and rows * nchunks * kk <= 5120 and rows * nchunks <= 512))
        if hasattr(ops, "topk_stage1") and x.is_cuda and fits:
            n = ncl * rows * (2 + 2 * kk)
            send, recv = self.buffers(n, x.device)
            # n_splits=1: the full lm_head (>= 256 row groups) is never split in K, and a slice must keep that summation
            # order whatever its own row count suggests to the launch planner -- else the logits depend on the world size
            local = ops.linear(x, self._head["packed"], n_splits=1) if hi > lo else None
            ops.topk_stage1(local, rows, kk, lo, ncl, send, dtype=x.dtype)
            allrec = self.exchange(send, recv)[:, :n].reshape(self.world * ncl, rows, 2 + 2 * kk)
            if not allrec.is_contiguous():
                allrec = allrec.contiguous()
            # an empty record adds exactly +0.0f to a row's sum and never holds a candidate: dropping it changes no bit
            return ops.topk_stage2(allrec[:nchunks], rows, V, kk, history.reshape(-1) if history is not None else None, argmax)
        # generic form: all-gather the logits (equal widths: the short / empty slices are padded with -inf)
        width = ncl * self.VOCAB_CHUNK
        pad = torch.full((rows, width), float("-inf"), dtype=x.dtype, device=x.device)
        if hi > lo:
            pad[:, :hi - lo] = torch.nn.functional.linear(x, self._head["w"])
        full = self.gather_rows(pad).permute(1, 0, 2).reshape(rows, self.world * width)[:, :V]
        if argmax:
            return ops.argmax_rows(full)
        return ops.logprob_topk(full.view(1, rows, V), history, k)


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

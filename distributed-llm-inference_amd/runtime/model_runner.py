"""Model runner: owns the weights, the paged KV cache and the per-step device buffers; executes a
``ScheduledBatch`` eagerly (mixed prefill/decode steps) or by replaying a captured hipGraph
(pure-decode steps).

Per step the host work is: C++ ``build_batch`` straight into pinned buffers → a handful of async H2D
copies → kernels (or one graph launch) → one D2H of the sampled ids.  All kernels read their
metadata from *static* device buffers, which is what makes the decode step capturable: a replay only
needs the buffers refreshed, never new launch arguments.

hipGraph buckets: one graph per padded batch size; padding rows have ``seq_len = 0`` and
``slot = -1`` so the kernels skip them.  The KV-split count of the decode kernel is a launch
dimension, hence fixed per bucket at capture time (small buckets → more splits).
"""
from __future__ import annotations

import os
import time
from typing import Optional

import numpy as np
import torch

from .. import ops
from ..config import EngineConfig, KV_BLOCK_SIZE
from ..models import build_model, torch_dtype
from ..ops import AttnMetadata
from ..ops.kv_layout import allocate_kv_cache, bytes_per_block
from ..parallel.state import SINGLE, ParallelContext


_DUMP_LOGITS = __import__("os").environ.get("DLI_DUMP_LOGITS", "")


def _pinned(shape, dtype, pin):
    t = torch.zeros(shape, dtype=dtype)
    return t.pin_memory() if pin else t


_DEBUG_BOUNDS = os.environ.get("DLI_DEBUG_BOUNDS") == "1"


class ModelRunner:
    MAX_COW = 256                      # copy-on-write page pairs per step (n > 1 forks)

    def __init__(self, cfg: EngineConfig, ctx: Optional[ParallelContext] = None, device: Optional[torch.device] = None):
        self.cfg = cfg
        self.ctx = ctx or SINGLE
        if device is None:
            use_gpu = cfg.device == "cuda" or (cfg.device == "auto" and torch.cuda.is_available())
            device = torch.device("cuda", torch.cuda.current_device()) if use_gpu else torch.device("cpu")
        self.device = device
        self.on_gpu = device.type == "cuda"
        mcfg = cfg.model
        self.dtype = torch_dtype(mcfg.dtype)
        if self.on_gpu and self.dtype != torch.bfloat16:
            self.dtype = torch.bfloat16                      # the HIP kernels are bf16
        t0 = time.perf_counter()
        self.model = build_model(mcfg, self.ctx, device, self.dtype, cfg.seed, cfg.weights_path)
        self.load_seconds = time.perf_counter() - t0
        self.kv_dtype = torch.float8_e4m3fn if cfg.cache.kv_dtype == "fp8" else self.dtype
        self.local_kv_heads = mcfg.num_kv_heads // self.ctx.tp_size
        self.local_heads = mcfg.num_heads // self.ctx.tp_size
        self.max_blocks_per_seq = (cfg.sched.max_model_len + KV_BLOCK_SIZE - 1) // KV_BLOCK_SIZE
        # every rank sizes from its OWN free memory; only the driver's count feeds the block manager, so the group
        # takes the minimum — a worker with fewer pages than the driver would be handed slots past its cache
        self.num_blocks = self.ctx.all_min_int(self._decide_num_blocks())
        self.kv_caches = allocate_kv_cache(mcfg.num_layers, self.num_blocks, self.local_kv_heads, mcfg.head_dim,
                                           self.kv_dtype, device)
        self._alloc_buffers()
        if self.ctx.tp_size > 1 and self.on_gpu and cfg.use_hip_graphs:
            import torch.distributed as dist
            if dist.get_backend(self.ctx.group) != "nccl" and self.ctx.custom_ar is None:
                cfg.use_hip_graphs = False                   # gloo collectives synchronise the host: not capturable
        self.graphs: dict[int, tuple] = {}
        self.graph_pool = None
        self.tile_tokens = ops.prefill_tile_tokens(self.local_heads, self.local_kv_heads)
        self.stats = {"graph_steps": 0, "eager_steps": 0, "ar_fallbacks": 0}
        self.selftest: dict = {}                             # start-up checks of the fast paths → what got disabled
        self._dumped = False
        self._ar_vote = 0                                    # this rank saw its all-reduce kernel time out
        self._comm_fault = False                             # driver: the step just collected cannot be trusted
        self.ctx.barrier()                                   # weights loaded everywhere before the first collective

    # ------------------------------------------------------------------ sizing
    def _decide_num_blocks(self) -> int:
        c = self.cfg
        m = c.model
        per_block = bytes_per_block(m.num_layers, self.local_kv_heads, m.head_dim, self.kv_dtype)
        if c.cache.num_blocks is not None:
            return int(c.cache.num_blocks)
        if not self.on_gpu:
            return 2048
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info(self.device)
        T = c.sched.max_num_batched_tokens
        tp = self.ctx.tp_size
        act = T * 2 * (4 * m.hidden_size + (m.num_heads + 2 * m.num_kv_heads) * m.head_dim // tp * 2
                       + 3 * m.intermediate_size // tp * max(m.top_k_experts, 1))
        logits = c.sched.max_num_seqs * m.vocab_size * 8
        reserve = max(3 * (act + logits), 4 << 30) + (2 << 30)       # + graph pools
        budget = free - (total * (1.0 - c.cache.gpu_memory_utilization)) - reserve
        n = int(budget // per_block)
        if n < self.max_blocks_per_seq:
            raise RuntimeError(f"not enough HBM for the KV cache: {budget / 2**30:.1f} GiB left after weights")
        return n

    @property
    def kv_cache_tokens(self) -> int:
        return self.num_blocks * KV_BLOCK_SIZE

    def _alloc_buffers(self):
        """One packed int32 step buffer (pinned host + device mirror); every metadata tensor the kernels
        read is a *view* into the device mirror, so a step needs a single H2D copy (and, under TP, a
        single broadcast), and all addresses are static for hipGraph replay."""
        s = self.cfg.sched
        T, B = s.max_num_batched_tokens, s.max_num_seqs
        self.max_T, self.max_B = T, B
        max_tiles = T // 16 + B + 1
        fields = [("input", T), ("pos", T), ("slot", T), ("seq_lens", B), ("qsl", B + 1), ("last", B),
                  ("top_k", B), ("temp", B), ("top_p", B), ("seeds", 2 * B), ("steps", 2 * B),
                  ("src", B), ("tile_seq", max_tiles), ("tile_q0", max_tiles), ("cow", 2 * self.MAX_COW),
                  ("bt", B * self.max_blocks_per_seq)]
        self.layout, off = {}, 0
        for name, n in fields:
            off = (off + 3) // 4 * 4                      # 16-byte aligned regions
            self.layout[name] = (off, n)
            off += n
        total = (off + 3) // 4 * 4
        # two host copies: step N+1 is staged while the H2D copy of step N may still be queued
        h_bufs = [_pinned(total, torch.int32, self.on_gpu) for _ in range(2)]
        self.d_buf = torch.zeros(total, dtype=torch.int32, device=self.device)

        def views(buf):
            v = {k: buf[o:o + n] for k, (o, n) in self.layout.items()}
            v["temp"], v["top_p"] = v["temp"].view(torch.float32), v["top_p"].view(torch.float32)
            v["seeds"], v["steps"] = v["seeds"].view(torch.int64), v["steps"].view(torch.int64)
            v["bt"] = v["bt"].view(B, self.max_blocks_per_seq)
            return v

        self.dv = views(self.d_buf)
        self._host_sets = []
        for hb in h_bufs:
            hv = views(hb)
            self._host_sets.append((hb, hv, {k: t.numpy() for k, t in hv.items()}))
        self._host_idx = 0
        self._bind_host(0)
        self.d_sampled = torch.zeros(B, dtype=torch.int64, device=self.device)
        self.h_sampled = [_pinned(B, torch.int64, self.on_gpu) for _ in range(2)]
        self._fetch_idx = 0
        self._prev_event = None
        max_splits = 16
        self.decode_ws = (torch.empty(B * self.local_heads * max_splits * 130, dtype=torch.float32, device=self.device)
                          if self.on_gpu else None)

    def _bind_host(self, i: int):
        self._host_idx = i
        self.h_buf, self.hv, self.nv = self._host_sets[i]
        nv = self.nv
        self.n_input, self.n_pos, self.n_slot = nv["input"], nv["pos"], nv["slot"]
        self.n_bt, self.n_seq_lens, self.n_qsl = nv["bt"], nv["seq_lens"], nv["qsl"]

    # ------------------------------------------------------------------ batch → device
    def _stage(self, batch, bm, cow=()):
        reqs = batch.requests
        if len(cow) > self.MAX_COW:
            raise RuntimeError(f"{len(cow)} copy-on-write pages in one step (limit {self.MAX_COW})")
        if cow:
            self.nv["cow"][:2 * len(cow)] = np.asarray(cow, dtype=np.int32).reshape(-1)
        B, T = len(reqs), batch.total_tokens
        nv = self.nv
        seq_ids = np.fromiter((r.seq_id for r in reqs), dtype=np.int64, count=B)
        computed = np.fromiter((r.num_computed for r in reqs), dtype=np.int32, count=B)
        bm.build_batch(seq_ids, computed, batch.num_new, self.n_pos, self.n_slot, self.n_bt, self.n_seq_lens, self.n_qsl)
        nd = batch.num_decode
        n_src = 0
        if nd:
            self.n_input[:nd] = np.fromiter((r.tokens[r.num_computed] for r in reqs[:nd]), dtype=np.int32, count=nd)
            # ids still on the GPU (sampled by the step in flight): row of the previous sampler output
            src = nv["src"][:nd]
            src[:] = np.fromiter((r.sample_row if r.pending else -1 for r in reqs[:nd]), dtype=np.int32, count=nd)
            n_src = int((src >= 0).sum())
            assert n_src == int((self.n_input[:nd] < 0).sum()), "in-flight token bookkeeping out of sync"
        off = nd
        for r, n in zip(reqs[nd:], batch.num_new[nd:]):
            self.n_input[off:off + n] = r.tokens[r.num_computed:r.num_computed + n]
            off += n
        idx = np.flatnonzero(batch.sample_mask)
        S = len(idx)
        if S:
            nv["last"][:S] = self.n_qsl[idx + 1] - 1
            sel = [reqs[i] for i in idx]
            nv["temp"][:S] = [r.params.temperature for r in sel]
            nv["top_p"][:S] = [r.params.top_p for r in sel]
            nv["top_k"][:S] = [r.params.top_k for r in sel]
            nv["seeds"][:S] = [r.seed for r in sel]
            nv["steps"][:S] = [r.num_output for r in sel]
        n_tiles = 0
        if B > nd:
            ts, t0 = ops.build_prefill_tiles(batch.num_new[nd:], nd, self.tile_tokens, self.n_seq_lens[nd:B])
            n_tiles = len(ts)
            nv["tile_seq"][:n_tiles] = ts
            nv["tile_q0"][:n_tiles] = t0
        return B, T, S, n_tiles, n_src

    def _copy_in(self):
        self.d_buf.copy_(self.h_buf, non_blocking=True)

    def _meta(self, B, T, num_decode, max_seq_len, splits, n_tiles) -> AttnMetadata:
        dv = self.dv
        return AttnMetadata(num_decode=num_decode, num_prefill=B - num_decode, num_tokens=T,
                            positions=dv["pos"][:T], slot_mapping=dv["slot"][:T], block_tables=dv["bt"],
                            seq_lens=dv["seq_lens"], query_start_loc=dv["qsl"], max_seq_len=max_seq_len,
                            decode_splits=splits, decode_workspace=self.decode_ws,
                            prefill_tile_seq=dv["tile_seq"], prefill_tile_q0=dv["tile_q0"],
                            num_prefill_tiles=n_tiles)

    # ------------------------------------------------------------------ forward + sample
    def _forward_sample(self, T, S, meta):
        dv = self.dv
        h = self.model(dv["input"][:T], meta, self.kv_caches)
        if S == 0:
            return None
        logits = self.model.compute_logits(h.index_select(0, dv["last"][:S]))
        if _DUMP_LOGITS and self.ctx.is_driver and not torch.cuda.is_current_stream_capturing() and not self._dumped:
            self._dumped = True                              # debug aid: first sampled row of the first eager step
            torch.save(logits[0].float().cpu(), _DUMP_LOGITS)
        if _DEBUG_BOUNDS and not (self.on_gpu and torch.cuda.is_current_stream_capturing()):
            # the sampler maps a NaN row to token 0 instead of an out-of-range id; in debug mode the NaN itself is the error
            assert bool(torch.isfinite(logits[: max(S, 1)].float()).all()), "non-finite logits"
        ops.sample(logits, dv["temp"][:S], dv["top_p"][:S], dv["top_k"][:S], dv["seeds"][:S], dv["steps"][:S],
                   out=self.d_sampled[:S])
        return logits

    @torch.inference_mode()
    def launch(self, batch, bm, cow=()):
        """Driver-side first half of a step: stage the batch, (broadcast it to the TP workers,) queue the
        forward pass, the sampler and the D2H copy of the sampled ids.  Returns a handle for ``collect``.
        At most one earlier step may still be uncollected (its sampled ids are then read on the device)."""
        self._bind_host(self._host_idx ^ 1)
        B, T, S, n_tiles, n_src = self._stage(batch, bm, cow)
        nd = batch.num_decode
        bucket = 0
        if nd == B and S == B and self.on_gpu and self.cfg.use_hip_graphs:
            bucket = self._bucket_for(B) or 0
        splits = 1
        if self.on_gpu and nd and not bucket:
            splits = ops.choose_decode_splits(nd, self.local_kv_heads, int(self.n_seq_lens[:nd].max()))
        header = (B, T, S, nd, n_tiles, bucket, splits, n_src, len(cow))
        if self.ctx.tp_size > 1:
            self._broadcast_step(header)
        return self.run_staged(header)

    def collect(self, handle) -> np.ndarray:
        """Second half: wait for the step's sampled token ids (one per ``sample_mask`` row)."""
        kind, a, b = handle
        if kind == "host":
            return a
        a.synchronize()                                      # the event recorded behind the D2H copy
        self._note_ar_error()
        buf, n = b
        return buf.numpy()[:n].copy()

    def _note_ar_error(self):
        ar = self.ctx.custom_ar
        if ar is not None and ar.enabled and not self._ar_vote and ar.error_seen:
            self._ar_vote = 1                                # travels with the next step header; the group switches together
            self._comm_fault = True                          # the engine discards this step and what is queued behind it
            print(f"[dli_amd] tp rank {self.ctx.tp_rank}: custom all-reduce timed out waiting for a peer", flush=True)

    def take_comm_fault(self) -> bool:
        """True once after a collected step was found to contain a timed-out all-reduce (``LLMEngine._finish_step``)."""
        f, self._comm_fault = self._comm_fault, False
        return f

    def execute(self, batch, bm) -> np.ndarray:
        return self.collect(self.launch(batch, bm))

    def _broadcast_step(self, header):
        """Driver → workers: the step header over the CPU (gloo) group, the packed step buffer over the device
        path (the IPC broadcast kernel when the custom collectives are up, else the device group's broadcast).

        The header travels as a SUM all-reduce of 10 int64: the driver contributes the 9 header fields, every rank
        contributes its "my all-reduce kernel timed out" vote in the last slot — so a time-out seen by ANY rank
        reaches all of them with the next step, and the whole group leaves the custom kernels at the same step."""
        import torch.distributed as dist
        ctx = self.ctx
        hdr = torch.zeros(10, dtype=torch.int64)
        if ctx.is_driver:
            hdr[:9] = torch.tensor(header, dtype=torch.int64)
        hdr[9] = self._ar_vote
        g = ctx.cpu_group if ctx.cpu_group is not None else ctx.group
        dist.all_reduce(hdr, group=g)
        if int(hdr[0]) < 0:
            return None
        if int(hdr[9]) > 0 and ctx.custom_ar is not None and ctx.custom_ar.enabled:
            ctx.disable_custom_ar("a rank's kernel gave up waiting for a peer")
            self.graphs.clear()                              # they hold the custom kernels
            self.stats["ar_fallbacks"] += 1
            if dist.get_backend(ctx.group) != "nccl":
                self.cfg.use_hip_graphs = False              # host-staged collectives cannot be captured
        head = tuple(int(x) for x in hdr[:9])
        if ctx.is_driver:
            self._pad_for(head)
            self._copy_in()
        ctx.broadcast_tensor(self.d_buf, 0)
        return head

    def worker_step(self) -> bool:
        """Worker-side step (TP rank > 0): wait for the driver's batch and execute it.  Returns False
        when the driver signalled shutdown."""
        header = self._broadcast_step(None)
        if header is None:
            return False
        _kind, ev, _ = self.run_staged(header, staged_on_device=True)
        if self._prev_event is not None:                     # stay at most one step ahead of the GPU, like the driver
            self._prev_event.synchronize()
            self._note_ar_error()
        self._prev_event = ev if _kind == "event" else None
        return True

    def shutdown_workers(self):
        if self.ctx.tp_size > 1 and self.ctx.is_driver:
            import torch.distributed as dist
            g = self.ctx.cpu_group if self.ctx.cpu_group is not None else self.ctx.group
            hdr = torch.zeros(10, dtype=torch.int64)
            hdr[0] = -1
            dist.all_reduce(hdr, group=g)

    def _pad_for(self, header):
        B, bucket = header[0], header[5]
        if bucket:
            self._pad_decode(B, bucket)

    @torch.inference_mode()
    def run_staged(self, header, staged_on_device: bool = False):
        B, T, S, nd, n_tiles, bucket, splits, n_src, n_cow = header
        tp_staged = self.ctx.tp_size > 1                    # _broadcast_step already padded + copied
        if bucket and not tp_staged:
            self._pad_decode(B, bucket)
        if not tp_staged:
            self._copy_in()
        if _DEBUG_BOUNDS:
            self._check_bounds(B if not bucket else bucket, T if not bucket else bucket, n_cow)
        if n_cow:
            # n > 1 forks: give the children private copies of the page they are about to write (every layer's K and V;
            # stream-ordered behind the step that filled the source pages, ahead of this step's kernels)
            pairs = self.dv["cow"][:2 * n_cow].view(n_cow, 2)
            for k_cache, v_cache in self.kv_caches:
                ops.copy_pages(k_cache, pairs)
                ops.copy_pages(v_cache, pairs)
        if n_src:
            # decode rows whose input id was sampled by the step still in flight: take it from the sampler's
            # device output (stream order puts this behind that step and ahead of this step's sampler)
            ops.gather_sampled(self.dv["input"][:nd], self.dv["src"][:nd], self.d_sampled)
        if bucket and not self.cfg.use_hip_graphs:
            # the bucket was chosen before graphs were switched off (all-reduce fallback in this very step): run the
            # padded batch eagerly — padding rows are inert, the split count is the bucket's on every rank
            B_pad, splits = bucket, self._splits_for_bucket(bucket)
            self._forward_sample(B_pad, B_pad, self._meta(B_pad, B_pad, B_pad, 0, splits, 0))
            self.stats["eager_steps"] += 1
            return self._fetch(B)
        if bucket:
            if bucket not in self.graphs:
                self.capture_graph(bucket)
            g = self.graphs[bucket]
            g[0].replay()
            for k, v in g[3].items():                        # collectives inside the graph, counted at capture
                self.ctx.comm[k] += v
            self.stats["graph_steps"] += 1
            return self._fetch(B)
        meta = self._meta(B, T, nd, 0, splits, n_tiles)
        self._forward_sample(T, S, meta)
        self.stats["eager_steps"] += 1
        return self._fetch(S)

    def _check_bounds(self, B: int, T: int, n_cow: int) -> None:
        """``DLI_DEBUG_BOUNDS=1``: before any kernel of the step runs, read the step's metadata back
        from the DEVICE buffer — what this rank's kernels will really see, on TP workers too — and check every index
        against this rank's cache: the kernels themselves do no bounds checks.  Synchronises; debug only."""
        dv, nb = self.dv, self.num_blocks
        slot = dv["slot"][:T].cpu().numpy()
        seq_lens = dv["seq_lens"][:B].cpu().numpy()
        bt = dv["bt"][:B].cpu().numpy()
        ids = dv["input"][:T].cpu().numpy()
        assert slot.size == 0 or (slot.max() < nb * KV_BLOCK_SIZE and slot.min() >= -1), \
            f"slot out of range: [{slot.min()}, {slot.max()}] with {nb} blocks"
        for i in range(B):
            used = (int(seq_lens[i]) + KV_BLOCK_SIZE - 1) // KV_BLOCK_SIZE
            row = bt[i, :used]
            assert row.size == 0 or (row.min() >= 0 and row.max() < nb), f"block table row {i}: {row.min()}..{row.max()} / {nb}"
        assert ids.size == 0 or ids.max() < self.cfg.model.vocab_size, f"token id {ids.max()} >= vocab"
        if n_cow:
            cow = dv["cow"][:2 * n_cow].cpu().numpy()
            assert cow.min() >= -1 and cow.max() < nb, f"copy-on-write page out of range: {cow}"

    def _fetch(self, S):
        """Queue the D2H copy of the sampled ids; the handle is resolved by ``collect``."""
        if not self.on_gpu:
            return ("host", self.d_sampled[:S].numpy().copy(), None)
        self._fetch_idx ^= 1
        buf = self.h_sampled[self._fetch_idx]
        if S:
            buf[:S].copy_(self.d_sampled[:S], non_blocking=True)
        ar = self.ctx.custom_ar
        if ar is not None and ar.enabled:
            ar.queue_error_check()                           # 4-byte D2H behind the step: read after its event
        ev = torch.cuda.Event()
        ev.record()
        return ("event", ev, (buf, S))

    # ------------------------------------------------------------------ hipGraph decode
    def _bucket_for(self, B: int) -> Optional[int]:
        for b in self.cfg.graph_batch_sizes:
            if b >= B and b <= self.cfg.sched.max_num_seqs:
                return b
        return None

    def _splits_for_bucket(self, bucket: int) -> int:
        return ops.choose_decode_splits(bucket, self.local_kv_heads, min(self.cfg.sched.max_model_len, 2048))

    def _pad_decode(self, B: int, bucket: int):
        if bucket > B:
            nv = self.nv
            self.n_seq_lens[B:bucket] = 0
            self.n_slot[B:bucket] = -1
            self.n_pos[B:bucket] = 0
            self.n_input[B:bucket] = 0
            self.n_qsl[B + 1:bucket + 1] = np.arange(B + 1, bucket + 1)
            nv["last"][B:bucket] = np.arange(B, bucket)
            nv["temp"][B:bucket] = 0.0
            nv["top_p"][B:bucket] = 1.0
            nv["top_k"][B:bucket] = 0
            nv["src"][B:bucket] = -1

    def capture_graph(self, bucket: int):
        """Capture one decode step at batch size ``bucket`` (inputs must be staged in the static buffers)."""
        meta = self._meta(bucket, bucket, bucket, 0, self._splits_for_bucket(bucket), 0)
        stream = torch.cuda.Stream(device=self.device)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):                      # warm-up outside capture (hipBLASLt workspaces, …)
            for _ in range(2):
                self._forward_sample(bucket, bucket, meta)
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        before = dict(self.ctx.comm)
        with torch.cuda.graph(g, pool=self.graph_pool, stream=stream):
            logits = self._forward_sample(bucket, bucket, meta)
        comm = {k: v - before[k] for k, v in self.ctx.comm.items() if v != before[k]}
        if self.graph_pool is None:
            self.graph_pool = g.pool()
        self.graphs[bucket] = (g, meta, logits, comm)        # `logits`: the graph's own output buffer (self-test reads it)
        return g

    # ------------------------------------------------------------------ fp8 KV calibration
    @torch.inference_mode()
    def calibrate_kv_scales(self, prompt_ids=None, num_tokens: int = 2048, held_out=None) -> dict:
        """fp8 (e4m3) KV cache: set every layer's per-kv-head K and V scales from one warm-up prefill (amax / 448 with 20 %
        head-room) instead of the unit scale — with unit scales small K/V values fall into e4m3's subnormals.  Runs the
        normal forward over ``prompt_ids`` (default: ``num_tokens`` random ids) into scratch pages 0…; each attention
        layer updates its scale tensor right before it writes the cache, so deeper layers are calibrated on activations
        that already went through calibrated layers.  Under TP every rank calls this (each calibrates its own heads).
        Scales live in device tensors that the kernels read at run time: calling it again later (e.g. on real prompts)
        needs no re-capture of the decode graphs."""
        from ..models.llama import Attention
        attns = [m for m in self.model.modules() if isinstance(m, Attention)]
        if self.kv_dtype != torch.float8_e4m3fn or not attns:
            return {}
        for a in attns:
            if a.kv_scales is None:
                a.enable_kv_scales(self.device)
        T = min(num_tokens if prompt_ids is None else len(prompt_ids), self.max_T, self.cfg.sched.max_model_len - 1,
                self.num_blocks * KV_BLOCK_SIZE)
        T = max(T // 16 * 16, 16) if T >= 16 else T
        if prompt_ids is None:
            ids = np.random.default_rng(1234).integers(3, self.cfg.model.vocab_size, size=T)
        else:
            ids = np.asarray(prompt_ids[:T])
            if held_out is None and len(prompt_ids) >= T + 16:
                held_out = np.asarray(prompt_ids[T:2 * T])
        def stage(tok):
            n = len(tok)
            nb = (n + KV_BLOCK_SIZE - 1) // KV_BLOCK_SIZE
            nv = self.nv
            self.n_input[:n] = tok
            self.n_pos[:n] = np.arange(n)
            self.n_slot[:n] = np.arange(n)                   # pages 0 … nb-1, in order
            self.n_bt[0, :nb] = np.arange(nb)
            self.n_seq_lens[0] = n
            self.n_qsl[:2] = (0, n)
            ts, t0 = ops.build_prefill_tiles(np.array([n], dtype=np.int32), 0, self.tile_tokens)
            nv["tile_seq"][:len(ts)] = ts
            nv["tile_q0"][:len(ts)] = t0
            self._copy_in()
            return self._meta(1, n, 0, 0, 1, len(ts))

        meta = stage(ids)
        for a in attns:
            a.calibrating, a._calibrated = True, False
        try:
            self.model(self.dv["input"][:T], meta, self.kv_caches)
        finally:
            for a in attns:
                a.calibrating = False
        # held-out check: a second stream (the prompt's unused tail, else fresh random ids) through the calibrated
        # model — the share of K/V values beyond the e4m3 range under these scales is what the cache write will clip
        if held_out is None:
            held_out = np.random.default_rng(4321).integers(3, self.cfg.model.vocab_size, size=T)
        held_out = np.asarray(held_out[:T])
        sat = None
        if len(held_out):
            meta = stage(held_out)
            for a in attns:
                a.measuring, a.sat_count, a.sat_total = True, None, None
            try:
                self.model(self.dv["input"][:len(held_out)], meta, self.kv_caches)
            finally:
                for a in attns:
                    a.measuring = False
            tot = sum(a.sat_total or 0 for a in attns)
            cnt = sum(float(a.sat_count) for a in attns if a.sat_count is not None)
            sat = cnt / max(tot, 1)
        if self.on_gpu:
            torch.cuda.synchronize()
        sc = torch.stack([a.kv_scales for a in attns]).float().cpu()          # [layers, 2, Hkv]
        self.selftest["fp8_kv_scales"] = {"tokens": int(T), "source": "prompts" if prompt_ids is not None else "random ids",
                                          "k_scale_min": float(sc[:, 0].min()), "k_scale_max": float(sc[:, 0].max()),
                                          "v_scale_min": float(sc[:, 1].min()), "v_scale_max": float(sc[:, 1].max()),
                                          "held_out_saturation_rate": sat}
        return self.selftest["fp8_kv_scales"]

    @torch.inference_mode()
    def prefill_logits(self, ids) -> torch.Tensor:
        """Test / tooling helper: logits ``[T, V]`` (fp32) of one prompt prefilled into scratch pages 0…, every position
        attending through the paged cache (so the KV-cache dtype and scales are in the loop)."""
        T = len(ids)
        assert T <= self.max_T and T < self.cfg.sched.max_model_len and T <= self.num_blocks * KV_BLOCK_SIZE
        nb = (T + KV_BLOCK_SIZE - 1) // KV_BLOCK_SIZE
        nv = self.nv
        self.n_input[:T] = np.asarray(ids)
        self.n_pos[:T] = np.arange(T)
        self.n_slot[:T] = np.arange(T)
        self.n_bt[0, :nb] = np.arange(nb)
        self.n_seq_lens[0] = T
        self.n_qsl[:2] = (0, T)
        ts, t0 = ops.build_prefill_tiles(np.array([T], dtype=np.int32), 0, self.tile_tokens)
        nv["tile_seq"][:len(ts)] = ts
        nv["tile_q0"][:len(ts)] = t0
        self._copy_in()
        h = self.model(self.dv["input"][:T], self._meta(1, T, 0, 0, 1, len(ts)), self.kv_caches)
        return self.model.compute_logits(h).float()

    # ------------------------------------------------------------------ start-up self-tests
    def _stage_synthetic_decode(self, B: int, ctx_len: int = 24):
        """A deterministic decode batch of ``B`` sequences (one page each) in the static buffers — the same on every
        TP rank, so no broadcast is needed."""
        nv = self.nv
        i = np.arange(B)
        page = (i % self.num_blocks).astype(np.int32)
        self.n_bt[:B] = 0
        self.n_bt[:B, 0] = page
        self.n_seq_lens[:B] = ctx_len
        self.n_pos[:B] = ctx_len - 1
        self.n_slot[:B] = page * KV_BLOCK_SIZE + ctx_len - 1
        self.n_input[:B] = (i * 7 + 3) % self.cfg.model.vocab_size
        self.n_qsl[:B + 1] = np.arange(B + 1)
        nv["last"][:B] = i
        nv["temp"][:B] = 0.0
        nv["top_p"][:B] = 1.0
        nv["top_k"][:B] = 0
        nv["seeds"][:B] = 1
        nv["steps"][:B] = 0
        nv["src"][:B] = -1
        self._copy_in()

    def _vote(self, ok: bool) -> bool:
        return bool(self.ctx.all_min_int(1 if ok else 0))

    @torch.inference_mode()
    def startup_selftest(self) -> dict:
        """Check the fast paths that change numerics-relevant execution against their plain counterparts, once, on
        every rank of the group; a path that fails (or raises) is switched off for the life of the engine and the
        reason is kept in ``self.selftest`` (bench.py prints it).  Covered: a captured decode hipGraph against the
        same step run eagerly (logit-level), and the GEMM/all-reduce pipeline of ``RowParallelLinear`` against
        GEMM → all-reduce."""
        res = self.selftest
        if not self.on_gpu:
            return res
        # --- hipGraph replay vs eager, same staged batch
        if self.cfg.use_hip_graphs and self.graphs:
            try:
                b = min([x for x in self.graphs if x >= 8] or list(self.graphs))
                self._stage_synthetic_decode(b)
                meta = self._meta(b, b, b, 0, self._splits_for_bucket(b), 0)
                le = self._forward_sample(b, b, meta).float().clone()
                ids_e = self.d_sampled[:b].clone()
                self.graphs[b][0].replay()
                lg = self.graphs[b][2].float()
                diff = float((le - lg).abs().max())
                scale = float(le.abs().max())
                same_ids = float((ids_e == self.d_sampled[:b]).float().mean())
                ok = bool(diff == diff and diff <= 2e-2 * scale + 1e-3)
                res["graph_vs_eager"] = {"bucket": b, "max_abs_logit_diff": round(diff, 6), "logit_scale": round(scale, 4),
                                         "greedy_agreement": same_ids}
            except Exception as e:
                ok = False
                res["graph_vs_eager"] = {"error": f"{type(e).__name__}: {e}"}
            ok = self._vote(ok)
            res["graph_vs_eager"]["ok"] = ok
            if not ok:
                self.cfg.use_hip_graphs = False
                self.graphs.clear()
                print(f"[dli_amd] hipGraph decode disabled by the start-up self-test: {res['graph_vs_eager']}", flush=True)
        # --- overlapped row-parallel linear vs plain
        if self.ctx.tp_size > 1:
            from ..models.layers import RowParallelLinear
            layer = next((m for m in self.model.modules() if isinstance(m, RowParallelLinear) and m.reduce), None)
            if layer is None or not layer._stream_ordered_backend():
                res["overlapped_row_parallel"] = {"ok": None, "skipped": f"device backend {self.ctx.device_backend}"}
                RowParallelLinear.OVERLAP_OK = False
            else:
                try:
                    g = torch.Generator(device=self.device).manual_seed(4321 + self.ctx.tp_rank)
                    rows = max(RowParallelLinear.OVERLAP_MIN_ROWS, 1024)
                    x = (torch.randn(rows, layer.weight.data.shape[1], generator=g, device=self.device) * 0.5).to(self.dtype)
                    want = self.ctx.all_reduce(ops.linear(x, layer.weight.data)).float()
                    got = layer._forward_overlapped(x, None).float()
                    torch.cuda.synchronize()
                    diff = float((want - got).abs().max())
                    scale = float(want.abs().max())
                    ok = bool(diff == diff and diff <= 3e-2 * scale + 1e-3)
                    res["overlapped_row_parallel"] = {"max_abs_diff": round(diff, 6), "scale": round(scale, 4)}
                except Exception as e:
                    ok = False
                    res["overlapped_row_parallel"] = {"error": f"{type(e).__name__}: {e}"}
                ok = self._vote(ok)
                res["overlapped_row_parallel"]["ok"] = ok
                RowParallelLinear.OVERLAP_OK = ok
                if not ok:
                    print(f"[dli_amd] GEMM/all-reduce overlap disabled by the start-up self-test: "
                          f"{res['overlapped_row_parallel']}", flush=True)
        return res

    def capture_all(self, buckets=None):
        """Pre-capture the decode graphs (largest first so that the pool is sized once).  Under TP every
        rank must call this with the same buckets (the captures contain collectives)."""
        if not (self.on_gpu and self.cfg.use_hip_graphs):
            return 0
        todo = [b for b in (buckets or self.cfg.graph_batch_sizes) if b <= self.cfg.sched.max_num_seqs]
        err = None
        try:
            for bucket in sorted(todo, reverse=True):
                if bucket in self.graphs:
                    continue
                self._pad_decode(0, bucket)
                self._copy_in()
                self.capture_graph(bucket)
            torch.cuda.synchronize()
        except Exception as e:                               # e.g. a collective that cannot be captured on this backend
            err = f"{type(e).__name__}: {e}"
        if not self._vote(err is None):                      # all ranks keep their graphs, or none does
            self.selftest["graph_capture"] = {"ok": False, "error": err or "failed on another rank"}
            print(f"[dli_amd] hipGraph capture failed ({self.selftest['graph_capture']['error']}): decode steps run eagerly",
                  flush=True)
            self.cfg.use_hip_graphs = False
            self.graphs.clear()
        return len(self.graphs)

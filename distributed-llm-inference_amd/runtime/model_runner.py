"""Model runner: owns the weights, the paged KV cache and the per-step device buffers; executes a
``ScheduledBatch`` eagerly (mixed prefill/decode steps) or by replaying a captured hipGraph
(pure-decode steps).

Per step the host work is: C++ ``build_batch`` straight into pinned buffers → a handful of async H2D
copies → kernels (or one graph launch) → one D2H of the sampled ids.  All kernels read their
metadata from *static* device buffers, which is what makes the decode step capturable: a replay only
needs the buffers refreshed, never new launch arguments.

hipGraphs.  *Decode* graphs: one per padded batch size (forward + LM head + sampler); padding rows have
``slot = -1`` and belong to no work item, so the kernels skip them.  *Mixed* graphs: a step that
holds prefill chunks replays a *body* graph keyed by its token count rounded up to a bucket (the scheduler already
aligns large steps to 256 tokens) followed by a *head* graph (LM head + sampler) keyed by the padded number of
sampled rows.  What makes one captured launch sequence serve any batch composition: every attention kernel takes its
work from lists in the static step buffer — decode page ranges (``ops.build_decode_work``), prefill tiles — with the
VALID counts in device memory (``counts``); grids are launched at the bucket's upper bound and surplus workgroups exit.
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
from ..parallel.state import SINGLE, ParallelContext, drain_collectives


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
        self.model = build_model(mcfg, self.ctx, device, self.dtype, cfg.seed, cfg.weights_path, cfg.quantization)
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
        if self.ctx.distributed and self.on_gpu and cfg.use_hip_graphs:
            import torch.distributed as dist
            if dist.get_backend(self.ctx.group) != "nccl" and self.ctx.custom_ar is None:
                cfg.use_hip_graphs = False                   # gloo collectives synchronise the host: not capturable
        self.graphs: dict[int, tuple] = {}
        self.graph_pool = None
        self.tile_tokens = ops.prefill_tile_tokens(self.local_heads, self.local_kv_heads)
        self.stats = {"graph_steps": 0, "mixed_graph_steps": 0, "eager_steps": 0, "ar_fallbacks": 0}
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
        n_dev = max(torch.cuda.device_count(), 1)
        # rehearsal mode: several ranks (of a TP group, or data-parallel replicas of a job started with DLI_SHARE_GPU=1) live
        # on this device and size their caches at the same time
        from . import launch
        group = self.ctx.tp_size if self.ctx.distributed else 1
        job = int(os.environ.get("WORLD_SIZE", "1")) if launch.share_gpu_requested() else 1
        sharing = (max(group, job) + n_dev - 1) // n_dev
        if sharing > 1:
            # whatever the others have loaded so far: an equal share of the device minus what THIS rank holds besides its cache
            # — and only 55 % of that share: every rank's hipGraph pools (≈ 11 GB each for a Mixtral shard at 8192-token
            # steps) come out of the same card, and a rehearsal needs no large cache (8 ranks of Mixtral at real shapes ran
            # the card out of memory during capture with the full share)
            mine = torch.cuda.memory_allocated(self.device)
            budget = min((budget - (sharing - 1) * reserve) / sharing,
                         0.55 * total * c.cache.gpu_memory_utilization / sharing - mine - reserve)
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
        self.items_cap = min(ops.decode_launch_caps(B, self.local_kv_heads)[0], 65535)   # decode work items of one step
        fields = [("input", T), ("pos", T), ("slot", T), ("seq_lens", B), ("qsl", B + 1), ("last", B),
                  ("top_k", B), ("temp", B), ("top_p", B), ("seeds", 2 * B), ("steps", 2 * B),
                  ("src", B), ("tile_seq", max_tiles), ("tile_q0", max_tiles), ("cow", 2 * self.MAX_COW),
                  ("counts", 4), ("dwork", 4 * self.items_cap), ("dred", 4 * B),
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
            v["dwork"], v["dred"] = v["dwork"].view(-1, 4), v["dred"].view(-1, 4)
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
        # partial results of the decode kernel: one slot per work item at most
        self.decode_ws = (torch.empty(self.items_cap * self.local_heads * 130, dtype=torch.float32, device=self.device)
                          if self.on_gpu else None)
        # hidden states of the sampled rows: written by a mixed step's body graph, read by its head graph
        self.h_last = torch.zeros(B, self.cfg.model.hidden_size, dtype=self.dtype, device=self.device) if self.on_gpu else None

    def _bind_host(self, i: int):
        self._host_idx = i
        self.h_buf, self.hv, self.nv = self._host_sets[i]
        nv = self.nv
        self.n_input, self.n_pos, self.n_slot = nv["input"], nv["pos"], nv["slot"]
        self.n_bt, self.n_seq_lens, self.n_qsl = nv["bt"], nv["seq_lens"], nv["qsl"]

    # ------------------------------------------------------------------ batch → device
    def _stage(self, batch, bm, cow=(), cap_items: int = 0):
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
        n_items, n_reds, n_parts = self._stage_decode_work(nd, cap_items)
        nv["counts"][:4] = (n_items, n_reds, n_tiles, 0)
        return B, T, S, n_tiles, n_src, n_items, n_reds, n_parts

    def _stage_decode_work(self, nd: int, cap_items: int = 0):
        """Work list of the decode kernel for rows ``0 … nd-1`` of the staged batch (GPU only: the CPU reference
        attention walks the block tables itself)."""
        if not (nd and self.on_gpu):
            return 0, 0, 0
        work, red, n_parts = ops.build_decode_work(self.n_seq_lens[:nd], self.local_kv_heads,
                                                   min(cap_items or self.items_cap, self.items_cap))
        self.nv["dwork"][:len(work)] = work
        self.nv["dred"][:len(red)] = red
        return len(work), len(red), n_parts

    def _copy_in(self):
        self.d_buf.copy_(self.h_buf, non_blocking=True)

    def _meta(self, B, T, num_decode, n_items, n_reds, n_tiles, counts_on_device: bool = False) -> AttnMetadata:
        """``n_items`` / ``n_reds`` / ``n_tiles`` are the LAUNCHED extents: the step's own counts for an eager step, a
        bucket's upper bounds (``counts_on_device``) inside a hipGraph."""
        dv = self.dv
        return AttnMetadata(num_decode=num_decode, num_prefill=B - num_decode, num_tokens=T,
                            positions=dv["pos"][:T], slot_mapping=dv["slot"][:T], block_tables=dv["bt"],
                            seq_lens=dv["seq_lens"], query_start_loc=dv["qsl"], max_seq_len=0,
                            decode_workspace=self.decode_ws, decode_work=dv["dwork"], decode_red=dv["dred"],
                            counts=dv["counts"], decode_items=n_items, decode_reds=n_reds, decode_max_parts=self.items_cap,
                            counts_on_device=counts_on_device,
                            prefill_tile_seq=dv["tile_seq"], prefill_tile_q0=dv["tile_q0"],
                            num_prefill_tiles=n_tiles)

    # ------------------------------------------------------------------ forward + sample
    def _sample_rows(self, h_rows, S):
        dv = self.dv
        logits = self.model.compute_logits(h_rows)
        if _DEBUG_BOUNDS and not (self.on_gpu and torch.cuda.is_current_stream_capturing()):
            # the sampler maps a NaN row to token 0 instead of an out-of-range id; in debug mode the NaN itself is the error
            assert bool(torch.isfinite(logits[: max(S, 1)].float()).all()), "non-finite logits"
        ops.sample(logits, dv["temp"][:S], dv["top_p"][:S], dv["top_k"][:S], dv["seeds"][:S], dv["steps"][:S],
                   out=self.d_sampled[:S])
        return logits

    def _forward_sample(self, T, S, meta):
        dv = self.dv
        h = self.model(dv["input"][:T], meta, self.kv_caches)
        if S == 0:
            return None
        return self._sample_rows(h.index_select(0, dv["last"][:S]), S)

    # ------------------------------------------------------------------ step planning
    def _plan(self, B: int, T: int, nd: int, S: int):
        """Which captured graphs (if any) serve a step of B sequences / T tokens / nd decode rows / S sampled rows:
        ``(decode bucket, mixed token bucket, head bucket)`` — all 0 for an eager step."""
        if not (self.on_gpu and self.cfg.use_hip_graphs):
            return 0, 0, 0
        if nd == B and S == B:
            return self._bucket_for(B) or 0, 0, 0
        if not self.cfg.mixed_step_graphs:
            return 0, 0, 0
        tb = self._mixed_bucket(T)
        sb = (self._bucket_for(S) or 0) if S else 0
        if not tb or (S and not sb) or B > min(tb, self.max_B):
            return 0, 0, 0
        return 0, tb, sb

    def _mixed_bucket(self, T: int) -> int:
        for b in self.mixed_buckets:
            if b >= T:
                return b
        return 0

    @property
    def mixed_buckets(self) -> tuple:
        """Token counts a mixed step is padded to: small powers of two, then every 256 (the scheduler's alignment)."""
        if not hasattr(self, "_mixed_buckets"):
            top = self.max_T
            bs = [b for b in (32, 64, 128) if b < top] + list(range(256, top + 1, 256))
            if top not in bs:
                bs.append(top)
            self._mixed_buckets = tuple(sorted(set(bs)))
        return self._mixed_buckets

    def _decode_caps(self, bucket: int):
        """(work items, reduce entries) a decode graph of ``bucket`` rows launches.  Where even the smallest batch this
        bucket serves (one more row than the next smaller bucket) fills the chip, no row is ever cut
        (``ops.decode_chunk_pages``) and the graph holds one workgroup per (row, kv head) and no reduce pass."""
        items, reds = ops.decode_launch_caps(bucket, self.local_kv_heads)
        smaller = [b for b in self.cfg.graph_batch_sizes if b < bucket]
        fewest_rows = (max(smaller) if smaller else 0) + 1   # the smallest batch that lands in this bucket
        if fewest_rows * self.local_kv_heads >= 512:
            return bucket, 0
        return min(items, self.items_cap), reds

    def _mixed_caps(self, tb: int):
        """Launch extents of a mixed graph of ``tb`` tokens: decode rows / work items / reduce entries / prefill tiles."""
        rows = min(tb, self.max_B)
        items, reds = ops.decode_launch_caps(rows, self.local_kv_heads)
        return rows, min(items, self.items_cap), reds, tb // self.tile_tokens + min(rows, 256)

    @torch.inference_mode()
    def launch(self, batch, bm, cow=()):
        """Driver-side first half of a step: stage the batch, (broadcast it to the TP workers,) queue the
        forward pass, the sampler and the D2H copy of the sampled ids.  Returns a handle for ``collect``.
        At most one earlier step may still be uncollected (its sampled ids are then read on the device)."""
        self._bind_host(self._host_idx ^ 1)
        nd = batch.num_decode
        bucket, tb, sb = self._plan(len(batch.requests), batch.total_tokens, nd, int(batch.sample_mask.sum()))
        cap = self._decode_caps(bucket)[0] if bucket else (self._mixed_caps(tb)[1] if tb else 0)
        B, T, S, n_tiles, n_src, n_items, n_reds, _ = self._stage(batch, bm, cow, cap)
        if tb and n_tiles > self._mixed_caps(tb)[3]:
            tb = sb = 0                                      # more prefill sequences than the graph's tile grid: eager
        header = (B, T, S, nd, n_tiles, bucket, n_items, n_src, len(cow), n_reds, tb, sb)
        if self.ctx.distributed:
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

    HEADER_FIELDS = 12

    def _broadcast_step(self, header):
        """Driver → workers: the step header over the CPU (gloo) group, the packed step buffer over the device
        path (the IPC broadcast kernel when the custom collectives are up, else the device group's broadcast).

        The header travels as a SUM all-reduce of 13 int64: the driver contributes the 12 header fields, every rank
        contributes its "my all-reduce kernel timed out" vote in the last slot — so a time-out seen by ANY rank
        reaches all of them with the next step, and the whole group leaves the custom kernels at the same step."""
        import torch.distributed as dist
        ctx = self.ctx
        n = self.HEADER_FIELDS
        hdr = torch.zeros(n + 1, dtype=torch.int64)
        if ctx.is_driver:
            hdr[:n] = torch.tensor(header, dtype=torch.int64)
        hdr[n] = self._ar_vote
        g = ctx.cpu_group if ctx.cpu_group is not None else ctx.group
        dist.all_reduce(hdr, group=g)
        if int(hdr[0]) < 0:
            return None
        if int(hdr[n]) > 0 and ctx.custom_ar is not None and ctx.custom_ar.enabled:
            ctx.disable_custom_ar("a rank's kernel gave up waiting for a peer")
            self.graphs.clear()                              # they hold the custom kernels
            self.stats["ar_fallbacks"] += 1
            if dist.get_backend(ctx.group) != "nccl":
                self.cfg.use_hip_graphs = False              # host-staged collectives cannot be captured
        head = tuple(int(x) for x in hdr[:n])
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
        if self.ctx.distributed and self.ctx.is_driver:
            import torch.distributed as dist
            g = self.ctx.cpu_group if self.ctx.cpu_group is not None else self.ctx.group
            hdr = torch.zeros(self.HEADER_FIELDS + 1, dtype=torch.int64)
            hdr[0] = -1
            dist.all_reduce(hdr, group=g)

    def _pad_for(self, header):
        B, T, S, bucket, tb, sb = header[0], header[1], header[2], header[5], header[10], header[11]
        if bucket:
            self._pad_decode(B, bucket)
        elif tb:
            self._pad_mixed(T, S, tb, sb)

    @torch.inference_mode()
    def run_staged(self, header, staged_on_device: bool = False):
        B, T, S, nd, n_tiles, bucket, n_items, n_src, n_cow, n_reds, tb, sb = header
        tp_staged = self.ctx.distributed                    # _broadcast_step already padded + copied
        if not tp_staged:
            self._pad_for(header)
            self._copy_in()
        if _DEBUG_BOUNDS:
            self._check_bounds(B, T, n_cow)
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
        if (bucket or tb) and not self.cfg.use_hip_graphs:
            # the graphs were chosen before they were switched off (all-reduce fallback in this very step): run the
            # padded batch eagerly — padding rows are inert, the work lists are exact on every rank
            if bucket:
                self._forward_sample(bucket, bucket, self._meta(bucket, bucket, bucket, n_items, n_reds, 0))
            else:
                self._forward_sample(tb, S, self._meta(B, tb, nd, n_items, n_reds, n_tiles))
            self.stats["eager_steps"] += 1
            return self._fetch(B if bucket else S)
        if bucket:
            if bucket not in self.graphs:
                self.capture_graph(bucket)
            self._maybe_dump(self._replay(bucket))
            self.stats["graph_steps"] += 1
            return self._fetch(B)
        if tb:
            if ("m", tb) not in self.graphs:
                self.capture_mixed_graph(tb)
            self._replay(("m", tb))
            if S:
                if ("h", sb) not in self.graphs:
                    self.capture_head_graph(sb)
                self._maybe_dump(self._replay(("h", sb)))
            self.stats["graph_steps"] += 1
            self.stats["mixed_graph_steps"] += 1
            return self._fetch(S)
        meta = self._meta(B, T, nd, n_items, n_reds, n_tiles)
        self._maybe_dump(self._forward_sample(T, S, meta))
        self.stats["eager_steps"] += 1
        return self._fetch(S)

    def _replay(self, key):
        g = self.graphs[key]
        g[0].replay()
        for k, v in g[3].items():                            # collectives inside the graph, counted at capture
            self.ctx.comm[k] += v
        return g[2]

    def _maybe_dump(self, logits):
        """Debug aid (DLI_DUMP_LOGITS=<file>): logits of the first sampled row of the first SERVED step — eager or
        replayed (the graphs keep their logits buffer) — on the driver; warm-up and capture runs never get here."""
        if _DUMP_LOGITS and not self._dumped and logits is not None and self.ctx.is_driver:
            self._dumped = True
            torch.save(logits[0].float().cpu(), _DUMP_LOGITS)

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
        counts = dv["counts"].cpu().numpy()
        n_items, n_reds = int(counts[0]), int(counts[1])
        if n_items:
            w = dv["dwork"][:n_items].cpu().numpy()
            pages = (seq_lens.astype(np.int64) + KV_BLOCK_SIZE - 1) // KV_BLOCK_SIZE
            assert n_items <= self.items_cap and w[:, 0].min() >= 0 and w[:, 0].max() < B, "decode work item: row out of range"
            assert (w[:, 1] >= 0).all() and (w[:, 1] <= w[:, 2]).all() and (w[:, 2] <= np.maximum(pages[w[:, 0]], 1)).all(), \
                "decode work item: page range outside its sequence"
            assert w[:, 3].max() < self.items_cap, "decode work item: partial slot past the workspace"
            cover = np.bincount(w[:, 0], weights=w[:, 2] - w[:, 1], minlength=B)
            nd_rows = np.unique(w[:, 0])
            assert (cover[nd_rows] == np.maximum(pages[nd_rows], 1)).all(), "decode work list does not cover every page once"
        if n_reds:
            r = dv["dred"][:n_reds].cpu().numpy()
            assert r[:, 0].max() < B and (r[:, 1] + r[:, 2]).max() <= self.items_cap, "decode reduce entry out of range"
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

    # ------------------------------------------------------------------ hipGraphs
    def _bucket_for(self, B: int) -> Optional[int]:
        for b in self.cfg.graph_batch_sizes:
            if b >= B and b <= self.cfg.sched.max_num_seqs:
                return b
        return None

    def _pad_sampler_rows(self, lo: int, hi: int):
        if hi > lo:
            nv = self.nv
            nv["temp"][lo:hi] = 0.0
            nv["top_p"][lo:hi] = 1.0
            nv["top_k"][lo:hi] = 0

    def _pad_decode(self, B: int, bucket: int):
        """Rows ``B … bucket-1`` of a decode graph: no work item names them, their KV write is suppressed."""
        if bucket > B:
            nv = self.nv
            self.n_seq_lens[B:bucket] = 0
            self.n_slot[B:bucket] = -1
            self.n_pos[B:bucket] = 0
            self.n_input[B:bucket] = 0
            self.n_qsl[B + 1:bucket + 1] = np.arange(B + 1, bucket + 1)
            nv["last"][B:bucket] = np.arange(B, bucket)
            nv["src"][B:bucket] = -1
            self._pad_sampler_rows(B, bucket)

    def _pad_mixed(self, T: int, S: int, tb: int, sb: int):
        """Token rows ``T … tb-1`` of a mixed graph belong to no sequence (no KV write, no attention); sampled rows
        ``S …`` point at token 0 and are sampled greedily into slots nobody reads."""
        if tb > T:
            self.n_slot[T:tb] = -1
            self.n_pos[T:tb] = 0
            self.n_input[T:tb] = 0
        rows = self._mixed_caps(tb)[0]
        self.nv["last"][S:max(rows, sb)] = 0
        self._pad_sampler_rows(S, sb)

    def _capture(self, key, run, warm: int = 2):
        stream = torch.cuda.Stream(device=self.device)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):                      # warm-up outside capture (hipBLASLt workspaces, …)
            for _ in range(warm):
                run()
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        drain_collectives()                                  # the warm-up's RCCL calls: out of the watchdog before the capture
        g = torch.cuda.CUDAGraph()
        before = dict(self.ctx.comm)
        with torch.cuda.graph(g, pool=self.graph_pool, stream=stream, capture_error_mode=ops.CAPTURE_MODE):
            out = run()
        comm = {k: v - before[k] for k, v in self.ctx.comm.items() if v != before[k]}
        if self.graph_pool is None:
            self.graph_pool = g.pool()
        return g, out, comm

    def capture_graph(self, bucket: int):
        """Capture one decode step at batch size ``bucket`` (inputs must be staged in the static buffers)."""
        items, reds = self._decode_caps(bucket)
        meta = self._meta(bucket, bucket, bucket, items, reds, 0, counts_on_device=True)
        g, logits, comm = self._capture(bucket, lambda: self._forward_sample(bucket, bucket, meta))
        # `logits`: the graph's own output buffer (the self-test and the debug dump read it).  Only small buckets keep
        # theirs alive: a [2048, 128 k] buffer per graph would pin gigabytes of the shared pool that later captures reuse
        self.graphs[bucket] = (g, meta, logits if (bucket <= 64 or _DUMP_LOGITS) else None, comm)
        return g

    def capture_mixed_graph(self, tb: int):
        """Body graph of a mixed (prefill + decode) step of up to ``tb`` tokens: forward pass over the padded token
        rows, then the hidden states of the rows to sample gathered into the static ``h_last``."""
        rows, items, reds, tiles = self._mixed_caps(tb)
        meta = self._meta(rows, tb, rows, items, reds, tiles, counts_on_device=True)
        dv = self.dv

        def run():
            h = self.model(dv["input"][:tb], meta, self.kv_caches)
            torch.index_select(h, 0, dv["last"][:rows], out=self.h_last[:rows])
            return h
        g, h, comm = self._capture(("m", tb), run)
        self.graphs[("m", tb)] = (g, meta, None, comm)
        return g

    def capture_head_graph(self, sb: int):
        """LM head + sampler over the first ``sb`` rows of ``h_last``."""
        g, logits, comm = self._capture(("h", sb), lambda: self._sample_rows(self.h_last[:sb], sb))
        self.graphs[("h", sb)] = (g, None, logits if (sb <= 64 or _DUMP_LOGITS) else None, comm)
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
            nv["counts"][:4] = (0, 0, len(ts), 0)
            self._copy_in()
            return self._meta(1, n, 0, 0, 0, len(ts))

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
        nv["counts"][:4] = (0, 0, len(ts), 0)
        self._copy_in()
        h = self.model(self.dv["input"][:T], self._meta(1, T, 0, 0, 0, len(ts)), self.kv_caches)
        return self.model.compute_logits(h).float()

    # ------------------------------------------------------------------ start-up self-tests
    def _stage_synthetic_decode(self, B: int, ctx_len: int = 24, prefill_len: int = 0, cap_items: int = 0):
        """A deterministic batch of ``B`` decode rows (one page each) — plus, with ``prefill_len``, one fresh prompt of
        that many tokens in the pages behind them — in the static buffers: the same on every TP rank, so no broadcast
        is needed.  Returns the header fields ``(B, T, S, n_tiles, n_items, n_reds)``."""
        nv = self.nv
        i = np.arange(B)
        page = (i % self.num_blocks).astype(np.int32)
        self.n_bt[:B + 1] = 0
        self.n_bt[:B, 0] = page
        self.n_seq_lens[:B] = ctx_len
        self.n_pos[:B] = ctx_len - 1
        self.n_slot[:B] = page * KV_BLOCK_SIZE + ctx_len - 1
        self.n_input[:B] = (i * 7 + 3) % self.cfg.model.vocab_size
        self.n_qsl[:B + 1] = np.arange(B + 1)
        nv["last"][:B] = i
        nv["src"][:B] = -1
        rows, T, n_tiles = B, B, 0
        if prefill_len:
            P = prefill_len
            npg = (P + KV_BLOCK_SIZE - 1) // KV_BLOCK_SIZE
            pages = B + np.arange(npg, dtype=np.int32)
            assert B + npg <= self.num_blocks
            t = np.arange(P)
            self.n_input[B:B + P] = (t * 13 + 5) % self.cfg.model.vocab_size
            self.n_pos[B:B + P] = t
            self.n_slot[B:B + P] = pages[t // KV_BLOCK_SIZE] * KV_BLOCK_SIZE + t % KV_BLOCK_SIZE
            self.n_bt[B, :npg] = pages
            self.n_seq_lens[B] = P
            self.n_qsl[B + 1] = B + P
            nv["last"][B] = B + P - 1
            ts, t0 = ops.build_prefill_tiles(np.array([P], dtype=np.int32), B, self.tile_tokens, self.n_seq_lens[B:B + 1])
            n_tiles = len(ts)
            nv["tile_seq"][:n_tiles] = ts
            nv["tile_q0"][:n_tiles] = t0
            rows, T = B + 1, B + P
        nv["temp"][:rows] = 0.0
        nv["top_p"][:rows] = 1.0
        nv["top_k"][:rows] = 0
        nv["seeds"][:rows] = 1
        nv["steps"][:rows] = 0
        n_items, n_reds, _ = self._stage_decode_work(B, cap_items)
        nv["counts"][:4] = (n_items, n_reds, n_tiles, 0)
        return rows, T, rows, n_tiles, n_items, n_reds

    def _vote(self, ok: bool) -> bool:
        return bool(self.ctx.all_min_int(1 if ok else 0))

    def _compare_logits(self, eager, graph, ids_e, ids_g, extra: dict) -> tuple:
        diff = float((eager - graph).abs().max())
        scale = float(eager.abs().max())
        ok = bool(diff == diff and diff <= 2e-2 * scale + 1e-3)
        return ok, {**extra, "max_abs_logit_diff": round(diff, 6), "logit_scale": round(scale, 4),
                    "greedy_agreement": float((ids_e == ids_g).float().mean())}

    @torch.inference_mode()
    def startup_selftest(self) -> dict:
        """Check the fast paths that change numerics-relevant execution against their plain counterparts, once, on
        every rank of the group; a path that fails (or raises) is switched off for the life of the engine and the
        reason is kept in ``self.selftest`` (bench.py prints it).  Covered: a captured decode hipGraph and a captured
        mixed (prefill + decode) body + head graph pair against the same steps run eagerly (logit-level).  (The
        GEMM/all-reduce pipeline of ``RowParallelLinear`` is checked earlier, by ``selftest_overlap``.)"""
        res = self.selftest
        if not self.on_gpu:
            return res
        # --- hipGraph replay vs eager, same staged batch
        decode_keys = [k for k in self.graphs if isinstance(k, int)]
        if self.cfg.use_hip_graphs and decode_keys:
            try:
                b = min([x for x in decode_keys if x >= 8] or decode_keys)
                rows, T, S, n_tiles, n_items, n_reds = self._stage_synthetic_decode(b, cap_items=self._decode_caps(b)[0])
                self._copy_in()
                le = self._forward_sample(b, b, self._meta(b, b, b, n_items, n_reds, 0)).float().clone()
                ids_e = self.d_sampled[:b].clone()
                self.graphs[b][0].replay()
                ok, res["graph_vs_eager"] = self._compare_logits(le, self.graphs[b][2].float(), ids_e, self.d_sampled[:b],
                                                                 {"bucket": b})
            except Exception as e:
                ok = False
                res["graph_vs_eager"] = {"error": f"{type(e).__name__}: {e}"}
            ok = self._vote(ok)
            res["graph_vs_eager"]["ok"] = ok
            if not ok:
                self.cfg.use_hip_graphs = False
                self.graphs.clear()
                print(f"[dli_amd] hipGraph decode disabled by the start-up self-test: {res['graph_vs_eager']}", flush=True)
        # --- mixed step: body + head graphs vs eager
        mixed_keys = [k[1] for k in self.graphs if isinstance(k, tuple) and k[0] == "m"]
        if self.cfg.use_hip_graphs and self.cfg.mixed_step_graphs and mixed_keys:
            try:
                nd, P = 5, 40
                tb = min([x for x in mixed_keys if x >= nd + P] or [max(mixed_keys)])
                sb = self._bucket_for(nd + 1)
                if tb < nd + P or not sb or self.num_blocks < nd + 3:
                    raise RuntimeError("no bucket fits the synthetic mixed step")
                rows, T, S, n_tiles, n_items, n_reds = self._stage_synthetic_decode(nd, prefill_len=P,
                                                                                    cap_items=self._mixed_caps(tb)[1])
                self._pad_mixed(T, S, tb, sb)
                self._copy_in()
                # eager over the SAME padded token rows (the GEMMs pick their kernels by row count: 45 rows and 64 rows
                # differ by bf16 rounding noise, which is not what this test is about), exact launch extents
                le = self._forward_sample(tb, S, self._meta(rows, tb, nd, n_items, n_reds, n_tiles)).float().clone()
                ids_e = self.d_sampled[:S].clone()
                if ("h", sb) not in self.graphs:
                    self.capture_head_graph(sb)
                self.graphs[("m", tb)][0].replay()
                self.graphs[("h", sb)][0].replay()
                ok, res["mixed_graph_vs_eager"] = self._compare_logits(le, self.graphs[("h", sb)][2][:S].float(), ids_e,
                                                                       self.d_sampled[:S], {"tokens": T, "bucket": tb, "head": sb})
            except Exception as e:
                ok = False
                res["mixed_graph_vs_eager"] = {"error": f"{type(e).__name__}: {e}"}
            ok = self._vote(ok)
            res["mixed_graph_vs_eager"]["ok"] = ok
            if not ok:
                self.cfg.mixed_step_graphs = False
                for k in [k for k in self.graphs if isinstance(k, tuple)]:
                    del self.graphs[k]
                print(f"[dli_amd] mixed-step hipGraphs disabled by the start-up self-test: {res['mixed_graph_vs_eager']}",
                      flush=True)
        return res

    @torch.inference_mode()
    def tune_gemms(self) -> dict:
        """Fill the measured GEMM dispatch table (``ops/gemm_plan.py``) for this model's projections at exactly the row
        counts the captured steps use (decode buckets, mixed token buckets; the LM head at the sampled-row buckets).
        TP rank 0 measures, the group shares its table."""
        from ..ops import gemm_plan
        if not (self.on_gpu and gemm_plan.ENABLED and self.cfg.tune_gemms):
            return {}
        plan = gemm_plan.PLAN
        blob = None
        tunable = None
        if self.cfg.library_gemm_tuning:                     # every rank tunes its own shard shapes (the first GEMM call of a shape does it)
            import torch.cuda.tunable as tunable
            path = self.cfg.library_gemm_tuning if self.ctx.tp_size == 1 else f"{self.cfg.library_gemm_tuning}.rank{self.ctx.tp_rank}"
            t0 = time.perf_counter()
            tunable.enable(True)
            tunable.set_filename(path)
            if os.path.exists(path):
                tunable.read_file(path)
            tunable.tuning_enable(True)
            tunable.set_max_tuning_iterations(20)
            tunable.set_max_tuning_duration(200)
            if hasattr(tunable, "write_file_on_exit"):
                tunable.write_file_on_exit(True)
            if not self.ctx.is_driver:                       # the driver's timing sweep below touches every shape; workers do it here
                self._touch_library_gemms()
        if self.ctx.is_driver:
            from ..models.llama import Attention, SwiGLUMLP
            groups: dict = {}

            def add(name, w, swiglu):
                groups.setdefault((name, tuple(w.shape), swiglu), []).append(w)
            for mod in self.model.modules():
                if self.cfg.quantization:                    # rows above the skinny range run W8A8: nothing to choose
                    break
                if isinstance(mod, Attention):
                    add("qkv", mod.qkv.weight.data, False)
                    add("o", mod.o.weight.data, False)
                elif isinstance(mod, SwiGLUMLP):
                    add("gate_up", mod.gate_up.weight.data, True)
                    add("down", mod.down.weight.data, False)
            B = self.cfg.sched.max_num_seqs
            decode_rows = [b for b in self.cfg.graph_batch_sizes if b <= B]
            rows = sorted(set(decode_rows) | set(self.mixed_buckets if self.cfg.mixed_step_graphs else ()))
            plan.tune([(k[0], v, k[2]) for k, v in groups.items()], rows, self.device,
                      budget_s=300.0 if tunable is not None else 60.0)   # a first start with library tuning spends seconds per shape
            head = getattr(self.model, "lm_head", None)
            if head is not None and getattr(head, "weight", None) is not None:
                plan.tune([("lm_head", [head.weight.data], False)], decode_rows, self.device, budget_s=20.0)
            blob = plan.export()
        if self.ctx.distributed:
            blob = self.ctx.broadcast_object(blob, src=0)
            if not self.ctx.is_driver:
                plan.load(blob)
        self.selftest["gemm_plan"] = plan.summary() if self.ctx.is_driver else {"entries": sum(len(v) - 1 for v in plan.table.values())}
        if tunable is not None:
            tunable.tuning_enable(False)                     # shapes first seen later (eager fall-backs) take the default solution
            try:
                if hasattr(tunable, "write_file"):
                    tunable.write_file()
            except Exception as e:                           # a read-only location: the tuning still holds for this process
                print(f"[dli_amd] could not write the TunableOp results: {e}", flush=True)
            self.selftest["gemm_plan"]["library_tuning"] = {"file": self.cfg.library_gemm_tuning, "entries": len(tunable.get_results()),
                                                            "seconds": round(time.perf_counter() - t0, 1)}
        return self.selftest["gemm_plan"]

    def _touch_library_gemms(self) -> None:
        """One library GEMM per (projection shape, captured row count): what TunableOp needs to see to tune them."""
        from ..models.layers import ColumnParallelLinear, RowParallelLinear
        import torch.nn.functional as F
        B = self.cfg.sched.max_num_seqs
        rows = sorted({b for b in self.cfg.graph_batch_sizes if 64 < b <= B} | set(self.mixed_buckets if self.cfg.mixed_step_graphs else ()))
        seen = set()
        for mod in self.model.modules():
            if isinstance(mod, (ColumnParallelLinear, RowParallelLinear)) and tuple(mod.weight.data.shape) not in seen:
                seen.add(tuple(mod.weight.data.shape))
                for m in rows:
                    F.linear(torch.zeros(m, mod.weight.data.shape[1], device=self.device, dtype=self.dtype), mod.weight.data)
        torch.cuda.synchronize()

    @torch.inference_mode()
    def selftest_overlap(self) -> dict:
        """GEMM / all-reduce pipeline of ``RowParallelLinear`` against GEMM → all-reduce, then the timing gate.  Runs
        BEFORE the graphs are captured: the captured steps hold whichever path this leaves switched on."""
        res = self.selftest
        if not self.on_gpu:
            return res
        if self.ctx.distributed:
            from ..models.layers import RowParallelLinear
            layer = next((m for m in self.model.modules() if isinstance(m, RowParallelLinear) and m.reduce), None)
            if layer is None or not layer.overlap_capable():
                res["overlapped_row_parallel"] = {"ok": None, "skipped": f"device backend {self.ctx.device_backend}, no custom all-reduce"}
                RowParallelLinear.OVERLAP_OK = False
            else:
                try:
                    g = torch.Generator(device=self.device).manual_seed(4321 + self.ctx.tp_rank)
                    rows = min(max(RowParallelLinear.OVERLAP_MIN_ROWS, 2048), max(self.max_T, RowParallelLinear.OVERLAP_MIN_ROWS))
                    x = (torch.randn(rows, layer.weight.data.shape[1], generator=g, device=self.device) * 0.5).to(self.dtype)
                    want = self.ctx.all_reduce(ops.linear(x, layer.weight.data)).float()
                    got = layer._forward_overlapped(x, None).float()
                    torch.cuda.synchronize()
                    diff = float((want - got).abs().max())
                    scale = float(want.abs().max())
                    ok = bool(diff == diff and diff <= 3e-2 * scale + 1e-3)
                    ar = self.ctx.custom_ar
                    res["overlapped_row_parallel"] = {"max_abs_diff": round(diff, 6), "scale": round(scale, 4), "rows": rows,
                                                      "chunk_rows": layer._chunk_rows(rows, want.shape[1] * 2),
                                                      "reduced_by": "custom IPC kernel on the side stream"
                                                      if (ar is not None and ar.enabled) else "RCCL on the side stream"}
                except Exception as e:
                    ok = False
                    res["overlapped_row_parallel"] = {"error": f"{type(e).__name__}: {e}"}
                ok = self._vote(ok)
                res["overlapped_row_parallel"]["ok"] = ok
                RowParallelLinear.OVERLAP_OK = ok
                if not ok:
                    print(f"[dli_amd] GEMM/all-reduce overlap disabled by the start-up self-test: "
                          f"{res['overlapped_row_parallel']}", flush=True)
                elif os.environ.get("DLI_OVERLAP_FORCE") != "1":
                    self._gate_overlap_by_time(layer, x, res["overlapped_row_parallel"])
        return res

    @torch.inference_mode()
    def selftest_sp_norm(self) -> dict:
        """Sequence-parallel reduce + add + norm kernel (``ParallelContext.sp_step``) against all-reduce →
        ``fused_add_rms_norm`` on random data of a large step's shape, on every rank; then both are timed and the fused
        path stays on only if it is not slower.  Runs before graph capture, like ``selftest_overlap``."""
        res, ctx = self.selftest, self.ctx
        mc = self.model.cfg
        H = mc.hidden_size
        rows = min(self.max_T, 2048)
        if not self.on_gpu or ctx.tp_size == 1 or not ctx.sp_ok:
            return res
        if ctx.emulated or not ctx.sp_wanted(rows, H, True):
            res["sp_add_norm"] = {"ok": None, "skipped": "emulated rank" if ctx.emulated else "no custom all-reduce / step too small"}
            return res
        ar, eps = ctx.custom_ar, mc.rms_eps
        lo, hi, _ = ctx.sp_rows(rows)
        try:
            g = torch.Generator(device=self.device).manual_seed(977 + ctx.tp_rank)
            y = (torch.randn(rows, H, generator=g, device=self.device) * 0.5).to(self.dtype)
            g0 = torch.Generator(device=self.device).manual_seed(55)
            resid = torch.randn(rows, H, generator=g0, device=self.device).to(self.dtype)
            w = (1 + 0.1 * torch.randn(H, generator=g0, device=self.device)).to(self.dtype)

            def plain(r):
                o = ctx.all_reduce(y.clone())
                ops.fused_add_rms_norm(o, r, w, eps)
                return o

            def fused(r):
                st = ar.staging_tensor(rows, H, y)
                st.copy_(y)
                return ar.all_reduce_add_norm(st, r, w, eps)
            r0, r1 = resid.clone(), resid.clone()
            want, got = plain(r0).float(), fused(r1).float()
            torch.cuda.synchronize()
            diff = max(float((want - got).abs().max()), float((r0[lo:hi].float() - r1[lo:hi].float()).abs().max()))
            scale = float(want.abs().max())
            ok = bool(diff == diff and diff <= 3e-2 * scale + 1e-3)
            res["sp_add_norm"] = {"max_abs_diff": round(diff, 6), "scale": round(scale, 4), "rows": rows}
            if ok:
                def timed(fn, iters=10):
                    r = resid.clone()
                    fn(r)
                    torch.cuda.synchronize()
                    ctx.barrier()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(iters):
                        fn(r)
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) * 1e3 / iters
                t = ctx.broadcast_object((timed(plain), timed(fused)), src=0)
                res["sp_add_norm"].update(plain_us=round(t[0], 1), fused_us=round(t[1], 1))
                if os.environ.get("DLI_SP_FORCE") != "1" and t[1] > 1.03 * t[0]:
                    res["sp_add_norm"]["enabled"] = False
                    ctx.sp_ok = False
        except Exception as e:
            ok = False
            res["sp_add_norm"] = {"error": f"{type(e).__name__}: {e}"}
        ok = self._vote(ok)
        res["sp_add_norm"]["ok"] = ok
        if not ok:
            ctx.sp_ok = False
            print(f"[dli_amd] sequence-parallel add-norm disabled by the start-up self-test: {res['sp_add_norm']}", flush=True)
        res["sp_add_norm"].setdefault("enabled", ctx.sp_ok)
        return res

    def _gate_overlap_by_time(self, layer, x, rec: dict) -> None:
        """Measured, not guessed: the pipeline stays on only where it beats the plain GEMM → all-reduce sequence on THIS
        system (two chunk counts tried; TP rank 0's clock decides for the group).  With every rank on its own GPU the
        reduction hides behind the next chunk's GEMM; with ranks sharing one GPU (rehearsals) there is no link time to
        hide and the co-resident kernels only take CUs from each other (profiles/overlap_ab_r3i.jsonl) — the same code
        then switches itself off.  ``DLI_OVERLAP_FORCE=1`` skips the gate (tests of the path itself)."""
        from ..models.layers import RowParallelLinear

        def timed(fn, iters=5):
            fn()
            torch.cuda.synchronize()
            self.ctx.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / iters

        try:
            # one layer per distinct shard shape (attention output and MLP down projection): the verdict is on their sum
            layers, seen = [], set()
            for m in self.model.modules():
                if isinstance(m, RowParallelLinear) and m.reduce and tuple(m.weight.data.shape) not in seen:
                    seen.add(tuple(m.weight.data.shape))
                    layers.append((m, torch.randn(x.shape[0], m.weight.data.shape[1], device=self.device).to(self.dtype)))
            plain = sum(timed(lambda: self.ctx.all_reduce(ops.linear(xi, m.weight.data))) for m, xi in layers)
            piped = {}
            for chunks in (2, 4):
                RowParallelLinear.OVERLAP_CHUNKS = chunks
                piped[chunks] = sum(timed(lambda: m._forward_overlapped(xi, None)) for m, xi in layers)
            verdict = self.ctx.broadcast_object((plain, piped), src=0)
        except Exception as e:
            verdict = None
            rec["timing_error"] = f"{type(e).__name__}: {e}"
        if not self._vote(verdict is not None):
            RowParallelLinear.OVERLAP_OK = False
            rec["enabled"] = False
            return
        plain, piped = verdict
        best = min(piped, key=piped.get)
        RowParallelLinear.OVERLAP_CHUNKS = best
        RowParallelLinear.OVERLAP_OK = bool(piped[best] < 0.97 * plain)
        rec.update(plain_us=round(plain, 1), pipelined_us={str(k): round(v, 1) for k, v in piped.items()}, chunks=best,
                   enabled=RowParallelLinear.OVERLAP_OK)
        if not RowParallelLinear.OVERLAP_OK:
            print(f"[dli_amd] GEMM/all-reduce overlap verified but not faster here ({piped[best]:.0f} µs against "
                  f"{plain:.0f} µs plain at {x.shape[0]} rows): off", flush=True)

    def capture_all(self, buckets=None, mixed=None):
        """Pre-capture the decode graphs, then the mixed-step body graphs and the head graphs (largest first so that the
        pool is sized once).  Under TP every rank must call this with the same buckets (the captures contain
        collectives).  Captures run on an empty step: no work item, no tile, every KV write suppressed."""
        if not (self.on_gpu and self.cfg.use_hip_graphs):
            return 0
        todo = [b for b in (buckets or self.cfg.graph_batch_sizes) if b <= self.cfg.sched.max_num_seqs]
        err = None
        try:
            self.nv["counts"][:] = 0
            for bucket in sorted(todo, reverse=True):
                if bucket in self.graphs:
                    continue
                self._pad_decode(0, bucket)
                self._copy_in()
                self.capture_graph(bucket)
            if self.cfg.mixed_step_graphs:
                for tb in sorted(self.mixed_buckets if mixed is None else mixed, reverse=True):
                    if ("m", tb) in self.graphs:
                        continue
                    self._pad_mixed(0, 0, tb, 0)
                    self._copy_in()
                    self.capture_mixed_graph(tb)
                for sb in sorted(todo, reverse=True):
                    if ("h", sb) not in self.graphs:
                        self._pad_sampler_rows(0, sb)
                        self.nv["last"][:sb] = 0
                        self._copy_in()
                        self.capture_head_graph(sb)
            torch.cuda.synchronize()
        except Exception as e:                               # e.g. a collective that cannot be captured on this backend
            err = f"{type(e).__name__}: {e}"
            # say so NOW: peers that did not fail may sit in a collective of their next warm-up run until its time-out,
            # and the vote below only completes after that
            print(f"[dli_amd] tp rank {self.ctx.tp_rank}: graph capture failed: {err}", flush=True)
        if not self._vote(err is None):                      # all ranks keep their graphs, or none does
            self.selftest["graph_capture"] = {"ok": False, "error": err or "failed on another rank"}
            print(f"[dli_amd] hipGraph capture failed ({self.selftest['graph_capture']['error']}): steps run eagerly",
                  flush=True)
            self.cfg.use_hip_graphs = False
            self.graphs.clear()
        return len(self.graphs)

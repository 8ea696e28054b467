"""Python side of the custom xGMI all-reduce (kernel: ``csrc/kernels/custom_all_reduce.hip``, K7).

Set-up (once per TP group): every rank allocates a staging buffer, a result buffer (plain device
memory) and a flag block (uncached memory), exports them as HIP IPC handles, the handles are
all-gathered over the CPU (gloo) group and every rank maps its peers' buffers.  A self-test against
sums exchanged over the CPU group decides whether the fast path is enabled; afterwards ``all_reduce``
/ ``all_gather`` / ``broadcast`` are single kernel launches on the current stream (capturable in the
decode hipGraph) and a TP step needs no library collective at all.

Launch shape is *measured*, not guessed: :meth:`CustomAllReduce.tune` sweeps workgroup counts and the
one-/two-shot algorithms over 8 KB … ``max_bytes`` inside hipGraphs (what the decode step sees), times
the library (RCCL) all-reduce on the same sizes when the device group is ``nccl``, installs the
winning (algorithm, workgroups) per size bucket on every rank and sets ``use_max_bytes`` to the size
where the library starts to win (bandwidth-optimised protocols win for prefill-sized
messages).  The sweep runs at start-up of every TP group, so numbers for real xGMI peers appear the
first time more than one GPU is visible; ``tuning`` keeps the table for the benchmark JSON.
"""
from __future__ import annotations

import os
import time

import torch
import torch.distributed as dist

from .. import ops
from .state import drain_collectives

_SIZES = (8 << 10, 32 << 10, 128 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20, 64 << 20)


class CustomAllReduce:
    def __init__(self, ctx, device, max_bytes: int = 64 << 20):
        lib = ops.hip()
        self.lib, self.ctx, self.device, self.max_bytes = lib, ctx, device, max_bytes
        self.use_max_bytes = max_bytes                      # custom path up to here, library above (set by tune())
        self.world, self.rank = ctx.tp_size, ctx.tp_rank
        self.tuning: dict = {}
        # Workgroups one call may launch.  The kernels spin until the same block of every peer has arrived, so all
        # peers' blocks must be able to be resident at once: trivially true with one GPU per rank; with several ranks
        # on ONE device (rehearsals) world × blocks workgroups of 512 threads share its 1024 slots with the GEMMs —
        # past about half of them a late rank's kernel cannot be scheduled and everybody times out waiting for it.
        n_dev = max(torch.cuda.device_count(), 1)
        sharing = (self.world + n_dev - 1) // n_dev if n_dev < self.world else 1
        self.block_cap = int(lib.ar_max_blocks()) if sharing == 1 else max(8, min(int(lib.ar_max_blocks()), 512 // sharing // 2))
        torch.cuda.set_device(device)
        group = self._cpu_group()
        # Set-up is collective, so a failure on one rank must surface on every rank at the same exchange: each phase
        # runs under try, the outcome travels with the next all-gather, and all ranks raise together (→ create() returns
        # None everywhere) instead of one rank leaving while its peers wait for it.
        own, handles, err = [], None, ""
        try:
            own = [lib.ar_alloc(max_bytes, False), lib.ar_alloc(max_bytes, False), lib.ar_alloc(lib.ar_signal_bytes(), True)]
            handles = [lib.ar_ipc_handle(p).tolist() for p in own]
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        self._own = own
        gathered = [None] * self.world
        dist.all_gather_object(gathered, handles if not err else {"error": err}, group=group)
        bad = {r: g["error"] for r, g in enumerate(gathered) if isinstance(g, dict)}
        if bad:
            raise RuntimeError(f"buffer allocation / IPC export failed on ranks {bad}")
        tables = [[], [], []]
        self._opened = []
        err = ""
        try:
            for r in range(self.world):
                for k in range(3):
                    if r == self.rank:
                        tables[k].append(own[k])
                    else:
                        p = lib.ar_ipc_open(torch.tensor(gathered[r][k], dtype=torch.uint8))
                        self._opened.append(p)
                        tables[k].append(p)
            t = [torch.tensor(x, dtype=torch.int64) for x in tables]
            self.state = lib.ar_create(self.rank, self.world, t[0], t[1], t[2], max_bytes)
        except Exception as e:
            err = f"{type(e).__name__}: {e}"
        errs = [None] * self.world
        dist.all_gather_object(errs, err, group=group)
        if any(errs):
            raise RuntimeError(f"mapping the peers' buffers failed on ranks { {r: e for r, e in enumerate(errs) if e} }")
        self.enabled = True
        self._err_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        dist.barrier(group=group)

    def _cpu_group(self):
        return self.ctx.cpu_group if self.ctx.cpu_group is not None else self.ctx.group

    @classmethod
    def create(cls, ctx, device, max_bytes: int = 64 << 20, tune: bool | None = None, tuning_rows: list | None = None):
        """Returns a verified (and tuned) instance, or ``None`` (→ library collectives only) if set-up or the
        self-test fails on any rank.  ``tuning_rows``: the launch-shape table a first-contact probe of the SAME group
        already measured (``comm_probe``) — installed as is instead of sweeping a second time."""
        try:
            ar = cls(ctx, device, max_bytes)
            if not ar.self_test():
                print(f"[dli_amd] custom all-reduce self-test FAILED on tp rank {ctx.tp_rank}: using the library collectives",
                      flush=True)
                return None
            if tune is None:
                tune = os.environ.get("DLI_AR_TUNE", "1") != "0"
            if tuning_rows:
                ar.install_table(ctx.broadcast_object(tuning_rows, src=0), source="probe")
            elif tune:
                ar.tune()
            return ar
        except Exception as e:                              # IPC unavailable, > 8 ranks, …
            print(f"[dli_amd] custom all-reduce disabled: {type(e).__name__}: {e}", flush=True)
            return None

    # ------------------------------------------------------------------ collectives
    def should_use(self, x: torch.Tensor) -> bool:
        return (self.enabled and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
                and 0 < x.numel() * 2 <= self.use_max_bytes and (x.numel() * 2) % 16 == 0 and x.data_ptr() % 16 == 0)

    def fits(self, nbytes: int) -> bool:
        return self.enabled and 0 < nbytes <= self.use_max_bytes and nbytes % 16 == 0

    def staging_tensor(self, rows: int, cols: int, like: torch.Tensor) -> torch.Tensor:
        """``[rows, cols]`` bf16 view of this rank's IPC staging buffer.  A GEMM that writes its output here has done the
        all-reduce's copy-in pass already (the kernels skip it when they are handed this pointer).  Valid until the next
        collective of this group writes the buffer: produce → ``all_reduce`` → done."""
        return self.lib.ar_staging_tensor(self.state, rows, cols, like)

    @property
    def one_shot_max_bytes(self) -> int:
        return int(self.tuning.get("one_shot_up_to_bytes", 256 << 10)) if self.tuning else (256 << 10)

    def all_reduce_add_norm(self, staged: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
        """Sequence-parallel all-reduce → residual add → RMSNorm in one kernel (``all_reduce_add_norm_kernel``).  ``staged``:
        the partial sums, a ``staging_tensor`` view.  ``residual`` is updated IN PLACE for this rank's rows only
        (rows ``[rank·⌈T/world⌉, …)``): afterwards its other rows are stale by design.  Returns the normalised ``[T, H]``."""
        out = torch.empty_like(residual)
        self.lib.ar_all_reduce_add_norm(self.state, staged, residual, weight, eps, out, min(64, self.block_cap))
        return out

    def all_reduce(self, x: torch.Tensor, algo: int = 0, blocks: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
        """Out-of-place sum over the TP group (``algo``: 0 tuned/auto, 1 one-shot, 2 two-shot; ``blocks``: 0 tuned, else
        a cap on the workgroups — every rank must pass the same value)."""
        if out is None:
            out = torch.empty_like(x)
        self.lib.ar_all_reduce(self.state, x, out, algo, min(blocks, self.block_cap) if blocks else 0)
        return out

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """``[world, *x.shape]`` (rank-major) — same IPC protocol, no library collective."""
        out = torch.empty((self.world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        self.lib.ar_all_gather(self.state, x, out)
        return out

    # ------------------------------------------------------------------ expert parallelism
    def ep_tokens_per_call(self, hidden: int, top_k: int) -> int:
        """Most tokens of a step one ``ep_moe`` call can take: in the worst routing every (token, expert) row lands on ONE
        rank, whose result buffer (``max_bytes``) must hold them all."""
        t = self.max_bytes // (top_k * hidden * 2)
        forced = int(os.environ.get("DLI_EP_TOKENS_PER_CALL", "0"))      # tests: exercise the several-calls-per-step path
        if forced > 0:
            t = min(t, forced)
        return max(t // (self.world * 8) * (self.world * 8), 0)

    def ep_moe(self, x_l: torch.Tensor, logits_l: torch.Tensor, w13: torch.Tensor, w2: torch.Tensor, top_k: int,
               n_cap: int) -> torch.Tensor:
        """Expert-parallel MoE block for this rank's token slice ``x_l`` (see ``csrc/kernels/custom_all_reduce.hip``:
        device-side routing, dispatch, local experts, combine — no host synchronisation, hipGraph-capturable).
        ``n_cap``: the largest (tokens · top_k) any rank passes in this call."""
        return self.lib.ep_moe(self.state, x_l.contiguous(), logits_l.contiguous(), w13, w2, top_k, n_cap, self.world, self.rank)

    def can_broadcast(self, buf: torch.Tensor) -> bool:
        nbytes = buf.numel() * buf.element_size()
        return (self.enabled and buf.is_cuda and buf.is_contiguous() and 0 < nbytes <= self.max_bytes
                and nbytes % 16 == 0 and buf.data_ptr() % 16 == 0)

    def broadcast(self, buf: torch.Tensor, root: int = 0) -> torch.Tensor:
        """In place: every rank's ``buf`` becomes the root's."""
        self.lib.ar_broadcast(self.state, buf, root)
        return buf

    # ------------------------------------------------------------------ failure detection
    def error(self) -> bool:
        """Blocking read of the kernels' timeout word (tests, shutdown)."""
        return bool(self.lib.ar_error(self.state))

    def queue_error_check(self) -> None:
        """Stream-ordered copy of the timeout word into pinned memory; read ``error_seen`` after the step's event."""
        self.lib.ar_error_async(self.state, self._err_host)

    @property
    def error_seen(self) -> bool:
        return bool(self._err_host[0].item())

    # ------------------------------------------------------------------ self-test
    def self_test(self) -> bool:
        """Compare against sums exchanged over the CPU group (works whatever the device backend is)."""
        ok = True
        group = self._cpu_group()
        g = torch.Generator().manual_seed(1234 + self.rank)
        for numel, algo in ((4096, 1), (1 << 20, 2), (3 * 4096 + 8, 2), (1 << 18, 0), (1 << 18, 1)):
            if numel * 2 > self.max_bytes:
                continue
            x_cpu = torch.randn(numel, generator=g).to(torch.bfloat16)
            parts = [None] * self.world
            dist.all_gather_object(parts, x_cpu, group=group)
            want = torch.stack([p.float() for p in parts]).sum(0)
            x = x_cpu.to(self.device)
            got = self.all_reduce(x, algo).float().cpu()
            gathered = self.all_gather(x).cpu()
            ok &= bool(torch.allclose(got, want, rtol=2e-2, atol=2e-2)) and not self.error()
            ok &= all(torch.equal(gathered[r], parts[r]) for r in range(self.world))
            b = (x if self.rank == 0 else torch.zeros_like(x)).view(torch.int32).clone()
            self.broadcast(b, 0)
            ok &= bool(torch.equal(b.cpu(), parts[0].view(torch.int32)))
        votes = [None] * self.world
        dist.all_gather_object(votes, bool(ok), group=group)
        self.enabled = all(votes)
        return self.enabled

    # ------------------------------------------------------------------ launch-shape sweep
    def _time_graph(self, fn, iters: int) -> float:
        """µs per call of ``fn`` replayed ``iters`` times inside one hipGraph (pure GPU time, as in a decode step)."""
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        drain_collectives()
        with torch.cuda.graph(g, stream=s, capture_error_mode=ops.CAPTURE_MODE):
            for _ in range(iters):
                fn()
        torch.cuda.synchronize()
        dist.barrier(group=self._cpu_group())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()                                           # aligns the GPUs of the group (host skew ends up in here) …
        a.record()
        g.replay()                                           # … so this one, queued behind it, is pure kernel time
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / iters

    def _time_eager(self, fn, iters: int) -> float:
        fn()
        torch.cuda.synchronize()
        dist.barrier(group=self._cpu_group())
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(iters):                               # aligns the GPUs of the group
            fn()
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / iters

    def _library_capturable(self) -> bool:
        """Can every rank capture a library all-reduce in a hipGraph?  Capture only — nothing is replayed before the
        group has voted, so a rank whose capture failed never leaves its peers waiting inside a collective."""
        ok = True
        try:
            x = torch.ones(4096, dtype=torch.bfloat16, device=self.device)
            dist.all_reduce(x, group=self.ctx.group)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            drain_collectives()
            with torch.cuda.graph(g, stream=s, capture_error_mode=ops.CAPTURE_MODE):
                dist.all_reduce(x, group=self.ctx.group)
            torch.cuda.synchronize()
        except Exception:
            ok = False
        votes = [None] * self.world
        dist.all_gather_object(votes, ok, group=self._cpu_group())
        return all(votes)

    def tune(self, sizes=_SIZES, iters: int = 10, with_library: bool | None = None, in_graph: bool = True) -> dict:
        """Sweep (algorithm × workgroups) per message size; install the winners on every rank.

        Every rank runs the same launches in the same order (the kernels synchronise peer to peer); the table is
        decided from TP rank 0's timings and broadcast, so the groups' launch shapes can never diverge."""
        t_begin = time.perf_counter()
        lib = self.lib
        max_blocks = self.block_cap
        sizes = [s for s in sizes if s <= self.max_bytes]
        if with_library is None:
            with_library = dist.get_backend(self.ctx.group) == "nccl"
        timer = self._time_graph if in_graph else self._time_eager
        lib_timer = timer
        if with_library and in_graph and not self._library_capturable():
            lib_timer = self._time_eager                     # agreed by the whole group: nobody replays what a peer lacks
        rows = []
        for nbytes in sizes:
            x = torch.ones(nbytes // 2, dtype=torch.bfloat16, device=self.device)
            out = torch.empty_like(x)
            cands = []
            need = max((nbytes // 16 + 511) // 512, 1)
            block_opts = sorted({min(b, need, max_blocks) for b in (8, 16, 32, 64, 128)})
            for algo in (1, 2):
                if algo == 1 and nbytes > (4 << 20) and self.world > 2:
                    continue                                 # one-shot moves world × the bytes: hopeless up here
                if algo == 2 and (self.world <= 2 or nbytes < (32 << 10)):
                    continue
                for blocks in block_opts:
                    us = timer(lambda: lib.ar_all_reduce(self.state, x, out, algo, blocks), iters)
                    cands.append({"algo": algo, "blocks": blocks, "us": round(us, 2)})
            row = {"bytes": nbytes, "candidates": cands}
            if with_library:
                # the same clock for both sides: RCCL replayed inside a hipGraph like the custom kernels (an eager
                # launch carries ≈ 10 µs of host work per call that a captured decode step does not pay); eager only
                # where the group found that the library cannot be captured
                fn_lib = lambda: dist.all_reduce(x, group=self.ctx.group)
                try:
                    row["library_us"] = round(lib_timer(fn_lib, iters), 2)
                    row["library_timed_in_graph"] = lib_timer is self._time_graph
                except Exception as e:                       # the sweep must never take the engine down
                    row["library_error"] = f"{type(e).__name__}: {e}"
            rows.append(row)
        # a time-out on ANY rank ends the sweep on EVERY rank at this same point (a rank that raised alone would leave
        # its peers waiting in the broadcast below, and the group with custom all-reduce on some ranks only)
        flags = [None] * self.world
        dist.all_gather_object(flags, bool(self.error()), group=self._cpu_group())
        if any(flags):
            raise RuntimeError(f"custom all-reduce kernel timed out during the launch-shape sweep (ranks {[i for i, f in enumerate(flags) if f]})")
        rows = self.ctx.broadcast_object(rows, src=0)        # rank 0's clock decides for everyone
        self.install_table(rows, source="sweep", timed_in_graph=in_graph, seconds=time.perf_counter() - t_begin)
        dist.barrier(group=self._cpu_group())
        return self.tuning

    def install_table(self, rows: list, source: str = "sweep", timed_in_graph: bool = True, seconds: float = 0.0) -> dict:
        """Per size bucket: the fastest (algorithm, workgroups); ``use_max_bytes`` = where the library starts to win on
        every larger size.  ``rows`` must be identical on every rank of the group."""
        thr, algos, blocks = [], [], []
        use_max = self.max_bytes
        lib_wins_from = None
        for row in rows:
            best = min(row["candidates"], key=lambda c: c["us"])
            row["best"] = best
            thr.append(row["bytes"])
            algos.append(best["algo"])
            blocks.append(best["blocks"])
            if "library_us" in row:
                if row["library_us"] < 0.95 * best["us"]:
                    lib_wins_from = row["bytes"] if lib_wins_from is None else lib_wins_from
                else:
                    lib_wins_from = None                     # must win on every larger size too
        if lib_wins_from is not None:
            prev = [r["bytes"] for r in rows if r["bytes"] < lib_wins_from]
            use_max = prev[-1] if prev else 0
        one_shot_max = max([r["bytes"] for r in rows if r["best"]["algo"] == 1], default=0)
        blocks = [min(b, self.block_cap) for b in blocks]
        self.lib.ar_set_tuning(self.state, torch.tensor(thr, dtype=torch.int64), torch.tensor(algos, dtype=torch.int32),
                               torch.tensor(blocks, dtype=torch.int32), min(int(blocks[-1]) if blocks else 64, self.block_cap),
                               int(one_shot_max))
        self.use_max_bytes = use_max
        self.tuning = {"world": self.world, "source": source, "timed_in_graph": timed_in_graph, "rows": rows,
                       "use_custom_up_to_bytes": use_max, "one_shot_up_to_bytes": one_shot_max,
                       "sweep_seconds": round(seconds, 2)}
        return self.tuning

    def close(self):
        if getattr(self, "state", None):
            self.lib.ar_destroy(self.state)
            self.state = None

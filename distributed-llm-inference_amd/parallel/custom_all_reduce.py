"""Python side of the custom xGMI all-reduce (kernel: ``csrc/kernels/custom_all_reduce.hip``, K7).

Set-up (once per TP group): every rank allocates a staging buffer, a result buffer (plain device
memory) and a flag block (uncached memory), exports them as HIP IPC handles, the handles are
all-gathered over the CPU (gloo) group and every rank maps its peers' buffers.  A self-test against
RCCL decides whether the fast path is enabled; afterwards ``all_reduce`` is a single kernel launch on
the current stream (capturable in the decode hipGraph).

Used for bf16 messages up to ``max_bytes`` (decode-sized: B·H·2 bytes ≤ 8 MiB); larger prefill
messages stay on RCCL, whose bandwidth-optimised protocols win there.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .. import ops


class CustomAllReduce:
    def __init__(self, ctx, device, max_bytes: int = 32 << 20):
        lib = ops.hip()
        self.lib, self.ctx, self.device, self.max_bytes = lib, ctx, device, max_bytes
        self.world, self.rank = ctx.tp_size, ctx.tp_rank
        torch.cuda.set_device(device)
        own = [lib.ar_alloc(max_bytes, False), lib.ar_alloc(max_bytes, False), lib.ar_alloc(lib.ar_signal_bytes(), True)]
        self._own = own
        handles = [lib.ar_ipc_handle(p).tolist() for p in own]
        gathered = [None] * self.world
        group = ctx.cpu_group if ctx.cpu_group is not None else ctx.group
        dist.all_gather_object(gathered, handles, group=group)
        tables = [[], [], []]
        for r in range(self.world):
            for k in range(3):
                if r == self.rank:
                    tables[k].append(own[k])
                else:
                    tables[k].append(lib.ar_ipc_open(torch.tensor(gathered[r][k], dtype=torch.uint8)))
        t = [torch.tensor(x, dtype=torch.int64) for x in tables]
        self.state = lib.ar_create(self.rank, self.world, t[0], t[1], t[2], max_bytes)
        self.enabled = True
        dist.barrier(group=group)

    @classmethod
    def create(cls, ctx, device, max_bytes: int = 32 << 20):
        """Returns a verified instance, or ``None`` (→ RCCL only) if set-up or the self-test fails."""
        try:
            ar = cls(ctx, device, max_bytes)
            return ar if ar.self_test() else None
        except Exception as e:                              # IPC unavailable, > 8 ranks, …
            print(f"[dli_amd] custom all-reduce disabled: {e}")
            return None

    def should_use(self, x: torch.Tensor) -> bool:
        return (self.enabled and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()
                and 0 < x.numel() * 2 <= self.max_bytes and (x.numel() * 2) % 16 == 0 and x.data_ptr() % 16 == 0)

    def all_reduce(self, x: torch.Tensor, algo: int = 0) -> torch.Tensor:
        """Out-of-place sum over the TP group (``algo``: 0 auto, 1 one-shot, 2 two-shot)."""
        out = torch.empty_like(x)
        self.lib.ar_all_reduce(self.state, x, out, algo)
        return out

    def all_gather(self, x: torch.Tensor) -> torch.Tensor:
        """``[world, *x.shape]`` (rank-major) — same IPC protocol, no library collective."""
        out = torch.empty((self.world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        self.lib.ar_all_gather(self.state, x, out)
        return out

    def error(self) -> bool:
        return bool(self.lib.ar_error(self.state))

    def self_test(self) -> bool:
        """Compare against sums exchanged over the CPU group (works whatever the device backend is)."""
        ok = True
        group = self.ctx.cpu_group if self.ctx.cpu_group is not None else self.ctx.group
        g = torch.Generator().manual_seed(1234 + self.rank)
        for numel, algo in ((4096, 1), (1 << 20, 2), (3 * 4096 + 8, 2), (1 << 18, 0)):
            x_cpu = torch.randn(numel, generator=g).to(torch.bfloat16)
            parts = [None] * self.world
            dist.all_gather_object(parts, x_cpu, group=group)
            want = torch.stack([p.float() for p in parts]).sum(0)
            x = x_cpu.to(self.device)
            got = self.all_reduce(x, algo).float().cpu()
            gathered = self.all_gather(x).cpu()
            ok &= bool(torch.allclose(got, want, rtol=2e-2, atol=2e-2)) and not self.error()
            ok &= all(torch.equal(gathered[r], parts[r]) for r in range(self.world))
        votes = [None] * self.world
        dist.all_gather_object(votes, ok, group=group)
        self.enabled = all(votes)
        return self.enabled

    def close(self):
        if getattr(self, "state", None):
            self.lib.ar_destroy(self.state)
            self.state = None

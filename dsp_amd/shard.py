"""One process per GPU: contiguous sharding of independent streams over the ranks of a torch.distributed
job (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path shards naturally -- streams never interact -- so there is NO data-plane
collective.  Collectives are used only for (i) broadcasting the chain description + filter taps from rank 0,
(ii) gathering per-stream digests, (iii) max/sum reduction of timings and sample counts.
"""
import hashlib
import os

import numpy as np


def stream_range(n_streams, rank, world):
    """Contiguous block of streams owned by `rank` (stream s -> rank s * world // n_streams)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    lo = n_streams * rank // world
    hi = n_streams * (rank + 1) // world
    return lo, hi


def owner_of(stream, n_streams, world):
    for r in range(world):
        lo, hi = stream_range(n_streams, r, world)
        if lo <= stream < hi:
            return r
    raise ValueError("stream out of range")


class Job:
    """Thin wrapper over torch.distributed that degrades to a single process when WORLD_SIZE == 1."""

    def __init__(self, backend=None, device=None):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = device
        self.dist = None
        self.backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if self.backend == "gloo":
            self.device = None              # gloo moves host tensors: the few bytes of setup / digests / timings go through the CPU
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if self.backend == "nccl" and device is not None:
                    kw["device_id"] = device
                dist.init_process_group(backend=self.backend, **kw)
            self.dist = dist

    def _dev(self):
        return self.device if self.device is not None else "cpu"

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def broadcast_setup(self, chain, taps):
        """Rank 0's chain text and filter taps to everyone (a few hundred bytes + <= 1 MiB)."""
        t = self.torch
        if self.dist is None:
            return chain, (None if taps is None else np.asarray(taps, dtype=np.float64))
        meta = t.zeros(2, dtype=t.int64, device=self._dev())
        payload = chain.encode() if self.rank == 0 else b""
        if self.rank == 0:
            meta[0] = len(payload)
            meta[1] = -1 if taps is None else len(taps)
        self.dist.broadcast(meta, src=0)
        n_chain, n_taps = int(meta[0]), int(meta[1])
        buf = t.zeros(n_chain, dtype=t.uint8, device=self._dev())
        if self.rank == 0:
            buf.copy_(t.tensor(list(payload), dtype=t.uint8))
        self.dist.broadcast(buf, src=0)
        chain = bytes(buf.cpu().tolist()).decode()
        out_taps = None
        if n_taps >= 0:
            tb = t.zeros(n_taps, dtype=t.float64, device=self._dev())
            if self.rank == 0:
                tb.copy_(t.from_numpy(np.asarray(taps, dtype=np.float64)))
            self.dist.broadcast(tb, src=0)
            out_taps = tb.cpu().numpy()
        return chain, out_taps

    def gather_digests(self, local, n_streams):
        """local: [S_local, 3] (sum, sum of squares, peak per stream) -> [n_streams, 3] on every rank."""
        t = self.torch
        if self.dist is None:
            return local
        width = max(stream_range(n_streams, r, self.world)[1] - stream_range(n_streams, r, self.world)[0] for r in range(self.world))
        home = local.device
        if self.backend == "gloo":
            local = local.cpu()
        pad = t.zeros((width, 3), dtype=t.float64, device=local.device)
        pad[: local.shape[0]] = local
        parts = [t.zeros_like(pad) for _ in range(self.world)]
        self.dist.all_gather(parts, pad)
        parts = [q.to(home) for q in parts]
        rows = []
        for r in range(self.world):
            lo, hi = stream_range(n_streams, r, self.world)
            rows.append(parts[r][: hi - lo])
        return t.cat(rows, dim=0)

    def max_time(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_count(self, n):
        if self.dist is None:
            return n
        t = self.torch.tensor([n], dtype=self.torch.int64, device=self._dev())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def taps_fingerprint(taps):
    return hashlib.sha256(np.asarray(taps, dtype="<f8").tobytes()).hexdigest()[:16]

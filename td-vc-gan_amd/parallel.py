"""Data-parallel gradient synchronisation over RCCL/xGMI.

The path shards by sample: every op is per-sample, every loss is a batch mean, so with equal shards
the global gradient is the mean of the rank gradients. Gradients already live in one flat fp32 arena
per model (arena.py), so the exchange is a few large sum-all-reduces (bucketed, default 16 MiB — ring
collectives over point-to-point xGMI links are per-link bound, large buckets amortise the 7-hop latency)
issued on a side stream; AdamW consumes `G * (1/world)` through its grad_scale argument, so no separate
scaling pass runs. One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm) or "gloo" (CPU tests).
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, bucket_bytes=16 << 20, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = group
        self.world = dist.get_world_size(group)
        self.scale = 1.0 / self.world
        self.bucket = max(1, bucket_bytes // 4)
        self._side = None

    def _buckets(self, flat, n):
        return [flat[o:min(n, o + self.bucket)] for o in range(0, n, self.bucket)]

    def all_reduce(self, arena):
        """Sum-all-reduce the live prefix of the arena's gradient buffer (dead parameters are never exchanged,
        so every rank skips the same tensors: Q7)."""
        flat, n = arena.G, arena.n_live
        if flat.is_cuda:
            cur = torch.cuda.current_stream(flat.device)
            if self._side is None:
                self._side = torch.cuda.Stream(flat.device)
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                for b in self._buckets(flat, n):
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
            cur.wait_stream(self._side)
        else:
            for b in self._buckets(flat, n):
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_params(self, arena, src=0):
        dist.broadcast(arena.P, src=src, group=self.group)

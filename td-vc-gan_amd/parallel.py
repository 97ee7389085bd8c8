"""Data-parallel gradient synchronisation over RCCL/xGMI.

The path shards by sample: every op is per-sample, every loss is a batch mean, so with equal shards
the global gradient is the mean of the rank gradients. Gradients already live in one flat fp32 arena
per model (arena.py), cut into contiguous SEGMENTS of ~8 MiB aligned to layer boundaries.

Schedule (train.py:287-291, 485-491 define what must be ordered: backward -> optimizer.step):
  * during a backward pass every weight-gradient launch reports its layer to the arena; as soon as the last
    contribution of a segment has been queued, the arena folds that segment's weight-norm gradients and calls
    `reduce_segment`: the side stream waits for the compute stream up to that point and sum-all-reduces the
    segment's slice of `G` while the compute stream carries on with the rest of the backward pass;
  * `wait()` — called right before the optimizer step — makes the compute stream wait for the side stream.
So the exchange of the late layers (which finish first) overlaps the backward of the early ones; only the last
segment's all-reduce is exposed. Ring collectives over point-to-point xGMI are per-link bound, hence few, large
messages (8 MiB segments; 68 MB of D gradients = 8 calls) rather than per-tensor ones. AdamW consumes
`G * (1/world)` through its grad_scale argument, so no separate scaling pass runs.

One process per GPU, torch.distributed backend "nccl" (= RCCL on ROCm) or "gloo" (CPU tests).
"""
import os

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
        self.calls = 0            # collectives issued (tests / diagnostics)
        # Diagnostic for one-GPU boxes (TDVC_DP_LOOPBACK=1, world size 1 only): RCCL launches NO kernel for a single-rank
        # all-reduce, so a kernel trace cannot show where the exchange sits. With the switch every segment hand-over also
        # copies the segment to a scratch buffer on the side stream -- a stand-in with the all-reduce's bytes and stream
        # position -- so that the trace shows side-stream work under the backward kernels. Never changes G.
        self.loopback = os.environ.get('TDVC_DP_LOOPBACK') == '1' and self.world == 1
        self._scratch = None

    # ------------------------------------------------------------------ segment pipeline (product path)
    def attach(self, arena):
        """Route the arena's completed gradient segments through this object (ParamArena.attach_sync)."""
        arena.attach_sync(self)

    def _side_stream(self, device):
        if self._side is None:
            # a HIGH-PRIORITY stream: HIP multiplexes the streams of a process onto a few hardware queues, and a plain side
            # stream can land on the compute stream's queue, where its kernels run strictly in turn with the backward pass
            # (seen in the kernel trace of `bench.py --force-dp`: profiles/r03_dp_overlap.txt); priority streams get
            # hardware queues of their own, so the exchange really runs beside the compute kernels
            self._side = torch.cuda.Stream(device, priority=-1)
        return self._side

    def reduce_segment(self, arena, seg):
        """Sum-all-reduce G[seg_bounds[seg] : seg_bounds[seg+1]] on the side stream, ordered after everything queued so
        far on the current stream (the weight-gradient kernels and the weight-norm fold of that segment)."""
        flat = arena.G[arena.seg_bounds[seg]:arena.seg_bounds[seg + 1]]
        self.calls += 1
        if flat.is_cuda:
            side = self._side_stream(flat.device)
            side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(side):
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                if self.loopback:
                    if self._scratch is None or self._scratch.numel() < flat.numel():
                        self._scratch = torch.empty(flat.numel(), dtype=flat.dtype, device=flat.device)
                    self._scratch[:flat.numel()].copy_(flat)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def wait(self, arena):
        """Before the optimizer step: the compute stream waits for every all-reduce issued for this arena."""
        if arena.G.is_cuda and self._side is not None:
            torch.cuda.current_stream(arena.G.device).wait_stream(self._side)

    # ------------------------------------------------------------------ whole-arena primitives
    def _buckets(self, flat, n):
        return [flat[o:min(n, o + self.bucket)] for o in range(0, n, self.bucket)]

    def all_reduce(self, arena):
        """Blocking variant: sum-all-reduce the whole live prefix of the arena's gradient buffer in buckets (dead
        parameters are never exchanged, so every rank skips the same tensors: Q7)."""
        flat, n = arena.G, arena.n_live
        if flat.is_cuda:
            cur = torch.cuda.current_stream(flat.device)
            side = self._side_stream(flat.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for b in self._buckets(flat, n):
                    dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                    self.calls += 1
            cur.wait_stream(side)
        else:
            for b in self._buckets(flat, n):
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
                self.calls += 1

    def broadcast_params(self, arena, src=0, include_frozen=None):
        """Rank `src`'s parameters to every rank. The dead tail of the arena travels too while it is small (the layers the
        forward never reads, SURVEY Q7: keeps checkpoints rank-independent); a large one (> 64 MB: a frozen feature extractor
        such as WavLM-Large, loaded from the same checkpoint on every rank) only on request."""
        if include_frozen is None:
            include_frozen = (arena.n_total - arena.n_live) <= (16 << 20)
        dist.broadcast(arena.P if include_frozen else arena.P[:arena.n_live], src=src, group=self.group)

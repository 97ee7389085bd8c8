"""Worker for tests/test_parallel_cpu.py — launched with torch.distributed.run, backend gloo, CPU only."""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


class FakeArena:
    def __init__(self, n_live, n_total, fill):
        self.G = torch.full((n_total,), float(fill))
        self.P = torch.arange(n_total, dtype=torch.float32) * (fill + 1)
        self.n_live, self.n_total = n_live, n_total


def main():
    torch.set_num_threads(2)
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = importlib.import_module('td-vc-gan_amd')
    sync = pkg.parallel.GradSync(bucket_bytes=64)          # tiny buckets: exercises the bucket loop
    a = FakeArena(n_live=1000, n_total=1010, fill=rank + 1)
    sync.all_reduce(a)
    tot = world * (world + 1) / 2
    assert bool((a.G[:1000] == tot).all()) and bool((a.G[1000:] == rank + 1).all()) and sync.scale == 1.0 / world
    sync.broadcast_params(a, include_frozen=False)          # live prefix only: a (large, frozen) dead tail stays as it was
    ar = torch.arange(1010, dtype=torch.float32)
    assert bool((a.P[:1000] == ar[:1000] * 2).all()) and bool((a.P[1000:] == ar[1000:] * (rank + 2)).all())
    sync.broadcast_params(a)                                 # default: a small dead tail travels too
    assert bool((a.P == ar * 2).all())      # rank 0 holds arange * (1 + 1)
    # segment pipeline (the product path's schedule): completed segments are reduced as the "backward pass" reports them,
    # the learning pass reduces everything at its end; either way every rank ends with the world sum, dead tail untouched
    A = pkg.arena

    class SegArena:
        def __init__(self, fill):
            self.seg_bounds = [0, 300, 700, 1000]
            self.nseg, self.n_live = 3, 1000
            self.G = torch.full((1010,), float(fill))
            self.track = A.SegmentTracker(3)

    seg_sync = pkg.parallel.GradSync()
    order = [2, 2, 1, 0, 1, 0, 0]
    for it in range(3):
        sa = SegArena(rank + 1 + it)
        sa.track.expected = None if it == 0 else sa_expected
        sa.track.begin()
        fired_early = []
        for s in order:
            if sa.track.note(s):
                fired_early.append(s)
                seg_sync.reduce_segment(sa, s)
        for s in sa.track.finish():
            seg_sync.reduce_segment(sa, s)
        seg_sync.wait(sa)
        sa_expected = sa.track.expected
        want = sum(r + 1 + it for r in range(world))
        assert bool((sa.G[:1000] == want).all()) and bool((sa.G[1000:] == rank + 1 + it).all()), it
        assert fired_early == ([] if it == 0 else [2, 1, 0]), (it, fired_early)      # late layers' segment first
    assert seg_sync.calls == 9
    # sharding contract with the oracle's D-step: gradient of the global batch == mean of shard gradients
    from common import filled_sd
    from oracle import losses as OL, model as OM
    per = 1
    bt = pkg.synth.make_batch(per * world, 8960, seed=3)
    sd = {k: v.clone().requires_grad_(True) for k, v in filled_sd('D').items()}
    sl = slice(per * rank, per * (rank + 1))
    x, lbl = bt['signal_real'][sl], bt['label_src'][sl]
    o, _ = OM.discriminator(sd, x, lbl, OM.disc_subsamples(x))
    OL.lsgan_to_one(o).backward()
    flat = torch.cat([sd[k].grad.reshape(-1) for k in sorted(sd)])
    arena = type('A', (), {})(); arena.G, arena.n_live = flat, flat.numel()
    big = pkg.parallel.GradSync(bucket_bytes=16 << 20)
    big.all_reduce(arena)
    if rank == 0:
        sd2 = {k: v.clone().requires_grad_(True) for k, v in filled_sd('D').items()}
        o2, _ = OM.discriminator(sd2, bt['signal_real'], bt['label_src'], OM.disc_subsamples(bt['signal_real']))
        OL.lsgan_to_one(o2).backward()
        ref = torch.cat([sd2[k].grad.reshape(-1) for k in sorted(sd2)])
        err = float((flat * sync.scale - ref).norm() / ref.norm())
        assert err < 1e-4, err
        print(f'DP_OK err={err:.2e}', flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

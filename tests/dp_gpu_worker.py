"""Worker for tests/test_dp_gpu.py — one process per GPU under torch.distributed.run, backend nccl (= RCCL).

Runs the PRODUCT step `TrainStep(grad_sync=GradSync())` data-parallel over the ranks (global batch sharded by sample,
SURVEY §8e) and, on every rank, the plain single-GPU step on the whole global batch; after `ITERS` iterations the two
must have produced the same parameters: with equal shards the mean of the rank gradients IS the global-batch gradient.
Also checked here: gradient segments really are handed to RCCL before the backward pass ends (from the second
iteration on), `capture()` refuses to graph-capture a data-parallel step, and the ranks stay bit-identical.
"""
import importlib
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
ITERS = 3


def shard(bt, lo, hi):
    return {k: v[lo:hi].contiguous() for k, v in bt.items()}


def main():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    local = int(os.environ.get('LOCAL_RANK', 0))
    dev = torch.device(f'cuda:{local}')
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', device_id=dev)
    rank, world = dist.get_rank(), dist.get_world_size()
    P = importlib.import_module('td-vc-gan_amd')
    from common import build_models, rel_l2, to_dev
    per, T = 2, 8960
    B = per * world
    cfg = P.train_step.StepConfig()
    bt_cpu = P.synth.make_batch(B, T, seed=77)
    ix = P.synth.contrastive_indices(B, T // 320, cfg.n_neg, seed=5)
    iy = P.synth.contrastive_indices(B, T // 320, cfg.n_neg, seed=6)

    # ---- reference: single-GPU step on the global batch (every rank computes it; no communication)
    G0, D0 = build_models(dev)
    ts0 = P.train_step.TrainStep(G0, D0, cfg, dev)
    bt_all = to_dev(bt_cpu, dev)
    for _ in range(ITERS):
        log0 = ts0.run(bt_all, ix.to(dev), iy.to(dev))
    torch.cuda.synchronize()

    # ---- data-parallel product step on this rank's shard
    G, D = build_models(dev)
    sync = P.parallel.GradSync()
    sync.broadcast_params(G.arena); sync.broadcast_params(D.arena)
    ts = P.train_step.TrainStep(G, D, cfg, dev, grad_sync=sync)
    lo, hi = rank * per, (rank + 1) * per
    bt = to_dev(shard(bt_cpu, lo, hi), dev)
    sx, sy = ix[lo:hi].contiguous().to(dev), iy[lo:hi].contiguous().to(dev)
    early = []
    orig = sync.reduce_segment

    handed = []                                # (arena, seg, copy of the segment's G slice as it was handed over)

    def spy(arena, seg):                       # was this segment handed over before the end-of-backward flush?
        early.append((arena is D.arena, seg, arena._finish_queued))
        # stream-ordered snapshot of what goes to the all-reduce: everything queued so far for this segment (its weight-grad
        # kernels, the deferred folds flushed for it, its weight-norm fold) and nothing later
        handed.append((arena, seg, arena.G[arena.seg_bounds[seg]:arena.seg_bounds[seg + 1]].clone()))
        return orig(arena, seg)
    sync.reduce_segment = spy
    for it in range(ITERS):
        n0 = len(early)
        log = ts.run(bt, sx, sy)
        fired = early[n0:]
        in_backward = [e for e in fired if e[2]]   # _finish_queued is still set while the backward pass is running
        if it == 0:
            assert not in_backward, 'learning pass must reduce at the end of backward only'
        else:
            assert len(in_backward) >= max(1, len(fired) - 2), (it, fired)
        assert len(fired) == G.arena.nseg + D.arena.nseg, (len(fired), G.arena.nseg, D.arena.nseg)
        if world == 1:
            # At world 1 the all-reduce is the identity, so a segment handed over BEFORE its last contribution (a weight-grad
            # kernel, a deferred fold or a weight-norm fold landing later) would still end up right in G and every parameter
            # check below would pass. Make that visible: the snapshot taken at hand-over must be bit-equal to the segment's
            # final gradient (G is not touched between the end of a backward pass and the next zero_grad).
            torch.cuda.synchronize()
            late = [(a is D.arena, s_) for a, s_, snap in handed
                    if not torch.equal(snap, a.G[a.seg_bounds[s_]:a.seg_bounds[s_ + 1]])]
            assert not late, f'iteration {it}: segments changed after their hand-over to the all-reduce: {late}'
        handed.clear()
    torch.cuda.synchronize()
    try:
        ts.capture(bt, sx, sy)
        raise AssertionError('capture() must refuse a data-parallel step')
    except RuntimeError as e:
        assert 'single-GPU' in str(e), e

    # ---- compare
    for k in ('D_loss', 'G_loss'):
        # the logged loss of a rank is the mean over ITS shard; the global-batch loss is the mean of the rank means
        t = log[k].detach().clone()
        dist.all_reduce(t)
        got, want = float(t) / world, float(log0[k])
        assert abs(got - want) <= 1e-3 * abs(want), (k, got, want)
    e_g = rel_l2(G.arena.P[:G.arena.n_live], G0.arena.P[:G0.arena.n_live])
    e_d = rel_l2(D.arena.P[:D.arena.n_live], D0.arena.P[:D0.arena.n_live])
    # parameters after ITERS AdamW steps; the UPDATE is the sensitive part (see tests/common.py), so compare that too
    from common import filled_sd
    upd = {}
    for name, m, m0 in (('G', G, G0), ('D', D, D0)):
        before = filled_sd(name)
        num = den = 0.0
        for (k, p), (_, p0) in zip(m.named_parameters(), m0.named_parameters()):
            b = before[k].to(dev)
            num += float(((p - b) - (p0 - b)).double().pow(2).sum()); den += float((p0 - b).double().pow(2).sum())
        upd[name] = (num / max(den, 1e-300)) ** 0.5
    assert e_g < 1e-4 and e_d < 1e-4, (e_g, e_d)
    assert upd['G'] < 0.15 and upd['D'] < 0.15, upd          # sign flips of noise-level gradient elements only
    # ranks hold identical parameters (same reduced gradients, same update)
    chk = torch.stack([G.arena.P.double().sum(), D.arena.P.double().sum()])
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    assert all(torch.equal(c, lst[0]) for c in lst), lst
    if rank == 0:
        print(f'DP_GPU_OK world={world} calls={sync.calls} params_rel=({e_g:.2e},{e_d:.2e}) update_rel=({upd["G"]:.3f},{upd["D"]:.3f})', flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""CPU: host-side logic of the data-parallel gradient segments (arena.plan_segments / SegmentTracker / the segment tables
of ParamArena) — SURVEY §8e: buckets fire as soon as their gradients are final, identical on every rank."""
import pytest
import torch

from common import D_ARGS, G_ARGS, pkg


def test_segment_tracker_learns_then_pipelines():
    A = pkg().arena
    tr = A.SegmentTracker(3)
    order = [2, 2, 1, 2, 1, 0, 0]                  # a backward pass: late layers (segment 2) first
    # learning pass: nothing fires early, everything is handed over at the end
    assert [tr.note(s) for s in order] == [False] * len(order)
    assert tr.finish() == [0, 1, 2] and tr.expected == [2, 2, 3]
    # pipelined pass: a segment fires exactly at its last contribution
    tr.begin()
    assert [tr.note(s) for s in order] == [False, False, False, True, True, False, True]
    assert tr.finish() == []
    # a contribution after its segment was handed over is an error (the reduced gradient would miss it)
    tr.begin()
    for s in order:
        tr.note(s)
    with pytest.raises(RuntimeError):
        tr.note(2)
    # fewer contributions than learned (part of the graph not taken): the rest is flushed at the end, profile adapts
    tr = A.SegmentTracker(2)
    for s in (1, 1, 0):
        tr.note(s)
    tr.finish()
    tr.begin()
    assert tr.note(1) is False and tr.note(0) is True
    assert tr.finish() == [1] and tr.expected == [1, 1]


@pytest.mark.parametrize('which', ['G', 'D'])
def test_arena_segments_partition_the_live_prefix(which):
    P = pkg()
    M = P.modules
    if which == 'G':
        model = M.Generator(**{**G_ARGS, 'decoder_channels': list(G_ARGS['decoder_channels'])})
    else:
        model = M.CollaborativeMultibandDiscriminator(**D_ARGS)
    a = P.arena.ParamArena(model, 'cpu', model.dead_prefixes)      # host tables only: no kernel is launched here
    b = a.seg_bounds
    assert b[0] == 0 and b[-1] == a.n_live and all(x < y for x, y in zip(b, b[1:]))
    # several buckets, none huge (D: one per discriminator, each dominated by its 21 MB 1024x1024x5 layer, which a
    # single weight-gradient launch produces and which therefore cannot usefully be split)
    assert a.nseg >= 3 and max(y - x for x, y in zip(b, b[1:])) * 4 < 32 << 20
    starts = {a.offsets[k] for k in a.live_keys}
    assert set(b[:-1]) <= starts                                     # boundaries fall on tensor starts ...
    for k in a.live_keys:                                            # ... and never inside a layer (bias, g, v together)
        prefix = k.rsplit('.', 1)[0]
        segs = {a.seg_of(a.offsets[kk]) for kk in a.live_keys if kk.rsplit('.', 1)[0] == prefix}
        assert len(segs) == 1, (prefix, segs)
    # weight-norm rows: contiguous ranges per segment that tile all rows, and each row's v offset lies in its segment
    assert a.seg_rows[0][0] == 0 and a.seg_rows[-1][1] == a.nrows
    assert all(x[1] == y[0] for x, y in zip(a.seg_rows, a.seg_rows[1:]))
    voff = a.row_voff.tolist()
    for i, (r0, r1) in enumerate(a.seg_rows):
        assert all(b[i] <= voff[r] < b[i + 1] for r in range(r0, r1))
    # every bound slot knows its segment
    name, mod = next((n, m) for n, m in model.named_modules() if isinstance(m, M.ConvParams))
    s = a.slot(name, mod.has_bias)
    assert 0 <= s.seg < a.nseg

"""Flat HBM arenas for one model: parameters, gradients, effective (weight-normed) weights.

MI355X-first memory layout: every parameter of a model is a view into ONE
contiguous fp32 buffer `P`; gradients mirror it in `G`; the effective weights w = g*v/||v|| of all
weight-normed convs live in `W` (rebuilt by one multi-tensor kernel per optimizer step) and their
gradients accumulate in `dW`. Consequences: AdamW is a single kernel over the live prefix of `P`,
gradient all-reduce is a handful of large RCCL calls over `G`, and zero_grad is two memsets.

Parameters the forward never reads (`decoder.excite_downsample.0.*`) sit at the tail of
the arena, outside the live prefix: their .grad stays None and AdamW skips them, exactly like
torch.optim.AdamW skips grad-less parameters in the reference (train.py:188, Q7).

Data-parallel runs (parallel.GradSync) additionally use the arena's SEGMENTS: the live prefix is cut into contiguous
pieces of ~8 MiB aligned to layer boundaries. Every weight-gradient launch reports its layer (note_grad); once all the
contributions a segment receives in one backward pass have been launched, the segment's weight-norm fold runs and its
slice of `G` is handed to the RCCL all-reduce on a side stream while the backward pass continues (SURVEY §8e schedule).
"""
import bisect

import torch

from . import _lib as L


class ConvSlot:
    """Raw device addresses of one conv layer's tensors inside the arenas."""
    __slots__ = ('w', 'b', 'dw', 'db', 'trainable', 'arena', 'wt', 'seg')

    def __init__(self, w=0, b=0, dw=0, db=0, trainable=True, arena=None, wt=0, seg=-1):
        self.w, self.b, self.dw, self.db, self.trainable, self.arena, self.wt = w, b, dw, db, trainable, arena, wt
        self.seg = seg          # arena segment holding this layer's parameters (gradient bucketing, -1 = none)


def plan_segments(keys, offsets, n_live, target):
    """Cut [0, n_live) into contiguous segments of >= `target` elements whose boundaries fall between layers (a layer =
    the parameters sharing one prefix: bias, weight_g, weight_v). Returns the nseg + 1 boundary offsets."""
    bounds, prev = [0], None
    for k in keys:
        prefix = k.rsplit('.', 1)[0]
        o = offsets[k]
        if prefix != prev and o - bounds[-1] >= target:
            bounds.append(o)
        prev = prefix
    if n_live - bounds[-1] < target // 4 and len(bounds) > 1:      # fold a short tail into the previous segment
        bounds.pop()
    bounds.append(n_live)
    return bounds


class SegmentTracker:
    """Host-side bookkeeping of which gradient segments are complete during a backward pass (no device work here, so
    the logic is covered by the CPU tests). The number of contributions each segment receives per pass is learned on
    the first pass after attach (that pass reduces everything at its end) and re-checked on every later pass."""

    def __init__(self, nseg):
        self.nseg = nseg
        self.expected = None
        self.begin()

    def begin(self):
        self.seen = [0] * self.nseg
        self.fired = [False] * self.nseg

    def note(self, seg):
        """A weight-gradient launch for a layer of `seg` was just queued. Returns True when the segment is complete."""
        if self.fired[seg]:
            raise RuntimeError(f'gradient contribution for arena segment {seg} after the segment was handed to the all-reduce '
                               '(the backward graph changed: call ParamArena.reset_segment_profile())')
        self.seen[seg] += 1
        if self.expected is not None and self.seen[seg] == self.expected[seg]:
            self.fired[seg] = True
            return True
        return False

    def finish(self):
        """End of the pass: the segments not yet handed over (all of them on the learning pass). Updates the profile."""
        rest = [i for i in range(self.nseg) if not self.fired[i]]
        self.expected = list(self.seen)
        for i in rest:
            self.fired[i] = True
        return rest


class ParamArena:
    def __init__(self, module: torch.nn.Module, device, dead_prefixes=(), transposable=(), seg_bytes=8 << 20):
        """transposable: prefixes of weight-normed stride-1, groups==1 convs whose effective weight is also kept
        pre-transposed as [Cin][Cout][K] (arena WT) for the input-gradient kernel.
        Layout of P: [trainable | frozen | dead]. Frozen = parameters with requires_grad == False (the reference freezes a
        sub-network that way, train.py:195-197): the forward reads them, they get no gradient and no optimizer update. Dead =
        parameters the forward never reads (`dead_prefixes`)."""
        named = list(module.named_parameters())
        is_dead = lambda k: any(k.startswith(d) for d in dead_prefixes)
        live = [(k, p) for k, p in named if not is_dead(k) and p.requires_grad]
        frozen = [(k, p) for k, p in named if not is_dead(k) and not p.requires_grad]
        dead = [(k, p) for k, p in named if is_dead(k)]
        if not live:
            raise L.TdvcError('ParamArena: the model has no trainable parameter')
        self.frozen_keys = set(k for k, _ in frozen)
        self.device = torch.device(device)
        self.offsets = {}
        off = 0
        for k, p in live + frozen + dead:
            self.offsets[k] = off
            off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
            if k == live[-1][0]:
                self.n_live = off
        self.n_total = off
        self.P = torch.zeros(self.n_total, dtype=torch.float32, device=self.device)
        # gradients exist for the live prefix only: the dead tail (never-read layers, a frozen feature extractor such as
        # WavLM-Large with ~316 M parameters) has no gradient storage and is never exchanged
        self.G = torch.zeros(max(self.n_live, 4), dtype=torch.float32, device=self.device)
        self.params = {}
        self.live_keys = [k for k, _ in live]
        with torch.no_grad():
            for k, p in live + frozen + dead:
                o = self.offsets[k]
                view = self.P[o:o + p.numel()].view(p.shape)
                view.copy_(p.data.to(self.device, torch.float32))
                p.data = view
                p.grad = None
                self.params[k] = p
        # effective-weight arena for weight-normed tensors
        self.wn = []          # (prefix, v_off, g_off, w_off, rows, cols)
        woff = 0
        self.nrows_live = 0            # weight-norm rows of trainable tensors come first: only they have a backward fold
        for k, p in live + frozen:
            if k.endswith('.weight_v'):
                pre = k[:-len('.weight_v')]
                rows, cols = p.shape[0], p.numel() // p.shape[0]
                self.wn.append((pre, self.offsets[k], self.offsets[pre + '.weight_g'], woff, rows, cols))
                woff += (p.numel() + 3) // 4 * 4
                if k not in self.frozen_keys:
                    self.nrows_live += rows
        self.n_w = max(woff, 4)
        self.W = torch.zeros(self.n_w, dtype=torch.float32, device=self.device)
        self.WT = torch.zeros(self.n_w, dtype=torch.float32, device=self.device)
        self.dW = torch.zeros(self.n_w, dtype=torch.float32, device=self.device)
        self.transposed = set(t for t in transposable)
        self.w_offsets = {pre: wo for pre, _, _, wo, _, _ in self.wn}
        if self.wn:
            rv, rg, rw, rl, tb, ts, tk = [], [], [], [], [], [], []
            for pre, vo, go, wo, rows, cols in self.wn:
                K = self.params[pre + '.weight_v'].shape[2]
                tr = pre in self.transposed
                for r in range(rows):
                    rv.append(vo + r * cols); rg.append(go + r); rw.append(wo + r * cols); rl.append(cols)
                    tb.append(wo + r * K); ts.append(rows * K); tk.append(K if tr else 0)
            t = lambda a, dt: torch.tensor(a, dtype=dt, device=self.device)
            self.row_voff, self.row_goff, self.row_woff = t(rv, torch.int64), t(rg, torch.int64), t(rw, torch.int64)
            self.row_len = t(rl, torch.int32)
            self.row_tbase, self.row_tstride, self.row_k = t(tb, torch.int64), t(ts, torch.int64), t(tk, torch.int32)
            self.nrows = len(rl)
        else:
            self.nrows = 0
        self.wgrad_enabled = True
        self.version = 0               # bumped by materialize(): derived copies of the weights (ops._weight_planes_x6) key on it
        self.x6_planes = {}            # effective-weight address -> (version, bf16 piece planes)
        self.token = torch.zeros(1, dtype=torch.float32, device=self.device, requires_grad=True)
        self._grads_attached = False
        self._finish_queued = False
        # gradient segments (data-parallel bucketing): boundaries in P/G offsets and the weight-norm rows of each
        self.seg_bounds = plan_segments(self.live_keys, self.offsets, self.n_live, max(1, seg_bytes // 4))
        self.nseg = len(self.seg_bounds) - 1
        self.seg_rows = []
        r = 0
        for i in range(self.nseg):
            r0 = r
            for pre, vo, _, _, rows, _ in self.wn:
                if self.seg_bounds[i] <= vo < self.seg_bounds[i + 1]:
                    r += rows
            self.seg_rows.append((r0, r))
        self._sync = None          # parallel.GradSync when attached
        self._track = None

    # ------------------------------------------------------------------ addresses
    def slot(self, prefix: str, has_bias: bool) -> ConvSlot:
        """Addresses for the conv whose parameters are `prefix`.{weight_v,weight_g | weight}[, bias]."""
        pb, gb = self.P.data_ptr(), self.G.data_ptr()
        if prefix in self.w_offsets:
            wo = self.w_offsets[prefix]
            w, dw = self.W.data_ptr() + 4 * wo, self.dW.data_ptr() + 4 * wo
        else:
            o = self.offsets[prefix + '.weight']
            w, dw = pb + 4 * o, gb + 4 * o
        b = db = 0
        if has_bias:
            o = self.offsets[prefix + '.bias']
            b, db = pb + 4 * o, gb + 4 * o
        wt = self.WT.data_ptr() + 4 * self.w_offsets[prefix] if prefix in self.transposed else 0
        wkey = prefix + ('.weight_v' if prefix in self.w_offsets else '.weight')
        first = self.offsets[wkey]
        if wkey in self.frozen_keys:      # read by the forward, never trained: no gradient slices (they would lie outside G)
            return ConvSlot(w, b, 0, 0, False, self, wt, -1)
        return ConvSlot(w, b, dw, db, True, self, wt, self.seg_of(first))

    def seg_of(self, offset: int) -> int:
        return max(0, min(self.nseg - 1, bisect.bisect_right(self.seg_bounds, offset) - 1))

    def owns(self, p: torch.Tensor) -> bool:
        a = p.data_ptr()
        return self.P.data_ptr() <= a < self.P.data_ptr() + 4 * self.n_total

    # ------------------------------------------------------------------ per-step kernels
    def materialize(self):
        """w = g * v / ||v|| for every weight-normed tensor: one launch."""
        self.version += 1
        if self.nrows:
            st = torch.cuda.current_stream(self.device).cuda_stream
            L.check(L.lib().tdvc_weight_norm_fwd_t(self.P.data_ptr(), self.W.data_ptr(), self.WT.data_ptr(),
                                                   self.row_voff.data_ptr(), self.row_goff.data_ptr(),
                                                   self.row_woff.data_ptr(), self.row_len.data_ptr(),
                                                   self.row_tbase.data_ptr(), self.row_tstride.data_ptr(),
                                                   self.row_k.data_ptr(), self.nrows, st))

    def zero_grad(self):
        # weight-grad folds still queued at this point can only be leftovers of a pass that did not finish (an exception in
        # backward, an abandoned graph capture): every completed pass flushes in finish_grads. Their slab pointers are stale.
        if self.device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():
            L.lib().tdvc_fold_reset(torch.cuda.current_stream(self.device).cuda_stream)
        self.G.zero_()
        self.dW.zero_()
        for p in self.params.values():
            p.grad = None
        self._grads_attached = False
        if self._track is not None:
            self._track.begin()

    # ------------------------------------------------------------------ data-parallel gradient segments
    def attach_sync(self, sync):
        """Hand completed gradient segments to `sync.reduce_segment(arena, seg)` as the backward pass produces them."""
        self._sync = sync
        self._track = SegmentTracker(self.nseg) if sync is not None else None

    def reset_segment_profile(self):
        if self._track is not None:
            self._track = SegmentTracker(self.nseg)

    def _fold_rows(self, r0, r1):
        """(v, g) gradients of weight-norm rows [r0, r1) from the effective-weight gradients in dW."""
        if r1 > r0:
            st = torch.cuda.current_stream(self.device).cuda_stream
            L.check(L.lib().tdvc_weight_norm_bwd(self.P.data_ptr(), self.dW.data_ptr(), self.G.data_ptr(),
                                                 self.row_voff.data_ptr() + 8 * r0, self.row_goff.data_ptr() + 8 * r0,
                                                 self.row_woff.data_ptr() + 8 * r0, self.row_len.data_ptr() + 4 * r0, r1 - r0, 0, st))

    def _segment_ready(self, seg):
        self._fold_rows(*self.seg_rows[seg])
        self._sync.reduce_segment(self, seg)

    def note_grad(self, slot):
        """Called by the operators right after a weight-gradient launch of `slot`'s layer was queued."""
        self.queue_finish()
        if self._track is not None and slot.seg >= 0 and self._track.note(slot.seg):
            self._flush_folds()
            self._segment_ready(slot.seg)

    def finish_grads(self):
        """Fold the effective-weight gradients into (v, g) gradients and expose .grad views."""
        self._finish_queued = False
        self._flush_folds()
        if self._track is not None:
            for seg in self._track.finish():
                self._segment_ready(seg)
        elif self.nrows_live:
            self._fold_rows(0, self.nrows_live)
        if not self._grads_attached:
            for k in self.live_keys:
                p = self.params[k]
                o = self.offsets[k]
                p.grad = self.G[o:o + p.numel()].view(p.shape)
            self._grads_attached = True

    def _flush_folds(self):
        """The weight-grad folds are deferred (ops.workspace): launch the queued ones before dW / G are read."""
        if self.device.type == 'cuda':
            from . import ops
            ops.fold_flush(self.device)

    def queue_finish(self):
        """Called from inside a backward: run finish_grads once, when this backward pass ends."""
        if not self._finish_queued:
            self._finish_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish_grads)


class FlatAdamW:
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps outside the sqrt) as one
    kernel over the arena's live prefix. The step counter lives on the device so that a captured
    hipGraph of the whole iteration replays correctly."""

    def __init__(self, arena: ParamArena, lr=1e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=1e-2):
        self.a, self.lr, self.betas, self.eps, self.wd = arena, lr, betas, eps, weight_decay
        self.m = torch.zeros(arena.n_live, dtype=torch.float32, device=arena.device)
        self.v = torch.zeros(arena.n_live, dtype=torch.float32, device=arena.device)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=arena.device)
        self._clip_ws = None           # [1024 partials | clip coefficient, total norm]

    def step(self, grad_scale=1.0, max_norm=None):
        """max_norm: torch.nn.utils.clip_grad_norm_(parameters, max_norm) before the step (train.py:289-290, 489-490): the
        total 2-norm of the gradient the optimizer sees (grad_scale included: the all-reduced mean under data parallelism) is
        computed on the device and the clip coefficient min(1, max_norm / (norm + 1e-6)) enters the update as one more
        gradient factor -- no extra pass over the gradients, no host synchronisation. `last_grad_norm` holds the norm."""
        a = self.a
        st = torch.cuda.current_stream(a.device).cuda_stream
        lib = L.lib()
        L.check(lib.tdvc_inc_i32(self.step_dev.data_ptr(), 1, st))
        coef = None
        if max_norm is not None:
            if self._clip_ws is None:
                self._clip_ws = torch.zeros(1024 + 2, dtype=torch.float32, device=a.device)
            coef = self._clip_ws.data_ptr() + 4 * 1024
            L.check(lib.tdvc_grad_clip_coef(a.G.data_ptr(), a.n_live, float(max_norm), grad_scale, self._clip_ws.data_ptr(), coef, st))
            self.last_grad_norm = self._clip_ws[1025:1026]
        L.check(lib.tdvc_adamw_clipped(a.P.data_ptr(), a.G.data_ptr(), self.m.data_ptr(), self.v.data_ptr(), a.n_live,
                                       self.lr, self.betas[0], self.betas[1], self.eps, self.wd, 0, self.step_dev.data_ptr(),
                                       grad_scale, coef, st))

    def zero_grad(self):
        self.a.zero_grad()

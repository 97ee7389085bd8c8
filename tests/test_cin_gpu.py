"""GPU parity of `modules.ConditionalInstanceNorm` (tdvc_cin_fwd / tdvc_cin_bwd + the Linear / Conv1d k=5 that
produce gamma, beta) — model/conditional_instance_norm.py:4-19 of the reference:

  * against tests/golden/cin.npz = outputs and gradients of the REFERENCE module itself (oracle/make_golden.py) on the
    same seeded inputs: y, dx and every parameter gradient, 2-D (Linear) and 3-D (Conv) conditioning, at 1e-3;
  * against a float64 restatement evaluated here, at 2e-5 (op-level tolerance of the other kernels).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from common import GOLDEN, filled_sd, pkg, rel_l2

pytestmark = pytest.mark.gpu


def _inputs():
    rs = np.random.RandomState(5)          # same draws, same order as oracle/make_golden.py
    x = torch.from_numpy(rs.randn(3, 32, 500).astype(np.float32))
    c2 = torch.from_numpy(rs.randn(3, 128).astype(np.float32))
    c3 = torch.from_numpy(rs.randn(3, 129, 500).astype(np.float32))
    cot = torch.from_numpy(rs.randn(3, 32, 500).astype(np.float32))
    return x, c2, c3, cot


def _ref64(sd, x, c, cot):
    sd = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    x = x.double().clone().requires_grad_(True)
    if c.dim() == 2:
        h = F.linear(c.double(), sd['embedding.weight'], sd['embedding.bias']).unsqueeze(2)
    else:
        h = F.conv1d(c.double(), sd['embedding_conv.weight'], sd['embedding_conv.bias'], padding=2)
    gamma, beta = h.chunk(2, dim=1)
    y = (1 + gamma) * F.instance_norm(x, eps=1e-5) + beta
    (y * cot.double()).mean().backward()
    return y, x.grad, {k: v.grad for k, v in sd.items() if v.grad is not None}


@pytest.mark.parametrize('kind', ['2d', '3d'])
def test_conditional_instance_norm_vs_reference_golden(kind, dev):
    P = pkg()
    gold = np.load(os.path.join(GOLDEN, 'cin.npz'))
    sd = filled_sd('CIN')
    cin = P.modules.ConditionalInstanceNorm(32, 128)
    cin.load_state_dict(sd)
    cin.ensure_arena(dev)
    x, c2, c3, cot = _inputs()
    c = c2 if kind == '2d' else c3
    cin.arena.zero_grad()
    xd = x.to(dev).requires_grad_(True)
    y = cin(xd, c.to(dev))
    (y * cot.to(dev)).mean().backward()
    torch.cuda.synchronize()
    y64, dx64, dp64 = _ref64(sd, x, c, cot)

    errs_gold = dict(y=rel_l2(y, torch.from_numpy(gold[f'y_{kind}'])), dx=rel_l2(xd.grad, torch.from_numpy(gold[f'dx_{kind}'])))
    errs_64 = dict(y=rel_l2(y, y64), dx=rel_l2(xd.grad, dx64))
    used = ('embedding.weight', 'embedding.bias') if kind == '2d' else ('embedding_conv.weight', 'embedding_conv.bias')
    params = dict(cin.named_parameters())
    for k in used:
        assert params[k].grad is not None, k
        errs_gold['d_' + k] = rel_l2(params[k].grad, torch.from_numpy(gold[f'd_{kind}_{k}']))
        errs_64['d_' + k] = rel_l2(params[k].grad, dp64[k])
    # the branch not taken contributes nothing: its gradient stays exactly zero (the reference leaves it None)
    for k, p in params.items():
        if k not in used and p.grad is not None:
            assert float(p.grad.abs().max()) == 0.0, k
    assert max(errs_gold.values()) < 1e-3, errs_gold
    assert max(errs_64.values()) < 2e-5, errs_64


def test_conditional_instance_norm_constant_row_and_long_sequence(dev):
    """Edge cases of the per-(sample, channel) statistics: a constant row (variance 0 -> eps only) and a sequence much
    longer than one block's stride (T = 16000), against float64."""
    P = pkg()
    sd = filled_sd('CIN')
    cin = P.modules.ConditionalInstanceNorm(32, 128)
    cin.load_state_dict(sd)
    cin.ensure_arena(dev)
    rs = np.random.RandomState(8)
    x = torch.from_numpy((rs.randn(2, 32, 16000) * 3 + 1.5).astype(np.float32))
    x[0, 3] = 0.25                              # constant row
    c = torch.from_numpy(rs.randn(2, 128).astype(np.float32))
    cot = torch.from_numpy(rs.randn(2, 32, 16000).astype(np.float32))
    cin.arena.zero_grad()
    xd = x.to(dev).requires_grad_(True)
    y = cin(xd, c.to(dev))
    (y * cot.to(dev)).mean().backward()
    torch.cuda.synchronize()
    y64, dx64, _ = _ref64(sd, x, c, cot)
    assert torch.isfinite(y).all() and torch.isfinite(xd.grad).all()
    assert rel_l2(y, y64) < 2e-5 and rel_l2(xd.grad, dx64) < 1e-4

"""CPU: `util.load_possible` / `util.load_model` (reference: util/__init__.py:64-89 driven by train.py:58-68) against
tests/golden/load_possible.* = what the reference's own function did to the same module and checkpoint
(oracle/make_golden_ssl.py), and the oracle's SSL content encoder against the reference module's output."""
import json
import os

import numpy as np
import torch

from common import GOLDEN, pkg


class Small(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv1d(4, 6, 3)
        self.b = torch.nn.Linear(5, 3)
        self.c = torch.nn.Conv1d(2, 2, 1)


def _fixture():
    g = np.load(os.path.join(GOLDEN, 'load_possible.npz'))
    part = lambda p: {k[len(p):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(p)}
    return part('before/'), part('ckpt/'), part('after/'), json.load(open(os.path.join(GOLDEN, 'load_possible.json')))


def test_load_model_falls_back_to_load_possible_like_the_reference(tmp_path, capsys):
    U = pkg().util
    before, ckpt, after, messages = _fixture()
    m = Small()
    m.load_state_dict(before)
    path = tmp_path / 'latest-G.pt'
    torch.save(ckpt, path)
    got = U.load_model(m, str(path))
    assert got == messages
    for k, v in m.state_dict().items():
        assert torch.equal(v, after[k]), k
    out = capsys.readouterr().out
    assert 'mismatched_size: a.weight' in out and 'unmatched_keys: zzz.weight' in out and 'missing_keys: c.bias' in out


def test_load_possible_alone_also_loads_matched_tensors():
    U = pkg().util
    before, ckpt, after, messages = _fixture()
    m = Small()
    m.load_state_dict(before)
    assert U.load_possible(m, ckpt) == messages          # no strict load before it: matched tensors must still arrive
    for k, v in m.state_dict().items():
        assert torch.equal(v, after[k]), k


def test_oracle_ssl_content_encoder_vs_reference_golden():
    from oracle import model as OM
    P = pkg()
    shapes = json.load(open(os.path.join(GOLDEN, 'shapes_SSLENC.json')))
    sd = {'encoder.encoder.' + k: v for k, v in P.synth.fill_state_dict(shapes).items()}
    rs = np.random.RandomState(77)
    c = torch.from_numpy(rs.randn(2, 1024, 100).astype(np.float32))
    gold = np.load(os.path.join(GOLDEN, 'ssl_encoder.npz'))
    m = OM.ssl_content_encoder(sd, c)
    assert float((m - torch.from_numpy(gold['m'])).norm() / torch.from_numpy(gold['m']).norm()) < 1e-5

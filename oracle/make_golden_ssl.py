"""ORACLE tooling — BUILD container only (imports /root/reference). Golden vectors for the SSL content encoder
 and the permissive checkpoint load (§8f-3):

  * the reference's own `model.ssl_encoder.Encoder` (pre 1x1 -> WN gated stack, 16 layers k=5 -> proj 1x1; :17-116) with
    deterministically filled weights on seeded random SSL features [2, 1024, 100]: output m, input gradient, parameter
    gradient norms + sampled elements, state_dict layout -> tests/golden/ssl_encoder.npz / .json, shapes_SSLENC.json;
    pins oracle.model.ssl_content_encoder against it. (SSLEncoder itself cannot be constructed: it loads
    wavlm/WavLM-Large.pt, which does not ship with the reference — README.md:6.)
  * `util.load_possible` driven exactly as train.py:58-68 drives it (strict load_state_dict first, permissive load on
    RuntimeError) on a small module with one matching, one shape-mismatched, one unknown and one missing tensor:
    resulting parameters + message lists -> tests/golden/load_possible.npz / .json.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_ssl
"""
import json

import numpy as np
import torch

from oracle import make_golden as MG
from oracle import model as OM

synth = MG.synth
OUT = MG.OUT


def main():
    MG.import_reference()
    from model.ssl_encoder import Encoder as RefEncoder
    import util as RU
    pin = {}
    enc = RefEncoder(1024, 128, 128, 5, 1, 16)
    shapes = MG.shapes_of(enc)
    json.dump(shapes, open(f'{OUT}/shapes_SSLENC.json', 'w'))
    sd = MG.load_filled(enc)
    rs = np.random.RandomState(77)
    c = torch.from_numpy(rs.randn(2, 1024, 100).astype(np.float32)).requires_grad_(True)
    cot = torch.from_numpy(rs.randn(2, 128, 100).astype(np.float32))
    torch.manual_seed(0)
    z, m, logs, _ = enc(c)
    (m * cot).mean().backward()
    so = {'encoder.encoder.' + k: v.clone().requires_grad_(True) for k, v in sd.items()}
    co = c.detach().clone().requires_grad_(True)
    mo = OM.ssl_content_encoder(so, co)
    (mo * cot).mean().backward()
    pin['ssl_encoder'] = dict(m=MG.rel(mo, m), dc=MG.rel(co.grad, c.grad),
                              grads=max(MG.rel(so['encoder.encoder.' + k].grad, p.grad) for k, p in enc.named_parameters()))
    norms, samples = MG.grad_summary({k: p.grad for k, p in enc.named_parameters()}, n_sample=16)
    np.savez_compressed(f'{OUT}/ssl_encoder.npz', m=m.detach().numpy(), dc=c.grad.numpy())
    json.dump(dict(norms=norms, samples=samples), open(f'{OUT}/ssl_encoder.json', 'w'))

    # ---------------- load_possible, as train.py:58-68 calls it
    class Small(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv1d(4, 6, 3)
            self.b = torch.nn.Linear(5, 3)
            self.c = torch.nn.Conv1d(2, 2, 1)

    torch.manual_seed(3)
    model = Small()
    before = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    ckpt = {'a.weight': torch.randn(8, 4, 5, generator=g), 'a.bias': torch.randn(6, generator=g),      # a.weight: larger in two dims
            'b.weight': torch.randn(2, 5, generator=g), 'b.bias': torch.randn(3, generator=g),         # b.weight: smaller in dim 0
            'zzz.weight': torch.randn(3, generator=g)}                                                  # unknown key; c.* missing
    try:
        model.load_state_dict(ckpt)
        raise AssertionError('strict load should have failed')
    except RuntimeError:
        messages = RU.load_possible(model, ckpt)
    after = {k: v.clone() for k, v in model.state_dict().items()}
    np.savez_compressed(f'{OUT}/load_possible.npz', **{'before/' + k: v.numpy() for k, v in before.items()},
                        **{'ckpt/' + k: v.numpy() for k, v in ckpt.items()}, **{'after/' + k: v.numpy() for k, v in after.items()})
    json.dump(messages, open(f'{OUT}/load_possible.json', 'w'))

    old = json.load(open(f'{OUT}/PINNING.json'))
    old.update(pin)
    json.dump(old, open(f'{OUT}/PINNING.json', 'w'), indent=1)
    print(json.dumps(pin, indent=1), messages)


if __name__ == '__main__':
    main()

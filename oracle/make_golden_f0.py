"""ORACLE tooling — BUILD container only (imports /root/reference). Golden vector for the device-side
`f0_to_excitation`: runs the reference's own util.f0_to_excitation (util/__init__.py:22-50) on CPU under a
fixed torch seed, replays its three random draws (start phase, voiced noise, unvoiced noise — in that order, the last one
only for the unvoiced samples in row-major order) and stores inputs + draws + output. Also pins the numpy restatement
`td-vc-gan_amd/synth.excitation_from_f0` against the reference with the same draws.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_f0
"""
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, ROOT)
synth = importlib.import_module('td-vc-gan_amd.synth')


def main():
    sys.path.insert(0, REF)
    sys.dont_write_bytecode = True
    sys.modules.setdefault('torchaudio', types.ModuleType('torchaudio'))
    import util as RU
    B, T, step = 3, 8960, 64
    f0 = synth.make_f0(np.random.RandomState(77), B, T)               # [B,1,T/64+1], ~30 % unvoiced
    f0_t = torch.from_numpy(f0)
    torch.manual_seed(4321)
    exc_ref = RU.f0_to_excitation(f0_t, step, 16000, True)
    # replay the draws
    torch.manual_seed(4321)
    phi0 = torch.rand(1) * 2 * torch.pi
    nv = torch.randn(B, 1, T)
    w = 2 * torch.pi * f0_t[:, :, :-1] / 16000
    up = torch.nn.functional.interpolate(w, scale_factor=step, mode='nearest')
    lin = torch.nn.functional.interpolate(w, scale_factor=step, mode='linear')
    msk = torch.nn.functional.interpolate(torch.log(w), scale_factor=step, mode='linear') != -torch.inf
    up[msk] = lin[msk]
    unv = up == 0
    nu = torch.zeros(B, 1, T)
    nu[unv] = torch.randn(int(unv.sum()))
    np.savez_compressed(f'{OUT}/f0_excitation.npz', f0=f0, start_phase=phi0.numpy(), noise_v=nv.numpy(), noise_u=nu.numpy(),
                        exc=exc_ref.numpy(), step=np.int32(step))
    # pin the numpy restatement (same formulas, float64) with the same draws
    class Replay:
        def __init__(self): self.k = 0
        def uniform(self): return float(phi0) / (2 * np.pi)
        def randn(self, *shape):
            self.k += 1
            return nv.numpy().astype(np.float64) if self.k == 1 else nu.numpy().astype(np.float64)
    ours = synth.excitation_from_f0(f0, Replay(), step)
    err = float(np.abs(ours - exc_ref.numpy()).max())
    pin_path = f'{OUT}/PINNING.json'
    pin = json.load(open(pin_path))
    pin['f0_to_excitation_restatement_max_abs'] = err
    pin['f0_to_excitation_unvoiced_frac'] = float(unv.float().mean())
    json.dump(pin, open(pin_path, 'w'), indent=1)
    print('restatement vs reference: max abs', err, ' unvoiced', float(unv.float().mean()))


if __name__ == '__main__':
    main()

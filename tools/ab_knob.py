"""A/B of a tuning knob (tdvc_debug_knob) over the launch shapes of the step: same process, same rotating operand sets,
knob off / on / off / on so that clock drift shows up as a difference between the repeats.

    python tools/ab_knob.py [knob=0]
"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402

import bench  # noqa: E402
import tile_sweep as ts  # noqa: E402

pkg = importlib.import_module('td-vc-gan_amd')
lib = pkg._lib.lib()


def main():
    knob = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    jobs = [(136, 32, 16000, 'dgrad', 3, 32), (136, 32, 16000, 'fwd', 3, 32), (136, 64, 8000, 'dgrad', 3, 32), (136, 64, 8000, 'fwd', 3, 32),
            (136, 128, 4000, 'dgrad', 3, 32), (136, 128, 4000, 'fwd', 3, 32), (136, 256, 500, 'dgrad', 3, 32), (136, 256, 500, 'fwd', 3, 32),
            (8, 136, 16000, 'fwd', 3, 32), (8, 136, 4000, 'fwd', 3, 32),
            (1024, 1024, 63, 'fwd', 5, 64), (1024, 1024, 63, 'dgrad', 5, 64),
            (16, 16, 16000, 'fwd', 3, 32), (16, 16, 16000, 'dgrad', 3, 32), (16, 16, 16000, 'fwd', 11, 32),
            (32, 32, 8000, 'fwd', 7, 32), (32, 32, 8000, 'dgrad', 7, 32), (64, 64, 4000, 'fwd', 7, 32), (64, 64, 4000, 'dgrad', 7, 32),
            (128, 128, 500, 'fwd', 7, 32), (128, 128, 500, 'dgrad', 7, 32), (256, 256, 50, 'fwd', 7, 32)]
    for cin, cout, T, which, k, bl in jobs:
        calls, keep = ts.conv_calls(cin, cout, T, which, k=k, BL=bl)
        res = []
        for v in (0, 1, 0, 1):
            lib.tdvc_debug_knob(knob, v)
            res.append(bench.time_launches(torch, calls, 40) * 1e3)
        lib.tdvc_debug_knob(knob, 0)
        off, on = min(res[0], res[2]), min(res[1], res[3])
        print(f'{cin:4d}->{cout:4d} k{k:<2d} T={T:5d} B={bl} {which:5s}  off {res[0]:7.1f} {res[2]:7.1f}   on {res[1]:7.1f} {res[3]:7.1f} us   on/off {on / off:.3f}', flush=True)
        del calls, keep
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()

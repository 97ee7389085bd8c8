"""Text summary of gpurun_out/kernel_table.json (written by bench.py): the classes by time, and time by kernel family.
    python tools/kernel_table_summary.py [N] > profiles/rNN_x_kernel_table_top.txt"""
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, 'gpurun_out', 'kernel_table.json')))
t, S = d['table'], d['step_ms']
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
tot = sum(r['ms_per_launch'] * r['launches_per_step'] for r in t)
print(f'step {S:.2f} ms, library {d["lib_sha16"]}; {len(t)} launch classes, sum(launches x time) = {tot:.2f} ms = {tot / S:.3f} of the step')
print('(each class timed on rotating operand sets > 600 MB: cold, so the sum exceeds the step, whose launches find part of their operands in the Infinity Cache)\n')
print(f'{"ms/step":>8s} {"share":>6s} {"n":>3s} {"us":>8s} {"frac":>6s} bound  kernel | op')
for r in t[:n]:
    ms = r['ms_per_launch'] * r['launches_per_step']
    print(f'{ms:8.3f} {100 * ms / S:5.1f}% {r["launches_per_step"]:3d} {r["ms_per_launch"] * 1e3:8.1f} {r["frac"]:6.3f} {r["bound"]:5s}  {r["kernel"]} | {r["op"]}')
fam = collections.Counter()
for r in t:
    fam[r['kernel'].split('<')[0].split(' + ')[0]] += r['ms_per_launch'] * r['launches_per_step']
print('\nby kernel family:')
for k, v in fam.most_common():
    print(f'{v:8.3f} ms {100 * v / S:5.1f}%  {k}')

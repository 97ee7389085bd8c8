"""bench.py's per-kernel roofline table as a standalone program, for rocprofv3 (kernel trace / PMC passes):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- \\
        python3 tools/microbench_kernels.py --iters 6 --manifest gpurun_out/pmc_manifest.json

Every entry launches the kernel(s) of one (op, launch shape) of the step on rotating operand sets; the manifest records, in
launch order, which kernel names and how many calls belong to each entry so that tools/pmc_traffic.py can assign the
profiler's per-dispatch counters back to the entries."""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=6)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--top', type=int, default=0, help='selection run: write the N largest table entries (by a quick timing pass) to --ops-file and exit')
    ap.add_argument('--ops-file', default=os.path.join(ROOT, 'gpurun_out', 'pmc_ops.json'), help='restrict the table to the entries listed in this JSON file (if it exists)')
    ap.add_argument('--manifest', default=os.path.join(ROOT, 'gpurun_out', 'pmc_manifest.json'))
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    pkg = importlib.import_module('td-vc-gan_amd')
    # the launch classes of one recorded iteration of config/conv_enc-stage1.yaml, exactly as bench.py records them
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from common import build_models, to_dev
    G, D = build_models(dev)
    ts = pkg.train_step.TrainStep(G, D, pkg.train_step.StepConfig(), dev)
    bt = to_dev(pkg.synth.make_batch(a.batch, 16000, seed=1234), dev)
    ix = pkg.synth.contrastive_indices(a.batch, 50, 100, 17).to(dev); iy = pkg.synth.contrastive_indices(a.batch, 50, 100, 917).to(dev)
    ts.run(bt, ix, iy)
    classes = bench.record_launches(pkg, lambda: ts.run(bt, ix, iy))
    del ts, G, D
    torch.cuda.empty_cache()
    if a.top:      # selection run (NOT under the profiler): the classes that matter by a quick timing pass -> --ops-file, then exit
        rows0, north = bench.kernel_table(pkg, dev, classes, 1.0, iters=4)
        ops_sel = [e['op'] for e in sorted(rows0, key=lambda e: -e['ms_per_launch'] * e['launches_per_step'])[:a.top]]
        if north['op'] not in ops_sel:      # the kernel that runs the north star's 16 -> 16 dilated conv is always part of the PMC set
            ops_sel.append(north['op'])
        json.dump(ops_sel, open(a.ops_file, 'w'), indent=1)
        print(f'{len(ops_sel)} table entries selected -> {a.ops_file}')
        return
    keep_ops = set(json.load(open(a.ops_file))) if a.ops_file and os.path.exists(a.ops_file) else None
    manifest = []
    rows, _ = bench.kernel_table(pkg, dev, classes, 1.0, iters=a.iters, manifest=manifest, only_ops=keep_ops)
    os.makedirs(os.path.dirname(a.manifest), exist_ok=True)
    json.dump(manifest, open(a.manifest, 'w'), indent=1)
    for e in sorted(rows, key=lambda e: -e['ms_per_launch'] * e['launches_per_step']):
        print(f"{e['ms_per_launch'] * 1e3:9.1f} us x{e['launches_per_step']:2d}  {e['frac']:.3f} of {e['bound']:4s} peak  {e['op']}  [{e['kernel']}]", flush=True)


if __name__ == '__main__':
    main()

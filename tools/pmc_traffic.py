"""Assign rocprofv3 --pmc counter values (one directory per counter pass, each run over tools/microbench_kernels.py with
the same manifest) to the entries of bench.py's kernel table and write profiles/r02_pmc.json.

    python tools/pmc_traffic.py gpurun_out/pmc_manifest.json gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq ... \\
        > profiles/r02_pmc.txt

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): on gfx950 FETCH_SIZE reports half the bytes of a wide
coalesced streaming read (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-B-per-lane stores."""
import collections
import csv
import glob
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
norm = importlib.import_module('td-vc-gan_amd._lib').normalize_kernel_name


def per_entry(manifest, d):
    """{op: {counter: mean per dispatch of the entry's FIRST listed kernel}} for one pass directory."""
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        return {}
    disp = collections.OrderedDict()          # dispatch id -> (name, {counter: value})
    for r in csv.DictReader(open(files[0], newline='')):
        k = int(r['Dispatch_Id'])
        disp.setdefault(k, (norm(r['Kernel_Name']), {}))[1][r['Counter_Name']] = float(r['Counter_Value'])
    seq = [v for _, v in sorted(disp.items())]
    # every entry ends with one pmc_marker_kernel dispatch (bench.kernel_table with a manifest): segment i <-> manifest[i]
    segs, cur = [], []
    for v in seq:
        if v[0] == 'pmc_marker_kernel':
            segs.append(cur); cur = []
        else:
            cur.append(v)
    segs = segs[1:]      # the first marker closes the recorded training iteration that precedes the table
    if len(segs) != len(manifest):
        raise SystemExit(f'{d}: {len(segs)} marker-delimited segments for {len(manifest)} manifest entries')
    out = {}
    for e, seg in zip(manifest, segs):
        mine = [v[1] for v in seg if v[0] == e['kernels'][0]]
        if len(mine) < 4:
            raise SystemExit(f'{d}: only {len(mine)} dispatches of {e["kernels"][0]} in the segment of {e["op"]!r}: {sorted({v[0] for v in seg})}')
        main = mine[-max(1, len(mine) - 3):]      # drop the warm-up launches
        agg = collections.defaultdict(float)
        for c in main:
            for k, v in c.items():
                agg[k] += v / len(main)
        out[e['op']] = dict(agg)
    return out


def main():
    manifest = json.load(open(sys.argv[1]))
    entries = {e['op']: dict(kernel=e['kernels'][0]) for e in manifest}
    for d in sys.argv[2:]:
        for op, vals in per_entry(manifest, d).items():
            entries[op].update(vals)
    for op, e in entries.items():
        if 'FETCH_SIZE' in e and 'WRITE_SIZE' in e:
            e['hbm_bytes_per_launch'] = (2.0 * e['FETCH_SIZE'] + e['WRITE_SIZE']) * 1024.0
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in e and 'GRBM_GUI_ACTIVE' in e and e['GRBM_GUI_ACTIVE'] > 0:
            # busy cycles are summed over the 4 SIMDs of every CU, GRBM_GUI_ACTIVE over the 8 XCDs (guide: DVFS section)
            e['mfma_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (256 * 4) / (e['GRBM_GUI_ACTIVE'] / 8)
    import bench
    tag = os.environ.get('TDVC_PROFILE_TAG', 'r03')
    doc = dict(entries=entries, lib_sha16=bench.lib_sha16(),
               note='rocprofv3 --pmc passes over tools/microbench_kernels.py; see tools/pmc_traffic.py. lib_sha16 = the libtdvc_hip.so build '
                    'the counters were taken on: bench.py reports `traffic` only when it runs that very build')
    for d in ('profiles', 'gpurun_out'):      # gpurun_out/ is what travels back from the GPU box; profiles/ is what is committed
        if os.path.isdir(os.path.join(ROOT, d)):
            json.dump(doc, open(os.path.join(ROOT, d, f'{tag}_pmc.json'), 'w'), indent=1)
    for op, e in entries.items():
        print(op)
        print('    ' + '  '.join(f'{k}={v:.4g}' if isinstance(v, float) else f'{k}={v}' for k, v in e.items()))


if __name__ == '__main__':
    main()

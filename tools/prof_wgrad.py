"""Per-template breakdown of the weight-gradient kernels from a rocprofv3 --kernel-trace CSV of bench.py (diagnostic tool)."""
import collections, csv, glob, re, sys
rows = list(csv.DictReader(open(glob.glob(f'{sys.argv[1]}/*/*_kernel_trace.csv')[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
lo, hi = ad[1] + 1, ad[-1] + 1
nsteps = (len(ad) - 2) // 2
pat = sys.argv[2] if len(sys.argv) > 2 else 'wgrad|slab'
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[lo:hi]:
    n = r['Kernel_Name']
    if not re.search(pat, n): continue
    m = re.search(r'(\w+)<([^>]*)>', n)
    key = ((m.group(1) + '<' + m.group(2) + '>') if m else n[:48], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    agg[key][0] += d; agg[key][1] += 1
tot = 0
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(f'{d/1e6/nsteps:7.3f} ms/step  n={c/nsteps:5.1f}  avg={d/c/1e3:8.1f} us  {k}')
for k, (d, c) in agg.items(): tot += d
print('total', tot/1e6/nsteps)

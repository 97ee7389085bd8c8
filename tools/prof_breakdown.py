"""Per-step kernel-time breakdown from a rocprofv3 --kernel-trace CSV of bench.py (diagnostic tool)."""
import collections, csv, glob, re, sys
d = sys.argv[1]
f = glob.glob(f'{d}/*/*_kernel_trace.csv*')[0]
import gzip, io
rows = list(csv.DictReader(io.TextIOWrapper(gzip.open(f)) if f.endswith('.gz') else open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
lo, hi = ad[1] + 1, ad[-1] + 1
nsteps = (len(ad) - 2) // 2
sub = rows[lo:hi]
cat, cnt, shapes = collections.Counter(), collections.Counter(), collections.defaultdict(collections.Counter)
def c(n):
    m = re.search(r'conv_lean_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)', n)
    if m: return 'lean M%sN%s W%sx%s xf%s epi%s' % m.group(1, 2, 3, 4, 5, 6)
    for key in ('wgrad_tile', 'wgrad_lean', 'conv_wgrad_kernel', 'conv_gemm_kernel', 'scalar', 'slab_reduce'):
        if key in n: return key
    if 'at::native' in n: return 'aten'
    return n.split('(')[0][-28:]
for r in sub:
    dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = c(r['Kernel_Name']); cat[k] += dur; cnt[k] += 1
    shapes[k][(r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])] += dur
tot = sum(cat.values())
for k, v in cat.most_common(40):
    print(f'{v / 1e6 / nsteps:8.2f} ms/step {100 * v / tot:5.1f}%  n/step={cnt[k] / nsteps:6.0f}  {k}')
    for g, dd in shapes[k].most_common(4): print(f'            grid={g}: {dd / 1e6 / nsteps:6.2f} ms/step')
print('kernel total/step', tot / 1e6 / nsteps, 'wall span/step', (int(sub[-1]['End_Timestamp']) - int(sub[0]['Start_Timestamp'])) / 1e6 / nsteps)

"""Summarise rocprofv3 --pmc passes of tools/microbench_conv.py: average counter value per dispatch, per kernel.
   python tools/pmc_summary.py gpurun_out/pmc_f gpurun_out/pmc_w"""
import collections, csv, glob, re, sys
for d in sys.argv[1:]:
    for f in glob.glob(f'{d}/*/*counter_collection.csv'):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            if 'conv_' not in n: continue
            m = re.search(r'(\w+)<([^>]*)>', n)
            key = ((m.group(1) + '<' + m.group(2) + '>') if m else n[:40], r['Grid_Size'] if 'Grid_Size' in r else '', r['Counter_Name'])
            agg[key][0] += float(r['Counter_Value']); agg[key][1] += 1
        for k, (v, c) in sorted(agg.items()):
            print(f'{k[2]:12s} avg={v / c:14.1f}  n={c:4d}  {k[0]}  grid={k[1]}')

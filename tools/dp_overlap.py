"""Where do the gradient all-reduces of a data-parallel run sit relative to the compute kernels? Reads a rocprofv3 --kernel-trace
CSV of `bench.py --force-dp` (or a multi-rank run) and reports, per RCCL kernel, the compute kernels of OTHER queues/streams that
ran during its interval.   python tools/dp_overlap.py gpurun_out/prof_dp"""
import csv, glob, gzip, io, sys
d = sys.argv[1]
f = glob.glob(f'{d}/**/*_kernel_trace.csv*', recursive=True)[0]
rows = list(csv.DictReader(io.TextIOWrapper(gzip.open(f)) if f.endswith('.gz') else open(f)))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
import collections
skey = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
ns = collections.Counter(r[skey] for r in rows)
main_s = max(ns, key=ns.get)
n_rccl = sum('nccl' in r['Kernel_Name'].lower() or 'rccl' in r['Kernel_Name'].lower() for r in rows)
# the exchange = RCCL kernels; at world size 1 RCCL launches none: then the TDVC_DP_LOOPBACK stand-in copies, which are the only
# dispatches of their (side) stream
if n_rccl:
    is_coll = lambda r: 'nccl' in r['Kernel_Name'].lower() or 'rccl' in r['Kernel_Name'].lower()
else:
    side = [s_ for s_ in ns if s_ != main_s and all('copyBuffer' in r['Kernel_Name'] for r in rows if r[skey] == s_) and ns[s_] >= 6]
    is_coll = lambda r: r[skey] in side
coll = [r for r in rows if is_coll(r)]
comp = [r for r in rows if not is_coll(r) and r[skey] == main_s]
print(f'{len(rows)} kernel dispatches; per stream {dict(ns)}; hardware queues {dict(collections.Counter(r["Queue_Id"] for r in rows))}; {n_rccl} RCCL kernels'
      + ('' if n_rccl else f' -> loopback stand-in copies on stream(s) {side} taken as the exchange'))
print('queue of the exchange:', sorted({r['Queue_Id'] for r in coll}), ' queue of the compute stream:', sorted({r['Queue_Id'] for r in comp}))
tot = ov = 0
for c in coll:
    dur = c['e'] - c['s']
    over = [(k, min(k['e'], c['e']) - max(k['s'], c['s'])) for k in comp if k['e'] > c['s'] and k['s'] < c['e']]
    o = sum(x for _, x in over)
    tot += dur; ov += min(o, dur)
    names = sorted({k['Kernel_Name'].split('(')[0][-44:] for k, _ in over})[:3]
    print(f"{c['Kernel_Name'][:40]:40s} stream {c[skey]} queue {c['Queue_Id']}: {dur / 1e3:8.1f} us, compute-stream kernels running meanwhile: {o / 1e3:8.1f} us  {names}")
if coll:
    print(f'exchange kernel time {tot / 1e3:.1f} us, of which {ov / 1e3:.1f} us ({100 * ov / max(tot, 1):.0f} %) ran while a compute-stream kernel of the step was running')

"""Timeline of one fused pass from a rocprofv3 --kernel-trace CSV: start / end of every kernel relative to the first
kernel of the pass (the last complete pass in the trace).  usage: python tools/pass_timeline.py TRACE.csv [first-kernel-prefix]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else 'k_prep_lds'
ks = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40], r.get('Stream_Id', r.get('Queue_Id', '?'))) for r in rows]
ks.sort()
starts = [i for i, k in enumerate(ks) if k[2].startswith(first)]
i0, i1 = starts[-2], starts[-1]
t0 = ks[i0][0]
for a, b, n, q in ks[i0:i1]:
    print('{:>8.1f} {:>8.1f} {:>7.1f} us  {:40s} queue {}'.format((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, n, q))
print('pass period {:.1f} us'.format((ks[i1][0] - t0) / 1e3))

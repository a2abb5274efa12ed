"""the kernels of one occupancy refresh (the side-stream burst of 8+ launches) from a rocprofv3 kernel trace: python tools/refresh_burst.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
by_q = {}
for r in rows:
    by_q.setdefault(r['Queue_Id'], []).append(r)
best = None
for q, rs in by_q.items():
    rs.sort(key=lambda r: r['s'])
    bursts = [[rs[0]]]
    for r in rs[1:]:
        if r['s'] - bursts[-1][-1]['e'] > 2e6:
            bursts.append([r])
        else:
            bursts[-1].append(r)
    for b in bursts:
        if any('opa_scatter_max' in r['Kernel_Name'] for r in b):
            best = b
if best is None:
    sys.exit('no refresh burst found')
print('launches %d, span %.1f us, busy %.1f us' % (len(best), (best[-1]['e'] - best[0]['s']) / 1e3, sum(r['e'] - r['s'] for r in best) / 1e3))
for r in best:
    print('%8.1f us  %s' % ((r['e'] - r['s']) / 1e3, r['Kernel_Name'][:100]))

"""print the top rows of a rocprofv3 kernel_stats.csv: calls, average us, share, short name"""
import csv
import re
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    name = re.sub(r'\(.*', '', r['Name']).replace('void ', '').replace('arcn::', '')
    if pat and not re.search(pat, name):
        continue
    print('%6.2f%% %6d calls %9.1f us avg  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, name[:90]))

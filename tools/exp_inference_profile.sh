R=$(pwd)
python tools/exp_render_image.py 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_i
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_i -o i --output-format csv -- python $R/tools/exp_render_image.py > /tmp/i.log 2>&1
tail -2 /tmp/i.log
f=$(find /tmp/prof_i -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel total ms', tot / 1e6)
for r in rows[:14]:
    print('%6.2f%% %6d calls %9.1f us  %s' % (float(r['Percentage']), int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:80]))
PY

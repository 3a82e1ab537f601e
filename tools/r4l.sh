cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/prof_stats.sh r4s6b_s1 --steps 30 --warmup 3 --no-legs --sustain-s 0 > gpurun_out/r4s6b_s1.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4s6b_s1/kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms/step', tot/33/1e6)
for r in rows[:24]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%8.3f ms/step' % (float(r['TotalDurationNs'])/33/1e6), '%8.1f us' % (float(r['AverageNs'])/1e3), r['Percentage'])
PY

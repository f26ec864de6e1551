"""Print the top kernels of the last tools/measure_wide.sh run (gpurun_out/mw) and its bench line."""
import csv, glob, json, os
f = max(glob.glob('gpurun_out/mw/prof/**/*kernel_stats.csv', recursive=True), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), '%9.2f ms' % (float(r['TotalDurationNs']) / 1e6), r['Percentage'])
d = json.load(open('gpurun_out/mw/prof.json'))
print('steps/s', d['value'], 'psi2 TF', d['roofline']['achieved'], 'ms/launch', d['roofline']['avg_launch_ms'])

"""Per-call durations of the chunk-pass kernels from a rocprofv3 --kernel-trace CSV (diagnostics)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(list)
for r in rows:
    per[r['Kernel_Name'].split('(')[0]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
tot = 0
for k, v in per.items():
    v.sort()
    d = [x[1] / 1000 for x in v]
    if len(d) > 20:
        tot += sum(d)
        print("%-22s n=%3d avg %6.1f us | every 4th: %s" % (k, len(d), sum(d) / len(d), ' '.join('%.0f' % x for x in d[::4])))
print("sum of chunk kernels: %.2f ms" % (tot / 1000))

"""Print a rocprofv3 kernel-stats CSV compactly: python tools/kstats.py <dir>/r_kernel_stats.csv ..."""
import csv, sys
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r['TotalDurationNs']) for r in rows)
    print(path, 'total %.1f ms' % (tot / 1e6))
    for r in rows[:6]:
        print('  %-14s calls %6s total %8.1f ms avg %8.3f ms max %8.3f %5.1f%%' % (r['Name'].split('(')[0][-14:], r['Calls'], int(r['TotalDurationNs']) / 1e6,
              float(r['AverageNs']) / 1e6, int(r['MaxNs']) / 1e6, float(r['Percentage'])))

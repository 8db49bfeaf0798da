"""print Calls / AverageNs / MinNs of the kernels whose name contains one of the given substrings, from a rocprofv3 kernel_stats.csv"""
import csv, sys
path, pats = sys.argv[1], sys.argv[2:]
for r in csv.DictReader(open(path)):
    if not pats or any(p in r["Name"] for p in pats):
        print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.2f} us  min {float(r["MinNs"])/1e3:9.2f} us  {r["Percentage"]}%')

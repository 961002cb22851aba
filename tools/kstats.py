"""Pretty-prints a rocprofv3 *_kernel_stats.csv"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0][-46:]
    print("%-48s calls=%4s avg_us=%10.1f tot_ms=%9.2f" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))

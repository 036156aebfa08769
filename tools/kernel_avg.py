"""Average duration of the kernels whose name contains a string, from a rocprofv3 kernel-stats CSV.
  python tools/kernel_avg.py <kernel_stats.csv> <substring>"""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Name"]:
        print(r["Name"].split("(")[0][-40:], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")

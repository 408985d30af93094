"""print (kernel, calls, average ms, share) from a rocprofv3 --kernel-trace --stats csv"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    name = r["Name"].split("(")[0].replace("void ", "").replace("natac::", "")
    print("%-34s calls %3s  avg %8.3f ms  %5.1f %%" % (name[:34], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["Percentage"])))

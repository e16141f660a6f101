"""print gpurun_out/ks.csv (tools/prof_step.sh) as microseconds per step"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ks.csv")))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 13
print("total per step ms %.3f" % (sum(int(r["TotalDurationNs"]) for r in rows) / n / 1e6))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-62s %5.1f calls %8.1f us/step" % (r["Name"][:62], int(r["Calls"]) / n, int(r["TotalDurationNs"]) / n / 1e3))

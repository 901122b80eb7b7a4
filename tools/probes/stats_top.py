import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print("%8.2f ms/step %7.1f calls %8.1f us  %s" % (float(r["TotalDurationNs"]) / div / 1e6, int(r["Calls"]) / div, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
print("total ms/step", tot / div / 1e6)

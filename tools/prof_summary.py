"""Summarise a rocprofv3 results DB (kernel-trace) into per-kernel totals; optionally write the CSV kept under profiles/."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
out = sys.argv[3] if len(sys.argv) > 3 else None
rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"total {tot/1e6:.2f} ms, per step {tot/1e6/steps:.2f} ms")
if out:
    with open(out, "w", newline="") as f:
        w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows: w.writerow([r[0], r[1], r[2], round(r[3], 1), round(100 * r[2] / tot, 3), r[4], r[5]])
for r in rows[:int(sys.argv[4]) if len(sys.argv) > 4 else 40]:
    print(f"{r[2]/1e6/steps:8.2f} ms/step {r[1]:6d} {r[3]/1e3:9.1f}us  {r[0][:100]}")

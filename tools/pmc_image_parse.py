"""rocprofv3 --pmc csv output of tools/pmc_image.sh -> per-kernel mean FETCH_SIZE / WRITE_SIZE per launch (KB as rocprofv3 reports them)."""
import csv, glob, json, sys
def load(d, ctr):
    acc = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"])
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        for (_, name), v in per.items():
            n, s = acc.get(name, (0, 0.0))
            acc[name] = (n + 1, s + v)
    return acc
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
rows = []
for name in sorted(set(f) | set(w), key=lambda n: -(f.get(n, (0, 0))[1] + w.get(n, (0, 0))[1])):
    nf, sf = f.get(name, (0, 0.0)); nw, sw = w.get(name, (0, 0.0))
    rows.append({"kernel": name[:120], "launches": nf or nw, "fetch_kb_per_launch_raw": sf / max(nf, 1), "write_kb_per_launch": sw / max(nw, 1)})
json.dump(rows, open(sys.argv[3], "w"), indent=1)
for r in rows[:25]:
    print(f"{r['launches']:5d} fetch {r['fetch_kb_per_launch_raw']/1e3:9.1f} MB(raw) write {r['write_kb_per_launch']/1e3:9.1f} MB  {r['kernel'][:80]}")

"""Per-stream busy timeline of one training step from a rocprofv3 results DB (two-stream schedule diagnostics)."""
import math, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = db.execute("select start, end, stream_id, name from kernels order by start").fetchall()
ad = [r for r in rows if 'adamw' in r[3]]
ends = [ad[i][1] for i in range(2, len(ad), 3)]
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]: out[-1][1] = max(out[-1][1], e)
        else: out.append([s, e])
    return out
def inter(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
t0, t1 = ends[k - 1], ends[k]
sel = [r for r in rows if r[1] > t0 and r[0] < t1]
sids = sorted(set(r[2] for r in sel))
per = {sid: union([(max(r[0], t0), min(r[1], t1)) for r in sel if r[2] == sid]) for sid in sids}
for sid in sids:
    print("stream", sid, "busy %.2f ms" % (sum(e - s for s, e in per[sid]) / 1e6))
if len(sids) > 1:
    print("both busy %.2f ms" % (inter(per[sids[0]], per[sids[1]]) / 1e6), "wall %.2f" % ((t1 - t0) / 1e6))
nb = int(math.ceil((t1 - t0) / 5e6))
for i in range(nb):
    s = t0 + i * 5e6; e = min(t1, s + 5e6)
    print(f"{i*5:4d} ms  " + "  ".join(f"s{sid}:{inter(per[sid], [[s, e]])/(e-s):4.2f}" for sid in sids))

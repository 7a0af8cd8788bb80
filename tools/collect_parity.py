"""gpurun_out/parity_r04.jsonl (written by the GPU tests through tests/parity_log.py) -> profiles/parity_r04.json:
one document {summary, rows: [{test, metric, measured, bound, ok}]}, the LAST record of every (test, metric) pair.

    python tools/collect_parity.py [in.jsonl] [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_r04.jsonl")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "parity_r04.json")
rows = {}
for line in open(src):
    line = line.strip()
    if line:
        d = json.loads(line)
        rows[(d["test"], d["metric"])] = d
rows = list(rows.values())
builds = sorted({(r.get("csrc", "?"), r.get("lib", "?")) for r in rows})
if len(builds) > 1:
    sys.exit(f"{src} mixes {len(builds)} builds of the library {builds}: re-run `pytest -m gpu` in one session")
north = [r for r in rows if abs(r["bound"] - 1e-2) < 1e-12]
for r in rows:
    r.pop("csrc", None); r.pop("lib", None)
doc = {"build": {"csrc_digest16": builds[0][0], "lib_sha16": builds[0][1]} if builds else None, "what": "every bound asserted by `pytest -m gpu` that goes through tests/parity_log.py: measured value next to its bound",
       "summary": {"records": len(rows), "failed": sum(not r["ok"] for r in rows),
                   "records_at_north_star_1e-2": len(north),
                   "worst_measured_at_north_star_1e-2": max((r["measured"] for r in north), default=None)},
       "rows": rows}
with open(dst, "w") as fh:
    json.dump(doc, fh, indent=1)
print(f"{len(rows)} records -> {dst}; failed: {doc['summary']['failed']}; worst at the 1e-2 bound: {doc['summary']['worst_measured_at_north_star_1e-2']}")

"""Assemble profiles/r0N_pmc_gemm_pp64.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv) over tools/pmc_gemm.py.
usage: pmc_gemm_report.py <fetch dir> <write dir> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch and summed over the XCDs' instances; FETCH_SIZE is doubled (MI355X_MICROARCH.md,
HBM section: gfx950 tallies the 128-byte requests of wide streaming reads at 64 bytes)."""
import csv, glob, json, sys
fdir, wdir, out = sys.argv[1:4]
def per_dispatch(d, ctr):
    rows = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr or "gemm_pp64" not in r["Kernel_Name"]:
                continue
            k = int(r["Dispatch_Id"])
            rows[k] = (int(r["Grid_Size"]), rows.get(k, (0, 0.0))[1] + float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]
fe, wr = per_dispatch(fdir, "FETCH_SIZE"), per_dispatch(wdir, "WRITE_SIZE")
shapes = [("qkv", 3072, 1024), ("o", 1024, 1024), ("ffn1", 4096, 1024), ("ffn2", 1024, 4096)]
M = 32768
assert len(fe) == 12 and len(wr) == 12, (len(fe), len(wr))      # 3 launches per shape; the last of each is taken
per = []
for i, (name, N, K) in enumerate(shapes):
    f_kb, w_kb = fe[3 * i + 2][1], wr[3 * i + 2][1]
    alg = (M * K + N * K + M * N) * 2
    tiles_m, tiles_n = M // 256, N // 256
    r, c = 8, 4
    model = (M * K * 2) * tiles_n * (1.0 / c) + (N * K * 2) * tiles_m * (1.0 / r)
    per.append({"shape": name, "M": M, "N": N, "K": K, "fetch_bytes": f_kb * 1024 * 2, "write_bytes": w_kb * 1024, "algorithmic_bytes": alg,
                "model_fetch_bytes": model})
mean_t = sum(p["fetch_bytes"] + p["write_bytes"] for p in per) / len(per)
mean_a = sum(p["algorithmic_bytes"] for p in per) / len(per)
json.dump({"kernel": "gemm_pp64_kernel<false,true,256>",
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, csv; tools/pmc_gemm.sh) over tools/pmc_gemm.py; FETCH_SIZE (KB) doubled per the gfx950 "
                     "correction of MI355X_MICROARCH.md (HBM section); counters are fabric-side and include Infinity-Cache hits",
           "per_shape": per, "mean_traffic_bytes_per_launch": mean_t, "mean_algorithmic_bytes_per_launch": mean_a, "ratio": mean_t / mean_a,
           "mean_hbm_bytes_per_launch": mean_t,
           "model": "fetch per tile = its A panel / c + its B panel / r with r x c = 8 x 4 tiles running at once per XCD (32 CUs): model_fetch_bytes"}, open(out, "w"), indent=1)
for p in per:
    print(p["shape"], f"fetch {p['fetch_bytes']/1e6:.1f} MB (model {p['model_fetch_bytes']/1e6:.1f})  write {p['write_bytes']/1e6:.1f} MB  algorithmic {p['algorithmic_bytes']/1e6:.1f} MB")
print("ratio", mean_t / mean_a)

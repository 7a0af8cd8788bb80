"""profiles/r0N_pmc_image_tower.json (python tools/pmc_image_report.py [tag, default r04] [one-stream csv]): measured fabric-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_image.sh)
of the image tower's largest kernels next to their ALGORITHMIC bytes and their in-step duration (one-stream rocprofv3 trace).

FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 tallies the 128-B requests of wide streaming reads at 64 B); WRITE_SIZE is
taken as reported.  Both counters sit on the L2's memory side and include Infinity-Cache hits.  Algorithmic bytes per launch are the mean over
the layers a kernel instantiation serves in EfficientNet-B4 @ 224, B = 256 (2-byte elements: fp16 forward tensors, bf16 gradients)."""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodalsimilar_amd.effnet import build_arch
B = 256
TAG = sys.argv[1] if len(sys.argv) > 1 else "r04"
ONE_STREAM = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", {"r03": "r03_b", "r04": "r04_c"}.get(TAG, TAG) + "_bench_cfg4_kernel_stats_one_stream.csv")
a = build_arch("efficientnet_b4")
h = 112
fam = {}
def add(key, rd, wr):
    n, r, w = fam.get(key, (0, 0, 0))
    fam[key] = (n + 1, r + rd, w + wr)
for b in a.blocks:
    ho = h // b.stride
    e_in, e_out = B * h * h * b.mid, B * ho * ho * b.mid
    plain = b.type == "ds"
    if b.stride == 1 and b.k == 5 and not plain and TAG >= "r04":      # round 4: the 5 x 5 blocks on the matrix cores (backward from 14^2 up)
        add("dwm_fwd_kernel<5, true, false>", 2 * e_in, 2 * e_out)
        add("dwm_bwd_kernel<5>" if ho >= 14 else "dwt_bwd_kernel<5, false, 1>", 2 * 3 * e_out, 2 * e_in)
    elif b.stride == 1:
        add(f"dwt_fwd_kernel<{b.k}, 1, {'false' if plain else 'true'}>", 2 * e_in, 2 * e_out)                       # z1 (or x) -> z2
        add(f"dwt_bwd_kernel<{b.k}, {'true' if plain else 'false'}, 1>", 2 * 3 * e_out + (2 * e_out if plain and b.skip else 0), 2 * e_in)   # dy, z2, z1 -> dpre1
    else:
        add(f"dwt_fwd_kernel<{b.k}, 2, true>", 2 * e_in, 2 * e_out)
        add(f"dwt_bwd_kernel<{b.k}, false, 2>", 2 * (2 * e_out + e_in), 2 * e_in)                                      # dy, z2 (output plane), z1 -> dpre1 (input plane)
    if not plain and not (b.mid <= 352 and b.cin <= 64):      # expand BatchNorm's backward as its own pass (where mmsim_pw_expand_bwd is not eligible)
        add("bn_bwd_apply_kernel", 2 * 2 * e_in, 2 * e_in)
    add("pool_bn_bwd_kernel", 2 * 2 * e_out, 0)                                                                        # z2, da2g -> [5][B][C] sums
    add("pool_bn_act_kernel", 2 * e_out, 0 if ho > 28 or (ho == 28 and b.mid <= 192) else 2 * e_out)                 # z2 -> squeeze (+ a2 where it is kept)
    add("bn_bwd_apply_kernel", 2 * 2 * B * ho * ho * b.cout, 2 * B * ho * ho * b.cout)                                # bn3: dy, z3 -> dz3
    h = ho
pmc = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_image.json")))
dur = {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(ONE_STREAM))}
def match(table, key):
    k2 = key.replace(" ", "")
    for name in table:
        n2 = name.replace(" ", "")
        if k2 in n2:
            return name
        # mangled names: dwt_fwd_kernelILi5ELi1ELb1EE...
        if "<" in key:
            base, args = key.split("<")
            args = args.rstrip(">").split(",")
            mang = base + "I" + "".join(("Lb1E" if x.strip() == "true" else "Lb0E" if x.strip() == "false" else f"Li{x.strip()}E") for x in args) + "E"
            if mang in n2:
                return name
    return None
rows = []
for key, (n, rd, wr) in fam.items():
    pn = match({r["kernel"]: 1 for r in pmc}, key)
    dn = match(dur, key)
    if pn is None or dn is None:
        continue
    p = next(r for r in pmc if r["kernel"] == pn)
    calls, tot = dur[dn]
    us = tot / calls / 1e3
    alg = (rd + wr) / n
    fetch, write = 2.0 * p["fetch_kb_per_launch_raw"] * 1024, p["write_kb_per_launch"] * 1024
    rows.append({"kernel": key, "launches_per_step": calls // 10, "mean_us_in_step_one_stream": round(us, 1),
                 "algorithmic_bytes_per_launch": int(alg), "algorithmic_read": int(rd / n), "algorithmic_write": int(wr / n),
                 "measured_fetch_bytes_per_launch": int(fetch), "measured_write_bytes_per_launch": int(write),
                 "traffic_over_algorithmic": round((fetch + write) / alg, 2), "achieved_algorithmic_TBps": round(alg / us / 1e6, 2),
                 "frac_of_8TBps": round(alg / us / 1e6 / 8.0, 3)})
rows.sort(key=lambda r: -r["mean_us_in_step_one_stream"] * r["launches_per_step"])
doc = {"what": __doc__.strip(), "note": "pool_bn_act / bn_bwd_apply serve more layers in the PMC run (image tower alone: also the head BatchNorm) than modelled rows; "
       "dwt_* rows: halo re-reads of the LDS-staged tiles and the per-block partial slabs (BatchNorm sums, tap-major weight gradients) are what the measured traffic adds to the algorithmic bytes",
       "rows": rows}
json.dump(doc, open(os.path.join(ROOT, "profiles", TAG + "_pmc_image_tower.json"), "w"), indent=1)
for r in rows:
    print(f"{r['kernel']:34s} x{r['launches_per_step']:3d} {r['mean_us_in_step_one_stream']:7.1f} us  alg {r['algorithmic_bytes_per_launch']/1e6:7.1f} MB  measured {(r['measured_fetch_bytes_per_launch']+r['measured_write_bytes_per_launch'])/1e6:7.1f} MB  ratio {r['traffic_over_algorithmic']:5.2f}  {r['achieved_algorithmic_TBps']:5.2f} TB/s")

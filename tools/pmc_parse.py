"""Sum a rocprofv3 --pmc counter (csv output) per kernel name over the LAST launch of each distinct grid size.
usage: pmc_parse.py <dir> <COUNTER> [kernel-substring]"""
import csv, glob, sys
d, ctr = sys.argv[1], sys.argv[2]
sub = sys.argv[3] if len(sys.argv) > 3 else "gemm_pp64"
rows = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr or sub not in r["Kernel_Name"]:
            continue
        key = int(r["Dispatch_Id"])
        rows[key] = (r["Kernel_Name"][:60], int(r["Grid_Size"]), rows.get(key, (0, 0, 0.0))[2] + float(r["Counter_Value"]))
for k in sorted(rows):
    print(k, rows[k][0], rows[k][1], rows[k][2])

"""Per-(kernel, launch geometry) totals from a rocprofv3 results DB: which SHAPES of a kernel the time goes to.
usage: prof_shapes.py run_results.db <steps> [name-substring ...]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]); pats = sys.argv[3:]
rows = db.execute("select name, grid_x, grid_y, grid_z, lds_size, vgpr_count + accum_vgpr_count, count(*), sum(end-start), avg(end-start) "
                  "from kernels group by 1,2,3,4,5 order by 8 desc").fetchall()
tot = 0.0
for n, gx, gy, gz, lds, vg, cnt, s, a in rows:
    if pats and not any(p in n for p in pats):
        continue
    tot += s
    print(f"{s/1e6/steps:7.3f} ms/step {cnt/steps:6.1f}/step {a/1e3:8.1f}us grid({gx//256 if gx%256==0 else gx},{gy},{gz}) lds {lds:6d} vgpr {vg:3d}  {n[:70]}")
print(f"selected total {tot/1e6/steps:.2f} ms/step")

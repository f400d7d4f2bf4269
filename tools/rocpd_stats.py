"""Per-kernel statistics (calls, average / min / max duration, grid, LDS, scratch) from a rocprofv3 rocpd SQLite file
(ROCm 7.2 writes <name>_results.db by default): python tools/rocpd_stats.py <results.db> [csv out]"""
import sqlite3
import subprocess
import sys

db = sqlite3.connect(sys.argv[1])
q = """select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), sum(d.end - d.start),
              max(d.grid_size_x), max(d.workgroup_size_x), max(d.group_segment_size), max(d.private_segment_size), max(s.arch_vgpr_count)
       from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
       group by s.kernel_name order by 6 desc"""
rows = db.execute(q).fetchall()
tot = sum(r[5] for r in rows)
names = subprocess.run(["c++filt"], input="\n".join(r[0].replace(".kd", "") for r in rows), capture_output=True, text=True).stdout.splitlines()
lines = ["kernel,calls,avg_us,min_us,max_us,share,grid,wg,lds_bytes,scratch_bytes,vgprs"]
for r, n in zip(rows, names):
    lines.append('"%s",%d,%.2f,%.2f,%.2f,%.4f,%d,%d,%d,%d,%d' % (n[:110], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / tot, r[6], r[7], r[8], r[9], r[10]))
out = "\n".join(lines)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
print(out)

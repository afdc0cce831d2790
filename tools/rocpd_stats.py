#!/usr/bin/env python3
"""Per-kernel summary (count, total, avg, min, max, share) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` writes DIR/NAME_results.db on ROCm 7.2).
Usage: tools/rocpd_stats.py results.db [--by-grid] > profiles/xyz_kernel_stats.txt
--by-grid splits a kernel symbol by launch configuration (grid and dynamic LDS bytes): the same template
instantiation serves several problem shapes (e.g. the 3072-d and the 2048-d SwiGLU GEMV)."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    cur = db.cursor()
    q = ("select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.end - d.start, d.group_segment_size "
         "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id")
    agg = {}
    tot = 0
    for name, gx, gy, gz, dur, lds in cur.execute(q):
        key = (name, gx, gy, gz, lds) if by_grid else (name,)
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
        tot += dur
    span = cur.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print(f"# kernels total busy {tot/1e6:.3f} ms over a {(span[1]-span[0])/1e6:.3f} ms span; {sum(a[0] for a in agg.values())} dispatches")
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        name = key[0]
        try:
            import subprocess
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            pass
        name = name.replace("(anonymous namespace)::", "").replace("ua2_linear_args", "args")
        if by_grid:
            name += f"  grid={key[1]}x{key[2]}x{key[3]} lds={key[4]}"
        print(f"{a[0]:8d} {a[1]/1e6:10.3f} {a[1]/a[0]/1e3:9.2f} {a[2]/1e3:9.2f} {a[3]/1e3:9.2f} {100*a[1]/tot:6.2f}  {name[:170]}")


if __name__ == "__main__":
    main()

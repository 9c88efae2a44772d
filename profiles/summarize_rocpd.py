#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (`*_results.db`, the default output of
`rocprofv3 --kernel-trace --stats`) into the small text summary committed under profiles/.

  python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/rNN_name.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}")
    print("# columns: calls | total_ms | avg_ms | pct | kernel")
    for name, calls, total, avg, pct in cur.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 12"):
        print(f"{calls:6d} | {total / 1e3:12.3f} | {avg / 1e3:12.4f} | {pct:6.2f} | {name[:150]}")
    print("# per-dispatch resources of the lzf kernels (first dispatch of each)")
    seen = set()
    for row in cur.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size, duration from kernels where name like '%lzf%' order by start"):
        if row[0] in seen:
            continue
        seen.add(row[0])
        print(f"{row[0][:90]} grid={row[1]} wg={row[2]} lds={row[3]} vgpr={row[4]} agpr={row[5]} sgpr={row[6]} scratch={row[7]} dur_ms={row[8] / 1e6:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.

    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r1_kernel_stats.md
"""
import sqlite3
import sys


def main(path, extra=""):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# rocprofv3 --kernel-trace --stats summary\n\nsource: `{path}` {extra}\n")
    print("| kernel | calls | total (us) | avg (us) | % |\n|---|---:|---:|---:|---:|")
    for name, calls, total, avg, pct in rows:
        short = name.replace("unsigned short", "bf16").replace("d2s::", "")
        if len(short) > 110:
            short = short[:107] + "..."
        print(f"| `{short}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")
    try:
        q = ("select name, grid_x, grid_y, grid_z, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
             "count(*), avg(duration) from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc limit 40")
        print("\n## by launch shape (top 40 by total time; duration in ns)\n")
        print("| kernel | grid | wg | vgpr | agpr | sgpr | lds | scratch | calls | avg (us) |\n|---|---|---:|---:|---:|---:|---:|---:|---:|---:|")
        for r in cur.execute(q):
            short = r[0].replace("unsigned short", "bf16").replace("d2s::", "").split("(")[0]
            print(f"| `{short}` | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11] / 1e3:.2f} |")
    except sqlite3.Error as e:  # pragma: no cover
        print(f"\n(no per-shape table: {e})")


if __name__ == "__main__":
    main(sys.argv[1], " ".join(sys.argv[2:]))

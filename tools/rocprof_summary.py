#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into small text/JSON files for profiles/.

usage: rocprof_summary.py <run_dir> <out_prefix>
  <run_dir>/trace/*.db       from  rocprofv3 --kernel-trace --stats
  <run_dir>/pmc_*/*.db       from  rocprofv3 --pmc <counters> --kernel-trace   (one pass per counter group)
Writes <out_prefix>_kernels.md (per-kernel calls / total / average) and <out_prefix>_pmc.json (per-kernel mean
counter values per dispatch).
"""
import glob
import json
import sqlite3
import sys


def main():
    run, out = sys.argv[1], sys.argv[2]
    lines = []
    for dbf in sorted(glob.glob(f"{run}/trace/*.db")):
        cur = sqlite3.connect(dbf).cursor()
        lines.append(f"# rocprofv3 --kernel-trace --stats ({dbf.split('/')[-1]})\n")
        lines.append("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|")
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            lines.append(f"| `{name[:110]}` | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |")
        lines.append("")
        lines.append("| kernel | grid | workgroup | lds_B | vgpr | sgpr |\n|---|---|---|---|---|---|")
        for r in cur.execute("select name,grid_x,grid_y,grid_z,workgroup_x,lds_size,vgpr_count,sgpr_count from kernels group by name"):
            lines.append(f"| `{r[0][:80]}` | {r[1]}x{r[2]}x{r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} |")
    open(out + "_kernels.md", "w").write("\n".join(lines) + "\n")
    pmc = {}
    for dbf in sorted(glob.glob(f"{run}/pmc_*/*.db")):
        cur = sqlite3.connect(dbf).cursor()
        q = "select kernel_name,counter_name,count(*),avg(value),avg(duration) from counters_collection group by kernel_name,counter_name"
        for k, c, n, v, dur in cur.execute(q):
            e = pmc.setdefault(k, {})
            e[c] = {"dispatches": n, "mean": v, "mean_duration_ns_under_pmc": dur}
    json.dump(pmc, open(out + "_pmc.json", "w"), indent=1, sort_keys=True)
    print(open(out + "_kernels.md").read())
    for k, e in pmc.items():
        print(k[:70], {c: round(v["mean"], 1) for c, v in e.items()})


if __name__ == "__main__":
    main()

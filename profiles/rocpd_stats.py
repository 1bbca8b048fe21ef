#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (ROCm 7.2 default output of
`rocprofv3 --kernel-trace --stats`) into the usual per-kernel stats table (CSV on stdout)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name if len(name) < 150 else name[:147] + "..."


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,VGPR,AGPR,LDS,GridX,WorkgroupX")
    for r in rows:
        print(f"\"{short(r[0])}\",{r[1]},{r[2]},{r[3]:.0f},{r[4]},{r[5]},{100.0 * r[2] / total:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")


if __name__ == "__main__":
    main(sys.argv[1])

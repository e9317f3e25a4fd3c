#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output into the small text summaries kept under profiles/.

    python profiles/summarize_rocprof.py stats  gpurun_out/prof_r1/r1_results.db   > profiles/rNN_kernel_stats.txt
    python profiles/summarize_rocprof.py pmc    gpurun_out/pmc_fetch/f_results.db  > profiles/rNN_pmc_fetch.txt
"""
import sqlite3
import sys


def stats(db):
    c = sqlite3.connect(db)
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
    for name, calls, total, avg, pct in c.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        short = name.replace("msim::(anonymous namespace)::", "msim::").split("(")[0]
        print(f"{short:60s} {calls:6d} {total:12.3f} {avg:10.3f} {pct:7.2f}")
    print("\n# per-dispatch durations of msim::k_rewrite (ns), grid size, LDS, VGPR/SGPR")
    rows = list(c.execute("select duration,grid_x,lds_size,vgpr_count,sgpr_count from kernels "
                          "where name like '%k_rewrite%' order by start"))
    for r in rows[:30]:
        print("  ", r)
    if rows:
        d = [r[0] for r in rows]
        print(f"  n={len(d)} sum={sum(d)/1e3:.1f} us  avg={sum(d)/len(d)/1e3:.3f} us")


def pmc(db):
    c = sqlite3.connect(db)
    print(f"# rocprofv3 --pmc summary of {db} (counter values summed over dispatches of each kernel)")
    q = ("select kernel_name, counter_name, count(*), sum(value), sum(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(value) desc")
    for name, ctr, n, total, dur in c.execute(q):
        short = name.replace("msim::(anonymous namespace)::", "msim::").split("(")[0]
        print(f"{short:50s} {ctr:12s} dispatches={n:4d} sum={total:16.1f} total_ns={dur}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])

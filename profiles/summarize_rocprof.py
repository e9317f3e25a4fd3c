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


def text(db, wall_line=""):
    """Kernel stats of a CLI run plus achieved GB/s of the text kernels: bytes in + out per kernel over
    its total duration.  Sizes come from the run's own output line ("CLI wall ... outputs: out_ms.fa N MB, out_ms.vcf M MB")
    and the input size (--mb): k_gather reads the FASTA body text and writes the bases, k_frame reads the mutated bases and
    writes the wrapped body, k_vcf_lines<true> reads 16-byte records + REF/ALT sources (~ the text it writes) and writes the
    VCF text, k_vcf_lines<false> (line lengths) reads the same sources without writing text."""
    import re
    stats(db)
    m = re.search(r"out_ms\.fa (\d+) MB, out_ms\.vcf (\d+) MB", wall_line)
    if not m:
        return
    fa_mb, vcf_mb = float(m.group(1)), float(m.group(2))
    bases_mb = fa_mb * 60.0 / 61.0
    c = sqlite3.connect(db)
    print("\n# text kernels: bytes moved (in + out, MB, whole run) / total duration -> achieved GB/s (HBM peak 8000)")
    for pat, mb, what in (("%k_gather%", fa_mb + bases_mb, "file text in, bases out"),
                          ("%k_frame%", bases_mb + fa_mb, "mutated bases in, wrapped text out"),
                          ("%k_vcf_lines<true>%", 2.0 * vcf_mb, "records + REF/ALT sources in (~ text size), text out"),
                          ("%k_vcf_lines<false>%", vcf_mb, "records + REF/ALT sources in (line lengths only)")):
        row = c.execute("select sum(total_calls), sum(total_duration) from top_kernels where name like ?", (pat,)).fetchone()
        if row and row[1]:
            print(f"{pat.strip('%'):24s} calls={int(row[0]):4d} total_us={row[1]:10.1f} {mb:9.1f} MB -> {mb * 1e6 / (row[1] * 1e-6) / 1e9:8.1f} GB/s  ({what})")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "text": text}[sys.argv[1]](*sys.argv[2:])

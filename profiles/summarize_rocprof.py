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
    VCF text.  k_vcf_lines<false> (line lengths): up to round 4 it read the same sources without writing; since round 5 it
    reads the record, one base per SNP and the ends of an inversion only -- its line keeps round 4's NOMINAL accounting (the
    text's size) so that rounds compare, and says so."""
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
                          ("%k_vcf_lines<true%", 2.0 * vcf_mb, "records + REF/ALT sources in (~ text size), text out"),
                          ("%k_vcf_lines<false%", vcf_mb, "line lengths only; NOMINAL bytes = the text's size (round 4's accounting: it "
                                                          "read the REF/ALT sources; now 16-byte records + one base per SNP)")):
        row = c.execute("select sum(total_calls), sum(total_duration) from top_kernels where name like ?", (pat,)).fetchone()
        if row and row[1]:
            print(f"{pat.strip('%'):24s} calls={int(row[0]):4d} total_us={row[1]:10.1f} {mb:9.1f} MB -> {mb * 1e6 / (row[1] * 1e-6) / 1e9:8.1f} GB/s  ({what})")


def timeline(db, which="-1", max_rows="400", groups="1", marker="k_state_init"):
    """Kernel dispatches of ONE step in start order: offset from the step's first kernel, duration, queue, name -- and per
    queue the busy time and the gaps between consecutive kernels (what a latency-bound chain is made of).  `which`: index of
    the k_state_init group the step starts with (negative: from the end); `groups`: how many such groups make one step (compatible
    mode reseeds the second stream lazily at the end of a step: 2)."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(c.execute(f"select start, end, {qcol}, name, grid_x from kernels order by start"))
    short = lambda n: n.replace("msim::(anonymous namespace)::", "").replace("msim::", "").split("(")[0]
    # a step starts with the stream kernels of the first contig's reseed: k_state_init
    starts = [i for i, r in enumerate(rows) if marker in r[3]]
    firsts = [starts[i] for i in range(len(starts)) if i == 0 or rows[starts[i]][0] - rows[starts[i - 1]][0] > 1_000_000]
    if not firsts:
        print("no", marker, "found"); return
    k = int(which)
    if k < 0:
        k = len(firsts) - 1 + k                      # (the last step may be cut off by the end of the run: take the one before)
    lo = firsts[max(k, 0)]
    hi = firsts[k + int(groups)] if k + int(groups) < len(firsts) else len(rows)
    step = rows[lo:hi]
    t0 = step[0][0]
    print(f"# step {k}: {len(step)} dispatches, {(max(r[1] for r in step) - t0) / 1e3:.1f} us from first start to last end")
    per_q = {}
    for st, en, q, name, grid in step:
        per_q.setdefault(q, []).append((st, en, short(name)))
    for q, lst in sorted(per_q.items(), key=lambda kv: kv[1][0][0]):
        busy = sum(e - s for s, e, _ in lst)
        gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
        pos = [g for g in gaps if g > 0]
        print(f"# queue {q}: {len(lst)} kernels, busy {busy / 1e3:.1f} us, first at {(lst[0][0] - t0) / 1e3:.1f}, last end {(lst[-1][1] - t0) / 1e3:.1f}, "
              f"positive gaps {len(pos)} sum {sum(pos) / 1e3:.1f} us")
    print(f"{'start_us':>10s} {'dur_us':>8s} {'queue':>6s}  kernel (grid)")
    for st, en, q, name, grid in step[:int(max_rows)]:
        print(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:8.1f} {q!s:>6s}  {short(name)} ({grid})")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "text": text, "timeline": timeline}[sys.argv[1]](*sys.argv[2:])

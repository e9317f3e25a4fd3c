#!/usr/bin/env python3
"""Headline benchmark: Mbases mutated / s on a synthetic 3 Gb, 24-contig genome, ARGS SNP rate 0.01,
titv 2.0 (BASELINE.json configs[1]), bit-compatible RNG streams (seeds 42/42).

A "step" is one pass of the hot path -- PLAN (position/type draw -> record table) + APPLY (HIP
rewrite kernel) -- over the whole genome, with the genome already resident in HBM as uint8 and the
results (mutated stream + record table) left in HBM.  The driver contract (flags, barrier + sync
around exactly K steps, max over ranks, ONE JSON line from rank 0) is kept; torch.distributed is
used only as rendezvous/barrier plumbing when launched with N > 1, the product path is ctypes ->
libmsim.so.

    python bench.py                      # 1 GPU, K=3, W=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Extra keys in the JSON line: "roofline" (rewrite kernel vs HBM peak, measured live with HIP events
on the library's stream) and "cpu_baseline" (the CPU oracle timed on a bounded sample, rank 0; its
"matches_gpu" says whether that sample genome, pushed through `one_step` -- the timed function, in the
timed call order -- on the GPU, gives the oracle's bytes).

What N > 1 means here.  The reference draws from two global generators contig after contig
(mutator.py:111-141), so in this bit-identical mode ONE genome over N GPUs is Amdahl-bound: every rank
walks the whole stream chain (msim_plan_chain for contigs it does not own) and only emission + APPLY are
divided -- at most 1.66x for this workload, ~1.05x for the SV mix, whatever N ("amdahl_ceiling",
measured).  That strong-scaling number is the N > 1 headline ("one genome", BASELINE's metric); the line
also carries what does scale: "weak_replicas" (N genomes at once, one per GPU, own seeds: linear) and
"one_genome_sharded_fast_rng" (--rng fast: counter-based draws, NOT the reference's numbers; nothing
chains, a rank plans and applies only what it owns).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

# GRCh38 primary chromosome lengths (chr1-22, X, Y); scaled so the total is exactly `total`
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
          138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
          83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def contig_lengths(total: int) -> list[int]:
    s = sum(GRCH38)
    out = [int(round(x * total / s)) for x in GRCH38]
    out[-1] += total - sum(out)
    return out


def workload_settings(lengths, snp=0.01, titv=2.0, extra=None):
    """Settings tree for `args -sn 0.01 -titv 2.0` over the synthetic contigs (host package)."""
    import mutation_simulator_amd as msa

    class Rec:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    class FakeFasta:     # only contig count + lengths are needed to build settings
        def __init__(self, ls):
            self.ls = ls

        def keys(self):
            return [f"chr{i+1}" for i in range(len(self.ls))]

        def __getitem__(self, k):
            return Rec(self.ls[k] if isinstance(k, int) else self.ls[int(k[3:]) - 1])

    argv = ["synthetic.fa", "args", "-sn", repr(snp), "-titv", repr(titv)] + list(extra or [])
    args = msa.get_args(argv)
    sim = msa.SimulationSettings.from_args(args, FakeFasta(lengths), True)
    return sim


C4_FIXTURE = ROOT / "tests" / "golden" / "c4_blocks.npz"


C4_STD_SNP = "sn 0.01"
C4_STD_SV = ("sn 0.005 in 0.001 inmin 1 inmax 50 de 0.001 demin 1 demax 50 du 0.0005 dumin 50 dumax 500 "
             "iv 0.0005 ivmin 50 ivmax 500")


def c4_rmt_text(lengths, seed=4, std_line=C4_STD_SNP) -> str:
    """BASELINE configs[3]: a NON-overlapping gene-blocking RMT (std `sn 0.01`, `a-b None` blocks) plus hot (`sn 0.05`)
    and cold (`sn 0.001`) ranges and 1 kb `sn 0.2` hot spots that take CPython's pool-path sample.

    For the bench genome (lengths == contig_lengths(3e9)) the blocks are the reference's own example after
    interval-merging (`data/Example RMT files/Homo_sapiens.rmt` as shipped overlaps and crashes the reference,
    SURVEY.md section 2): the data fixture tests/golden/c4_blocks.npz written by tests/golden/make_c4_blocks.py in the
    build container (33.9 k ranges, 43-67 % of a contig blocked).  For any other genome (the scaled-down test
    genomes) or without the fixture: a synthetic file in the same style from this function's own deterministic
    generator (log-normal block lengths, median 3.6 kb)."""
    if C4_FIXTURE.exists():
        z = np.load(C4_FIXTURE)
        if list(z["lengths"]) == list(lengths):
            what = ("None", "sn 0.05", "sn 0.001", "sn 0.2")
            out = ["std", "it None", std_line, ""]
            for ci in range(len(lengths)):
                out.append(f"chr {ci + 1}")
                for a, e, k in zip(z[f"s{ci}"].tolist(), z[f"e{ci}"].tolist(), z[f"k{ci}"].tolist()):
                    out.append(f"{a}-{e} {what[k]}")
            return "\n".join(out) + "\n"
    rs = np.random.RandomState(seed)
    out = ["std", "it None", std_line, ""]
    for ci, L in enumerate(lengths):
        out.append(f"chr {ci + 1}")
        n_blocks = max(1, int(L / 75_000))
        block_len = np.minimum(np.exp(rs.normal(np.log(3600.0), 2.09, n_blocks)).astype(np.int64) + 50, 2_000_000)
        kinds = rs.choice(4, size=n_blocks, p=[0.96, 0.015, 0.015, 0.01])      # None / hot / cold / pool-path hot spot
        block_len[kinds == 3] = 1000
        block_len[kinds == 1] = rs.randint(1_000, 100_000, int((kinds == 1).sum()))
        block_len[kinds == 2] = rs.randint(100_000, 1_000_000, int((kinds == 2).sum()))
        free = L - int(block_len.sum()) - 2 * n_blocks - 1000
        while free < L // 4:                                                     # keep at least a quarter unblocked
            block_len = np.maximum(block_len // 2, 50)
            free = L - int(block_len.sum()) - 2 * n_blocks - 1000
        gaps = rs.dirichlet(np.ones(n_blocks + 1)) * free
        at = 1                                                                   # RMT positions are 1-based inclusive
        for b in range(n_blocks):
            at += int(gaps[b]) + 2
            a, e = at, at + int(block_len[b]) - 1
            what = ("None", "sn 0.05", "sn 0.001", "sn 0.2")[kinds[b]]
            out.append(f"{a}-{e} {what}")
            at = e + 1
        assert at < L
    return "\n".join(out) + "\n"


def workload_settings_rmt(lengths, rmt_text: str):
    """Settings tree of an RMT file over the synthetic contigs, through the package's own RMT parser."""
    import tempfile

    import mutation_simulator_amd as msa

    class Rec:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    class FakeFasta:
        def __init__(self, ls):
            self.ls = ls

        def keys(self):
            return [f"chr{i+1}" for i in range(len(self.ls))]

        def __getitem__(self, k):
            return Rec(self.ls[k] if isinstance(k, int) else self.ls[int(k[3:]) - 1])

    with tempfile.NamedTemporaryFile("w", suffix=".rmt", delete=False) as f:
        f.write(rmt_text)
        path = f.name
    try:
        return msa.SimulationSettings.from_rmt(Path(path), FakeFasta(lengths), True)
    finally:
        os.unlink(path)


def one_step(eng, sim, cids, my_contigs, seed=42, plan_descriptors=None, lengths=None):
    """Walk the chain of both RNG streams over every contig in order; PLAN with emission + APPLY this rank's contigs
    (`lengths` given: the others through msim_plan_chain -- stream positions only)."""
    from mutation_simulator_amd import mutator as mm
    from mutation_simulator_amd.sharding import run_sharded_pass
    eng.seed(seed, seed)
    run_sharded_pass(eng, sim, cids, my_contigs, plan_descriptors or mm.plan_descriptors, lengths=lengths)
    eng.sync()


C3_FLAGS = ["-in", "0.001", "-inmin", "1", "-inmax", "50", "-de", "0.001", "-demin", "1", "-demax", "50",
            "-du", "0.0005", "-dumin", "50", "-dumax", "500", "-iv", "0.0005", "-ivmin", "50", "-ivmax", "500"]

README_FLAGS = ["-in", "0.01", "-de", "0.01", "-du", "0.01", "-iv", "0.01", "-tl", "0.01"]     # + -sn 0.01: README.md:430-446

WORKLOADS = {
    "c2": {"mode": "ARGS", "what": "-sn 0.01 -titv 2.0 (BASELINE configs[1])", "kernel": "msim::k_rewrite_snp_b",   # (an emission group per launch: four contigs)
           "metric": "Mbases mutated/sec on 3 Gb synthetic genome, ARGS SNP rate 0.01"},
    "c3": {"mode": "ARGS", "what": "full SV mix (BASELINE configs[2]): -sn 0.005 -in/-de 0.001 len 1-50, -du/-iv 0.0005 len 50-500",
           "kernel": "msim::k_rewrite_b<140>", "metric": "Mbases mutated/sec on 3 Gb synthetic genome, ARGS full SV mix"},   # (threes)
    "c4": {"mode": "RMT", "what": "RMT mode, gene-blocking file with hot/cold spots (BASELINE configs[3])",
           "kernel": "msim::k_rewrite_snp_b", "metric": "Mbases mutated/sec on 3 Gb synthetic genome, RMT hot/cold/blocked ranges"},
    "c4sv": {"mode": "RMT", "what": "RMT mode, the configs[3] gene-blocking file with the configs[2] SV mix as its std line "
                                    "(every gap between two blocked genes draws SN/IN/DE/DU/IV; hot/cold ranges keep their own settings)",
             "kernel": "msim::k_rewrite_b<140>", "metric": "Mbases mutated/sec on 3 Gb synthetic genome, RMT gene blocks + SV std line"},
    "readme": {"mode": "ARGS", "what": "the reference README's benchmark flags: -sn -in -de -du -iv -tl 0.01 each, default lengths "
                                       "(180 M candidates per 3 Gb, translocations linked per contig)",
               "kernel": "msim::k_rewrite_b<140|1024>", "metric": "Mbases mutated/sec on 3 Gb synthetic genome, ARGS every type at 0.01"},
}


def build_settings(workload: str, lengths):
    if workload == "c2":
        return workload_settings(lengths)
    if workload == "c3":
        return workload_settings(lengths, snp=0.005, titv=1.0, extra=C3_FLAGS)
    if workload == "readme":
        return workload_settings(lengths, snp=0.01, titv=1.0, extra=README_FLAGS)
    if workload == "c4sv":
        return workload_settings_rmt(lengths, c4_rmt_text(lengths, std_line=C4_STD_SV))
    return workload_settings_rmt(lengths, c4_rmt_text(lengths))


def host_info():
    model = "unknown"
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model, os.cpu_count()


def cpu_baseline(sample_total: int, workload: str = "c2", n_contigs: int = 4, device=None):
    """CPU oracle (single-threaded C restatement of the reference path) on a bounded sample of the same
    workload: `n_contigs` contigs totalling `sample_total` bases with the workload's settings, seeds 42/42.
    The reference itself cannot travel to the GPU box; its wall time measured in the build container
    (tests/golden/reference_timing.json, written by `make_goldens.py timing`) is carried along.

    `device` given: the same sample genome then goes through `one_step` -- the function the timed steps run: plan + apply
    every contig, ONE synchronisation, reads afterwards (emission groups, batched rewrite launches) -- on that GPU, and
    `matches_gpu` says whether every contig's framed Fasta body and VCF text equal the oracle's (SHA-256 per contig) and
    whether both MT19937 streams end where the oracle's do.  The oracle stays the checker: nothing of it is inside `value`."""
    import hashlib

    from oracle import oracle as orc
    from oracle.support import dump_sim, synth_host      # (synth_host: the device generator's host twin)
    lengths = [sample_total // n_contigs] * n_contigs
    sim = build_settings(workload, lengths)
    dump = dump_sim(sim)
    by_number = {ch["number"]: ch for ch in dump["chromosomes"]}
    o = orc.Oracle()
    o.seed(42, 42)
    o.configure(dump)
    dt = 0.0
    want = {}
    for chrom in sim.chromosomes:                      # mutate()'s contig loop, mutator.py:111-141
        i = chrom.number
        name = f"chr{i+1}"
        bases = synth_host(lengths[i], 1000 + i)
        t0 = time.perf_counter()
        fa, vcf, _ = o.mutate_contig_stream(bases, name, f"{name} synthetic", 60, by_number[i]["ranges"])
        dt += time.perf_counter() - t0
        head = len(f">{name} synthetic\n")
        want[i] = (hashlib.sha256(memoryview(fa)[head:]).hexdigest(), hashlib.sha256(vcf).hexdigest(), vcf.count(b"\n"))
        del bases, fa, vcf
    total = sum(lengths)
    model, cores = host_info()
    out = {"value": round(total / dt / 1e6, 3), "unit": "Mbases/s", "cores": 1, "kind": "port",
           "cpu_model": model, "node_cores": cores,
           "sample": f"{n_contigs} contigs x {lengths[0]/1e6:.0f} Mb, {WORKLOADS[workload]['what']}, seeds 42/42, "
                     f"Fasta framing + VCF text included ({dt:.1f} s of CPU work)"}
    if device is not None:
        try:
            out["matches_gpu"], out["matches_gpu_detail"] = gpu_matches_oracle(device, sim, lengths, want, o)
        except Exception as e:  # noqa: BLE001
            out["matches_gpu"], out["matches_gpu_detail"] = False, {"error": f"{type(e).__name__}: {e}"}
    rt = ROOT / "tests" / "golden" / "reference_timing.json"
    if rt.exists():
        try:
            ref = json.loads(rt.read_text())
            out["reference_survey_mbases_s"] = {r["name"]: r["mbases_per_s"] for r in ref["runs"]}
            out["reference_survey_note"] = (f"real reference CLI, 1 core of {ref['cpu_model']} in the build container, "
                                            f"in-memory pyfaidx stand-in; README publishes {ref['published_readme_mbases_s']} Mbases/s")
        except Exception:  # noqa: BLE001
            pass
    return out


def gpu_matches_oracle(device, sim, lengths, want, oracle):
    """The cpu_baseline sample genome through `one_step` (the timed function, in the timed order) on the GPU; per contig the
    SHA-256 of the framed Fasta body and of the device-rendered VCF lines against the oracle's (`want`), line counts, and the
    final positions of both MT19937 streams."""
    import hashlib
    import random

    from mutation_simulator_amd import _ffi
    from mutation_simulator_amd import mutator as mm
    eng = _ffi.Engine(device)
    try:
        cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
        eng.set_params(mm.params_descriptor(sim))
        eng.reset_stats()
        one_step(eng, sim, cids, list(range(len(lengths))), 42, mm.plan_table)
        st = eng.stats()
        bad = []
        lines = 0
        for chrom in sim.chromosomes:
            i = chrom.number
            name = f"chr{i+1}"
            text = eng.fetch_sequence_framed(cids[i], 60, guess_len=lengths[i])
            _, n_rec, _ = eng.result_sizes(cids[i], applied=False)
            vcf = eng.render_vcf_device(cids[i], name, guess=n_rec * 48 + 256)
            got = (hashlib.sha256(memoryview(text)).hexdigest(), hashlib.sha256(memoryview(vcf)).hexdigest(), bytes(vcf).count(b"\n"))
            lines += got[2]
            if got != want[i]:
                bad.append(name)
            del text, vcf
        streams_ok = True
        for stream in (0, 1):
            mt, pos = eng.get_mt_state(stream)
            omt, opos = oracle.get_state(stream)
            a, b = random.Random(), random.Random()
            a.setstate((3, tuple(int(x) for x in mt) + (int(pos),), None))
            b.setstate((3, tuple(omt) + (opos,), None))
            streams_ok = streams_ok and [a.getrandbits(32) for _ in range(8)] == [b.getrandbits(32) for _ in range(8)]
        detail = {"what": "the sample genome through bench.one_step (plan + apply all contigs, one sync, then reads) on the GPU: "
                          "SHA-256 of each contig's framed Fasta body and device VCF text == the oracle's, both MT19937 streams "
                          "end where the oracle's do",
                  "contigs": len(lengths), "contigs_differing": bad, "vcf_lines": lines, "streams_end_equal": streams_ok,
                  "apply_launches": st["apply_launches"], "plan_engines": engines_of(st, 1)}
        return (not bad) and streams_ok, detail
    finally:
        eng.close()


def e2e_cli(eng, total_bases=1_200_000_000, n_contigs=6, where="tmpfs"):
    """End to end through the product CLI: a FASTA file -> `_ms.fa` + `_ms.vcf` (`args -sn 0.01 -titv 2.0`, seeds 42/42), wall
    time of `__main__.main` with its stage split.  Secondary number: H2D / D2H, FASTA parsing, line framing,
    VCF text and the file writes are all inside; `value` of the headline is not.  `where`: "tmpfs" (/dev/shm: input and
    outputs are memory pages; a write costs the kernel's per-page work of shmem) or "tmpdir" (the system's temp dir: the
    page cache of a disk file system; nobody waits for the writeback -- as for any CLI that exits after close())."""
    import shutil
    import tempfile

    from mutation_simulator_amd import __main__ as cli
    base = None
    if where == "tmpfs":
        base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    td = Path(tempfile.mkdtemp(prefix="msim_e2e_", dir=base))
    try:
        fa = td / "in.fa"
        L = total_bases // n_contigs
        with open(fa, "wb") as f:                      # synthesised + line-wrapped on the device (not timed)
            for i in range(n_contigs):
                cid = eng.add_contig_synthetic(L, 5000 + i)
                eng.set_params(_snp_only_params())
                eng.plan_contig(cid, [])
                eng.apply_contig(cid)
                text = eng.fetch_sequence_framed(cid, 60, guess_len=L)
                f.write(f">chr{i+1} synthetic e2e\n".encode())
                f.write(memoryview(text))
                if L % 60:
                    f.write(b"\n")
                eng.clear()
        in_bytes = fa.stat().st_size
        stats_path = td / "stats.json"
        argv = ["--seed", "42", "-q", "--bench-json", str(stats_path), "-o", str(td / "out"), str(fa), "args", "-sn", "0.01",
                "-titv", "2.0"]
        best = None
        for _ in range(3):                             # (the first run also brings up the output channels' threads and rings)
            for o in td.glob("out_ms*"):
                o.unlink()
            t0 = time.perf_counter()
            cli.main(argv)
            dt = time.perf_counter() - t0
            st = json.loads(stats_path.read_text())
            if best is None or dt < best[0]:
                best = (dt, st)
        dt, st = best
        outs = {o.name: o.stat().st_size for o in td.glob("out_ms*")}
        return {"metric": "Mbases/s end to end through the CLI: FASTA file " + ("in tmpfs" if base else "in the temp dir") + " -> mutated Fasta + VCF files",
                "value": round(total_bases / dt / 1e6, 1), "unit": "Mbases/s", "wall_s": round(dt, 4),
                "workload": f"{total_bases/1e9:.1f} Gb, {n_contigs} contigs, args -sn 0.01 -titv 2.0, --seed 42 (best of 3 runs)",
                "files_in": str(td.parent) + (" (tmpfs)" if base else " (temp dir: page cache of its file system)"),
                "input_bytes": in_bytes, "output_bytes": outs, "cli_s": st.get("cli_s"),
                "contig_path_s": st.get("contig_path_s"), "plan_engines": engines_of(st, 1),
                "device_ms": {k: round(st[k], 2) for k in ("plan_gpu_ms", "apply_ms") if k in st}}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def fast_rng_steps(lengths, mm, steps=5, warmup=3):
    """c2 / c3 / c4 / c4sv with `--rng fast` (msim.h: MSIM_RNG_FAST): the counter-based PLAN engine (plan_fast.hip) instead of the
    reference's two MT19937 streams.  NOT a parity number -- same construction and distributions, different draws -- and never the
    headline: it shows what the step costs once PLAN has no sequential chain and no host work (plan_host = 0 by construction)."""
    from mutation_simulator_amd import _ffi
    eng = _ffi.Engine(int(os.environ.get("MSIM_BENCH_DEVICE", 0)), _ffi.RNG_FAST)
    out = {"what": "NOT stream-compatible with the reference (Philox4x32-10 draws; same construction and distributions): PLAN without "
                   "its sequential chain, same APPLY kernels; results left in HBM", "steps": steps, "warmup": warmup}
    try:
        cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
        for w in ("c2", "c3", "c4", "c4sv"):
            sim = build_settings(w, lengths)
            eng.set_params(mm.params_descriptor(sim))

            def step():
                eng.set_fast_key(42)
                for ch in sim.chromosomes:
                    eng.plan_contig(cids[ch.number], mm.plan_table(ch))      # (the range tables are made inside the step)
                    eng.apply_contig(cids[ch.number])
                eng.sync()
            for _ in range(warmup):
                step()
            eng.reset_stats()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            dt = time.perf_counter() - t0
            st = eng.stats()
            k_ms = st["apply_kernel_ms"]
            alg = st["bytes_in"] + st["bytes_out"] + 16 * st["records"]
            out[w] = {"value": round(sum(lengths) * steps / dt / 1e6, 3), "unit": "Mbases/s", "ms_per_step": round(dt / steps * 1e3, 3),
                      "records_per_step": st["records"] // steps, "stages_ms_per_step": stages_of(st, steps),
                      "plan_engines": engines_of(st, steps),
                      "rewrite_kernel_frac_of_hbm_peak": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else None,
                      "step_roofline": step_roofline(st, dt)}
    finally:
        eng.close()
    return out


def fast_rank_steps(lengths, mm, full_ms, n_list=(2, 4, 8)):
    """`--rng fast`: step of the most loaded rank of an N-way LPT partition (it plans and applies what it owns and skips the rest),
    for `predicted_one_genome_scaling`."""
    from mutation_simulator_amd import _ffi
    from mutation_simulator_amd.sharding import lpt_partition
    out = {}
    eng = _ffi.Engine(int(os.environ.get("MSIM_BENCH_DEVICE", 0)), _ffi.RNG_FAST)
    try:
        cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
        for w in ("c2", "c3"):
            sim = build_settings(w, lengths)
            tables = [mm.plan_table(ch) for ch in sim.chromosomes]
            eng.set_params(mm.params_descriptor(sim))
            row = {"full_step_ms": full_ms[w]}
            for n in n_list:
                parts = lpt_partition(lengths, n)
                mine = set(max(parts, key=lambda part: sum(lengths[i] for i in part)))

                def step():
                    eng.set_fast_key(42)
                    for ch, t in zip(sim.chromosomes, tables):
                        if ch.number in mine:
                            eng.plan_contig(cids[ch.number], t)
                            eng.apply_contig(cids[ch.number])
                        else:
                            eng.plan_chain(lengths[ch.number], t)
                    eng.sync()
                for _ in range(2):
                    step()
                t0 = time.perf_counter()
                for _ in range(5):
                    step()
                ms = (time.perf_counter() - t0) / 5 * 1e3
                row[str(n)] = {"rank_step_ms": round(ms, 3), "speedup": round(full_ms[w] / ms, 3)}
            out[w] = row
    finally:
        eng.close()
    return out


def fast_rng_isolated(total_bases):
    """The fast-mode numbers from a PROCESS OF THEIR OWN (`bench.py --fast-only`): what `--rng fast` is for a user.  Inside this
    process they came out 10-15 % slower (c2 1.74 instead of 1.54 ms, c3 2.55 instead of 2.22; round 6, scratch A/B in NOTES section 10):
    the bit-compatible context's hardware queues outlive it (even closed), and the fast context's four normal-priority streams
    then share what is left of the runtime's four queues per priority -- while with more queues per priority
    (GPU_MAX_HW_QUEUES >= 6) the host-chain workloads of THIS process lose 4 ms per step.  Falls back to this process on failure."""
    import subprocess
    cmd = [sys.executable, str(ROOT / "bench.py"), "--fast-only", "--total-bases", str(total_bases)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=dict(os.environ, MSIM_BENCH_NO_PMC="1"))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode == 0 and lines:
            out = json.loads(lines[-1])
            out["measured_in"] = "a process of its own (bench.py --fast-only)"
            return out
        err = (r.stderr or "")[-300:]
    except Exception as e:  # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    from mutation_simulator_amd import mutator as mm
    out = fast_rng_steps(contig_lengths(total_bases), mm)
    out["measured_in"] = f"this process (the child failed: {err})"
    return out


def fast_rng_sharded(lengths, owned, local_rank, barrier, max_over_ranks, mm, steps=5, warmup=2):
    """ONE genome over the N GPUs with `--rng fast`: every rank holds, plans and applies only the contigs it owns
    (msim_plan_chain for the others: their ordinal, nothing else).  value = 3 Gb / max-over-ranks step time."""
    from mutation_simulator_amd import _ffi
    eng = _ffi.Engine(int(os.environ.get("MSIM_BENCH_DEVICE", local_rank)), _ffi.RNG_FAST)
    out = {"what": "NOT stream-compatible with the reference (counter-based draws, same construction and distributions); ONE genome, "
                   "contigs sharded (LPT), no chain and no host work: a rank skips what it does not own; results left in HBM on the "
                   "owning GPU", "steps": steps, "warmup": warmup, "scaling": "strong"}
    try:
        mine = set(owned)
        cids = {i: eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths) if i in mine}
        for w in ("c2", "c3"):
            sim = build_settings(w, lengths)
            tables = [mm.plan_table(ch) for ch in sim.chromosomes]
            eng.set_params(mm.params_descriptor(sim))

            def step():
                eng.set_fast_key(42)
                for ch, t in zip(sim.chromosomes, tables):
                    if ch.number in mine:
                        eng.plan_contig(cids[ch.number], t)
                        eng.apply_contig(cids[ch.number])
                    else:
                        eng.plan_chain(lengths[ch.number], t)
                eng.sync()
            for _ in range(warmup):
                step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            eng.sync()
            dt = max_over_ranks(time.perf_counter() - t0)
            out[w] = {"value": round(sum(lengths) * steps / dt / 1e6, 3), "unit": "Mbases/s", "ms_per_step": round(dt / steps * 1e3, 3)}
    finally:
        eng.close()
    return out


def predict_scaling(measure, lengths, steps, dt_full_c2, sec, mm, n_list=(2, 4, 8)):
    """Step time of the most loaded rank of an N-way LPT partition, on this GPU: c2 and c3 in the bit-compatible mode (the rank
    walks every contig's stream positions, msim_plan_chain for what it does not own) and in `--rng fast` (it skips them)."""
    from mutation_simulator_amd.sharding import lpt_partition
    out = {"what": "predicted strong scaling of ONE 3 Gb genome, from 1 GPU: full step / step of the most loaded rank of an N-way LPT "
                   "partition of the 24 contigs (results left in HBM on the owning GPU; the RCCL gather to rank 0 is extra)",
           "n_gpus": list(n_list)}
    heavy = {}
    for n in n_list:
        parts = lpt_partition(lengths, n)
        heavy[n] = max(parts, key=lambda part: sum(lengths[i] for i in part))
    full = {"c2": dt_full_c2 / steps * 1e3, "c3": sec["c3"]["ms_per_step"] if "c3" in sec else None}
    for w, k in (("c2", min(steps, 10)), ("c3", 3)):
        if full[w] is None:
            continue
        row = {"full_step_ms": round(full[w], 3)}
        for n in n_list:
            dtn, _ = measure(w, k, 2, owned=heavy[n], step_seed=42)
            ms = dtn / k * 1e3
            row[str(n)] = {"rank_owns": f"{len(heavy[n])} of {len(lengths)} contigs, {sum(lengths[i] for i in heavy[n]) / 1e6:.0f} Mb",
                           "rank_step_ms": round(ms, 3), "speedup": round(full[w] / ms, 3)}
        out[w] = row
    fr = sec.get("fast_rng") or {}
    for w, row in (fr.get("rank_steps") or {}).items():   # (measured with the fast steps themselves, in their own process)
        out[f"fast_rng_{w}"] = row
    return out


def _snp_only_params():
    from mutation_simulator_amd import _ffi
    p = _ffi.Params()
    for i in range(8):
        p.block[i] = 1
    p.ti_lim = 1 << 52
    return p


LIVE_PMC = [False]            # set by main(): the headline's roofline.traffic is measured in child rocprofv3 passes


def roofline_of(st, workload, steps):
    launches = max(st["apply_launches"], 1)
    alg_bytes = st["bytes_in"] + st["bytes_out"] + 16 * st["records"]
    k_ms = st["apply_kernel_ms"]
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, source = None, None
    live = live_traffic(workload, WORKLOADS[workload]["kernel"]) if LIVE_PMC[0] else None
    if live is not None:
        traffic = live["traffic"]
        source = ("measured during this run: two child passes `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` of "
                  f"`bench.py --workload {workload} --steps 1`, per launch of the kernel; FETCH_SIZE x 2 (gfx950) = {live['fetch_bytes_x2']} "
                  f"+ WRITE_SIZE {live['write_bytes']} bytes")
    for name in (() if live is not None else ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json")):
        tf = ROOT / "profiles" / name
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text())[workload]["traffic_bytes_per_launch"]
                source = f"profiles/{name} (rocprofv3 --pmc TCC counters of an earlier run of this command, not this run)"
                break
            except Exception:  # noqa: BLE001
                traffic = None
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
            "kernel": WORKLOADS[workload]["kernel"], "algorithmic_bytes_per_launch": alg_bytes // launches,
            "avg_launch_ms": round(k_ms / launches, 4), "launches": launches}


def live_traffic(workload, kernel):
    """HBM bytes per launch of the rewrite kernel, measured NOW: two child runs of this file under `rocprofv3 --pmc` (FETCH_SIZE
    and WRITE_SIZE in passes of their own, never with a trace domain -- MI355X_MICROARCH.md, HBM section), one step each of the
    same workload; FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of a wide coalesced stream at 64 bytes).  None when
    rocprofv3 is not there or a pass fails (the line then says where the number comes from instead)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if prof is None or os.environ.get("MSIM_BENCH_NO_PMC"):
        return None
    per = {}
    td = tempfile.mkdtemp(prefix="msim_pmc_", dir="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(td, ctr)
            env = dict(os.environ, TMPDIR="/tmp", MSIM_BENCH_NO_PMC="1")
            cmd = [prof, "--pmc", ctr, "-d", out, "-o", "p", "--", sys.executable, str(ROOT / "bench.py"), "--workload", workload,
                   "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not dbs:
                return None
            c = sqlite3.connect(dbs[0])
            # (the exact kernel: "k_rewrite<140>" must not average over k_rewrite_snp / k_rewrite_b launches of the same run)
            short = kernel.split("::")[-1]
            pat = f"%{short}%" if "<" in short else f"%{short}(%"
            row = c.execute("select count(*), sum(value) from counters_collection where counter_name = ? and kernel_name like ?",
                            (ctr, pat)).fetchone()
            if (not row or not row[0]) and "<" not in short:     # (kernel names without their argument list in this rocprofv3)
                row = c.execute("select count(*), sum(value) from counters_collection where counter_name = ? and "
                                "(kernel_name = ? or kernel_name like ?)", (ctr, kernel, f"%::{short}")).fetchone()
            c.close()
            if not row or not row[0]:
                return None
            per[ctr] = row[1] / row[0] * 1024.0                # KB per dispatch -> bytes
    except Exception:  # noqa: BLE001
        return None
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return {"traffic": int(2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]), "fetch_bytes_x2": int(2.0 * per["FETCH_SIZE"]),
            "write_bytes": int(per["WRITE_SIZE"])}


def step_roofline(st, dt):
    """The same algorithmic bytes over the WHOLE timed region (PLAN chain included), for orientation: the step is bound by
    the latency of the per-contig stream-position chain (and, for c3 / c4, by a host core), not by HBM."""
    alg_bytes = st["bytes_in"] + st["bytes_out"] + 16 * st["records"]
    achieved = alg_bytes / dt / 1e9 if dt > 0 else 0.0
    return {"achieved": round(achieved, 1), "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "what": "algorithmic bytes of all APPLY launches / wall time of the timed steps"}


def spread_of(per_step_s):
    """Spread of the timed steps' own wall times (each step ends in a synchronisation), beside the mean the headline is made of."""
    v = sorted(x * 1e3 for x in per_step_s)
    n = len(v)
    if not n:
        return None
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"ms_per_step_min": round(v[0], 3), "ms_per_step_median": round(med, 3), "ms_per_step_max": round(v[-1], 3), "steps": n}


def host_walk_of(st):
    """The host-sequential stages apart from their surroundings (msim_timing, ABI 8): ns of one host core per walked item and the
    time the walks waited for the device -- a slower host shows in the first, a slower link / device in the second."""
    out = {}
    run, wait = st.get("host_walk_run_ms", 0.0), st.get("host_walk_wait_ms", 0.0)
    if st.get("host_walk_candidates"):
        out["host_walk_ns_per_candidate"] = round(run * 1e6 / st["host_walk_candidates"], 3)
        out["candidates_per_step"] = None                # (filled by the caller, which knows the step count)
    elif st.get("host_cut_words"):
        out["host_cut_ns_per_word"] = round(run * 1e6 / st["host_cut_words"], 3)
    if out:
        out["run_ms_total"] = round(run, 3)
        out["wait_ms_total"] = round(wait, 3)
    return out


def engines_of(st, steps):
    """Which PLAN engine the contigs of a step went through (msim_timing.contigs_*)."""
    return {k[len("contigs_"):]: st[k] // steps for k in st if k.startswith("contigs_") and st[k]}


def stages_of(st, steps):
    out = {"plan_host": round(st["plan_host_ms"] / steps, 3), "plan_gpu": round(st["plan_gpu_ms"] / steps, 3),
           "record_upload": round(st["upload_ms"] / steps, 3), "apply_all_kernels": round(st["apply_ms"] / steps, 3),
           "apply_rewrite_kernel": round(st["apply_kernel_ms"] / steps, 3)}
    hw = host_walk_of(st)
    if hw:                                              # engines with a host chain: the walk itself / its waits, per step
        out["host_walk_run"] = round(hw.pop("run_ms_total") / steps, 3)
        out["host_walk_wait"] = round(hw.pop("wait_ms_total") / steps, 3)
        hw.pop("candidates_per_step", None)
        if st.get("host_walk_candidates"):
            hw["host_walk_candidates_per_step"] = st["host_walk_candidates"] // steps
        out.update(hw)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)             # (the driver's own: --steps 20 --warmup 5)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--total-bases", type=int, default=3_000_000_000)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the config-3 / config-4 secondary measurements")
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c4sv", "readme"], default="c2",
                    help="c2 = BASELINE configs[1] (headline: -sn 0.01 -titv 2.0); c3 = configs[2], the full SV mix; "
                         "c4 = configs[3], RMT mode with ~40 k blocked ranges + hot/cold spots; c4sv = the c4 file with the "
                         "c3 SV mix as its std line")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="N > 1: 'strong' (default) = BASELINE configs[4]'s shape: ONE genome, contigs sharded (LPT), the MT19937 "
                         "stream chain walked on every rank, RCCL gather to rank 0 (Amdahl-bound by the chain: amdahl_ceiling); "
                         "'weak' = N independent genomes, one per GPU with its own seeded streams (per-GPU work fixed; scales "
                         "linearly).  The other mode is measured too and reported in the same line")
    ap.add_argument("--gather", action="store_true", help="(kept for compatibility: the gather is always measured for N > 1, strong)")
    ap.add_argument("--fast-only", action="store_true", help="(child of the N = 1 run) only the `--rng fast` secondary numbers, as one JSON line")
    a = ap.parse_args()
    if a.fast_only:
        from mutation_simulator_amd import _ffi
        from mutation_simulator_amd import mutator as mm
        device = int(os.environ.get("MSIM_BENCH_DEVICE", 0))
        _ffi.warm_up_async(device, pin=True).join()
        lengths = contig_lengths(a.total_bases)
        out = fast_rng_steps(lengths, mm)
        try:
            out["rank_steps"] = fast_rank_steps(lengths, mm, {w: out[w]["ms_per_step"] for w in ("c2", "c3")})
        except Exception as e:  # noqa: BLE001
            out["rank_steps_error"] = f"{type(e).__name__}: {e}"
        print(json.dumps(out))
        return

    # ONE JSON line on stdout, nothing else: gloo and RCCL print banners to the process's stdout from C, so file descriptor 1
    # is pointed at stderr for the whole run and the line goes to a private duplicate of the real stdout
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
        # launched by torch.distributed.run: one rank per GPU.  torch.distributed (gloo, CPU) is the CONTROL plane
        # only -- rendezvous, barriers, max-over-ranks; the data plane is libmsim's own RCCL communicator
        # (msim_comm_*: ncclSend/ncclRecv over xGMI), no torch tensor ever holds genome bytes.
        import torch.distributed as dist
        dist.init_process_group("gloo")

    from mutation_simulator_amd import _ffi
    from mutation_simulator_amd import mutator as mm
    from mutation_simulator_amd.sharding import lpt_partition

    lengths = contig_lengths(a.total_bases)
    # N > 1.  "strong" (default since round 4: BASELINE's metric is "3 Gb synthetic genome ... 1 -> 8 GPUs" and configs[4] is ONE
    # genome sharded): contigs sharded (LPT), the stream chain walked by every rank (msim_plan_chain for contigs it does not
    # own), RCCL gather to rank 0 (streams + record tables + insert pools) -- Amdahl-bound by the chain in compatible mode
    # (amdahl_ceiling), near-linear in fast mode, where nothing chains (one_genome_sharded_fast_rng).  "weak": N genomes at
    # once, one per GPU, each with its own seeded streams -- the cohort case, no exchange step.  Whichever is the headline,
    # the other one is measured too and rides in the same line (weak_replicas / one_genome_sharded).
    scaling = a.scaling or ("strong" if world > 1 else "weak")      # N > 1: BASELINE's metric is ONE 3 Gb genome over the N GPUs
    strong = scaling == "strong" and world > 1
    parts = lpt_partition(lengths, world)               # ownership of the one-genome mode
    everything = list(range(len(lengths)))
    mine = parts[rank] if strong else everything
    seed = 42 if strong else 42 + rank       # replicas: every GPU mutates its own genome

    # (MSIM_BENCH_DEVICE: put every rank on one GPU -- lets the N > 1 control flow run on a 1-GPU box)
    device = int(os.environ.get("MSIM_BENCH_DEVICE", local_rank))
    # The calling thread moves onto the CPUs of its GPU's NUMA node, as the CLI's does (_ffi.warm_up_async: sysfs numa_node of
    # the PCI device; MSIM_NO_PIN=1 leaves it where the scheduler put it).  A c2 step is ~160 dependent launches: from the
    # other socket of the pool's two-socket hosts it takes 4.37-4.45 ms, from the GPU's own 4.18-4.26 -- unpinned, a
    # process lands on either (the two "modes" of the round-3 and round-4 notes).
    _ffi.warm_up_async(device, pin=True).join()
    eng = _ffi.Engine(device)
    # genome resident in HBM before the timed region (3 GB; every rank holds every contig so that
    # contig numbering is global -- 288 GB of HBM make the replica free)
    cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
    eng.sync()
    comm, comm_error = None, None                       # the RCCL communicator: opened LAST (gather phase at the end of main)

    def barrier():
        eng.sync()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x: float) -> float:
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    marshal_ms = {}
    spread = {}                                         # per workload: this rank's wall time of every timed step (last measure() of it)

    def measure(workload, steps, warmup, gather=False, owned=None, step_seed=None):
        owned = mine if owned is None else owned
        step_seed = seed if step_seed is None else step_seed
        sim = build_settings(workload, lengths)
        # msim_range table per contig: settings -> integers INSIDE every timed step since round 4 (mutator.plan_table ->
        # msim_build_ranges: the reference's int((stop - start + 1) * rate), cdf and setsize per range are part of its hot path,
        # mutator.py:160-174,225).  "descriptor_marshalling_ms" = what that is per step (it is inside ms_per_step).  The walk
        # over the settings tree's RangeDefinition objects happens once, at the first table (warm-up): mutator._range_triples.
        marsh = [0.0]

        def plan_descs(chrom):
            t_m = time.perf_counter()
            d = mm.plan_table(chrom)
            marsh[0] += time.perf_counter() - t_m
            return d
        eng.set_params(mm.params_descriptor(sim))

        def step():
            one_step(eng, sim, cids, owned, step_seed, plan_descs, lengths if len(owned) < len(lengths) else None)
            if gather:
                comm.gather_to_root(cids, parts)

        for _ in range(max(warmup, 1)):                 # (at least one: the first table of a settings tree walks its objects)
            step()
        eng.reset_stats()
        barrier()
        marsh[0] = 0.0
        per_step = []
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            step()                                      # (ends in the context's synchronisation: a step's own wall time)
            per_step.append(time.perf_counter() - ts)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        marshal_ms[workload] = round(marsh[0] / steps * 1e3, 3)
        spread[workload] = spread_of(per_step)
        return dt, eng.stats()

    dt, st = measure(a.workload, a.steps, a.warmup)
    line = None
    LIVE_PMC[0] = world == 1 and not a.no_secondary and not os.environ.get("MSIM_BENCH_NO_PMC")
    if rank == 0:
        total = sum(lengths) * (1 if strong else world)      # weak: every rank mutated a whole genome
        W = WORKLOADS[a.workload]
        line = {
            "metric": W["metric"],
            "value": round(total * a.steps / dt / 1e6, 3), "unit": "Mbases/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            **{k: v for k, v in (spread.get(a.workload) or {}).items() if k != "steps"},      # rank 0's own steps: min / median / max
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{W['mode']} mode, 3 Gb 24-contig synthetic genome (GRCh38-proportioned), {W['what']}"
                                   ", CPython/NumPy-compatible MT19937 streams seeded 42/42",
                       "total_bases": total, "contigs": len(lengths),
                       "host_thread": (f"on the CPUs of the GPU's NUMA node ({len(os.sched_getaffinity(0))} of {os.cpu_count()})"
                                       if hasattr(os, "sched_getaffinity") and len(os.sched_getaffinity(0)) < (os.cpu_count() or 0)
                                       else "where the scheduler put it"),
                       "parallelism": (f"one genome, contigs sharded over {world} GPUs (LPT), stream chain walked per rank, "
                                       "results left in HBM on the owning GPU ('with_gather' adds the RCCL gather to rank 0)"
                                       if strong else f"{world} independent replica(s): one whole genome per GPU, "
                                       f"streams seeded 42+rank, results left in HBM")},
            "stages_ms_per_step": stages_of(st, a.steps),
            "descriptor_marshalling_ms": marshal_ms[a.workload],      # per step, INSIDE ms_per_step (msim_build_ranges)
            "plan_engines": engines_of(st, a.steps),
            "records_per_step": st["records"] // a.steps,
            "roofline": roofline_of(st, a.workload, a.steps),
        }
        if st.get("snp_samples_ahead"):
            # (anchored windows: samples whose heavy kernels ran off the stream-position chain, and how close the worst exact
            #  start came to an edge of the +-(8 sigma + 256 words) interval the host planned for: 1000 = the edge = an error)
            line["plan_ahead"] = {"samples_ahead_per_step": st["snp_samples_ahead"] // a.steps,
                                  "margin_used_permille_worst": st.get("snp_ahead_margin_permille")}
        if world == 1:
            line["step_roofline"] = step_roofline(st, dt)
    LIVE_PMC[0] = False                                 # (the secondary workloads: profiles/r05_traffic.json)
    gather_jobs = []                                    # (workload, its dict in the line): measured in the gather phase

    def sharded_numbers(dt_sharded, workload=None):
        """One genome over the N GPUs: throughput without and with the exchange step north_star names (every peer sends its
        mutated contigs to rank 0 over its own xGMI link)."""
        workload = workload or a.workload
        out = None
        if rank == 0:
            out = {"value": round(sum(lengths) * a.steps / dt_sharded / 1e6, 3), "unit": "Mbases/s", "scaling": "strong",
                   "ms_per_step": round(dt_sharded / a.steps * 1e3, 3),
                   "what": f"ONE genome, contigs sharded over {world} GPUs (LPT), results left in HBM on the owning GPU"}
        if rank == 0:
            out["with_gather"] = {"error": "not measured: the RCCL phase did not complete"}      # (filled in by the gather phase)
        gather_jobs.append((workload, out))
        return out

    if strong:
        sh = sharded_numbers(dt)
        if rank == 0:
            line["with_gather"] = sh["with_gather"]          # (replaced by the gather phase at the end)
        # what DOES scale: N independent replicas (one whole genome per GPU, own seeds) -- the weak-scaling number
        dtw, _ = measure(a.workload, a.steps, 1, owned=everything, step_seed=42 + rank)
        if rank == 0:
            line["weak_replicas"] = {"value": round(sum(lengths) * world * a.steps / dtw / 1e6, 3), "unit": "Mbases/s",
                                     "ms_per_step": round(dtw / a.steps * 1e3, 3), "scaling": "weak",
                                     "what": f"{world} independent replicas, one whole genome per GPU, streams seeded 42+rank"}
        if a.workload == "c2" and not a.no_secondary:
            # BASELINE configs[4] as written: the full SV mix (c3) on ONE genome, contigs sharded over the N GPUs, RCCL gather
            dt4, st4 = measure("c3", a.steps, 2, owned=parts[rank], step_seed=42)
            sh4 = sharded_numbers(dt4, "c3")
            if rank == 0:
                sh4["metric"] = WORKLOADS["c3"]["metric"]
                sh4["plan_engines_rank0"] = engines_of(st4, a.steps)
                line["configs4_sv_mix_sharded"] = sh4
            # the same genome in fast mode (--rng fast: counter-based draws, NOT the reference's numbers): nothing chains, a rank
            # plans and applies its own contigs and skips the others -- the mode in which per-contig scaling can be near-linear
            try:
                fr = fast_rng_sharded(lengths, parts[rank], local_rank, barrier, max_over_ranks, mm)
            except Exception as e:  # noqa: BLE001
                fr = {"error": f"{type(e).__name__}: {e}"}
            if rank == 0:
                line["one_genome_sharded_fast_rng"] = fr
    elif world > 1:
        dts_, _ = measure(a.workload, a.steps, 1, owned=parts[rank], step_seed=42)
        sh = sharded_numbers(dts_)
        if rank == 0:
            line["one_genome_sharded"] = sh
        if a.workload == "c2" and not a.no_secondary:
            # BASELINE configs[4] as written: the full SV mix (c3) on ONE genome, contigs sharded over the N GPUs, RCCL gather
            dt4, st4 = measure("c3", a.steps, 2, owned=parts[rank], step_seed=42)
            sh4 = sharded_numbers(dt4, "c3")
            if rank == 0:
                sh4["metric"] = WORKLOADS["c3"]["metric"]
                sh4["plan_engines_rank0"] = engines_of(st4, a.steps)
                line["configs4_sv_mix_sharded"] = sh4
    # What N GPUs can give ONE genome (strong scaling): the chain of both MT19937 streams runs over every contig on every
    # rank whatever N is; only emission + APPLY of the owned contigs shrink.  Measured, not modelled: a step that owns
    # nothing (msim_plan_chain for all 24 contigs) against the full step.
    dtc, _ = measure(a.workload, a.steps, 1, owned=[], step_seed=42)
    dtf = dt
    if strong:
        dtf, _ = measure(a.workload, a.steps, 1, owned=everything, step_seed=42)
    if rank == 0:
        line["amdahl_ceiling"] = {"value": round(dtf / dtc, 3), "chain_only_ms_per_step": round(dtc / a.steps * 1e3, 3),
                                  "full_step_ms": round(dtf / a.steps * 1e3, 3),
                                  "what": "upper bound of the strong-scaling speed-up of one genome over N GPUs: full 1-GPU step / "
                                          "step of a rank that owns no contig (stream chain only); N independent genomes "
                                          "(weak_replicas / --scaling weak) scale linearly instead"}
    if world == 1 and not a.no_secondary and a.workload == "c2":
        # BASELINE configs[2] and [3] on the same resident genome: 8 steps each after 2 warm-up steps (the first steps of a
        # workload still grow scratch buffers), same definition of a step
        sec = {}
        n_sec, w_sec = 8, 2                              # (eight: one hiccup of the host in five steps moved c3 by 15 %)
        for w in ("c3", "c4", "c4sv"):
            dts, sts = measure(w, n_sec, w_sec)
            sec[w] = {"metric": WORKLOADS[w]["metric"], "value": round(sum(lengths) * n_sec / dts / 1e6, 3), "unit": "Mbases/s",
                      "ms_per_step": round(dts / n_sec * 1e3, 3), **{k: v for k, v in (spread.get(w) or {}).items() if k != "steps"},
                      "steps": n_sec, "warmup": w_sec,
                      "stages_ms_per_step": stages_of(sts, n_sec), "descriptor_marshalling_ms": marshal_ms[w],
                      "plan_engines": engines_of(sts, n_sec),
                      "records_per_step": sts["records"] // n_sec,
                      "roofline": roofline_of(sts, w, n_sec), "step_roofline": step_roofline(sts, dts)}
        try:
            sec["fast_rng"] = fast_rng_isolated(a.total_bases)
        except Exception as e:  # noqa: BLE001
            sec["fast_rng"] = {"error": f"{type(e).__name__}: {e}"}
        # What a SCALE run should show, from this one GPU: the step of the most loaded rank of an N-way LPT partition (it plans
        # / applies what it owns and walks the others' contigs for their stream positions) against the full step -- the strong-
        # scaling speed-up of ONE genome that 2 / 4 / 8 GPUs can reach before the gather, per mode.  A prediction to hold a
        # measured curve against, not a measurement of N GPUs.
        try:
            line["predicted_one_genome_scaling"] = predict_scaling(measure, lengths, a.steps, dtf, sec, mm)
        except Exception as e:  # noqa: BLE001
            line["predicted_one_genome_scaling"] = {"error": f"{type(e).__name__}: {e}"}
        # (in front of the CLI runs: e2e_cli clears this context's contigs)
        for key, where in (("e2e", "tmpfs"), ("e2e_tmpdir", "tmpdir")):
            try:
                sec[key] = e2e_cli(eng, where=where)
            except Exception as e:  # noqa: BLE001  (no tmpfs / disk space: the kernels' numbers above still stand)
                sec[key] = {"error": f"{type(e).__name__}: {e}"}
        line["secondary"] = sec
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(a.cpu_sample, a.workload, device=device)      # bounded sample: ~6-10 s of CPU work
    printed = [False]

    def emit_line():
        if rank == 0 and not printed[0]:
            printed[0] = True
            if strong and gather_jobs:
                line["with_gather"] = gather_jobs[0][1]["with_gather"]
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(line) + "\n").encode())

    # ---- the RCCL phase: LAST, and under a watchdog.  Everything above needs no communicator; the gather over xGMI is the one
    # part of this file that has never run on more than one GPU (1-GPU boxes only), so a hang in ncclCommInitRank or in the
    # grouped send / recv must not cost the line: when the limit passes, rank 0 prints what it has and every rank leaves.
    if world > 1 and gather_jobs:
        import threading
        limit = float(os.environ.get("MSIM_BENCH_RCCL_TIMEOUT", "240"))

        def on_timeout():
            if rank == 0:
                for _, out in gather_jobs:
                    if "value" not in out["with_gather"]:
                        out["with_gather"] = {"error": f"the RCCL phase did not finish within {limit:.0f} s"}
                emit_line()
            os._exit(0)
        watchdog = threading.Timer(limit, on_timeout)
        watchdog.daemon = True
        watchdog.start()
        from mutation_simulator_amd.gather import Communicator
        try:
            comm = Communicator(eng, rank, world, dist)
        except Exception as e:  # noqa: BLE001  (no RCCL / duplicate GPUs ...)
            comm_error = f"{type(e).__name__}: {e}"
        flags = [None] * world
        dist.all_gather_object(flags, comm_error)
        if any(flags):                                   # every rank takes the same branch
            comm_error = next(f for f in flags if f)
            if comm is not None:
                comm.close()
            comm = None
        for workload, out in gather_jobs:
            if comm is not None:
                dtg, _ = measure(workload, a.steps, 1, gather=True, owned=parts[rank], step_seed=42)
                if rank == 0:
                    out["with_gather"] = {"value": round(sum(lengths) * a.steps / dtg / 1e6, 3), "unit": "Mbases/s",
                                          "ms_per_step": round(dtg / a.steps * 1e3, 3), "transport": comm.describe()}
            elif rank == 0:
                out["with_gather"] = {"error": comm_error}
        watchdog.cancel()
    emit_line()
    if world > 1:                                        # (the same for the teardown: the line is out, nothing may hang now)
        import threading
        bye = threading.Timer(60.0, lambda: os._exit(0))
        bye.daemon = True
        bye.start()
    if comm is not None:
        comm.close()
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

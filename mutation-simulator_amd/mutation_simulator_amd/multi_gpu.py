"""``--gpus N``: one genome over the N GPUs of a node, byte-identical to a 1-GPU run (BASELINE configs[4]).

What shards and what does not.  mutate()'s contig loop (reference mutator.py:111-141) is the unit of work, but both
MT19937 streams are consumed sequentially ACROSS contigs with data-dependent word counts, so where contig c+1 starts in
the streams is only known once contig c has been planned.  Every worker therefore walks the whole chain -- for contigs it
does not own through ``msim_plan_chain`` (stream positions only: no records, no insert pool, no SNP outcomes, no contig in
HBM) -- and does the heavy part only for the units it owns: FASTA text to HBM, PLAN with emission, APPLY, line framing and
VCF text on the device, D2H over ITS OWN PCIe link, file writes.  The end-to-end run is bound by exactly those (DESIGN.md
section 4b), which is where N GPUs help; the 1-2 ms of rewrite kernel per 3 Gb are not.

Process model: the parent never touches a GPU.  It starts one worker interpreter per GPU (``python -m
mutation_simulator_amd.multi_gpu``; never a fork: a forked child of a process that initialised HIP is not allowed on this
platform), hands each the argument namespace and the states of Python's two global generators through a job file, and afterwards concatenates the workers' part files in contig order (the newline rule
between two records is FastaWriter's), prints the collected warnings in contig order, re-raises the first error in contig
order (the files then hold what a 1-GPU run would have written by then) and puts the advanced generator states back.

Knobs for boxes without N GPUs (tests): ``MSIM_SHARD_DEVICES=0,0`` maps ranks to device ordinals; ``MSIM_SHARD_HOST_ONLY=1``
runs the workers on host-only contexts (PLAN + VCF text, no sequence) so the control flow runs on the CPU tier.
"""
from __future__ import annotations

import os
import random
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np

from . import _ffi
from .fasta_writer import FastaWriter
from .mutator import (Mutator, export_python_streams, import_python_streams, params_descriptor, plan_descriptors,
                      plan_table)
from .sharding import lpt_partition
from .vcf_writer import VcfWriter


def unit_owners(unit_bases: list, world: int) -> list:
    """Owner rank of every unit: longest-processing-time packing on the units' bases (sharding.lpt_partition)."""
    owner = [0] * len(unit_bases)
    for r, part in enumerate(lpt_partition(list(unit_bases), world)):
        for u in part:
            owner[u] = r
    return owner


def _devices(args, world: int) -> list:
    env = os.environ.get("MSIM_SHARD_DEVICES")
    if env:
        devs = [int(x) for x in env.split(",")]
        if len(devs) < world:
            raise _ffi.MsimError(f"MSIM_SHARD_DEVICES names {len(devs)} devices for {world} workers")
        return devs[:world]
    first = int(getattr(args, "device", 0) or 0)
    return [first + r for r in range(world)]


class ShardWorker(Mutator):
    """The Mutator of one rank: same per-contig code paths, but only for the units it owns, into part files of its own."""

    def __init__(self, args, fasta, sim, rank: int, world: int, part_dir: Path):
        self._args, self._fasta, self._sim = args, fasta, sim
        self.rank, self.world = rank, world
        self._fasta_writer = FastaWriter(part_dir / f"part{rank}.fa")
        self._vcf_writer = VcfWriter(part_dir / f"part{rank}.vcf")      # record lines only: the parent owns the header
        self._engine = None
        self._own_engine = True
        self.stats: dict = {}
        self.warned: list = []
        self._t = {"ingest_s": 0.0, "plan_apply_s": 0.0, "fasta_egress_s": 0.0, "vcf_egress_s": 0.0}
        self._host_only = os.environ.get("MSIM_SHARD_HOST_ONLY") == "1"

    def _warn_empty(self, chrom):
        self.warned.append(chrom.number)               # printed by the parent, in contig order

    def _batchable(self, chrom) -> bool:
        return not self._host_only and super()._batchable(chrom)

    def _run_contig(self, eng, chrom, done):
        if not self._host_only:
            return super()._run_contig(eng, chrom, done)
        rec = self._fasta[chrom.number]                # CPU tier: PLAN on a host-only context, VCF by the host renderer
        cid = eng.add_contig(rec.bases)
        eng.plan_contig(cid, plan_descriptors(chrom))
        if eng.plan_was_empty(cid) and "warned" not in done:
            self._warn_empty(chrom)
            done.add("warned")
        if "header" not in done:
            self._fasta_writer.set_bpl(self._fasta.faidx.index[rec.name].lenc)
            self._fasta_writer.write_header(rec.long_name)
            done.add("header")
        recs, pool = eng.fetch_records(cid)
        self._vcf_writer.write_raw(_ffi.render_vcf(recs, pool, rec.bases, rec.name))
        eng.clear()

    def _chain_only(self, eng, chroms, k):
        """Advance the streams over contig k (not ours).  A device engine's window overflow is recovered like
        ``_mutate_one`` does: streams back to the run's start, everything up to k again through the host planner."""
        rec = self._fasta[chroms[k].number]
        try:
            eng.plan_chain(len(rec), plan_table(chroms[k]))
        except _ffi.MsimError as e:
            if "overflowed its" not in str(e):
                raise
            eng.clear()
            export_python_streams(eng)
            eng.set_plan_mode(_ffi.PLAN_HOST)
            try:
                for prev in chroms[:k + 1]:
                    eng.plan_chain(len(self._fasta[prev.number]), plan_table(prev))
            finally:
                eng.set_plan_mode(_ffi.PLAN_AUTO)

    def run(self, device: int) -> dict:
        self._engine = eng = self._open_engine(-1 if self._host_only else device)
        eng.set_params(params_descriptor(self._sim))
        eng.reset_stats()
        chroms = self._chromosomes()
        units = self._units(chroms)
        tab = getattr(self._fasta, "index_table", None)
        if tab is not None and len(tab) == len(chroms) and not isinstance(chroms, list):
            cum = np.concatenate(([0], np.cumsum(tab["n_bases"].astype(np.int64))))
            unit_bases = [int(cum[j] - cum[i]) for i, j in units]
        else:
            unit_bases = [sum(len(self._fasta[c.number]) for c in chroms[i:j]) for i, j in units]
        owner = unit_owners(unit_bases, self.world)
        segments, error = [], None
        try:
            for u, (i, j) in enumerate(units):
                if owner[u] != self.rank:
                    try:
                        for k in range(i, j):
                            self._chain_only(eng, chroms, k)
                    except ValueError as e:            # the reference's ValueError is raised by PLAN: every rank meets it
                        error = {"unit": u, "type": "ValueError", "args": e.args}
                        break
                    continue
                self._fasta_writer.begin_segment()
                f0, v0 = self._fasta_writer.tell(), self._vcf_writer.tell()
                try:
                    self._process_unit(eng, chroms, i, j)
                except (KeyError, ValueError) as e:
                    error = {"unit": u, "type": type(e).__name__, "args": e.args}
                segments.append((u, f0, self._fasta_writer.tell(), v0, self._vcf_writer.tell()))
                if error:
                    break
            eng.file_wait()                            # the part files hold everything queued for them (or: MsimError)
        finally:
            if error is None and not self._fast_rng:
                import_python_streams(eng)
            self.stats = eng.stats()
        rng = (random.getstate(), np.random.get_state()) if self.rank == 0 and error is None else None
        return {"segments": segments, "error": error, "warned": self.warned, "stats": self.stats, "rng": rng,
                "units": len(units), "owned": sum(1 for o in owner if o == self.rank)}


def _worker_main(job_path: str, rank: int) -> int:
    """A worker process: ``python -m mutation_simulator_amd.multi_gpu <job.pkl> <rank>``.  Its result (or what went
    wrong) goes to ``result<rank>.pkl`` next to the job file."""
    import pickle
    job = pickle.loads(Path(job_path).read_bytes())
    part_dir = Path(job["part_dir"])
    try:
        from . import SimulationSettings, load_fasta
        args, world = job["args"], job["world"]
        random.setstate(job["rng"][0])
        np.random.set_state(job["rng"][1])
        fasta = load_fasta(args.infile)
        if args.mode == "args":
            sim = SimulationSettings.from_args(args, fasta, True)
        else:
            sim = SimulationSettings.from_rmt(args.rmtfile, fasta, True)
        w = ShardWorker(args, fasta, sim, rank, world, part_dir)
        try:
            res = w.run(_devices(args, world)[rank])
        finally:
            w.close()
    except BaseException as e:  # noqa: BLE001  (reported to the parent, which raises it as a MsimError)
        import traceback
        res = {"fatal": f"{type(e).__name__}: {e}", "traceback": traceback.format_exc()}
    tmp = part_dir / f"result{rank}.tmp"
    tmp.write_bytes(pickle.dumps(res))
    tmp.rename(part_dir / f"result{rank}.pkl")
    return 0


def _copy_range(src, offset: int, nbytes: int, write):
    src.seek(offset)
    left = nbytes
    while left:
        chunk = src.read(min(left, 64 << 20))
        if not chunk:
            raise _ffi.MsimError("part file shorter than its index says")
        write(chunk)
        left -= len(chunk)


def die_with_parent():
    """``preexec_fn`` of a worker: SIGKILL when the parent goes away (PR_SET_PDEATHSIG) -- a parent that is killed from outside
    (a watchdog's ``os._exit``, a lease's limit) never reaches the ``finally`` that ends its workers, and an orphan would go on
    holding its GPU."""
    try:
        import ctypes
        import signal
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)      # PR_SET_PDEATHSIG = 1
    except Exception:  # noqa: BLE001  (not Linux: the worker simply lives as before)
        pass


def mutate_sharded(m: Mutator):
    """``Mutator.mutate()`` for ``--gpus N`` (the parent side; see the module docstring)."""
    import pickle
    import subprocess
    args = m._args
    world = int(args.gpus)
    part_dir = Path(tempfile.mkdtemp(prefix=".msim_parts_", dir=str(Path(args.outfasta).resolve().parent)))
    job = part_dir / "job.pkl"
    job.write_bytes(pickle.dumps({"args": args, "world": world, "part_dir": str(part_dir),
                                  "rng": (random.getstate(), np.random.get_state())}))
    # plain child interpreters (not multiprocessing: nothing of this process -- its __main__, an initialised HIP runtime
    # of an embedding application -- may leak into a worker)
    env = dict(os.environ)
    pkg_parent = str(Path(__file__).resolve().parent.parent)
    env["PYTHONPATH"] = pkg_parent + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    procs = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen([sys.executable, "-m", "mutation_simulator_amd.multi_gpu", str(job), str(r)],
                                          env=env, stdout=subprocess.DEVNULL, preexec_fn=die_with_parent))
        # All workers are watched together: the first one that dies without a result or reports a fatal error (no device,
        # a bad MSIM_SHARD_DEVICES) ends the run at once -- the others are killed (finally:) instead of being left to finish
        # their whole share -- and a worker that hangs (a stuck GPU) is bounded by MSIM_SHARD_TIMEOUT seconds (0 = none).
        import time
        limit = float(os.environ.get("MSIM_SHARD_TIMEOUT", "0") or 0)
        t_start = time.monotonic()
        results = [None] * world
        while any(res is None for res in results):
            progressed = False
            for r, p in enumerate(procs):
                if results[r] is not None:
                    continue
                code = p.poll()
                if code is None:
                    continue
                progressed = True
                out = part_dir / f"result{r}.pkl"
                res = pickle.loads(out.read_bytes()) if out.exists() else {
                    "fatal": f"worker process ended with exit code {code} and no result"}
                if "fatal" in res:
                    raise _ffi.MsimError(f"--gpus {world}: worker {r} failed: {res['fatal']}\n{res.get('traceback', '')}")
                results[r] = res
            if limit and time.monotonic() - t_start > limit:
                stuck = [r for r, res in enumerate(results) if res is None]
                raise _ffi.MsimError(f"--gpus {world}: worker(s) {stuck} still running after MSIM_SHARD_TIMEOUT={limit:g} s")
            if not progressed:
                time.sleep(0.002)
        errors = [res["error"] for res in results if res["error"]]
        fail_unit = min((e["unit"] for e in errors), default=None)
        where = {}
        for r, res in enumerate(results):
            for u, f0, f1, v0, v1 in res["segments"]:
                where[u] = (r, f0, f1, v0, v1)
        fa = [open(part_dir / f"part{r}.fa", "rb") for r in range(world)]
        vcf = [open(part_dir / f"part{r}.vcf", "rb") for r in range(world)]
        try:
            for u in sorted(where):
                if fail_unit is not None and u > fail_unit:
                    break
                r, f0, f1, v0, v1 = where[u]
                m._fasta_writer.append_segment(fa[r], f0, f1 - f0)
                _copy_range(vcf[r], v0, v1 - v0, m._vcf_writer.write_raw)
        finally:
            for f in fa + vcf:
                f.close()
        numbers = {c.number: c for c in m._sim.chromosomes}
        for n in sorted({n for res in results for n in res["warned"]}):
            m._warn_empty(numbers[n])
        st = dict(results[0]["stats"])                   # the chain is replicated: word counts are rank 0's ...
        for key in st:
            if key.startswith("contigs_") or key in ("apply_launches", "bytes_in", "bytes_out", "records"):
                st[key] = sum(res["stats"][key] for res in results)      # ... owned work adds up
            elif key.endswith("_ms"):
                st[key] = max(res["stats"][key] for res in results)
        st["gpus"] = world
        st["units_per_rank"] = [res["owned"] for res in results]
        m.stats = st
        if fail_unit is not None:
            at_unit = [e for e in errors if e["unit"] == fail_unit]
            owner_rank = where[fail_unit][0] if fail_unit in where else None
            e = results[owner_rank]["error"] if owner_rank is not None and results[owner_rank]["error"] in at_unit else at_unit[0]
            raise (KeyError if e["type"] == "KeyError" else ValueError)(*e["args"])
        random.setstate(results[0]["rng"][0])
        np.random.set_state(results[0]["rng"][1])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        shutil.rmtree(part_dir, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(_worker_main(sys.argv[1], int(sys.argv[2])))

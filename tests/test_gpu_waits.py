"""Host waits have a deadline (ctx.h: wait_stream / wait_event, spin_until / spin_event; MSIM_WAIT_TIMEOUT_S, default 120 s).

The reference cannot hang -- `Mutator.mutate` is a plain loop (mutator.py:105-142) -- so neither may the product: a wait that is
kept waiting beyond the limit returns MSIM_ERR_HIP naming the call and the stream / event it waited for, instead of blocking
inside a HIP call where not even a signal handler runs.  `msim_dbg_stall` keeps a stream busy for a bounded time (the device's
100 MHz wall clock), so the wait is really kept waiting and the queue still drains by itself."""
from __future__ import annotations

import ctypes as C
import time

import pytest

import bench
from mutation_simulator_amd import _ffi
from mutation_simulator_amd import mutator as mm

pytestmark = pytest.mark.gpu


def _stall(eng, which, ms):
    eng.lib.msim_dbg_stall.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
    eng.lib.msim_dbg_stall.restype = C.c_int
    assert eng.lib.msim_dbg_stall(eng.h, which, ms) == _ffi.OK


@pytest.mark.parametrize("which,name", [(0, "c->stream"), (1, "c->emit_stream")])
def test_sync_gives_up_at_the_deadline_and_names_the_stream(which, name, monkeypatch):
    monkeypatch.setenv("MSIM_WAIT_TIMEOUT_S", "0.5")
    with _ffi.Engine(0) as eng:
        eng.sync()
        _stall(eng, which, 2500)
        t0 = time.perf_counter()
        with pytest.raises(_ffi.MsimError) as ei:
            eng.sync()
        dt = time.perf_counter() - t0
        assert 0.4 < dt < 2.0, dt
        msg = str(ei.value)
        assert "no completion within 0.5 s" in msg and "MSIM_WAIT_TIMEOUT_S" in msg, msg
        assert name in msg, msg                      # which stream the host was waiting for
        monkeypatch.setenv("MSIM_WAIT_TIMEOUT_S", "30")
        eng.sync()                                   # the stall ends by itself; the context is usable again
        assert time.perf_counter() - t0 > 2.0


def test_a_step_behind_a_stalled_plan_stream_is_still_right(monkeypatch):
    """A wait that is kept waiting for less than the limit changes nothing: a c2-shaped step queued behind a one-second stall
    gives the records of a step that was not."""
    lengths = [3_000_000, 5_000_000]
    sim = bench.build_settings("c2", lengths)
    with _ffi.Engine(0) as eng:
        cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
        eng.set_params(mm.params_descriptor(sim))
        bench.one_step(eng, sim, cids, [0, 1], 42, mm.plan_table)
        want = [eng.fetch_records(c)[0].tobytes() for c in cids]
        _stall(eng, 0, 1000)
        t0 = time.perf_counter()
        bench.one_step(eng, sim, cids, [0, 1], 42, mm.plan_table)
        assert time.perf_counter() - t0 > 0.9
        assert [eng.fetch_records(c)[0].tobytes() for c in cids] == want


def test_host_chain_poll_gives_up_at_the_deadline(monkeypatch):
    """The SV-mix engine's host polls a mailbox word the plan stream raises (spin_until): with the stream stalled beyond the
    limit the plan call returns the error; afterwards the same plan succeeds."""
    lengths = [4_000_000]
    sim = bench.build_settings("c3", lengths)
    with _ffi.Engine(0) as eng:
        cids = [eng.add_contig_synthetic(L, 1000 + i) for i, L in enumerate(lengths)]
        eng.set_params(mm.params_descriptor(sim))
        bench.one_step(eng, sim, cids, [0], 42, mm.plan_table)
        want = eng.fetch_records(cids[0])[0].tobytes()
        eng.seed(42, 42)
        monkeypatch.setenv("MSIM_WAIT_TIMEOUT_S", "0.5")
        _stall(eng, 0, 2500)
        t0 = time.perf_counter()
        with pytest.raises(_ffi.MsimError) as ei:
            bench.one_step(eng, sim, cids, [0], 42, mm.plan_table)
        assert time.perf_counter() - t0 < 2.2
        assert "no completion within 0.5 s" in str(ei.value), str(ei.value)
        monkeypatch.setenv("MSIM_WAIT_TIMEOUT_S", "30")
        time.sleep(2.5)
        bench.one_step(eng, sim, cids, [0], 42, mm.plan_table)
        assert eng.fetch_records(cids[0])[0].tobytes() == want

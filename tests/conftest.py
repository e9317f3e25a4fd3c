"""pytest configuration: markers and import paths.

``-m "not gpu"`` : oracle vs goldens, host logic, C-ABI load/symbol checks (runs anywhere).
``-m gpu``       : parity tests proper -- HIP path through the C-ABI vs the oracle (needs an MI355X).
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "mutation-simulator_amd", ROOT / "tests", ROOT / "tests" / "golden"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# A run that is killed from outside (a lease's limit, pytest-timeout's os._exit) leaves no report: every test's id is appended
# to a flushed file as it STARTS and as it ends, so what was running is known afterwards (round 5: one run of seven sat in a HIP
# call for 25 minutes and not even the test was known).  MSIM_PYTEST_PROGRESS=<path> moves the file; "0" switches it off.
def _progress_path():
    import os
    p = os.environ.get("MSIM_PYTEST_PROGRESS")
    if p == "0":
        return None
    if p:
        return Path(p)
    d = ROOT / "gpurun_out"
    try:
        d.mkdir(exist_ok=True)
    except OSError:
        return None
    return d / "pytest_progress.log"


def _progress(line: str):
    import os
    import time
    p = _progress_path()
    if p is None:
        return
    try:
        fd = os.open(str(p), os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
        try:
            os.write(fd, f"{time.strftime('%H:%M:%S')} pid {os.getpid()} {line}\n".encode())
            os.fsync(fd)
        finally:
            os.close(fd)
    except OSError:
        pass


def pytest_runtest_logstart(nodeid, location):
    _progress(f"START {nodeid}")


def pytest_runtest_logfinish(nodeid, location):
    _progress(f"END   {nodeid}")

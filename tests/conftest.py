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

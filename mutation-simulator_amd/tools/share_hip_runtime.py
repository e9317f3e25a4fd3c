#!/usr/bin/env python3
"""Post-link step: make libmsim.so ask for ``libamdhip64.so`` instead of ``libamdhip64.so.7``.

PyTorch-ROCm wheels bundle their own HIP runtime as ``libamdhip64.so`` (no SONAME).  A process that
loads both that and the system's ``libamdhip64.so.7`` has two HIP/HSA runtimes, and the second one
finds no GPU.  libmsim never needs torch, but harnesses (bench.py's RCCL gather, users' scripts) may
put both in one process.  Requesting the unversioned name makes the dynamic loader hand libmsim the
runtime that is already loaded -- torch's if torch came first, the system's (found through RUNPATH ->
/opt/rocm/lib/libamdhip64.so) otherwise, in which case torch in turn reuses that one.

Implementation: the DT_NEEDED string lives in .dynstr; shortening it in place ("...so.7" ->
"...so\\0\\0") is a safe edit (same offset, still NUL-terminated).  Equivalent to
``patchelf --replace-needed libamdhip64.so.7 libamdhip64.so``.
"""
import sys

OLD, NEW = b"libamdhip64.so.7\0", b"libamdhip64.so\0\0\0"


def main(path):
    data = bytearray(open(path, "rb").read())
    n = data.count(OLD)
    if n == 0:
        if data.count(b"libamdhip64.so\0") >= 1:
            print(f"{path}: already requests libamdhip64.so")
            return 0
        print(f"{path}: DT_NEEDED libamdhip64.so.7 not found", file=sys.stderr)
        return 1
    if n != 1:
        print(f"{path}: expected exactly one occurrence, found {n}", file=sys.stderr)
        return 1
    i = data.index(OLD)
    data[i:i + len(OLD)] = NEW
    open(path, "wb").write(data)
    print(f"{path}: DT_NEEDED libamdhip64.so.7 -> libamdhip64.so")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))

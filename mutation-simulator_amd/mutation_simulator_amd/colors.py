"""ANSI colours, active only when both stdout and stderr are terminals (reference colors.py:4-15)."""
import sys

_tty = sys.stdout.isatty() and sys.stderr.isatty()


class Colors:
    error = "\033[0;31m" if _tty else ""
    warn = "\033[0;33m" if _tty else ""
    ok = "\033[0;32m" if _tty else ""
    norm = "\033[0m" if _tty else ""

"""Opt-in literal-name alias (put `<repo>/compat` on sys.path) of the reference's `precise_runner` package
(/root/reference/runner/precise_runner/__init__.py:1): `from precise_runner import PreciseRunner, PreciseEngine,
ReadWriteStream` (runner/example.py:17) resolves to mycroft_precise_amd.runner.  Not a component."""
from mycroft_precise_amd.runner import PreciseRunner, PreciseEngine, ReadWriteStream      # noqa: F401

__version__ = '0.3.1'

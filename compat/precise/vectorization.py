"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.vectorization` IS `mycroft_precise_amd.vectorization`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/vectorization.py), so unchanged reference-side code -- `from precise.vectorization import vectorize_raw` -- resolves to this framework."""
import sys
import mycroft_precise_amd.vectorization as _impl

sys.modules[__name__] = _impl

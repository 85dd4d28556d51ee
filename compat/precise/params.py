"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.params` IS `mycroft_precise_amd.params`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/params.py), so unchanged reference-side code -- `from precise.params import pr, inject_params` -- resolves to this framework."""
import sys
import mycroft_precise_amd.params as _impl

sys.modules[__name__] = _impl

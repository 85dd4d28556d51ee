"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.functions` IS `mycroft_precise_amd.functions`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/functions.py), so unchanged reference-side code -- `from precise.functions import sigmoid, asigmoid, pdf` -- resolves to this framework."""
import sys
import mycroft_precise_amd.functions as _impl

sys.modules[__name__] = _impl

"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.model` IS `mycroft_precise_amd.model`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/model.py), so unchanged reference-side code -- `from precise.model import load_precise_model` -- resolves to this framework."""
import sys
import mycroft_precise_amd.model as _impl

sys.modules[__name__] = _impl

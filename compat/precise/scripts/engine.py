"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.scripts.engine` IS `mycroft_precise_amd.scripts.engine`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/scripts/engine.py), so unchanged reference-side code -- `python -m precise.scripts.engine model.pb 2048` -- resolves to this framework."""
import sys
import mycroft_precise_amd.scripts.engine as _impl

sys.modules[__name__] = _impl

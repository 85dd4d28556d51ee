"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.util` IS `mycroft_precise_amd.util`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/util.py), so unchanged reference-side code -- `from precise.util import buffer_to_audio` -- resolves to this framework."""
import sys
import mycroft_precise_amd.util as _impl

sys.modules[__name__] = _impl

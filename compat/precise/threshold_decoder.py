"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.threshold_decoder` IS `mycroft_precise_amd.threshold_decoder`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/threshold_decoder.py), so unchanged reference-side code -- `from precise.threshold_decoder import ThresholdDecoder` -- resolves to this framework."""
import sys
import mycroft_precise_amd.threshold_decoder as _impl

sys.modules[__name__] = _impl

"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise_runner.runner` IS `mycroft_precise_amd.runner`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/runner/precise_runner/runner.py), so unchanged reference-side code -- `from precise_runner.runner import ListenerEngine, TriggerDetector` -- resolves to this framework."""
import sys
import mycroft_precise_amd.runner as _impl

sys.modules[__name__] = _impl

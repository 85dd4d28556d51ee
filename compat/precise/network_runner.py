"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.network_runner` IS `mycroft_precise_amd.network_runner`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/network_runner.py), so unchanged reference-side code -- `from precise.network_runner import Listener` -- resolves to this framework."""
import sys
import mycroft_precise_amd.network_runner as _impl

sys.modules[__name__] = _impl

"""Opt-in literal-name alias (put `<repo>/compat` on sys.path): `precise.scripts.engine` IS `mycroft_precise_amd.scripts.engine`.
Not a component: one line that hands the import system the MI355X module under the reference's module name
(/root/reference/precise/scripts/engine.py), so unchanged reference-side code -- `python -m precise.scripts.engine model.pb 2048` -- resolves to this framework."""
import sys
import mycroft_precise_amd.scripts.engine as _impl

if __name__ == '__main__':      # `python -m precise.scripts.engine model.pb 2048`: the alias runs as __main__, the implementation's own guard does not fire
    sys.exit(_impl.main())
sys.modules[__name__] = _impl

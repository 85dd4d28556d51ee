"""Opt-in literal-name alias (put `<repo>/compat` on sys.path) of the hot-path modules of the reference's `precise`
package (/root/reference/precise/__init__.py): network_runner, vectorization, params, threshold_decoder, util, functions,
model, scripts.engine.  Not a component; modules outside the hot path (training, datasets, CLI) do not exist here."""
__version__ = '0.3.0'

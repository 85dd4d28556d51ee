"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares; the
product path fails loudly (no CPU fallback) when the library or the GPU is missing."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO
from mycroft_precise_amd import _lib, _build


def declared_functions():
    names = []
    for fn in os.listdir(os.path.join(REPO, 'include')):
        if fn.endswith('.h'):
            text = open(os.path.join(REPO, 'include', fn)).read()
            text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
            names += re.findall(r'\b(pe_[a-z_0-9]+)\s*\(', text)
    return sorted(set(names))


def test_library_is_built_in_tree():
    path = _build.build()
    assert path == _lib.LIB_PATH and os.path.exists(path)
    assert os.path.dirname(path) == os.path.join(REPO, 'mycroft_precise_amd')


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 19
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'library does not export ' + n
    assert sorted(_lib.EXPORTS) == names, 'ctypes table and header disagree'
    lib = _lib.load()
    assert lib.pe_abi_version() == _lib.ABI_VERSION


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_lib.PeParams) == 12 * 4
    assert ctypes.sizeof(_lib.PeGruLayer) == 8 + 3 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(_lib.PeInfo) == 9 * 4 + 4 + 8


def test_create_argument_validation_without_gpu():
    """Argument errors are reported before any device work, with the reference's exception types."""
    from mycroft_precise_amd import synth
    from mycroft_precise_amd.params import pr
    w = synth.make_weights()
    with pytest.raises(ValueError):
        _lib.HipEngine(pr, w, n_streams=0)
    bad = pr.copy()
    bad.__dict__['n_fft'] = 1500                    # (powers of two from 64 to 2048 and any other length from 16 to 1024 have kernels)
    with pytest.raises((NotImplementedError, ValueError)):
        _lib.HipEngine(bad, w, n_streams=1, mel_filters=np.zeros((20, 751)))
    bad.__dict__['n_fft'] = 12
    with pytest.raises((NotImplementedError, ValueError)):
        _lib.HipEngine(bad, w, n_streams=1, mel_filters=np.zeros((20, 7)))
    bad.__dict__['n_fft'] = 4096
    with pytest.raises((NotImplementedError, ValueError)):
        _lib.HipEngine(bad, w, n_streams=1, mel_filters=np.zeros((20, 2049)))
    delta = pr.copy()
    delta.__dict__['use_delta'] = True
    with pytest.raises(ValueError):                 # use_delta needs a layer with 2 * n_mfcc inputs
        _lib.HipEngine(delta, w, n_streams=1)
    big = synth.make_weights(units=(320,))               # the streamed-weight kernel holds up to 256 units per layer
    with pytest.raises(NotImplementedError):
        _lib.HipEngine(pr, big, n_streams=1)
    deep = synth.make_weights(units=(64, 64, 64))        # one or two GRU layers have kernels
    with pytest.raises(NotImplementedError):
        _lib.HipEngine(pr, deep, n_streams=1)


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='a GPU is present')
def test_no_gpu_is_a_loud_error_not_a_fallback():
    from mycroft_precise_amd import synth
    from mycroft_precise_amd.params import pr
    with pytest.raises(_lib.EngineError) as ei:
        _lib.HipEngine(pr, synth.make_weights(), n_streams=4)
    assert 'hipSetDevice' in str(ei.value)


def test_missing_library_is_a_loud_error(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', os.path.join(REPO, 'does', 'not', 'exist.so'))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.load()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, 'mycroft_precise_amd')
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'oracle.' not in src or f == 'synth.py', f

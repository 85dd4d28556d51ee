"""CPU: the one-frame-per-wave MFCC kernel's tables and data flow, replayed lane by lane on the host
(tools/emulate_mfcc_wave.cpp compiles the SAME table builder and the SAME per-lane arithmetic as the HIP kernel,
mycroft_precise_amd/csrc/mfcc_wave_{tables,core}.h) against the oracle's MFCC of the same frames."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO
from mycroft_precise_amd import synth
from mycroft_precise_amd import vectorization as V
from oracle import sonopy_restated as so, speechpy_restated as sp


@pytest.fixture(scope='module')
def emulator(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('emu') / 'emulate_mfcc_wave')
    subprocess.run(['g++', '-O2', '-std=c++17', '-I', os.path.join(REPO, 'mycroft_precise_amd', 'csrc'),
                    os.path.join(REPO, 'tools', 'emulate_mfcc_wave.cpp'), '-o', exe], check=True)
    return exe


def _frames():
    rng = np.random.default_rng(1)
    rows = [synth.stream_pcm(s, 512, k) for s, k in ((0, 'tone_noise'), (5, 'tone_noise'), (12, 'square'), (1, 'quiet'), (0, 'zeros'))]
    rows.append(rng.integers(-32768, 32767, 512).astype('<i2'))
    rows.append(np.full(512, -32768, '<i2'))
    return np.stack(rows).astype('<i2')


@pytest.mark.parametrize('name,n_filt', [('sonopy', 20), ('speechpy', 20), ('sonopy', 40), ('sonopy', 26), ('speechpy', 40), ('sonopy', 48)])
def test_wave_data_flow_matches_the_oracle(emulator, tmp_path, name, n_filt):
    frames = _frames()
    frames.tofile(str(tmp_path / 'frames.bin'))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', V.UnverifiedFilterbank)
        bank = V.mel_filterbank(16000, n_filt, 257) if name == 'sonopy' else V.speechpy_filterbank(16000, n_filt, 257)
    ref = so.mfcc_from_frames if name == 'sonopy' else sp.mfcc_from_frames
    want = ref(frames.astype(np.float64) / 32768.0, num_filt=n_filt)
    bank.astype('<f8').tofile(str(tmp_path / 'filt.bin'))
    for prec, tol in (('f64', 1e-10), ('f32', 5e-4)):
        out = str(tmp_path / ('out_%s.bin' % prec))
        r = subprocess.run([emulator, prec, str(n_filt), '13', '0' if name == 'sonopy' else '1', str(tmp_path / 'filt.bin'),
                            str(tmp_path / 'frames.bin'), out], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        got = np.fromfile(out).reshape(len(frames), 13 + n_filt)
        assert np.abs(got[:, :13] - want).max() <= tol, (prec, np.abs(got[:, :13] - want).max())


def test_filterbanks_that_do_not_fit_one_wave_are_refused(emulator, tmp_path):
    """64 lanes x 16 bins per lane is the mel pass's capacity: 64 sonopy filters over 257 bins need more runs."""
    _frames().tofile(str(tmp_path / 'frames.bin'))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', V.UnverifiedFilterbank)
        V.mel_filterbank(16000, 64, 257).astype('<f8').tofile(str(tmp_path / 'filt.bin'))
    r = subprocess.run([emulator, 'f64', '64', '13', '0', str(tmp_path / 'filt.bin'), str(tmp_path / 'frames.bin'),
                        str(tmp_path / 'o.bin')], capture_output=True, text=True)
    assert r.returncode == 3 and 'too wide for one wave' in r.stderr

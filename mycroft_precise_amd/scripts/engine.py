#!/usr/bin/env python3
"""
precise-engine for MI355X: same wire protocol as the reference's
``precise/scripts/engine.py`` (/root/reference/precise/scripts/engine.py:14-67).

stdin is a stream of raw little-endian int16 mono 16 kHz audio, consumed CHUNK_SIZE bytes at a
time (all of stdin at once if CHUNK_SIZE is omitted); for every chunk one confidence is written
to stdout as ``str(float) + '\\n'`` and flushed.  Everything else goes to stderr.

    python -m mycroft_precise_amd.scripts.engine MODEL [CHUNK_SIZE] < audio.raw
"""
import argparse
import sys

from .. import __version__


def build_parser():
    p = argparse.ArgumentParser(prog='precise-engine', description=__doc__,
                                formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('-v', '--version', action='version', version=__version__)
    p.add_argument('model_name', help='model to read from (with its .params)')
    p.add_argument('chunk_size', type=int, nargs='?', default=-1,
                   help='Number of bytes to read before making a prediction. '
                        'Higher values are less computationally expensive')
    return p


class EngineScript:
    """``create(model_name=..., chunk_size=...)`` mirrors how the reference's tests drive its scripts
    (base_script.py:13-30); ``run()`` serves predictions until stdin ends."""

    def __init__(self, args):
        if sys.stdin.isatty():
            raise ValueError('Please pipe audio via stdin using < audio.wav')
        self.args = args

    @classmethod
    def create(cls, **kwargs):
        defaults = dict(model_name=None, chunk_size=-1)
        defaults.update(kwargs)
        return cls(argparse.Namespace(**defaults))

    @staticmethod
    def _emit(sink, confidence):
        sink.write((str(float(confidence)) + '\n').encode('ascii'))
        sink.flush()

    def run(self):
        from ..network_runner import Listener
        real_stdout = sys.stdout
        sys.stdout = sys.stderr            # library chatter must never reach the prediction pipe
        try:
            listener = Listener(self.args.model_name, self.args.chunk_size)
            audio_in, sink = sys.stdin.buffer, real_stdout.buffer
            while True:
                self._emit(sink, listener.update(audio_in))
        except (EOFError, KeyboardInterrupt):
            pass
        finally:
            sys.stdout = real_stdout


def main(argv=None):
    EngineScript(build_parser().parse_args(argv)).run()


if __name__ == '__main__':
    main()

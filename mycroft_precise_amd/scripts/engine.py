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
    def __init__(self, args):
        self.args = args
        if sys.stdin.isatty():
            raise ValueError('Please pipe audio via stdin using < audio.wav')

    @classmethod
    def create(cls, **kwargs):
        ns = argparse.Namespace(model_name=None, chunk_size=-1)
        ns.__dict__.update(kwargs)
        return cls(ns)

    def run(self):
        from ..network_runner import Listener
        stdout = sys.stdout
        sys.stdout = sys.stderr            # only predictions may reach the real stdout
        try:
            listener = Listener(self.args.model_name, self.args.chunk_size)
            while True:
                conf = listener.update(sys.stdin.buffer)
                stdout.buffer.write((str(float(conf)) + '\n').encode('ascii'))
                stdout.buffer.flush()
        except (EOFError, KeyboardInterrupt):
            pass
        finally:
            sys.stdout = stdout


def main(argv=None):
    EngineScript(build_parser().parse_args(argv)).run()


if __name__ == '__main__':
    main()

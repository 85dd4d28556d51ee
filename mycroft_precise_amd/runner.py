"""
Client-side wrapper: drop-in for ``precise_runner``
(/root/reference/runner/precise_runner/runner.py:22-243): ``Engine``, ``PreciseEngine``
(subprocess speaking the stdin/stdout chunk protocol), ``ListenerEngine`` (in-process),
``ReadWriteStream``, ``TriggerDetector`` and ``PreciseRunner``.  Standard library only.

``PreciseEngine`` works unchanged against ``python -m mycroft_precise_amd.scripts.engine`` (the
MI355X engine executable); ``ListenerEngine`` wraps an in-process
``mycroft_precise_amd.network_runner.Listener``.
"""
import atexit
import threading
import time
from subprocess import PIPE, Popen


class Engine(object):
    """Interface: ``chunk_size`` bytes in, one confidence out (runner.py:22-33)."""

    def __init__(self, chunk_size=2048):
        self.chunk_size = chunk_size

    def start(self):
        pass

    def stop(self):
        pass

    def get_prediction(self, chunk):
        raise NotImplementedError


class PreciseEngine(Engine):
    """
    Wraps an engine executable (runner.py:36-67).

    Args:
        exe_file (Union[str, list]): executable, or argv prefix such as
            ``['python', '-m', 'mycroft_precise_amd.scripts.engine']``
        model_file (str): model to load (with its ``.params``)
        chunk_size (int): *bytes* per prediction
    """

    def __init__(self, exe_file, model_file, chunk_size=2048):
        Engine.__init__(self, chunk_size)
        prefix = list(exe_file) if isinstance(exe_file, list) else [exe_file]
        self.exe_args = prefix + [model_file, str(self.chunk_size)]
        self.proc = None

    def start(self):
        self.proc = Popen(self.exe_args, stdin=PIPE, stdout=PIPE)

    def stop(self):
        if self.proc:
            self.proc.kill()
            self.proc = None

    def get_prediction(self, chunk):
        if len(chunk) != self.chunk_size:
            raise ValueError('Invalid chunk size: ' + str(len(chunk)))
        self.proc.stdin.write(chunk)
        self.proc.stdin.flush()
        return float(self.proc.stdout.readline())


class ListenerEngine(Engine):
    """In-process engine around a Listener (runner.py:70-73)."""

    def __init__(self, listener, chunk_size=2048):
        Engine.__init__(self, chunk_size)
        self.get_prediction = listener.update


class ReadWriteStream(object):
    """
    Byte pipe that can be written at any pace; ``read(n)`` blocks until n bytes are there.  When
    ``chop_samples`` is set and more than that is buffered, a read first drops everything but the
    trailing ``len % chop_samples`` bytes (runner.py:76-112).
    """

    def __init__(self, s=b'', chop_samples=-1):
        self.buffer = s
        self.chop_samples = chop_samples
        self._cond = threading.Condition()

    def __len__(self):
        return len(self.buffer)

    def read(self, n=-1, timeout=None):
        with self._cond:
            if n == -1:
                n = len(self.buffer)
            if 0 < self.chop_samples < len(self.buffer):
                keep = len(self.buffer) % self.chop_samples
                self.buffer = self.buffer[-keep:]          # keep == 0 keeps everything (slice [-0:])
            deadline = None if timeout is None else time.time() + timeout
            while len(self.buffer) < n:
                remaining = None if deadline is None else deadline - time.time()
                if remaining is not None and remaining <= 0:
                    return b''
                if not self._cond.wait(remaining):
                    return b''
            chunk, self.buffer = self.buffer[:n], self.buffer[n:]
            return chunk

    def write(self, s):
        with self._cond:
            self.buffer += s
            self._cond.notify_all()

    def flush(self):
        """sys.stdout compatibility"""


class TriggerDetector:
    """
    Debounces per-chunk predictions into activations (runner.py:115-142): an activation fires when
    more than ``trigger_level`` chunks net of decays were above ``1 - sensitivity``; after firing
    the counter is parked at -(8*2048)//chunk_size and counts back up to zero, swallowing chunks.
    """

    def __init__(self, chunk_size, sensitivity=0.5, trigger_level=3):
        self.chunk_size = chunk_size
        self.sensitivity = sensitivity
        self.trigger_level = trigger_level
        self.activation = 0

    def update(self, prob):
        # type: (float) -> bool
        hot = prob > 1.0 - self.sensitivity
        if not hot and self.activation >= 0:
            if self.activation > 0:
                self.activation -= 1
            return False
        self.activation += 1
        fired = self.activation > self.trigger_level
        if fired or (hot and self.activation < 0):
            self.activation = -(8 * 2048) // self.chunk_size
        return fired


class PreciseRunner(object):
    """
    Reads audio from ``stream`` in ``engine.chunk_size``-byte chunks on a daemon thread, feeds the
    engine, reports every prediction and debounced activations (runner.py:145-243).

    Args:
        engine (Engine)
        trigger_level (int): chunk activations needed to trigger on_activation
        sensitivity (float): 0.0 .. 1.0
        stream (BinaryIO): 16 kHz mono int16 audio source; the microphone (pyaudio) if None
        on_prediction (Callable[[float], None])
        on_activation (Callable[[], None])
    """

    def __init__(self, engine, trigger_level=3, sensitivity=0.5, stream=None,
                 on_prediction=lambda x: None, on_activation=lambda: None):
        self.engine = engine
        self.trigger_level = trigger_level
        self.stream = stream
        self.on_prediction = on_prediction
        self.on_activation = on_activation
        self.chunk_size = engine.chunk_size
        self.pa = None
        self.thread = None
        self.running = False
        self.is_paused = False
        self.detector = TriggerDetector(self.chunk_size, sensitivity, trigger_level)
        atexit.register(self.stop)

    def _wrap_stream_read(self, stream):
        """pyaudio streams count samples, not bytes: read(n) -> read(n // 2)."""
        try:
            import pyaudio
        except ImportError:
            return
        if getattr(stream.read, '__func__', None) is pyaudio.Stream.read:
            stream.read = lambda x: pyaudio.Stream.read(stream, x // 2, False)

    def start(self):
        if self.stream is None:
            from pyaudio import PyAudio, paInt16
            self.pa = PyAudio()
            self.stream = self.pa.open(16000, 1, paInt16, True, frames_per_buffer=self.chunk_size)
        self._wrap_stream_read(self.stream)
        self.engine.start()
        self.running = True
        self.is_paused = False
        self.thread = threading.Thread(target=self._handle_predictions)
        self.thread.daemon = True
        self.thread.start()

    def stop(self):
        if self.thread:
            self.running = False
            if isinstance(self.stream, ReadWriteStream):
                self.stream.write(b'\0' * self.chunk_size)      # unblock the reader
            self.thread.join()
            self.thread = None
        self.engine.stop()
        if self.pa:
            self.pa.terminate()
            self.stream.stop_stream()
            self.stream = self.pa = None

    def pause(self):
        self.is_paused = True

    def play(self):
        self.is_paused = False

    def _handle_predictions(self):
        while self.running:
            chunk = self.stream.read(self.chunk_size)
            if self.is_paused:
                continue
            prob = self.engine.get_prediction(chunk)
            self.on_prediction(prob)
            if self.detector.update(prob):
                self.on_activation()

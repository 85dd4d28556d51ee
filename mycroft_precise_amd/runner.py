"""
Client side of the engine protocol -- API-compatible stand-in for the ``precise_runner`` package
(/root/reference/runner/precise_runner/runner.py:22-243, exported names per ``__init__.py:1``).

What a user of that package relies on, and gets here unchanged:

* ``Engine`` -- ``chunk_size`` (bytes per prediction), ``start()``, ``stop()``,
  ``get_prediction(chunk) -> float``;
* ``PreciseEngine(exe_file, model_file, chunk_size=2048)`` -- child process fed raw PCM on stdin,
  answering one ASCII float per line; ``ValueError`` for a chunk of the wrong length;
* ``ListenerEngine(listener, chunk_size=2048)`` -- same interface around an in-process ``Listener``;
* ``ReadWriteStream`` -- thread-safe byte pipe with blocking ``read(n, timeout)``;
* ``TriggerDetector(chunk_size, sensitivity, trigger_level).update(prob) -> bool``;
* ``PreciseRunner(engine, trigger_level, sensitivity, stream, on_prediction, on_activation)`` with
  ``start / stop / pause / play``.

Only the standard library is needed (``pyaudio`` solely when no ``stream`` is supplied).  The MI355X
engine executable is ``python -m mycroft_precise_amd.scripts.engine``.
"""
import atexit
import subprocess
import threading
import time

_REARM_BYTES = 8 * 2048        # audio swallowed after an activation, in bytes (runner.py:137)


class Engine(object):
    """Anything that turns one chunk of audio into one confidence."""

    def __init__(self, chunk_size=2048):
        self.chunk_size = chunk_size

    def start(self):
        """Acquire resources (no-op by default)."""

    def stop(self):
        """Release resources (no-op by default)."""

    def get_prediction(self, chunk):
        raise NotImplementedError


class PreciseEngine(Engine):
    """Engine living in a child process.

    ``exe_file`` is an executable path or an argv prefix (list); the child is started as
    ``exe_file... model_file chunk_size`` and must speak the stdin/stdout protocol of
    ``precise/scripts/engine.py``.
    """

    def __init__(self, exe_file, model_file, chunk_size=2048):
        super(PreciseEngine, self).__init__(chunk_size)
        argv = list(exe_file) if isinstance(exe_file, list) else [exe_file]
        self.exe_args = argv + [model_file, str(chunk_size)]
        self.proc = None

    def start(self):
        self.proc = subprocess.Popen(self.exe_args, stdin=subprocess.PIPE, stdout=subprocess.PIPE)

    def stop(self):
        child, self.proc = self.proc, None
        if child is not None:
            child.kill()

    def get_prediction(self, chunk):
        if len(chunk) != self.chunk_size:
            raise ValueError('Invalid chunk size: ' + str(len(chunk)))
        pipe_in, pipe_out = self.proc.stdin, self.proc.stdout
        pipe_in.write(chunk)
        pipe_in.flush()
        return float(pipe_out.readline())


class ListenerEngine(Engine):
    """Engine backed by ``listener.update`` in this process."""

    def __init__(self, listener, chunk_size=2048):
        super(ListenerEngine, self).__init__(chunk_size)
        self.get_prediction = listener.update


class ReadWriteStream(object):
    """Byte pipe: producers ``write`` whenever they like, a consumer ``read(n)`` blocks until ``n``
    bytes are available (or ``timeout`` seconds passed, returning ``b''``).  With ``chop_samples`` set
    and more than that buffered, a read first discards all but the trailing
    ``len(buffer) % chop_samples`` bytes -- and nothing at all when that remainder is zero."""

    def __init__(self, s=b'', chop_samples=-1):
        self.buffer = s
        self.chop_samples = chop_samples
        self._cond = threading.Condition()

    def __len__(self):
        return len(self.buffer)

    def _chop(self):
        if 0 < self.chop_samples < len(self.buffer):
            keep = len(self.buffer) % self.chop_samples
            if keep:
                self.buffer = self.buffer[-keep:]

    def read(self, n=-1, timeout=None):
        with self._cond:
            if n == -1:
                n = len(self.buffer)
            self._chop()
            give_up = None if timeout is None else time.time() + timeout
            while len(self.buffer) < n:
                left = None if give_up is None else give_up - time.time()
                if left is not None and left <= 0:
                    return b''
                if not self._cond.wait(left):
                    return b''
            head, self.buffer = self.buffer[:n], self.buffer[n:]
            return head

    def write(self, s):
        with self._cond:
            self.buffer += s
            self._cond.notify_all()

    def flush(self):
        """Present so the object can stand in for ``sys.stdout``."""


class TriggerDetector:
    """Turns a run of per-chunk confidences into discrete activations.

    A counter climbs by one for every chunk above ``1 - sensitivity`` and decays by one otherwise;
    passing ``trigger_level`` fires, after which the counter is parked ``8 * 2048 / chunk_size`` chunks
    below zero and has to climb back before anything can fire again (runner.py:115-142)."""

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
            self.activation = -_REARM_BYTES // self.chunk_size
        return fired


class BatchTriggerDetector:
    """``TriggerDetector`` for many streams at once (numpy, host side): ``update(probs[n])`` returns the
    boolean activations of this round.  Same state machine per stream; ``pe_decode`` keeps an identical
    one on the device for ``BatchedListener.update_detect``."""

    def __init__(self, n_streams, chunk_size, sensitivity=0.5, trigger_level=3):
        import numpy as np
        self._np = np
        self.chunk_size = chunk_size
        self.sensitivity = sensitivity
        self.trigger_level = trigger_level
        self.activation = np.zeros(int(n_streams), dtype=np.int64)

    def update(self, probs):
        np = self._np
        hot = np.asarray(probs, dtype=np.float64) > 1.0 - self.sensitivity
        act = self.activation
        counting = hot | (act < 0)                      # streams whose counter moves up this round
        decayed = np.where(~counting & (act > 0), act - 1, act)
        bumped = act + 1
        fired = counting & (bumped > self.trigger_level)
        rearm = counting & (fired | (hot & (bumped < 0)))
        self.activation = np.where(rearm, -_REARM_BYTES // self.chunk_size, np.where(counting, bumped, decayed))
        return fired


class PreciseRunner(object):
    """Pumps ``stream`` into ``engine`` on a daemon thread.

    Every ``engine.chunk_size`` bytes read produce one ``on_prediction(prob)``; a ``TriggerDetector``
    built from ``sensitivity`` / ``trigger_level`` decides when to call ``on_activation()``.  Without a
    ``stream`` the default microphone is opened through pyaudio (16 kHz, mono, int16).
    """

    def __init__(self, engine, trigger_level=3, sensitivity=0.5, stream=None,
                 on_prediction=lambda x: None, on_activation=lambda: None):
        self.engine = engine
        self.trigger_level = trigger_level
        self.stream = stream
        self.on_prediction = on_prediction
        self.on_activation = on_activation
        self.chunk_size = engine.chunk_size
        self.detector = TriggerDetector(self.chunk_size, sensitivity, trigger_level)
        self.pa = None
        self.thread = None
        self.running = False
        self.is_paused = False
        atexit.register(self.stop)

    # -- microphone plumbing (only without a caller-supplied stream) ---------------------------------
    def _open_microphone(self):
        import pyaudio
        self.pa = pyaudio.PyAudio()
        self.stream = self.pa.open(16000, 1, pyaudio.paInt16, True, frames_per_buffer=self.chunk_size)

    def _wrap_stream_read(self, stream):
        """pyaudio counts samples where everything else here counts bytes."""
        try:
            import pyaudio
        except ImportError:
            return
        if getattr(stream.read, '__func__', None) is pyaudio.Stream.read:
            stream.read = lambda nbytes: pyaudio.Stream.read(stream, nbytes // 2, False)

    # -- lifecycle -----------------------------------------------------------------------------------
    def start(self):
        if self.stream is None:
            self._open_microphone()
        self._wrap_stream_read(self.stream)
        self.engine.start()
        self.is_paused = False
        self.running = True
        worker = threading.Thread(target=self._handle_predictions)
        worker.daemon = True
        self.thread = worker
        worker.start()

    def stop(self):
        worker, self.thread = self.thread, None
        if worker is not None:
            self.running = False
            if isinstance(self.stream, ReadWriteStream):
                self.stream.write(b'\0' * self.chunk_size)       # wake a reader blocked on an empty pipe
            worker.join()
        self.engine.stop()
        if self.pa is not None:
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

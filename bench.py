#!/usr/bin/env python3
"""
bench.py -- windows/sec of the MI355X wake-word hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (pe_update_device_keep: int16 PCM chunk -> MFCC frames -> feature
window -> GRU -> probability; the PCM slabs are resident and outlive the calls, so the leftover samples of
Listener.update_vectors stay in them -- `--keep 0` / the line's "carry_copy_path": pe_update_device, which copies
them to the engine's carry in every call, same bits) over one batch of synthetic streams: one 1024-sample chunk for each
of ``--streams`` (4096) streams per GPU = BASELINE.json configs[1] at N=1 and configs[2] at N=8
(weak scaling, streams sharded across ranks, no collective on the data path; the per-step
probabilities of the timed region are gathered to rank 0 once at its end, inside the timing).
The PCM of all W+K steps is resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line: metric/value (whole-job windows/s), ms_per_step, plus
  "roofline"       the launch of the timed region (network || MFCC || bookkeeping roles in one kernel) vs the dense
                   fp32 MFMA peak (HIP-event time on the launch stream); with --gru-precision bf16 (configs[4]) vs the
                   HBM peak, which is what binds that configuration,
  "roofline_gru" / "roofline_mfcc"   the two stages launched separately,
  "cpu_baseline"   the numpy oracle ("port") timed on this box's host cores (N=1 only),
  "cpu_baseline_single_stream"   one stream through Listener.update the way the reference runs.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import numpy as np
import torch                                   # first: libprecise_engine binds torch's HIP runtime
import torch.distributed as dist

from mycroft_precise_amd import synth
from mycroft_precise_amd._lib import HipEngine
from mycroft_precise_amd.dist import env_world, gather_probabilities
from mycroft_precise_amd.params import pr

METRIC = 'MFCC+GRU windows/sec (node); max concurrent real-time 16 kHz streams'
CHUNK = 1024                                   # samples per update (2048-byte chunks, runner.py:48)
REALTIME_WINDOWS_PER_S = 16000.0 / CHUNK       # 15.625 updates/s keep one stream real-time
# algorithmic work per window (one update of one stream)
MFCC_BYTES_PER_WINDOW = 2048 + 1.28 * 13 * 4   # PCM read + fp32 feature rows written = 2114.6 B
GRU_FLOP_PER_WINDOW = 2 * 29 * (13 * 60 + 20 * 60) + 2 * 20      # 114 880
HBM_PEAK_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3                   # dense fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0                 # dense bf16 matrix peak (MI355X_MICROARCH.md)


def synth_pcm_device(n_updates, n_streams, first_stream, device):
    """Seeded synthetic PCM of the SURVEY 8(d) shape, generated on the device:
    N(0, 3000^2) + 8000 sin(2 pi f_s t), f_s = 200 + 37 (s mod 97) Hz  ->  int16 [n_updates, B, CHUNK]."""
    g = torch.Generator(device=device)
    g.manual_seed(1234 + first_stream)
    out = torch.empty((n_updates, n_streams, CHUNK), dtype=torch.int16, device=device)
    sid = torch.arange(first_stream, first_stream + n_streams, device=device, dtype=torch.float64)
    freq = (200.0 + 37.0 * torch.remainder(sid, 97.0)).view(n_streams, 1)
    for u in range(n_updates):
        t = (torch.arange(CHUNK, device=device, dtype=torch.float64) + u * CHUNK).view(1, CHUNK) / 16000.0
        x = 8000.0 * torch.sin(2.0 * np.pi * freq * t)
        x = x + 3000.0 * torch.randn((n_streams, CHUNK), generator=g, device=device, dtype=torch.float64)
        out[u] = torch.clamp(torch.round(x), -32768, 32767).to(torch.int16)
    return out


def _affinity_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _physical_cores():
    """One logical CPU per PHYSICAL core of this process's affinity set (the first hardware thread of every
    /sys/devices/system/cpu/cpuN/topology/thread_siblings_list group); falls back to the affinity set itself."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return list(range(os.cpu_count() or 1))
    seen, picked = set(), []
    for c in allowed:
        try:
            with open('/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list' % c) as f:
                txt = f.read().strip()
            sib = set()
            for part in txt.split(','):
                if '-' in part:
                    lo, hi = part.split('-')
                    sib.update(range(int(lo), int(hi) + 1))
                else:
                    sib.add(int(part))
            key = min(sib)
        except (OSError, ValueError):
            key = c
        if key not in seen:
            seen.add(key)
            picked.append(c)
    return picked or allowed


def _cpu_worker(args):
    """One host core's share of the cpu_baseline: the numpy oracle ("port") on its own slice of streams.
    Import, weights and PCM synthesis happen BEFORE the start barrier; only the arithmetic is timed.  The loop
    runs until `seconds` have passed (so the sample is bounded whatever the loaded per-core rate turns out to
    be); returns (updates done, own compute time)."""
    first, n_streams, seconds, n_distinct, seed, barrier, cpu = args
    if cpu is not None:
        try:
            os.sched_setaffinity(0, {cpu})                  # one worker per physical core, pinned
        except (AttributeError, OSError):
            pass
    from oracle import listener as oracle_listener          # checker / baseline only
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)                                # one core per worker, no BLAS oversubscription
    except ImportError:
        pass
    weights = synth.make_weights(seed=seed)
    base = synth.batch_pcm(min(n_streams, 64), n_distinct, CHUNK, first_stream=first)     # [n_distinct, <=64, CHUNK]
    pcm = np.ascontiguousarray(np.tile(base, (1, (n_streams + base.shape[1] - 1) // base.shape[1], 1))[:, :n_streams])
    oracle = oracle_listener.BatchedOracle(weights, n_streams)
    for u in range(5):                                       # first touches (page faults, allocator growth: the first
        oracle.update_raw(pcm[u % n_distinct])               # updates of a 1024-stream batch run 10-20x slower), untimed
    if barrier is not None:
        barrier.wait()
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.update_raw(pcm[(n + 5) % n_distinct])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return n, dt


def _cpu_round(cpus, streams_per_core, seconds):
    import multiprocessing as mp
    ctx = mp.get_context('fork')
    cores = len(cpus)
    barrier = ctx.Barrier(cores)
    procs, results = [], ctx.Queue()

    def run(job):
        results.put(_cpu_worker(job))

    t0 = time.perf_counter()
    for i, c in enumerate(cpus):
        pr_ = ctx.Process(target=run, args=((i * streams_per_core, streams_per_core, seconds, 8, 42, barrier, c),))
        pr_.start()
        procs.append(pr_)
    done = [results.get() for _ in procs]
    for pr_ in procs:
        pr_.join()
    wall = time.perf_counter() - t0
    windows = sum(n for n, _ in done) * streams_per_core
    compute = max(dt for _, dt in done)
    return {'streams_per_core': streams_per_core, 'windows': windows, 'compute_s': compute, 'wall_s': wall,
            'value': windows / compute}


def _cpu_torch_batched(threads, n_streams, seconds):
    from oracle.torch_batched import TorchBatchedOracle
    prev = torch.get_num_threads()
    torch.set_num_threads(int(threads))
    try:
        weights = synth.make_weights(seed=42)
        base = synth.batch_pcm(64, 8, CHUNK)
        pcm = torch.from_numpy(np.ascontiguousarray(np.tile(base, (1, n_streams // 64, 1))))
        oracle = TorchBatchedOracle(weights, n_streams)
        for u in range(5):
            oracle.update_raw(pcm[u % 8])
        n, t0 = 0, time.perf_counter()
        while True:
            oracle.update_raw(pcm[n % 8])
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds:
                break
    finally:
        torch.set_num_threads(prev)
    return {'value': n_streams * n / dt, 'unit': 'windows/s', 'threads': int(threads), 'streams': n_streams, 'compute_s': dt,
            'sample': 'torch-CPU restatement, %d intra-op threads, %d streams per update, %d updates in %.1f s' % (threads, n_streams, n, dt)}


def cpu_baseline(seconds=4.0, streams_per_core=128):
    """Oracle ("port" of the reference's sonopy + Keras arithmetic, vectorised over streams) on the host cores, on a
    bounded sample of the same workload: ONE process per PHYSICAL core, pinned to it (sched_setaffinity; the second
    hardware thread of a core adds nothing to a float64 numpy loop and un-pinned workers migrate), ONE round at the
    batch shape that measured fastest on this pool's hosts (128 streams per core: cache-resident; 1024 per core -- the
    GPU's own regime -- runs at half that rate because the float64 temporaries fall out of cache).  All workers start
    their timed loops together (barrier after fork / synthesis) and run for `seconds`; rate = windows of all workers /
    the longest worker's compute time.  One core alone is timed first; `linear_scaling_reference` = that rate x the
    physical cores (what the box would do if the cores did not share memory bandwidth / clocks)."""
    logical = _affinity_cores()
    cpus = _physical_cores()
    cores = len(cpus)
    from oracle import listener as _warm_import      # noqa: F401  (imported once here: the forked workers inherit it)
    try:
        prev = os.sched_getaffinity(0)
    except AttributeError:
        prev = None
    n_alone, t_alone = _cpu_worker((0, 1024, 1.0, 8, 42, None, cpus[0]))        # one core, nothing else running
    n_a128, t_a128 = _cpu_worker((0, streams_per_core, 1.0, 8, 42, None, cpus[0]))
    if prev is not None:
        os.sched_setaffinity(0, prev)
    r = _cpu_round(cpus, streams_per_core, seconds)
    alone = max(1024 * n_alone / t_alone, streams_per_core * n_a128 / t_a128)
    # BASELINE.md B2 "numpy / torch-CPU": the same restatement on torch's multi-threaded CPU kernels, ONE process, the GPU's own
    # batch (4096 streams per update), every physical core as an intra-op thread -- after the forked numpy workers are gone
    # (forking a process whose thread pool is already running can hang).  Whichever is faster is the baseline's `value`.
    tb = None
    try:
        tb = _cpu_torch_batched(cores, 4096, 2.0)
    except Exception as ex:                                  # noqa: BLE001  (the baseline must not cost the bench its line)
        tb = {'error': repr(ex)}
    best_is_torch = bool(tb and tb.get('value', 0.0) > r['value'])
    return {'value': tb['value'] if best_is_torch else r['value'], 'unit': 'windows/s', 'cores': cores, 'physical_cores': cores, 'logical_cpus': logical,
            'pinned': True, 'kind': 'port', 'implementation': 'torch-CPU batched (oracle/torch_batched.py)' if best_is_torch else 'numpy, one pinned process per physical core (oracle/listener.py)',
            'numpy_multiprocess': {'value': r['value'], 'unit': 'windows/s'}, 'torch_batched': tb,
            'compute_s': r['compute_s'], 'wall_s': r['wall_s'] + t_alone + t_a128,
            'per_core': r['value'] / cores, 'single_core_alone': alone,
            'single_core_alone_1024_streams': 1024 * n_alone / t_alone,
            'single_core_alone_%d_streams' % streams_per_core: streams_per_core * n_a128 / t_a128,
            'linear_scaling_reference': alone * cores,
            'rounds': [r],
            'sample': 'numpy oracle (float64 MFCC + float32 GRU), one pinned process per physical core on %d cores (%d logical '
                      'CPUs), %d streams per core, timed loops start together after fork/synthesis and run %.0f s: %d windows in '
                      '%.1f s = %.0f windows/s; one core alone: %.0f windows/s (x %d cores = %.0f if the cores scaled linearly)'
                      % (cores, logical, streams_per_core, seconds, r['windows'], r['compute_s'], r['value'], alone, cores, alone * cores)}


def cpu_baseline_single_stream(seconds=3.0):
    """ONE stream through Listener.update, the way the reference runs (batch 1, one chunk per
    call).  With /root/reference present (the build container) it is the reference's own unmodified
    precise.network_runner.Listener with the restated third-party arithmetic plugged into its two seams
    (sonopy module, runner_cls); on the GPU box, where the reference does not exist, the oracle's
    restatement of that class."""
    from oracle import listener as oracle_listener, keras_gru, sonopy_restated
    weights = synth.make_weights()
    pcm = synth.stream_pcm(0, 64 * CHUNK)
    chunks = [pcm[i * CHUNK:(i + 1) * CHUNK].tobytes() for i in range(64)]
    kind, listener = 'port', None
    ref_root = '/root/reference'
    if os.path.isdir(os.path.join(ref_root, 'precise')):
        try:
            sys.dont_write_bytecode = True
            import warnings
            warnings.simplefilter('ignore', DeprecationWarning)
            sys.modules.setdefault('sonopy', sonopy_restated)
            if ref_root not in sys.path:
                sys.path.insert(0, ref_root)
            from precise.network_runner import Listener as RefListener
            listener = RefListener('synthetic-model-not-on-disk', 2 * CHUNK, runner_cls=keras_gru.make_runner_cls(weights))
            kind = 'reference-glue'
        except Exception:                                    # noqa: BLE001  (any import problem: fall back, labelled)
            listener = None
    if listener is None:
        listener = oracle_listener.OracleListener(weights)
    for c in chunks[:40]:
        listener.update(c)                                   # fill the feature window, untimed
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        listener.update(chunks[n % 64])
        n += 1
    dt = time.perf_counter() - t0
    return {'value': n / dt, 'unit': 'windows/s', 'cores': 1, 'kind': kind, 'ms_per_update': 1e3 * dt / n,
            'realtime_factor': (n / dt) / REALTIME_WINDOWS_PER_S,
            'sample': '%d Listener.update calls of one %d-sample chunk on one stream in %.1f s; %s'
                      % (n, CHUNK, dt, "the reference's unmodified precise.network_runner.Listener with restated sonopy / "
                         "Keras-GRU arithmetic in its seams" if kind == 'reference-glue' else
                         "oracle.listener.OracleListener (the reference tree is not on this box)")}


def wait_for_gpu():
    """End of a timed region: the contract's torch.cuda.synchronize().  (Measured, tools/gpu_host_overhead.py: a 20-step
    region costs the host ~20 us more than the GPU's own first-start-to-last-end time either way; spinning on an event
    query first and synchronizing afterwards costs 15 us MORE than synchronizing alone.)"""
    torch.cuda.synchronize()


def extra_config(name, device, dev_index, units, streams, mfcc_precision, gru_precision, ring_precision, steps, warmup, n_res, tol,
                 params_kw=None, roofline_kind=None, gru_tiling=-1, keep=False):
    """One non-headline configuration on this GPU (N = 1 only, after the headline's timed region): the same step
    definition, its own roofline object, and a parity spot-check of the timed region's last probabilities against
    the oracle (the checker: never inside a timed region) on the first 256 streams (32 for the wide network).
    params_kw: ListenerParams overrides (n_fft / n_filt / n_mfcc ...: the general front end, params.py:28-118).
    roofline_kind: 'hbm' (fused launch vs HBM), 'mfma' (network launch vs fp32 MFMA), 'mfma_fused' (fused launch vs fp32
    MFMA: the capacity point, where the update IS the network + MFCC roles of one launch), 'mfma_update_x3' (the same for
    the float32 network on the bf16 pipe, whose update is two launches), 'hbm_mfcc' (MFCC launch vs HBM).
    gru_tiling: pe_set_gru_tiling (-1 = the engine's own choice).
    keep: pe_update_device_keep -- the resident slabs outlive every call, so the leftover samples stay in them."""
    import warnings
    from oracle import listener as oracle_listener
    hpr, opr = pr, None
    if params_kw:
        hpr = pr.copy()
        hpr.__dict__.update(params_kw)
        opr = oracle_listener.Params(**params_kw)
    n_mfcc = int(hpr.n_mfcc)
    weights = synth.make_weights(n_in=n_mfcc, units=units)
    stock = units == (20,)
    flop_per_window = 2 * sum(29 * 3 * h * (f + h) for f, h in zip((n_mfcc,) + units[:-1], units)) + 2 * units[-1]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        engine = HipEngine(hpr, weights, n_streams=streams, device=dev_index, mfcc_precision=mfcc_precision,
                           gru_precision=gru_precision, ring_precision=ring_precision)
    if gru_tiling >= 0:
        engine.set_gru_tiling(gru_tiling)
    tiling_used = engine.gru_tiling()
    pcm = synth_pcm_device(n_res, streams, 0, device)
    out = torch.zeros((streams,), dtype=torch.float32, device=device)
    st = torch.cuda.current_stream().cuda_stream
    chunk_bytes = streams * CHUNK * 2

    def run(first, n):
        for i in range(n):
            engine.update_device(pcm.data_ptr() + ((first + i) % n_res) * chunk_bytes, CHUNK, out.data_ptr(), st, keep=keep)

    run(0, warmup)
    torch.cuda.synchronize()
    # two timed passes of `steps` steps, both reported, the better one is `value`: a one-off stall of the host or the allocator
    # inside a 10 ms region (seen once in a while right after the 4 GB slabs of a 65 536-stream configuration are allocated)
    # is not the configuration's throughput
    n_check = min(256 if units == (20,) else 32, streams)      # streams replayed by the oracle afterwards (the wide network costs it 35 Mflop per window)
    passes, got = [], None
    for k in range(2):
        t0 = time.perf_counter()
        run(warmup + k * steps, steps)
        wait_for_gpu()
        passes.append(time.perf_counter() - t0)
        if k == 0:          # the parity spot-check is on the first pass's last probabilities (the oracle replays warmup + steps updates)
            got = out[:n_check].cpu().numpy().astype(np.float64)
    elapsed = min(passes)
    # launch durations: HIP events on the launch stream (bracket of back-to-back updates; the two stages apart)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n_b = min(steps, 100)
    ev0.record()
    run(warmup + 2 * steps, n_b)
    ev1.record()
    ev1.synchronize()
    update_ms = ev0.elapsed_time(ev1) / n_b
    engine.set_fused(False)
    engine.set_timing(True)
    g_ms, m_ms = [], []
    for i in range(min(steps, 40)):
        run(warmup + 2 * steps + n_b + i, 1)
        tm = engine.last_timing()
        m_ms.append(tm[0])
        g_ms.append(tm[1])
    engine.set_timing(False)
    engine.close()
    gru_ms, mfcc_ms = float(np.mean(g_ms)), float(np.mean(m_ms))
    # parity spot-check: the oracle replays the same chunks for streams 0..n_check-1
    host = pcm[:, :n_check].cpu().numpy()
    oracle = oracle_listener.BatchedOracle(weights, n_check, opr) if opr is not None else oracle_listener.BatchedOracle(weights, n_check)
    want = None
    for i in range(warmup + steps):
        want = oracle.update_raw(host[i % n_res])
    err = float(np.abs(np.asarray(want, dtype=np.float64) - got).max())
    if roofline_kind is None:
        roofline_kind = 'hbm' if gru_precision == 'bf16' else 'mfma'
    mfcc_name = 'double' if mfcc_precision == 'f64' else 'float'
    bytes_per_window = 2048 + 1.28 * n_mfcc * (2 if ring_precision == 'bf16' else 4)
    if roofline_kind in ('hbm', 'hbm_mfcc'):
        ms = update_ms if roofline_kind == 'hbm' else mfcc_ms
        ach = bytes_per_window * streams / (ms * 1e-3) / 1e9
        kern = ('fused_update_bf16_kernel%s<%s, ShapeStock>' % ('_nopk' if mfcc_name == 'float' else '', mfcc_name) if roofline_kind == 'hbm'
                else 'mfcc_general_stream_kernel<%s> (one HIP event pair per launch)' % mfcc_name)
        roof = {'kernel': kern, 'bound': 'hbm',
                'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS, 'traffic': None,
                'avg_launch_ms': ms, 'algorithmic': '%.1f B/window x %d windows/launch' % (bytes_per_window, streams)}
    else:
        fused = roofline_kind in ('mfma_fused', 'mfma_update_x3')
        ms = update_ms if fused else gru_ms
        ach = flop_per_window * streams / (ms * 1e-3) / 1e12
        kern = ('mfcc_kernel<%s, ShapeStock, true> then gru_x3_kernel<1> (two dependent launches; the float32 products are formed on the bf16 '
                'matrix pipe, priced here against the fp32 MFMA peak like the other float32 lines)' % mfcc_name if roofline_kind == 'mfma_update_x3'
                else 'fused_update_kernel<%s, ShapeStock, 5, false, false, false> (network || MFCC || bookkeeping roles)' % mfcc_name if fused
                else ('gru_wide_kernel<%d, 1, 4>' % ((units[0] + 63) // 64) if not stock else 'network launch'))
        roof = {'kernel': kern, 'bound': 'mfma',
                'achieved': ach, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / MFMA_F32_PEAK_TFLOPS, 'traffic': None,
                'avg_launch_ms': ms, 'algorithmic': '%d flop/window x %d windows/launch' % (flop_per_window, streams)}
    res = {'name': name, 'value': streams * steps / elapsed, 'unit': 'windows/s', 'ms_per_step': 1e3 * elapsed / steps,
           'realtime_streams': streams * steps / elapsed / REALTIME_WINDOWS_PER_S,
           'steps': steps, 'warmup': warmup, 'dtype': gru_precision,
           'timed_passes_ms_per_step': [1e3 * p_ / steps for p_ in passes],      # two passes of `steps` steps; `value` is the better one
           'config': {'workload': name, 'streams_per_gpu': streams, 'gru': 'H=%s' % ','.join(map(str, units)),
                      'mfcc_dtype': mfcc_precision, 'feature_rows': ring_precision,
                      'gru_form': ({0: 'bf16 operands, eight gate values per lane (gru_bf16_device.h)', 1: 'bf16 operands, five gate values per lane (gru_b20_device.h)'}[tiling_used]
                                   if gru_precision == 'bf16' else
                                   {0: 'streamed float32 weights, v_mfma_f32_16x16x4_f32 (gru_wide_device.h)',
                                    2: 'streamed float32 weights split into 3 bf16 pieces in registers, products on v_mfma_f32_16x16x32_bf16 (gru_wide_x3_device.h)'}[tiling_used]
                                   if len(units) > 1 or units[0] > 32 else
                                   {0: 'classic tiling, v_mfma_f32_16x16x4_f32', 1: 're-tiled stock width, v_mfma_f32_16x16x4_f32',
                                    2: 'float32 operands as 3 x bf16 pieces, 6 piece products on v_mfma_f32_16x16x32_bf16, float32 accumulate / gates / state'}[tiling_used]),
                      'resident_pcm_mb': n_res * chunk_bytes / 1e6},
           'entry_point': 'pe_update_device_keep' if keep else 'pe_update_device',
           'stage_ms': {'update_back_to_back': update_ms, 'mfcc_launch_alone': mfcc_ms, 'network_launch_alone': gru_ms},
           'roofline': roof,
           'parity': {'max_abs_err': err, 'tol': tol, 'streams_checked': n_check, 'ok': bool(err <= tol),
                      'against': 'oracle.listener.BatchedOracle on the same chunks (checker, outside the timed region)'}}
    if params_kw:
        res['config']['listener_params'] = dict(params_kw)
    return res


PCIE_GEN5_X16_GBS = 63.0                       # one direction, the number DESIGN.md section 5 prices the host-fed path against


def host_fed_extra(device, dev_index, streams=4096, steps=300, warmup=30):
    """The host-fed path: chunks arrive in HOST memory, as the reference's engine receives them
    (precise/scripts/engine.py:60-63, runner/precise_runner/runner.py:62-67) -- pe_update_async / pe_wait with the PCM in
    pinned buffers of the engine (pe_host_alloc: zero-copy DMA), three updates in flight, chunk u + 1 crossing PCIe under
    update u.  Reported: windows/s, achieved PCIe GB/s (2048 B per window host -> device), bit-identity with the
    device-resident path on the same chunks; beside it the same loop from pageable numpy arrays and the synchronous pe_update."""
    weights = synth.make_weights()
    engine = HipEngine(pr, weights, n_streams=streams, device=dev_index)
    n_res = 8
    dev_pcm = synth_pcm_device(n_res, streams, 0, device)
    host_pcm = dev_pcm.cpu().numpy()                                   # pageable copy
    bufs = [engine.host_array((streams, CHUNK), '<i2') for _ in range(n_res)]
    for b, h in zip(bufs, host_pcm):
        b[:] = h
    outs = engine.host_array((3, streams), np.float32)

    def loop(n, first, src):
        for i in range(n):
            engine.update_async(src[(first + i) % n_res], outs[i % 3])
        engine.wait()

    loop(warmup, 0, bufs)
    t0 = time.perf_counter()
    loop(steps, warmup, bufs)
    dt = time.perf_counter() - t0
    last = outs[(steps - 1) % 3].copy()
    # the same chunks through the device-resident entry point on a second engine: the bits must agree
    ref = HipEngine(pr, weights, n_streams=streams, device=dev_index)
    out_d = torch.zeros((streams,), dtype=torch.float32, device=device)
    st = torch.cuda.current_stream().cuda_stream
    for i in range(warmup + steps):
        ref.update_device(dev_pcm[i % n_res].data_ptr(), CHUNK, out_d.data_ptr(), st)
    torch.cuda.synchronize()
    same = bool(np.array_equal(out_d.cpu().numpy(), last))
    ref.close()
    # pageable sources (copied through the engine's pinned staging at the call), and the synchronous pe_update
    n2 = max(20, steps // 4)
    pageable = [host_pcm[i] for i in range(n_res)]
    loop(10, 0, pageable)
    t1 = time.perf_counter()
    loop(n2, 10, pageable)
    dt_pageable = time.perf_counter() - t1
    t2 = time.perf_counter()
    for i in range(n2):
        engine.update(pageable[i % n_res])
    dt_sync = time.perf_counter() - t2
    engine.close()
    gbs = streams * CHUNK * 2 * steps / dt / 1e9
    return {'name': 'host-fed: stock GRU fp32 + f64 MFCC, batch=%d streams, chunks in pinned HOST memory (pe_update_async, 3 updates in flight)' % streams,
            'value': streams * steps / dt, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
            'realtime_streams': streams * steps / dt / REALTIME_WINDOWS_PER_S,
            'pcie': {'bound': 'pcie', 'achieved': gbs, 'peak': PCIE_GEN5_X16_GBS, 'unit': 'GB/s', 'frac': gbs / PCIE_GEN5_X16_GBS,
                     'algorithmic': '2048 B/window host -> device + 4 B/window back'},
            'bit_identical_to_device_resident_path': same,
            'pageable_sources': {'value': streams * n2 / dt_pageable, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt_pageable / n2,
                                 'note': 'numpy arrays in pageable memory: one CPU copy into the pinned ring per update'},
            'synchronous_pe_update': {'value': streams * n2 / dt_sync, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt_sync / n2,
                                      'note': 'hipMemcpy -> launch -> hipMemcpy, nothing overlapped (the drop-in Listener / BatchedListener.update path)'},
            'config': {'workload': 'host-fed stock configuration', 'streams_per_gpu': streams, 'resident_pcm_mb': 0.0}}


def single_stream_latency_extra(n_calls=400):
    """BASELINE configs[0] on the GPU side: ONE stream through the drop-in Listener.update(bytes) (the call precise-engine makes
    per 2048-byte chunk, scripts/engine.py:60-63) -- host bytes in, decoded probability out, per-call latency."""
    import tempfile
    from mycroft_precise_amd.model import save_weights
    from mycroft_precise_amd.network_runner import Listener
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, 'synthetic.npz')
        save_weights(path, synth.make_weights())
        from mycroft_precise_amd.params import save_params
        save_params(path)                          # (a model without its .params makes inject_params print a warning -- to stdout, as the reference does)
        lis = Listener(path, 2 * CHUNK)
        pcm = synth.stream_pcm(0, 64 * CHUNK)
        chunks = [pcm[i * CHUNK:(i + 1) * CHUNK].tobytes() for i in range(64)]
        for c in chunks[:40]:
            lis.update(c)
        lat = []
        for i in range(n_calls):
            t0 = time.perf_counter()
            lis.update(chunks[i % 64])
            lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e6
    return {'name': 'configs[0] on the GPU: 1 stream, Listener.update(2048 bytes) per call (host bytes in, decoded probability out)',
            'value': 1e6 / float(np.mean(lat)), 'unit': 'windows/s', 'latency_us': {'median': float(np.median(lat)), 'mean': float(np.mean(lat)), 'p99': float(np.percentile(lat, 99))},
            'realtime_budget_us': 64000.0, 'steps': n_calls,
            'config': {'workload': 'one stream, one 2048-byte chunk per call, synchronous', 'streams_per_gpu': 1}}


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): become
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same argv>`.
    Refuses first, by name, when the node has fewer GPUs than ranks (PE_BENCH_SHARED_GPU=1: every rank on cuda:0, gloo)."""
    import socket
    shared = os.environ.get('PE_BENCH_SHARED_GPU') == '1'
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if shared else n_gpus):
        sys.exit('bench.py --gpus %d: this node exposes %d GPU(s) (torch.cuda.device_count()); one rank per MI355X needs %d'
                 '%s' % (n_gpus, have, n_gpus, '' if shared else ' (PE_BENCH_SHARED_GPU=1 runs every rank on cuda:0 over gloo: a plumbing test, not a measurement)'))
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def headline_parity(pcm, fed, n_timed, got, weights, tol, n_check=256):
    """Parity object of the headline line: the oracle (the checker; never inside a timed region) replays the chunks the
    first `n_check` streams were fed and every probability of the timed region is compared.  `fed` is the whole sequence of
    resident-slab indices since the engine was created (roofline pass, warm-up, timed region); the replay starts late: at an
    update index that is a multiple of 25 (25 x 1024 samples = 32 hops: a fresh stream frames the audio exactly as the
    running one does from there) at least 30 updates before the timed region (the 29-frame window holds only frames from
    the replayed part by then -- 1.28 frames per update)."""
    from oracle import listener as oracle_listener
    n_check = min(n_check, got.shape[1])
    n_pre = len(fed) - n_timed
    start = max(0, (n_pre - 30) // 25 * 25)
    if CHUNK * 25 % pr.hop_samples != 0:
        start = 0
    slabs = sorted(set(fed[start:]))
    host = {i: pcm[i, :n_check].cpu().numpy() for i in slabs}
    oracle = oracle_listener.BatchedOracle(weights, n_check)
    want = []
    for k, i in enumerate(fed[start:]):
        w = oracle.update_raw(host[i])
        if start + k >= n_pre:
            want.append(np.asarray(w, dtype=np.float64))
    want = np.stack(want)
    err = float(np.abs(want - got[:, :n_check].astype(np.float64)).max())
    return {'max_abs_err': err, 'tol': tol, 'ok': bool(err <= tol), 'streams_checked': n_check, 'steps_checked': int(n_timed),
            'oracle_lead_in_updates': int(n_pre - start),
            'against': "oracle.listener.BatchedOracle replaying the timed region's own chunks (checker, outside the timed region); "
                       "every probability of the timed region's probs[steps][0:%d]" % n_check}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=40)
    ap.add_argument('--streams', type=int, default=4096, help='streams per GPU')
    ap.add_argument('--mfcc-precision', choices=['f64', 'f32'], default='f64')
    ap.add_argument('--gru-precision', choices=['f32', 'bf16'], default='f32',
                    help="bf16 = BASELINE configs[4] arithmetic (bf16 MFMA operands, tol 1e-2); not the headline")
    ap.add_argument('--ring-precision', choices=['f32', 'bf16'], default='f32',
                    help="bf16 = 32-byte bf16 feature rows (BASELINE configs[4]: bf16 MFCC+GRU); needs --gru-precision bf16")
    ap.add_argument('--units', default='20', help="GRU widths, e.g. 20 (stock, default) or 256,256 (BASELINE configs[3])")
    ap.add_argument('--gru-tiling', type=int, default=-1, help='pe_set_gru_tiling: -1 automatic (default), 0 classic, 1 re-tiled stock width, 2 float32 products on the bf16 pipe')
    ap.add_argument('--gru-waves', type=int, default=0, help='pe_set_gru_waves: 0 automatic (default), 1 or 4 waves per tile')
    ap.add_argument('--keep', type=int, default=1, choices=[0, 1],
                    help='1 (default): every update goes through pe_update_device_keep -- the PCM slabs are resident in HBM and outlive the calls, so the '
                         'leftover samples stay in them; 0: pe_update_device (the engine copies them to its carry in every call)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra-configs', action='store_true', help='skip the non-headline BASELINE configurations (wide 256x2, bf16) after the headline')
    ap.add_argument('--only-extra', default='', help='run only the extra configurations whose name contains this text')
    ap.add_argument('--no-batched', action='store_true', help='skip the pe_update_many extra (profiling runs: keeps per-kernel means clean)')
    ap.add_argument('--roofline-launches', type=int, default=2000,
                    help='back-to-back launches of the roofline pass that precedes the warm-up (HIP events around the run)')
    ap.add_argument('--resident-updates', type=int, default=256,
                    help='distinct PCM chunks kept in HBM per stream (reused cyclically beyond that)')
    ap.add_argument('--gather-every-step', choices=['rccl', 'host', 'direct', 'none'], default='direct',
                    help="after the headline's timed region, a second one in which step u's probabilities leave the GPU while update "
                         "u + 1 runs: 'rccl' = one asynchronous gather to rank 0 per step (gloo on host copies with PE_BENCH_SHARED_GPU=1), "
                         "'host' = every rank copies its own [B] floats into its own pinned host ring on a side stream (no collective per step), "
                         "'direct' = the update's own output pointer IS a row of the rank's pinned host ring: the network role's final store "
                         "crosses PCIe itself, no copy, no side stream, no call beside the update")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args.gpus)                       # does not return
    rank, local_rank, world = env_world()
    # host-core baseline first, before this process owns a GPU context (it forks workers)
    cpu = cpu_baseline() if (world == 1 and not args.no_cpu_baseline) else None
    cpu_b1 = cpu_baseline_single_stream() if (world == 1 and not args.no_cpu_baseline) else None
    if args.gpus != world:
        sys.exit('bench.py --gpus %d was started as one of %d rank(s) (WORLD_SIZE): launch it bare -- it starts its own ranks -- or '
                 'with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        sys.exit('bench.py needs an MI355X (torch.cuda.is_available() is False)')
    # PE_BENCH_SHARED_GPU=1 (test aid for 1-GPU boxes): every rank uses cuda:0 and the collectives run over gloo
    # on host copies, so the N > 1 control flow can be exercised where RCCL (one device per rank) cannot run.
    shared_gpu = os.environ.get('PE_BENCH_SHARED_GPU') == '1'
    dev_index = 0 if shared_gpu else local_rank
    if dev_index >= torch.cuda.device_count():
        sys.exit('bench.py rank %d: local rank %d has no GPU (this node exposes %d); one rank per MI355X'
                 % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    comm_device = torch.device('cpu') if shared_gpu else device
    # PE_BENCH_FORCE_DIST=1 (test aid for 1-GPU boxes): a world of ONE rank still creates its process group and issues every
    # collective of the N > 1 path -- over RCCL, which two ranks on one GPU cannot use.  Nothing crosses xGMI; what runs is the
    # plumbing (communicator with device_id, barrier(device_ids), the settle-the-collective handshake, gather, clock exchange).
    force_dist = os.environ.get('PE_BENCH_FORCE_DIST') == '1'
    multi = world > 1 or force_dist
    if multi:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', str(rank))
        os.environ.setdefault('WORLD_SIZE', str(world))
        if shared_gpu:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=device)

    B = args.streams
    n_global = B * world
    steps, warmup = args.steps, args.warmup
    n_res = max(2, args.resident_updates)          # independent of --steps: the input working set (n_res x B x 2 KB) is the same for every command line

    units = tuple(int(u) for u in args.units.split(','))
    weights = synth.make_weights(units=units)
    stock = units == (20,)
    flop_per_window = 2 * sum(29 * 3 * h * (f + h) for f, h in zip((13,) + units[:-1], units)) + 2 * units[-1]
    engine = HipEngine(pr, weights, n_streams=B, device=dev_index, mfcc_precision=args.mfcc_precision,
                       gru_precision=args.gru_precision, ring_precision=args.ring_precision)
    if args.gru_tiling != -1:
        engine.set_gru_tiling(args.gru_tiling)
    if args.gru_waves:
        engine.set_gru_waves(args.gru_waves)
    first_stream = int(os.environ.get('PE_BENCH_FIRST_STREAM', '0')) + rank * B      # (test aid: shard offset of a solo run)
    pcm = synth_pcm_device(n_res, B, first_stream, device)
    probs = torch.zeros((steps, B), dtype=torch.float32, device=device)
    scratch = torch.zeros((B,), dtype=torch.float32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    chunk_bytes = B * CHUNK * 2
    pcm_base = pcm.data_ptr()
    probs_base = probs.data_ptr()

    def barrier():
        if multi:
            if shared_gpu:
                dist.barrier()
            else:
                dist.barrier(device_ids=[dev_index])      # (RCCL: names the device, no guess from the rank)

    # which slab of the resident PCM every update since the engine's creation was fed (headline_parity replays its tail)
    fed = []
    keep = bool(args.keep)
    last_slab = [-1]

    def upd(u, out_ptr, keep_=None):
        """One update from resident slab u.  pe_update_device_keep needs the PREVIOUS call's slab untouched and distinct from this
        one (the engine refuses overlapping chunks): where two regions of this script meet on the same slab, that one call
        takes pe_update_device."""
        k = keep if keep_ is None else keep_
        engine.update_device(pcm_base + u * chunk_bytes, CHUNK, out_ptr, stream, keep=k and u != last_slab[0])
        last_slab[0] = u

    def run(first_step, n, out_rows):
        for i in range(n):
            u = (first_step + i) % n_res
            upd(u, probs_base + i * B * 4 if out_rows else scratch.data_ptr())

    # The launch the timed region uses (MFCC || GRU roles in one kernel): HIP events on the launch
    # stream bracketing a run of launches (per-launch events would add ~2.5 us of their own to a 22 us
    # kernel); the launches are back to back, so elapsed / n is the average launch duration.
    def bracket_pass(n):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for i in range(n):
            u = (warmup + steps + i) % n_res
            upd(u, scratch.data_ptr())
        ev1.record()
        ev1.synchronize()
        return ev0.elapsed_time(ev1) / n

    collective = {'collective': None, 'fallback_reason': None}
    if multi:
        # Warm the communicator with the timed gather's own shape -- and settle HERE, before anything is timed, which
        # collective the job uses: a gather to rank 0 (one send per peer over its own xGMI link), or, should this RCCL build
        # refuse it, an all-gather.  Every rank must take the same one: the outcome is agreed with an all-reduce (MIN).
        ok = 1
        try:
            gather_probabilities(probs.to(comm_device), n_global, dst=0, force_collective=force_dist)
        except Exception as ex:                               # noqa: BLE001
            ok = 0
            collective['fallback_reason'] = repr(ex)[:200]
        flag = torch.tensor([ok], dtype=torch.int32, device=comm_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        collective['collective'] = 'gather' if int(flag.item()) == 1 else 'all_gather'
        if collective['collective'] == 'all_gather':
            gather_probabilities(probs.to(comm_device), n_global, dst=None, force_collective=force_dist)
    gather_dst = 0 if collective['collective'] != 'all_gather' else None
    # ---- roofline pass FIRST, on every rank: `roofline.achieved` comes from here.  It also is what takes the GPU out of
    # idle: a 25-launch region entered from an idle GPU runs 8 % slower than the same region after >= 20 ms of
    # back-to-back launches (19.9 vs 18.3 us/step, tools/gpu_cold_start*.py), and the metric is sustained throughput.
    roofline_launches = args.roofline_launches
    fused_ms = bracket_pass(roofline_launches)          # wide networks: MFCC launch + network launch per update
    fed += [(warmup + steps + i) % n_res for i in range(roofline_launches)]

    # ---- warm-up: `warmup` untimed steps --------------------------------------------------------
    run(0, warmup, False)
    torch.cuda.synchronize()
    barrier()

    # ---- timed region: exactly `steps` steps + the final gather of their probabilities -------
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(warmup, steps, True)
    gathered = gather_probabilities(probs.to(comm_device), n_global, dst=gather_dst, force_collective=force_dist) if multi else probs      # rank 0 (all ranks after a fallback)
    wait_for_gpu()
    barrier()
    elapsed = time.perf_counter() - t0
    fed += [i % n_res for i in range(warmup)] + [(warmup + i) % n_res for i in range(steps)]
    rank_ms = [1e3 * elapsed / steps]
    ranks_seen = 1
    if multi:
        ranks_seen = dist.get_world_size()
        mine = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        every = [torch.zeros_like(mine) for _ in range(ranks_seen)]
        dist.all_gather(every, mine)             # every rank's own clock around the same region
        rank_ms = [1e3 * float(x.item()) / steps for x in every]
        elapsed = max(float(x.item()) for x in every)
    assert ranks_seen == world, 'the communicator reports %d ranks, the launcher %d' % (ranks_seen, world)
    if rank == 0:
        assert gathered.shape == (steps, n_global)
    finite = bool(torch.isfinite(gathered if rank == 0 else probs).all().item())
    if rank == 0 and os.environ.get('PE_BENCH_DUMP'):        # test aid: the timed region's probabilities, rank-ordered
        np.save(os.environ['PE_BENCH_DUMP'], gathered.cpu().numpy())
    # parity of the headline itself (rank 0's own shard: global streams 0 .. 255), from the timed region's own probabilities
    # (the oracle replay itself runs AFTER the GPU regions below: it keeps the host busy for seconds, the GPU would fall back to
    #  its idle clocks, and the two secondary regions -- 3 ms each -- would measure the clock ramp: 16.9 instead of 15.2 us per step)
    parity = None
    parity_inputs = (list(fed), probs.cpu().numpy()) if rank == 0 else None

    # ---- --gather-every-step: a second region, same K steps, step u's probabilities leaving while update u + 1 runs ----------
    per_step = None
    if args.gather_every_step != 'none':
        mode = args.gather_every_step
        side = torch.cuda.Stream(device=device)
        host_ring = torch.empty((steps, B), dtype=torch.float32).pin_memory()
        recv = None
        if mode == 'rccl' and multi and rank == 0:
            recv = torch.empty((world, steps, B), dtype=torch.float32, device=comm_device)
        # (the host work between the regions -- copies, the pinned allocation above -- lets the GPU fall back to its idle
        #  clocks, and a 3 ms region entered from there measures the clock ramp: 16.9 instead of 15.2 us per step.  The same
        #  untimed run of back-to-back launches that precedes the headline's region precedes this one)
        bracket_pass(roofline_launches)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        works, evs = [], []
        direct = mode == 'direct'
        ring_base = host_ring.data_ptr()
        for i in range(steps):
            u = (warmup + i) % n_res
            if direct:                                             # pinned host memory is device-visible: the kernel writes it
                upd(u, ring_base + i * B * 4)
                continue
            upd(u, probs_base + i * B * 4)
            if mode == 'host' or (mode == 'rccl' and shared_gpu) or not multi:
                ev = torch.cuda.Event()
                ev.record()                                       # behind update i on the launch stream
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    host_ring[i].copy_(probs[i], non_blocking=True)
                    done = torch.cuda.Event()
                    done.record()
                evs.append(done)
                if mode == 'rccl' and multi:                      # shared-GPU test path: gloo needs host tensors, so the host waits per step
                    done.synchronize()
                    dist.gather(host_ring[i], [recv[r][i] for r in range(world)] if rank == 0 else None, dst=0)
            else:
                # RCCL: torch's collective stream waits for the launch stream's work so far (update i) and runs the gather beside
                # update i + 1; async_op keeps the launch stream itself from waiting for it
                works.append(dist.gather(probs[i], [recv[r][i] for r in range(world)] if rank == 0 else None, dst=0, async_op=True))
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        barrier()
        dt = time.perf_counter() - t1
        fed += [(warmup + i) % n_res for i in range(steps)]
        if multi:
            mine = torch.tensor([dt], dtype=torch.float64, device=comm_device)
            dist.all_reduce(mine, op=dist.ReduceOp.MAX)
            dt = float(mine.item())
        delivered_ok = None
        if direct:
            # the last step again from the feature windows as they stand (pe_run_device: the network alone, nothing advances),
            # into device memory: must be the bits the last update wrote to the host ring; every other row must be finite
            engine.run_device(scratch.data_ptr(), stream)
            torch.cuda.synchronize()
            delivered_ok = bool(torch.equal(host_ring[steps - 1].to(device), scratch)) and bool(torch.isfinite(host_ring).all().item())
        elif rank == 0:
            if recv is not None:
                delivered_ok = bool(torch.equal(recv[0].to(device), probs))
            else:
                delivered_ok = bool(torch.equal(host_ring.to(device), probs))
        per_step = {'mode': mode if multi or mode in ('host', 'direct') else 'host (one rank: nothing to gather)',
                    'transport': ("the update's output pointer is a row of the rank's own pinned host ring: the network role's final store crosses PCIe "
                                  "(4 B x %d per step), no copy, no side stream" % B if direct else
                                  'gloo on host copies (PE_BENCH_SHARED_GPU=1: plumbing only, the host waits for every step)' if (mode == 'rccl' and shared_gpu and multi)
                                  else 'RCCL gather to rank 0 per step, asynchronous beside the next update' if (mode == 'rccl' and multi)
                                  else "each rank's own pinned host ring, one 4 B x %d copy per step on a side stream" % B),
                    'ms_per_step': 1e3 * dt / steps, 'value': n_global * steps / dt, 'unit': 'windows/s',
                    'final_gather_ms_per_step': 1e3 * elapsed / steps, 'delivered_equals_device': delivered_ok}

    # ---- the same K steps through pe_update_device (the leftover samples copied to the engine's carry in every call): the figure
    #      a caller gets whose chunks do NOT outlive the call -- reported beside the headline, N = 1 only
    carry_path = None
    if keep and world == 1:
        def plain(first, n):
            for i in range(n):
                upd((first + i) % n_res, scratch.data_ptr(), False)
        plain(warmup + steps, roofline_launches)         # (the first of them moves the kept leftovers to the carry; as many untimed launches as in front of the headline's region)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        plain(warmup + steps + roofline_launches, steps)
        wait_for_gpu()
        dt2 = time.perf_counter() - t2
        carry_path = {'entry_point': 'pe_update_device', 'value': B * steps / dt2, 'unit': 'windows/s', 'ms_per_step': 1e3 * dt2 / steps,
                      'note': 'same engine, same slabs, %d steps right after the timed region: every call copies the leftover samples into the '
                              "engine's carry (bit-identical results; tests/test_gpu_parity.py::test_kept_leftovers_*)" % steps}

    # ---- instrumented passes: HIP-event time per launch, on the launch stream --------------------
    def timed_pass(fused):
        engine.set_fused(fused)
        engine.set_timing(True)
        first, second = [], []
        for i in range(min(steps, 100)):
            u = (warmup + steps + i) % n_res
            upd(u, scratch.data_ptr())
            a, b = engine.last_timing()
            first.append(a)
            second.append(b)
        engine.set_timing(False)
        engine.set_fused(True)
        return float(np.mean(first)), float(np.mean(second))

    # the two roles as separate dependent launches, one HIP event pair per kernel (engine-side events,
    # same stream); each figure carries the event overhead
    mfcc_ms, gru_ms = timed_pass(False)

    # ---- extra (not the headline): the same updates issued 8 per call (pe_update_many_device) ----------
    # Three launches per 8 updates: every MFCC frame of the call, the per-stream bookkeeping, then the network for all 8 x B windows
    # at once.  Results are bit-identical to single updates; a caller pays 8 chunks of buffering latency.
    time_batched = None
    depth = 8
    if world == 1 and not args.no_batched and (stock or args.gru_precision == 'bf16') and n_res >= 2 * depth:      # N = 1 only: no collectives outside the timed region
        def batched_run(tiling):
            """8 updates per call on a fresh engine reserved for it (the engine picks ONE network form for all of its launches when it
            is reserved: engine.hip gru_args); tiling -1 = that automatic choice, otherwise forced."""
            eng = HipEngine(pr, weights, n_streams=B, device=dev_index, mfcc_precision=args.mfcc_precision,
                            gru_precision=args.gru_precision, ring_precision=args.ring_precision)
            try:
                eng.reserve_updates(depth, CHUNK)
                if tiling >= 0:
                    eng.set_gru_tiling(tiling)
                form = eng.gru_tiling()
                many_out = torch.zeros((depth, B), dtype=torch.float32, device=device)
                rounds = max(4, min(steps, 200) // depth)
                for i in range(4):
                    eng.update_many_device(pcm_base + ((i * depth) % max(1, n_res - depth)) * chunk_bytes, CHUNK, depth, many_out.data_ptr(), stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(rounds):
                    eng.update_many_device(pcm_base + ((i * depth) % max(1, n_res - depth)) * chunk_bytes, CHUNK, depth, many_out.data_ptr(), stream)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                # parity: the same 4 + rounds calls replayed by the oracle on 256 streams, last call's 8 x 256 probabilities
                from oracle import listener as oracle_listener
                n_chk = min(256, B)
                oracle = oracle_listener.BatchedOracle(weights, n_chk)
                host = pcm[:, :n_chk].cpu().numpy()
                want = None
                for i in list(range(4)) + list(range(rounds)):
                    first = (i * depth) % max(1, n_res - depth)
                    want = np.stack([oracle.update_raw(host[first + k]) for k in range(depth)])
                err = float(np.abs(want.astype(np.float64) - many_out[:, :n_chk].cpu().numpy().astype(np.float64)).max())
                tol_b = 1e-2 if args.gru_precision == 'bf16' else 1e-4
                return {'gru_form': form, 'value': n_global * rounds * depth / dt, 'unit': 'windows/s', 'ms_per_update': 1e3 * dt / (rounds * depth),
                        'parity': {'max_abs_err': err, 'tol': tol_b, 'ok': bool(err <= tol_b), 'streams_checked': n_chk}}
            finally:
                eng.close()
        try:
            auto = batched_run(-1)
            time_batched = dict(auto, updates_per_call=depth,
                                note='pe_update_many_device on an engine reserved for %d updates per call: same results as single updates, 2 launches per %d '
                                     'updates; the reserved engine picks its network form for the batched launch (gru_form: 1 = re-tiled f32-input MFMAs, '
                                     '2 = float32 products on the bf16 pipe); not the headline' % (depth, depth))
            if args.gru_precision == 'f32' and stock:
                other = 1 if auto['gru_form'] == 2 else 2
                time_batched['other_form'] = batched_run(other)
        except (ValueError, NotImplementedError) as ex:
            time_batched = {'error': repr(ex)}

    # ---- the other BASELINE.json configurations that fit one GPU (N = 1 only; each a few seconds) -------------------
    extras = []
    only = args.only_extra
    if world == 1 and rank == 0 and not args.no_extra_configs and stock and args.gru_precision == 'f32' and B == 4096:
        for cfg in (dict(name='configs[3]: wide GRU 256x2 fp32, batch=4096 synthetic 16 kHz streams on 1 MI355X', units=(256, 256), streams=4096,
                         mfcc_precision='f64', gru_precision='f32', ring_precision='f32', steps=100, warmup=40, n_res=64, tol=1e-4),
                    dict(name='configs[4] shard: bf16 MFCC rows + bf16 GRU (f32 front end), batch=8192 streams per MI355X (65536 / 8)', units=(20,), streams=8192,
                         mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16', steps=200, warmup=40, n_res=64, tol=1e-2),
                    dict(name='configs[4] on one GPU: bf16 MFCC rows + bf16 GRU (f32 front end), batch=65536 streams', units=(20,), streams=65536,
                         mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16', steps=100, warmup=40, n_res=32, tol=1e-2),
                    # the metric's second half ("max concurrent real-time streams") is a large-batch figure: the stock
                    # configuration (float64 MFCC as the reference computes it, float32 GRU) with the machine full.  At this
                    # size the engine takes the float32 network on the bf16 matrix pipe by itself (gru_x3_device.h: every
                    # operand as three bf16 pieces; the f32-input MFMAs keep their whole SIMD from issuing) ...
                    dict(name='capacity: stock GRU fp32 + f64 MFCC, batch=65536 streams on 1 MI355X (max concurrent real-time streams)', units=(20,), streams=65536,
                         mfcc_precision='f64', gru_precision='f32', ring_precision='f32', steps=100, warmup=40, n_res=32, tol=1e-4,
                         roofline_kind='mfma_update_x3'),
                    # ... the same point through pe_update_device_keep: the resident slabs outlive the calls, the leftover samples stay
                    # in them (no carry copy: ~200 of the ~390 bytes the MFCC launch writes per stream)
                    dict(name='capacity, leftovers kept in the resident chunks (pe_update_device_keep), batch=65536 streams', units=(20,), streams=65536,
                         mfcc_precision='f64', gru_precision='f32', ring_precision='f32', steps=100, warmup=40, n_res=32, tol=1e-4,
                         roofline_kind='mfma_update_x3', keep=True),
                    # ... and the same point forced onto the f32-input MFMAs (classic tiling, fused launch) for comparison
                    dict(name='capacity, classic tiling forced (pe_set_gru_tiling 0: v_mfma_f32_16x16x4_f32, fused launch), batch=65536 streams', units=(20,), streams=65536,
                         mfcc_precision='f64', gru_precision='f32', ring_precision='f32', steps=100, warmup=40, n_res=32, tol=1e-4,
                         roofline_kind='mfma_fused', gru_tiling=0),
                    # a non-stock .params file (params.py:28-118): the general front end (mfcc_general_device.h)
                    dict(name='general front end: n_fft=1024, n_filt=40, n_mfcc=20 (non-stock ListenerParams), stock-width GRU fp32, batch=4096 streams', units=(20,), streams=4096,
                         mfcc_precision='f64', gru_precision='f32', ring_precision='f32', steps=100, warmup=40, n_res=64, tol=1e-4,
                         params_kw=dict(n_fft=1024, n_filt=40, n_mfcc=20), roofline_kind='hbm_mfcc')):
            if only and only not in cfg['name']:
                continue
            try:
                extras.append(extra_config(device=device, dev_index=dev_index, **cfg))
            except Exception as ex:                                  # noqa: BLE001  (an extra must not cost the headline line)
                extras.append({'name': cfg['name'], 'error': repr(ex)})
        for label, fn in (('host-fed', lambda: host_fed_extra(device, dev_index)), ('configs[0]', single_stream_latency_extra)):
            if only and only not in label:
                continue
            try:
                extras.append(fn())
            except Exception as ex:                                  # noqa: BLE001
                extras.append({'name': label, 'error': repr(ex)})

    # the headline's parity object, last: every GPU region of this run is behind us
    if rank == 0:
        try:
            parity = headline_parity(pcm, parity_inputs[0], steps, parity_inputs[1], weights, 1e-2 if args.gru_precision == 'bf16' else 1e-4,
                                     n_check=256 if stock else 16)
        except Exception as ex:                                  # noqa: BLE001  (the checker must not cost the bench its line)
            parity = {'error': repr(ex)}

    def pmc_traffic(kernel):
        """HBM bytes per launch from the committed rocprofv3 PMC summary (bench.py cannot collect PMC
        counters itself); None when the profile is missing or was taken at another batch size."""
        try:
            with open(os.path.join(REPO, 'profiles', 'pmc_latest.json')) as f:
                pmc = json.load(f)
            if pmc.get('streams') != B:
                return None
            k = pmc[kernel]
            return (k['fetch_kb'] + k['write_kb']) * 1024.0
        except (OSError, KeyError, ValueError):
            return None

    if rank == 0:
        value = n_global * steps / elapsed
        def tflops(ms):
            return flop_per_window * B / (ms * 1e-3) / 1e12

        def gbs(ms):
            return MFCC_BYTES_PER_WINDOW * B / (ms * 1e-3) / 1e9

        mfcc_name = 'double' if args.mfcc_precision == 'f64' else 'float'
        mfma_peak = MFMA_F32_PEAK_TFLOPS if args.gru_precision == 'f32' else MFMA_BF16_PEAK_TFLOPS      # (float32 products on the bf16 pipe are priced as float32)
        # the engine's own rule (engine.hip: gru_args): stock width re-tiled while tiles <= 2 x CUs (or forced), four waves per
        # tile while tiles <= 2 x CUs on the re-tiled shapes / <= CUs on the classic tiling (or forced)
        n_cus = torch.cuda.get_device_properties(device).multi_processor_count
        tiles = (B + 15) // 16
        retiled = (tiles <= 2 * n_cus) if args.gru_tiling < 0 else args.gru_tiling == 1
        four_waves = (tiles <= (2 * n_cus if retiled else n_cus)) if not args.gru_waves else args.gru_waves == 4
        nopk = '_nopk' if mfcc_name == 'float' else ''        # kernels.hip: the float32 frame role's kernels are compiled without packed float32
        mfcc_kernel = 'mfcc_kernel%s<%s, ShapeStock, true>' % (nopk, mfcc_name)
        fused_name = ('fused_update_kernel%s<%s, ShapeStock, 5, %s, false, %s>' % (nopk, mfcc_name, 'true' if four_waves else 'false', 'true' if retiled else 'false')
                      if args.gru_precision == 'f32' else 'fused_update_bf16_kernel%s<%s, ShapeStock>' % (nopk, mfcc_name))
        gru_name = ((('gru_cw_kernel' if retiled else 'gru_mw_kernel<5, false>') if four_waves else
                     ('gru_v_kernel<1>' if retiled else 'gru_small_kernel<5, 1, false>')) if args.gru_precision == 'f32'
                    else 'gru_bf16_kernel<1>')
        x3 = args.gru_precision == 'f32' and stock and (args.gru_tiling == 2 or (args.gru_tiling < 0 and tiles > 4 * n_cus))
        if x3:
            fused_name = '%s then gru_x3_kernel<1> (two dependent launches)' % mfcc_kernel
            gru_name = 'gru_x3_kernel<1>'
        if not stock:
            fused_name = '%s + gru_wide_kernel<%d, 1, 4>' % (mfcc_kernel, (units[0] + 63) // 64)
            gru_name = 'gru_wide_kernel<%d, 1, 4>' % ((units[0] + 63) // 64)
        feat_bytes = 2 if args.ring_precision == 'bf16' else 4
        bytes_per_window = 2048 + 1.28 * 13 * feat_bytes       # PCM read + feature rows written (2114.6 / 2081.3 B)

        def gbs_w(ms):
            return bytes_per_window * B / (ms * 1e-3) / 1e9
        # BASELINE configs[4] (bf16 network): 21 G windows/s of MFMA headroom, so the HBM side of the MFCC stage is the
        # roofline that binds; every other configuration is priced against the matrix cores
        hbm_bound = args.gru_precision == 'bf16'
        line = {
            'metric': METRIC, 'value': value, 'unit': 'windows/s',
            'n_gpus': world, 'steps': steps, 'warmup': warmup,
            'ms_per_step': 1e3 * elapsed / steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': args.gru_precision, 'data': 'synthetic',
            'config': {'workload': '%s%s GRU (default ListenerParams) %s, batch=%d synthetic 16 kHz '
                                   'streams per MI355X%s, one 1024-sample chunk per stream per step'
                                   % ('BASELINE configs[1]: ' if (stock and world == 1 and B == 4096 and args.gru_precision == 'f32') else
                                      'BASELINE configs[2]: ' if (stock and world == 8 and B == 4096 and args.gru_precision == 'f32') else '',
                                      'stock' if stock else 'wide %s' % 'x'.join(map(str, units)), args.gru_precision, B,
                                      ' (batch=%d streams sharded across %d x MI355X, RCCL gather over xGMI)' % (n_global, world) if world > 1 else ''),
                       'streams_per_gpu': B, 'global_streams': n_global, 'chunk_samples': CHUNK,
                       'entry_point': 'pe_update_device_keep' if keep else 'pe_update_device',
                       'gru': 'H=%s, T=29, F=13, ' % args.units + ('f32 operands as 3 x bf16 pieces, 6 piece products on bf16 MFMA 16x16x32, f32 accumulate / gates / state' if x3 else 'f32 MFMA 16x16x4' if args.gru_precision == 'f32' else 'bf16 MFMA 16x16x32, f32 accumulate'),
                       'mfcc_dtype': args.mfcc_precision, 'feature_rows': args.ring_precision,
                       'parallelism': 'streams sharded over %d rank(s), final RCCL gather of probabilities to rank 0' % world},
            'realtime_streams': value / REALTIME_WINDOWS_PER_S,
            'outputs_finite': finite,
            'parity': parity,
            # what the communicator reported and every rank's own clock around the timed region (value uses the max)
            'sequence': 'roofline pass (%d back-to-back launches, HIP events) -> %d warm-up steps -> %d timed steps; a region entered '
                        'from an idle GPU measures ~8 %% slower (DESIGN.md 5)' % (roofline_launches, warmup, steps),
            'untimed_launches_before_timed_region': roofline_launches + warmup,
            'resident_pcm': {'slabs': n_res, 'mb': n_res * chunk_bytes / 1e6,
                             'note': 'distinct [B][1024] int16 slabs cycled by every pass; independent of --steps'},
            'ranks_seen': ranks_seen, 'rank_ms_per_step': {'min': min(rank_ms), 'max': max(rank_ms)},
            'collective_backend': (dist.get_backend() if multi else None),      # 'nccl' = RCCL over xGMI (one device per rank)
            'collective': collective['collective'], 'collective_fallback_reason': collective['fallback_reason'],
            'streams_per_rank': [B] * world,
            # dominant kernel of the timed region: the fused launch (GRU role is its long pole)
            'roofline': ({'kernel': fused_name, 'bound': 'hbm', 'achieved': gbs_w(fused_ms), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                          'frac': gbs_w(fused_ms) / HBM_PEAK_GBS, 'traffic': None, 'avg_launch_ms': fused_ms,
                          'algorithmic': '%.1f B/window x %d windows/launch' % (bytes_per_window, B)} if hbm_bound else
                         {'kernel': fused_name, 'bound': 'mfma',
                          'achieved': tflops(fused_ms), 'peak': mfma_peak, 'unit': 'TFLOP/s',
                          'frac': tflops(fused_ms) / mfma_peak,
                          'traffic': pmc_traffic('fused_update_kernel') if (args.gru_precision == 'f32' and stock) else None,
                          'traffic_unit': 'bytes/launch (rocprofv3 PMC of this command, committed as profiles/pmc_latest.json; not measured in this run)',
                          'avg_launch_ms': fused_ms,
                          'algorithmic': '%d flop/window x %d windows/launch' % (flop_per_window, B)}),
            'roofline_gru': {'kernel': gru_name, 'bound': 'mfma', 'achieved': tflops(gru_ms),
                             'peak': mfma_peak, 'unit': 'TFLOP/s',
                             'frac': tflops(gru_ms) / mfma_peak, 'traffic': pmc_traffic('gru_cw_kernel' if retiled else 'gru_mw_kernel') if (args.gru_precision == 'f32' and stock and four_waves) else None,
                             'avg_launch_ms': gru_ms},
            'roofline_mfcc': {'kernel': mfcc_kernel, 'bound': 'hbm',
                              'achieved': gbs_w(mfcc_ms), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                              'frac': gbs_w(mfcc_ms) / HBM_PEAK_GBS, 'traffic': pmc_traffic('mfcc_kernel'),
                              'avg_launch_ms': mfcc_ms,
                              'algorithmic': '%.1f B/window x %d windows/launch' % (bytes_per_window, B)},
        }
        try:     # spin / streaming microbenchmarks of the same pool ("also report against measured peaks")
            with open(os.path.join(REPO, 'profiles', 'measured_peaks.json')) as f:
                mp = json.load(f)
            m_mfma = mp['mfma_f32_16x16x4_tflops'] if args.gru_precision == 'f32' else mp['mfma_bf16_16x16x32_tflops']
            line['measured_peaks'] = {'mfma_tflops': m_mfma, 'hbm_read_gbs': mp['hbm_read_gbs'], 'source': 'profiles/measured_peaks.json',
                                      'roofline_frac': line['roofline']['achieved'] / (mp['hbm_read_gbs'] if hbm_bound else m_mfma),
                                      'roofline_gru_frac': line['roofline_gru']['achieved'] / m_mfma,
                                      'roofline_mfcc_frac': line['roofline_mfcc']['achieved'] / mp['hbm_read_gbs']}
        except (OSError, KeyError, ValueError):
            pass
        if extras:
            line['extra_configs'] = extras
        if time_batched is not None:
            line['time_batched'] = time_batched
        if per_step is not None:
            line['per_step_delivery'] = per_step
        if carry_path is not None:
            line['carry_copy_path'] = carry_path
        if cpu is not None:
            line['cpu_baseline'] = cpu
        if cpu_b1 is not None:
            line['cpu_baseline_single_stream'] = cpu_b1
        print(json.dumps(line), flush=True)

    engine.close()
    if multi:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

"""Position soak of the fused float32-frame launches (round 5; the gate for shipping the five-values bf16 network).

Every stream of an engine gets the SAME audio, so every position must produce the same bits:
  * after EVERY fused update the raw outputs are compared with stream 0 on the device (no host round trip);
  * every `--every` updates the whole feature window of every stream is read back (pe_get_vectors) and compared with
    stream 0's -- a frame stays in the window for 29 frames (~22 updates), so every frame is looked at.
A wrong float32 frame (profiles/round4/r4v_b20_fused_corruption.log: ~1 % off in 13 coefficients) shows up in both.

    [PE_LIB=...] [PE_B20=1] python tools/gpu_frame_soak.py --streams 8192 --frames 1e8 [--ring bf16] [--gru bf16] [--mfcc f32]
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr

ap = argparse.ArgumentParser()
ap.add_argument('--streams', type=int, default=8192)
ap.add_argument('--frames', type=float, default=1e7, help='float32 frames to compute (1.28 per stream and update)')
ap.add_argument('--mfcc', default='f32')
ap.add_argument('--gru', default='bf16')
ap.add_argument('--ring', default='bf16')
ap.add_argument('--every', type=int, default=16)
ap.add_argument('--tiling', type=int, default=None)
args = ap.parse_args()

B = args.streams
n_up = int(args.frames / (B * 1.28)) + 1
dev = torch.device('cuda', 0)
w = synth.make_weights()
kw = dict(mfcc_precision=args.mfcc, gru_precision=args.gru)
if args.gru == 'bf16':
    kw['ring_precision'] = args.ring
eng = _lib.HipEngine(pr, w, n_streams=B, **kw)
if args.tiling is not None:
    eng.set_gru_tiling(args.tiling)
n_res = 8
base = synth.batch_pcm(1, n_res)                                   # [n_res][1][1024]
pcm = torch.from_numpy(np.ascontiguousarray(base[:, 0, :])).to(dev)[:, None, :].expand(n_res, B, 1024).contiguous()
out = torch.zeros(B, device=dev)
bad_dev = torch.zeros((), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
bad_windows = 0
worst = 0.0
t0 = time.time()
for u in range(n_up):
    eng.update_device(pcm[u % n_res].data_ptr(), 1024, out.data_ptr(), st)
    bad_dev += (out != out[0]).sum()
    if (u + 1) % args.every == 0 or u == n_up - 1:
        torch.cuda.synchronize()
        feats = eng.get_vectors()
        diff = np.any(feats != feats[0], axis=(1, 2))
        nb = int(diff.sum())
        if nb:
            bad_windows += nb
            worst = max(worst, float(np.abs(feats[diff] - feats[0]).max()))
torch.cuda.synchronize()
print('%s network form %s mfcc=%s gru=%s ring=%s: %d streams x %d fused updates = %.3g frames: output positions that ever disagreed %d, '
      'feature windows that differed at a checkpoint %d (worst |delta| %.3g), %.1f s'
      % (os.path.basename(os.environ.get('PE_LIB', 'in-tree')), eng.gru_tiling(), args.mfcc, args.gru, args.ring, B, n_up,
         B * n_up * 1.28, int(bad_dev.item()), bad_windows, worst, time.time() - t0), flush=True)
eng.close()

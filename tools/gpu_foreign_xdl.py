"""(round 4) Do frames stay bit-reproducible while ANOTHER engine's XDL-heavy network launches run on a second stream?
Engine B (the one under test) gets the same audio on every stream, so its feature windows and probabilities must agree
across positions; engine A (float32 network on the bf16 pipe, 65 536 streams) keeps the machine's matrix pipes busy.
    python tools/gpu_foreign_xdl.py [streams of B]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
dev = torch.device('cuda', 0)
w = synth.make_weights()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
n_up = 12
base = synth.batch_pcm(1, n_up)
a = _lib.HipEngine(pr, w, n_streams=65536)
a.set_gru_tiling(2)
out_a = torch.zeros(65536, device=dev)
pcm_a = (torch.randn((4, 65536, 1024), device=dev) * 3000).to(torch.int16)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for i in range(35):
    a.update_device(pcm_a[i % 4].data_ptr(), 1024, out_a.data_ptr(), sa.cuda_stream)
torch.cuda.synchronize()
for name, kw in (('f32 frames + bf16 network (fused)', dict(mfcc_precision='f32', gru_precision='bf16', ring_precision='bf16')),
                 ('f32 frames + f32 network (fused)', dict(mfcc_precision='f32')),
                 ('f64 frames + f32 network (fused, the headline path)', dict())):
    for busy in (False, True):
        b = _lib.HipEngine(pr, w, n_streams=B, **kw)
        out_b = torch.zeros(B, device=dev)
        bad_p, bad_f = 0, 0
        for u in range(n_up):
            pcm_b = torch.from_numpy(np.repeat(base[u], B, axis=0)).to(dev)
            torch.cuda.synchronize()
            if busy:
                for i in range(6):
                    a.run_device(out_a.data_ptr(), sa.cuda_stream)          # ~0.3 ms of XDL-heavy launches on the other stream
            b.update_device(pcm_b.data_ptr(), 1024, out_b.data_ptr(), sb.cuda_stream)
            torch.cuda.synchronize()
            p = out_b.cpu().numpy()
            bad_p += int((p != p[0]).sum())
        feats = b.get_vectors()
        bad_f = int(np.any(feats != feats[0], axis=(1, 2)).sum())
        print('%-52s foreign XDL launches %-5s: positions whose probability differs (summed over %d updates) %d, feature windows that differ at the end %d'
              % (name, busy, n_up, bad_p, bad_f), flush=True)
        b.close()
a.close()

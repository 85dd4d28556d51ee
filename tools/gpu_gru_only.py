"""Network stage alone (pe_run_device on warmed-up feature windows) N times at one size, for rocprofv3 runs.
    python tools/gpu_gru_only.py <streams> [n] [waves: 0|1|4|16] [proj: -1|0|1]      (PE_GRU_PREC=f32|bf16, PE_GRU_TILING=0|1|2)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from mycroft_precise_amd import synth, _lib
from mycroft_precise_amd.params import pr
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
waves = int(sys.argv[3]) if len(sys.argv) > 3 else 0
proj = int(sys.argv[4]) if len(sys.argv) > 4 else -1
dev = torch.device('cuda', 0)
eng = _lib.HipEngine(pr, synth.make_weights(), n_streams=B, gru_precision=os.environ.get('PE_GRU_PREC', 'f32'))
if os.environ.get('PE_GRU_TILING'):
    eng.set_gru_tiling(int(os.environ['PE_GRU_TILING']))
if proj >= 0:
    eng.set_input_projection(bool(proj))
if waves:
    eng.set_gru_waves(waves)
pcm = (torch.randn((16, B, 1024), device=dev) * 3000).to(torch.int16)
out = torch.zeros(B, device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(40):
    eng.update_device(pcm[i % 16].data_ptr(), 1024, out.data_ptr(), st)
for i in range(n):
    eng.run_device(out.data_ptr(), st)
torch.cuda.synchronize()
eng.close()

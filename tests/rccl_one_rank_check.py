"""The RCCL calls of the N > 1 path on a ONE-GPU box: a process group of one rank over the `nccl` backend (= RCCL on ROCm),
and every collective bench.py / dist.py issue at N > 1, with their argument shapes, on cuda:0 -- communicator creation with
device_id, barrier(device_ids), the gather / all-gather of the timed region's probabilities through the product's own
gather_probabilities (force_collective), the MIN all-reduce that settles the collective, the all-gather of per-rank clocks,
and the asynchronous per-step gather.  One rank exchanges nothing over xGMI; what this pins is that this RCCL build accepts
the calls (run by tests/test_gpu_parity.py::test_rccl_calls_with_one_rank in a subprocess; prints 'ok ...')."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import torch.distributed as dist
from mycroft_precise_amd.dist import gather_probabilities


def main():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    dist.init_process_group('nccl', device_id=dev)
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    dist.barrier(device_ids=[0])
    steps, B = 20, 4096
    probs = torch.rand((steps, B), device=dev)
    got = gather_probabilities(probs, B, dst=0, force_collective=True)              # dist.gather on RCCL
    assert torch.equal(got, probs)
    got = gather_probabilities(probs, B, dst=None, force_collective=True)           # all_gather_into_tensor on RCCL
    assert torch.equal(got, probs)
    flag = torch.tensor([1], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)                                     # how the ranks agree on the collective
    assert int(flag.item()) == 1
    mine = torch.tensor([0.25], dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine)]
    dist.all_gather(every, mine)                                                    # every rank's own clock
    assert float(every[0].item()) == 0.25
    dist.all_reduce(mine, op=dist.ReduceOp.MAX)
    recv = torch.empty((1, steps, B), dtype=torch.float32, device=dev)
    works = [dist.gather(probs[i], [recv[0][i]], dst=0, async_op=True) for i in range(steps)]      # --gather-every-step rccl
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    assert torch.equal(recv[0], probs)
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()
    print('ok: nccl backend (RCCL), 1 rank: barrier, gather, all_gather_into_tensor, all_reduce MIN / MAX, all_gather, %d async gathers' % steps)


if __name__ == '__main__':
    main()

"""Probe (not product): what bench.py's closing barrier costs under torch.distributed (RCCL) on this box.
   python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/barrier_probe.py"""
import os, time, torch
import torch.distributed as dist
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
torch.cuda.set_device(dev)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("nccl", device_id=dev)
t = torch.zeros(1, device=dev)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6
print("dist.barrier() + synchronize: %.0f us" % timeit(lambda: (dist.barrier(), torch.cuda.synchronize())))
print("all_reduce(1 float) + synchronize: %.0f us" % timeit(lambda: (dist.all_reduce(t), torch.cuda.synchronize())))
print("synchronize alone: %.0f us" % timeit(lambda: torch.cuda.synchronize()))
dist.destroy_process_group()

"""One plain product (default 8192 x 8192 x 4096) launched a number of times: the thing tools/pmc_sq.sh / rocm-smi watch.
   python3 tools/gemm_loop.py [launches] [M N K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import _lib
L = _lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
M, N, K = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (8192, 8192, 4096)
A = torch.rand(M, K, device="cuda:0"); B = torch.rand(N, K, device="cuda:0"); C = torch.empty(M, N, device="cuda:0")
torch.cuda.synchronize()
t0 = time.time()
for _ in range(n):
    L.dspfft_gemm_nt_f32(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N, 1, 1, 0, 0, 0, 1.0, None)
torch.cuda.synchronize()
dt = time.time() - t0
print("launches", n, "ms each", 1e3 * dt / n, "TF", 2.0 * M * N * K * n / dt / 1e12)

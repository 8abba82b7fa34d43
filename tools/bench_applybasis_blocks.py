"""applybasis' partial sums with SMALL blocks (-P 8x8, 16x16: applybasis/applybasis.c:410-431) -- the case north_star names for the matrix cores: per
configuration the time of one call, the rate at which the result is written and the arithmetic rate, so that the bound can be named.
   python3 tools/bench_applybasis_blocks.py            (REPS=n for rocprofv3 runs)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dspfun_amd import _lib
from dspfun_amd.applybasis import partsums, FUNCTIONS
L = _lib.load()
REPS = int(os.environ.get("REPS", "10"))
# (image n x n, block P x P, terms K x K, function)
CASES = [(512, 8, 64, "dct2"), (512, 16, 64, "dct2"), (1024, 16, 64, "dct2"), (1024, 8, 32, "dct2"), (512, 8, 8, "dct2"), (512, 8, 64, "dft"), (256, 8, 256, "dct2")]
for (n, P, K, fn) in CASES:
    img = torch.rand(n, n, 3, device="cuda:0") * 2 - 1
    N = n // P
    func = FUNCTIONS.index(fn)
    out = torch.empty((K, K, N, N, 3, 2), dtype=torch.float32, device="cuda:0")
    work = torch.empty(L.dspfft_applybasis_work_floats(n, n, K, K, P, P, func), dtype=torch.float32, device="cuda:0")
    def call():
        rc = L.dspfft_applybasis_partsums(out.data_ptr(), img.data_ptr(), n, n, func, 1, K, K, P, P, 0, 0, work.data_ptr(), None)
        assert rc == 0, L.dspfft_zoom_last_error()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / REPS
    cplx = 2 if func <= 1 else 1
    out_bytes = out.numel() * 4
    flops = 2.0 * 3 * cplx * (K * n * n + cplx * K * (K * N) * n)          # step 1: (K x P)(P x n) per block column; step 2: (K x P)(P x K N) per block row
    print(json.dumps({"image": n, "block": P, "terms": K, "function": fn, "blocks": N * N, "ms": round(dt * 1e3, 4), "result_MB": round(out_bytes / 1e6, 1),
                      "result_GBps": round(out_bytes / dt / 1e9, 1), "TFLOPs": round(flops / dt / 1e12, 2)}))
    del out, work, img

"""CPU: the product C-ABI library loads and exports every symbol include/dspfft.h declares
(no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b((?:dspfft|fftw[fl]?)_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_match_binding_list():
    from dspfun_amd import _lib
    assert set(declared("dspfft.h")) == set(_lib.SYMBOLS)


def test_product_library_exports_every_declared_symbol():
    from dspfun_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(_lib.LIB_PATH)
    for name in declared("dspfft.h"):
        assert hasattr(lib, name), name
    assert b"gfx950" in C.c_char_p(C.cast(lib.dspfft_version, C.CFUNCTYPE(C.c_char_p))()).value


def test_product_library_exports_the_fftw_named_boundary():
    """every function include/fftw3.h declares (the FFTW functions the tools use, x {fftwf_, fftw_, fftwl_})"""
    from dspfun_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    names = declared("fftw3.h")
    assert len(names) == 36, names      # 12 functions x {fftwf_, fftw_, fftwl_}
    for name in names:
        assert hasattr(lib, name), name


def test_host_harnesses_build():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "host")])
    for exe in ("spec_gpu", "scan_gpu", "motion_gpu"):
        assert os.access(os.path.join(ROOT, "host", exe), os.X_OK)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dspfun_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_bench_and_smoke_fail_loudly_without_a_gpu():
    """no CPU path hides behind the entry points the driver runs: bench.py exits with an error, smoke() asserts"""
    import subprocess, sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True)
    assert r.returncode == 2 and "no GPU" in r.stderr, (r.returncode, r.stderr[-200:])
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    with pytest.raises(AssertionError, match="needs a GPU"):
        g.smoke()


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` called directly (no torchrun) starts two ranks as a child process; without a GPU each
    rank stops with "no GPU visible" and the parent returns non-zero"""
    import subprocess, sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    # both ranks start; torchrun stops the second as soon as the first has failed, so one or two of them get to say why
    assert 1 <= r.stderr.count("no GPU visible") <= 2 and "nproc_per_node" not in r.stderr[:0], r.stderr[-600:]
    assert "rank      : 0" in r.stderr or "local_rank: 0" in r.stderr or "exitcode" in r.stderr



def test_planning_effort_override_is_the_calling_threads_own_and_two_threads_plan_and_execute_side_by_side():
    """dspfft_set_plan_effort is process-wide (a plan made on any thread sees it: worker pools, Python threads -- ADVICE r05); the override
    dspfft_set_thread_plan_effort is the calling thread's own (the FFTW shim sets and restores it around each fftw(plan_many_r2r)): what one thread
    overrides another never sees, and two threads that plan and execute at the same time -- one of them flipping its override all the while -- both
    get the transform.  On the test-only emulation library (the engine's planner and pass logic are the product's; ctypes releases the GIL for the calls)."""
    import threading
    import numpy as np
    import oracle_lib as ol
    from emul_lib import emul
    from dspfun_amd.engine import Plan, REDFT10, REDFT01
    L = emul()
    L.dspfft_set_plan_effort.restype = None
    L.dspfft_set_thread_plan_effort.restype = None
    L.dspfft_set_plan_effort(3)
    assert L.dspfft_get_plan_effort() == 3 and L.dspfft_get_thread_plan_effort() == -1
    seen, errors = {}, []
    start = threading.Barrier(2)

    def worker(name, effort, h, w):
        try:
            assert L.dspfft_get_plan_effort() == 3 and L.dspfft_get_thread_plan_effort() == -1      # the process-wide value reaches a new thread
            start.wait()
            x = ol.synth_f32(hash(name) & 0xffff, h * w * 3).reshape(h, w, 3)
            ref = ol.dct2d_interleaved(x.astype(np.float64), REDFT10, impl="port", threads=1)
            for it in range(12):
                L.dspfft_set_thread_plan_effort(effort if it % 2 == 0 else effort + 1)
                fwd = Plan.image(h, w, 3, REDFT10, lib=L)
                inv = Plan.image(h, w, 3, REDFT01, lib=L).set_scale(1.0 / (4 * w * h))
                assert L.dspfft_get_plan_effort() == (effort if it % 2 == 0 else effort + 1)       # nobody else touched it
                d = x.copy()
                fwd.execute(d.ctypes.data)
                assert np.abs(d - ref).max() / np.abs(ref).max() < 2e-6
                inv.execute(d.ctypes.data)
                assert np.abs(d - x).max() < 5e-6
            seen[name] = L.dspfft_get_plan_effort()
            L.dspfft_set_thread_plan_effort(-1)
            assert L.dspfft_get_plan_effort() == 3
        except Exception as e:              # (an assertion in a thread would otherwise vanish)
            errors.append((name, repr(e)))

    ts = [threading.Thread(target=worker, args=("a", 0, 48, 64)), threading.Thread(target=worker, args=("b", 4, 60, 96))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    L.dspfft_set_plan_effort(0)
    assert not errors, errors
    assert seen == {"a": 1, "b": 5}
    assert L.dspfft_get_plan_effort() == 0 and L.dspfft_get_thread_plan_effort() == -1      # ... and the main thread never had an override

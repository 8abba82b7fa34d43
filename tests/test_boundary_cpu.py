"""CPU: completeness of the FFTW-named boundary.
  * fftwl_ (COEFF_PRECISION=L, reference include/precision.h:73-79) is a host-only long-double path of the product
    (dspfun_amd/csrc/fftwl_cpu.cpp): checked against the oracle's long-double definitions.
  * the names the reference's link lines ask for (libfftw3f.so, libfftw3f_threads.so, ... and fftw3f.pc) are build outputs of
    dspfun_amd/csrc/Makefile; a translation unit that includes the REFERENCE's own include/precision.h with
    COEFF_PRECISION=F|D|L and <fftw3.h> from include/ compiles and links with `-lfftw3f -lfftw3f_threads` etc.
    (skipped when /root/reference is absent)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dspfun_amd", "csrc")
REF = "/root/reference"


@pytest.fixture(scope="module")
def lib():
    from dspfun_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    L = C.CDLL(_lib.LIB_PATH)
    ip = C.POINTER(C.c_int)
    L.fftwl_alloc_real.restype = C.c_void_p
    L.fftwl_alloc_real.argtypes = [C.c_size_t]
    L.fftwl_free.argtypes = [C.c_void_p]
    L.fftwl_plan_many_r2r.restype = C.c_void_p
    L.fftwl_plan_many_r2r.argtypes = [C.c_int, ip, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, C.c_void_p, ip, C.c_int, C.c_int, ip, C.c_uint]
    L.fftwl_execute.argtypes = [C.c_void_p]
    L.fftwl_destroy_plan.argtypes = [C.c_void_p]
    return L


def ia(v):
    return None if v is None else (C.c_int * len(v))(*v)


@pytest.mark.parametrize("n,howmany,inembed,istride,idist,onembed,ostride,odist,kinds,inplace", [
    ([12, 10], 3, None, 3, 1, None, 3, 1, [5, 5], True),          # spec.c:63: interleaved image, in place
    ([12, 10], 3, None, 3, 1, None, 3, 1, [4, 4], False),         # scan.c:359: out of place
    ([4, 6, 5], 1, [6, 8, 7], 1, 0, [6, 8, 7], 1, 0, [5, 5, 5], True),   # motion.c:535: block embedded in a larger buffer
    ([7, 13], 2, None, 2, 1, None, 2, 1, [5, 4], True),           # primes, mixed kinds
    ([17], 4, None, 1, 17, None, 1, 17, [4], False),
])
def test_fftwl_against_the_long_double_definitions(lib, n, howmany, inembed, istride, idist, onembed, ostride, odist, kinds, inplace):
    def span(embed, stride, dist):
        idx = 0
        for a in range(len(n)):
            idx = idx * (embed[a] if embed else n[a]) + (n[a] - 1)
        return idx * stride + (howmany - 1) * dist + 1
    ilen, olen = span(inembed, istride, idist), span(onembed, ostride, odist)
    x = ol.synth_f32(99 + len(n), max(ilen, olen)).astype(np.float64) - 0.25
    ref = ol.r2r_many(x[:ilen], n, kinds, howmany, inembed, istride, idist, onembed, ostride, odist, out=x[:olen].copy(), impl="direct")
    # numpy owns the long double arrays (x86-64: 80-bit extended in 16 bytes, the C `long double`); the plan takes any host pointer
    a = x.astype(np.longdouble)
    b = a if inplace else x[:olen].astype(np.longdouble)      # out of place: elements the transform does not write must survive
    plan = lib.fftwl_plan_many_r2r(len(n), ia(n), howmany, a.ctypes.data, ia(inembed), istride, idist, b.ctypes.data, ia(onembed), ostride, odist, ia(kinds), 1 << 6)
    assert plan
    keep = a.copy()
    lib.fftwl_execute(plan)
    got = b[:olen].astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max())
    if not inplace:
        assert np.array_equal(a, keep)                 # the input is not touched
        lib.fftwl_execute(plan)                        # repeatable (scan.c:447)
        assert np.abs(b[:olen].astype(np.float64) - ref).max() <= 1e-14 * max(1.0, np.abs(ref).max())
    lib.fftwl_destroy_plan(plan)
    p = lib.fftwl_alloc_real(100)
    assert p and p % 16 == 0
    lib.fftwl_free(p)


def test_link_aliases_and_pkgconfig_files_are_build_outputs():
    subprocess.check_call(["make", "-s", "-C", CSRC, "aliases"])
    for name in ("libfftw3f.so", "libfftw3f_threads.so", "libfftw3.so", "libfftw3_threads.so", "libfftw3l.so", "libfftw3l_threads.so"):
        p = os.path.join(CSRC, name)
        assert os.path.islink(p) and os.path.realpath(p) == os.path.join(CSRC, "libdspfft_hip.so"), name
    for pc in ("fftw3f", "fftw3", "fftw3l"):
        txt = open(os.path.join(CSRC, "pkgconfig", pc + ".pc")).read()
        assert f"-l{pc}" in txt and "Cflags: -I${prefix}/include" in txt and f"prefix={ROOT}" in txt


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "include", "precision.h")), reason="reference tree absent")
@pytest.mark.parametrize("prec,inter,fftw", [("F", "D", "fftw3f"), ("D", "L", "fftw3"), ("L", "L", "fftw3l")])      # scan/Makefile:1-2, spec/Makefile:1-2
def test_reference_precision_header_compiles_and_links_against_the_engine(tmp_path, prec, inter, fftw):
    """the reference's OWN include/precision.h (typedefs :102-105, fftw(call) :115) + our include/fftw3.h; link line as in
    scan/Makefile:4,12 (`pkg-config --libs` + -l<fftw>_threads -lpthread).  Link only: executing needs the GPU."""
    subprocess.check_call(["make", "-s", "-C", CSRC, "aliases"])
    src = tmp_path / "tu.c"
    src.write_text(
        '#include <stdlib.h>\n#include <fftw3.h>\n#include "precision.h"\n'
        "int main(int argc, char **argv) {\n"
        "  int n[2] = {4, 6};\n"
        "  coeff *c = fftw(alloc_real)(4 * 6 * 3);                              /* spec/spec.c:59 */\n"
        "  fftw(init_threads)(); fftw(plan_with_nthreads)(2);                      /* scan/scan.c:289-290 */\n"
        "  fftw(plan) p = fftw(plan_many_r2r)(2, n, 3, c, NULL, 3, 1, c, NULL, 3, 1, (fftw(r2r_kind)[]){FFTW_REDFT10, FFTW_REDFT10}, FFTW_ESTIMATE);\n"
        "  fftw_r2r_kind k2 = FFTW_REDFT01; (void)k2;                              /* spec/spec.c:63 spells it without the precision */\n"
        "  if (argc > 100) { fftw(execute)(p); }\n"
        "  fftw(destroy_plan)(p); fftw(free)(c); fftw(cleanup)(); fftw(cleanup_threads)();\n"
        "  intermediate x = mi(sqrt)(2); (void)x;\n"
        "  return 0;\n}\n")
    exe = tmp_path / "tu"
    cmd = ["gcc", "-std=c11", "-D_GNU_SOURCE", f"-DCOEFF_PRECISION={prec}", f"-DINTERMEDIATE_PRECISION={inter}", f"-I{ROOT}/include", f"-I{REF}/include", str(src), "-o", str(exe),
           f"-L{CSRC}", f"-Wl,-rpath,{CSRC}", f"-l{fftw}", f"-l{fftw}_threads", "-lpthread", "-lm",
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    out = subprocess.check_output(["nm", "-u", str(exe)]).decode()
    prefix = {"F": "fftwf_", "D": "fftw_", "L": "fftwl_"}[prec]
    assert f"{prefix}plan_many_r2r" in out and f"{prefix}alloc_real" in out

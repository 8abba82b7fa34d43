"""TEST-ONLY: loads tests/emul/libdspfft_emul.so (CPU emulation of the kernel phases)."""
import ctypes as C
import os
import subprocess

from dspfun_amd import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_emul = None


def emul():
    global _emul
    if _emul is None:
        # one build at a time: pytest-xdist workers would otherwise race on the same output file
        import fcntl
        with open(os.path.join(_HERE, "emul", ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "emul")], stdout=subprocess.DEVNULL)
        _emul = L.bind(C.CDLL(os.path.join(_HERE, "emul", "libdspfft_emul.so")))
    return _emul

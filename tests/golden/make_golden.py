"""Regenerates tests/golden/*.npz and *.json.  Run HERE (needs scipy); never on the GPU box.

The DCT fixtures are NOT reference outputs (FFTW is not installable here and the reference
holds no vectors): they come from scipy.fft (pocketfft), an independent implementation of the
same published REDFT10/REDFT01 definitions, evaluated in f64.  Inputs follow SURVEY.md 8(d):
splitmix64(seed) -> f32 (u>>40)*2^-24.

The scan fixtures ARE reference data: the two 8x8 `diagonal` listings are the text of
scan/README.md:121-129 and :136-150, and the zigzag FNV hashes were recorded from the
compiled reference by the survey.
"""
import json
import os
import sys

import numpy as np
import scipy.fft as sf

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import synth_f32, synth_u8  # noqa: E402


def main():
    out = {}
    # 2-D interleaved (spec.c:63 / ispec.c:165 / scan.c:292,359 plan shape), mixed-radix sizes
    for i, (h, w, c) in enumerate([(48, 64, 3), (15, 27, 3), (30, 45, 3), (60, 135, 1), (16, 16, 3), (7, 13, 2), (1, 8, 3), (8, 1, 3)]):
        x = synth_f32(0xD5F0100 + i, h * w * c).reshape(h, w, c)
        out[f"img{i}_in"] = x
        out[f"img{i}_redft10"] = sf.dctn(x.astype(np.float64), type=2, axes=(0, 1))
        out[f"img{i}_redft01"] = sf.dctn(x.astype(np.float64), type=3, axes=(0, 1))
    # 1-D lengths incl. 270 and primes
    for N in (2, 3, 4, 5, 6, 8, 9, 10, 12, 15, 16, 17, 27, 30, 31, 45, 60, 64, 97, 135, 270, 540):
        x = synth_f32(0xD5F0200 + N, N)
        out[f"vec{N}_in"] = x
        out[f"vec{N}_redft10"] = sf.dct(x.astype(np.float64), type=2)
        out[f"vec{N}_redft01"] = sf.dct(x.astype(np.float64), type=3)
    # 3-D planar embedded (motion.c:535-552): block {8,12,10} inside minbuf {10,12,16}
    d, h, w, md, mh, mw = 8, 12, 10, 10, 12, 16
    buf = synth_u8(0xD5F0300, md * mh * mw).astype(np.float64).reshape(md, mh, mw)
    out["vol_in"] = buf
    o2 = buf.copy(); o2[:d, :h, :w] = sf.dctn(buf[:d, :h, :w], type=2)
    o3 = buf.copy(); o3[:d, :h, :w] = sf.dctn(buf[:d, :h, :w], type=3)
    out["vol_redft10"] = o2
    out["vol_redft01"] = o3
    out["vol_dims"] = np.array([d, h, w, md, mh, mw])
    np.savez_compressed(os.path.join(HERE, "dct_golden.npz"), **out)

    scan = {
        "source": "SURVEY.md 8c (hashes recorded from the compiled reference scan_methods.c); "
                  "FNV-1a-64 word variant: h ^= (y*w+x); h *= 1099511628211; offset basis 1469598103934665603",
        "zigzag_fnv": {
            "8x8": "3429f64e9a8101d3", "6x4": "6dd8126560949135", "4x6": "cd8f55d6751398bd",
            "16x9": "6c07653352e593cb", "9x16": "47e101f5ab15f253", "256x256": "615fed655143fc83",
            "1920x1080": "3d6530a0743c1063", "1080x1920": "4d67c11c18b4c7b3", "7680x4320": "5107222c372da523",
        },
        "diagonal_8x8_index": [  # scan/README.md:121-129
            " 0  1  2  3  4  5  6  7", " 1  2  3  4  5  6  7  8", " 2  3  4  5  6  7  8  9", " 3  4  5  6  7  8  9 10",
            " 4  5  6  7  8  9 10 11", " 5  6  7  8  9 10 11 12", " 6  7  8  9 10 11 12 13", " 7  8  9 10 11 12 13 14"],
        "diagonal_8x8_coordinate": [  # scan/README.md:136-150 (x,y pairs)
            "0,0", "0,1 1,0", "0,2 1,1 2,0", "0,3 1,2 2,1 3,0", "0,4 1,3 2,2 3,1 4,0", "0,5 1,4 2,3 3,2 4,1 5,0",
            "0,6 1,5 2,4 3,3 4,2 5,1 6,0", "0,7 1,6 2,5 3,4 4,3 5,2 6,1 7,0", "1,7 2,6 3,5 4,4 5,3 6,2 7,1",
            "2,7 3,6 4,5 5,4 6,3 7,2", "3,7 4,6 5,5 6,4 7,3", "4,7 5,6 6,5 7,4", "5,7 6,6 7,5", "6,7 7,6", "7,7"],
    }
    with open(os.path.join(HERE, "scan_golden.json"), "w") as f:
        json.dump(scan, f, indent=1)
    print("wrote", os.path.join(HERE, "dct_golden.npz"), os.path.getsize(os.path.join(HERE, "dct_golden.npz")))


if __name__ == "__main__":
    main()

"""TEST INFRASTRUCTURE ONLY: f64 restatement of one motion block with scaled != block (motion/motion.c:535-552 plans, :559-572
constants, :617-647 load + forward + uniform range, :652-668 top-N, :740-753 quantiser + inverse, :755-776 output), composed from
the oracle's pieces (oracle/dct_oracle.c via r2r_many with embedding, oracle_motion_uniform_f64)."""
import numpy as np

import oracle_lib as ol


def consts(block, scaled):
    nb, ns = float(np.prod(block)), float(np.prod(scaled))
    scalefactor = ns / nb                                  # :566
    normalization = 1.0 / np.sqrt(ns * 8)                  # :567
    return scalefactor, normalization


def block_roundtrip(pix_u8, block, scaled, minbuf, quant=0.0, topn=0, spec="none", impl="port"):
    """pix_u8: uint8 array of shape minbuf (only the block corner is read).  Returns (uint8 output of shape minbuf with the scaled
    corner written, uniform-range coefficients after the filters as float64 of shape minbuf)."""
    bd, bh, bw = block
    sd, sh, sw = scaled
    md, mh, mw = minbuf
    ad, ah, aw = min(bd, sd), min(bh, sh), min(bw, sw)
    scalefactor, normalization = consts(block, scaled)
    c = np.zeros(minbuf, dtype=np.float64)                  # :619
    c[:bd, :bh, :bw] = pix_u8[:bd, :bh, :bw]               # :620-638
    c = ol.r2r_many(c, list(block), [ol.REDFT10] * 3, inembed=list(minbuf), onembed=list(minbuf), impl=impl).reshape(minbuf)   # :641
    ol.lib().oracle_motion_uniform_f64(c.ctypes.data, ad, ah, aw, mh, mw, 1)     # :644-647
    dc = c[0, 0, 0]
    if topn:                                                # :652-668 over the whole buffer; ties: earliest in buffer order
        flat = c.ravel()
        order = np.argsort(-np.abs(flat), kind="stable")
        keep = np.zeros(flat.size, dtype=bool)
        keep[order[:topn]] = True
        flat[~keep] = 0.0
    if quant:                                               # :570,740-744
        q = quant * 8 * np.sqrt(float(np.prod(scaled)))
        a = c[:ad, :ah, :aw]
        a[...] = np.round(a / q) * q
    coeffs = c.copy()
    out = np.zeros(minbuf, dtype=np.uint8)
    if spec == "none":
        ol.lib().oracle_motion_uniform_f64(c.ctypes.data, ad, ah, aw, mh, mw, -1)    # :748-751
        c = ol.r2r_many(c, list(scaled), [ol.REDFT01] * 3, inembed=list(minbuf), onembed=list(minbuf), impl=impl).reshape(minbuf)   # :753
    pel = c[:sd, :sh, :sw] * scalefactor * normalization   # :759
    ns = float(np.prod(scaled))
    if spec == "abs":
        cc = 255.0 / np.log1p(abs(dc * scalefactor * normalization))                # :755
        pel = cc * np.log1p(np.abs(pel))
    elif spec == "shift":
        cc = 127.5 / np.log1p(ns * normalization * 255 * 8)                          # :568
        pel = cc * np.copysign(np.log1p(np.abs(pel)), pel) + 127.5
    elif spec == "flat":
        cc = 0.0
        pel = pel * normalization / 2 + 127.5
    else:
        cc = 0.0
        pel = pel * normalization
    r = np.where(pel >= 0, np.floor(pel + 0.5), -np.floor(-pel + 0.5))              # lround: halves away from zero
    out[:sd, :sh, :sw] = np.clip(r, 0, 255).astype(np.uint8)
    return out, coeffs, cc

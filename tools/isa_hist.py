#!/usr/bin/env python3
"""Instruction histogram of gfx950 kernels from `hipcc -S --offload-device-only` output.

    hipcc <flags of csrc/Makefile> -S --offload-device-only spec_inst_split.hip -o /tmp/split.s
    python3 tools/isa_hist.py /tmp/split.s 'row_pair_kernel.*7680ELi3' [--phases]

Counts are STATIC (per thread and per pass through the text; a loop body counts once -- row_pair_kernel's two-line loop
therefore shows the instructions of ONE line).  Classes:
  fma/mul/add   v_fma / v_fmac / v_mul_f / v_add_f / v_sub_f / v_pk_{fma,mul,add}_f32   (the butterflies and twiddles)
  mov           v_mov / v_accvgpr / v_swap / v_perm / v_readlane ... (register shuffling)
  int           v_add_u32 / v_mad_u / v_lshl / v_and / v_mul_lo / v_mul_hi / v_sub_u / v_add_co / v_lshl_add ... (index arithmetic)
  sel           v_cndmask / v_cmp*                                                     (predication)
  cvt           v_cvt*
  lds           ds_*
  vmem          global_* / buffer_* / flat_* / scratch_*
  salu          s_* except s_waitcnt / s_barrier / s_nop
  wait          s_waitcnt / s_nop / s_barrier
--phases splits the text at every s_barrier.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("ds_",)):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op in ("s_waitcnt", "s_nop", "s_barrier") or op.startswith("s_waitcnt") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        if re.match(r"v_(pk_)?(fma|fmac|mul|add|sub|mad|mac|subrev|max|min|rcp|rsq|sqrt|floor|fract|rndne|trunc|ceil|ldexp|div)_(f|legacy_f|dx9)", op) or re.match(r"v_(pk_)?(fma|fmac)_", op) or op.startswith("v_dot"):
            return "fma/mul/add"
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            return "mfma"
        if op.startswith(("v_mov", "v_accvgpr", "v_swap", "v_perm", "v_readlane", "v_readfirstlane", "v_writelane", "v_bfi", "v_alignbit", "v_pk_mov")):
            return "mov"
        if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")):
            return "sel"
        if op.startswith("v_cvt"):
            return "cvt"
        return "int"
    return "other"


def kernels(path):
    """yield (name, [ops])"""
    name, ops = None, []
    lab = re.compile(r"^(_Z[\w$.]+|[A-Za-z_][\w$.]*):\s*(;.*)?$")
    for line in open(path):
        m = lab.match(line)
        if m and not line.startswith(".L"):
            if name is not None:
                yield name, ops
            name, ops = m.group(1), []
            continue
        if name is None:
            continue
        s = line.strip()
        if not s or s[0] in ".;" or s.endswith(":"):
            if s.startswith(".Lfunc_end"):
                yield name, ops
                name, ops = None, []
            continue
        op = s.split()[0]
        if op in ("s_endpgm", "s_code_end"):
            if op == "s_endpgm":
                ops.append(op)
            continue
        ops.append(op)
    if name is not None:
        yield name, ops


def main():
    path, pat = sys.argv[1], re.compile(sys.argv[2])
    phases = "--phases" in sys.argv
    order = ["fma/mul/add", "mfma", "mov", "int", "sel", "cvt", "lds", "vmem", "salu", "wait", "other"]
    for name, ops in kernels(path):
        if not pat.search(name) or not ops:
            continue
        print(f"== {name}")
        segs = [[]]
        for op in ops:
            segs[-1].append(op)
            if phases and op == "s_barrier":
                segs.append([])
        tot = collections.Counter()
        rows = []
        for seg in segs:
            c = collections.Counter(classify(o) for o in seg)
            tot.update(c)
            rows.append(c)
        hdr = "  seg " + "".join(f"{k:>12}" for k in order) + f"{'VALU':>8}{'all':>8}"
        print(hdr)
        valu_keys = ("fma/mul/add", "mfma", "mov", "int", "sel", "cvt")
        for i, c in enumerate(rows + [tot]):
            if not phases and i < len(rows):
                continue
            lab = "total" if i == len(rows) else f"{i:5d}"
            print(f"{lab:>5} " + "".join(f"{c.get(k, 0):12d}" for k in order) + f"{sum(c.get(k, 0) for k in valu_keys):8d}{sum(c.values()):8d}")
        v = sum(tot.get(k, 0) for k in valu_keys)
        if v:
            print("  VALU share: " + ", ".join(f"{k} {100.0 * tot.get(k, 0) / v:.0f}%" for k in valu_keys if tot.get(k, 0)))
        top = collections.Counter(o for o in ops if classify(o) in ("mov", "int", "sel", "cvt")).most_common(12)
        print("  top non-arithmetic VALU: " + ", ".join(f"{o} {n}" for o, n in top))


if __name__ == "__main__":
    main()

"""CPU (emulation backend): dspfft_execute_many_repeat -- the frame loop of a clip inside the library (bench.py's step loop) --
and the argument checks of the per-pass events (timed windows that run past the batch, one-pass block plans)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol
from dspfun_amd.engine import Plan, Batch, Events, DspfftError, REDFT10, REDFT01
from emul_lib import emul


def make(nframes, h=24, w=32, c=3):
    L = emul()
    fwd = Plan.image(h, w, c, REDFT10, lib=L)
    inv = Plan.image(h, w, c, REDFT01, lib=L).set_scale(1.0 / (4 * w * h))
    frames = [ol.synth_f32(77 + f, h * w * c).reshape(h, w, c).copy() for f in range(nframes)]
    # two "streams" (the emulation ignores them, the library still has to tell them apart)
    batch = Batch([(pl, fr.ctypes.data, None, 1 + (i % 2)) for i, fr in enumerate(frames) for pl in (fwd, inv)], lib=L)
    return L, fwd, inv, frames, batch


def test_repeats_are_roundtrips_and_forward_only_batches_compose():
    L, fwd, inv, frames, batch = make(4)
    ref = [f.copy() for f in frames]
    batch.run_repeat(5, rejoin_every=2)
    for f, r in zip(frames, ref):
        assert np.abs(f - r).max() <= 2e-5            # five in-place roundtrips
    # a batch of forward plans only: two repeats = the transform applied twice
    x = ref[0].copy()
    b2 = Batch([(fwd, x.ctypes.data, None, 0)], lib=L)
    b2.run_repeat(2)
    want = ol.dct2d_interleaved(ol.dct2d_interleaved(ref[0].astype(np.float64), REDFT10, impl="port"), REDFT10, impl="port")
    assert np.abs(x - want).max() <= 2e-6 * np.abs(want).max()
    b2.run_repeat(0)                                   # zero repeats: nothing happens
    assert np.abs(x - want).max() <= 2e-6 * np.abs(want).max()


def test_event_windows_rotate_and_are_checked():
    L, fwd, inv, frames, batch = make(4)
    npass = fwd.num_passes + inv.num_passes
    steps, every, window = 6, 2, 4                      # windows on steps 0, 2, 4: items 0-3, 4-7, 0-3
    ev = Events(2 * npass * 2 * 3, lib=L)
    batch.run_repeat(steps, 8, every, window, ev)
    assert ev.elapsed_ms(0, 1) == 0.0                   # the emulation's events carry no time; the call must accept them
    # a window that does not divide the batch (3 frames, two bracketed per step: ADVICE r2 bench.py:229) is refused, not overrun
    L3, _, _, _, batch3 = make(3)
    with pytest.raises(DspfftError, match="must divide"):
        batch3.run_repeat(4, 0, 1, 4, Events(64, lib=L3))
    # dspfft_execute_many: a timed window past the end of the batch
    with pytest.raises(DspfftError, match="outside the batch"):
        batch3.run(timed_item=4, timed_count=4, events=Events(64, lib=L3))


def test_block_plans_cannot_be_bracketed_per_pass():
    """a one-pass small-block plan runs ONE fused kernel in dspfft_execute: bracketing its axis passes would time other kernels"""
    L = emul()
    blk = Plan.guru([(8, 64 * 64, 64 * 64), (8, 64, 64), (8, 1, 1)], [(2, 8 * 64 * 64, 8 * 64 * 64), (8, 8 * 64, 8 * 64), (8, 8, 8)], [REDFT10] * 3, lib=L)
    if "BLOCK" not in blk.describe():
        pytest.skip("no one-pass block plan for this shape in this build")
    x = np.zeros(16 * 64 * 64, dtype=np.float32)
    b = Batch([(blk, x.ctypes.data, None, 0)], lib=L)
    b.run()                                             # untimed: fine
    with pytest.raises(DspfftError, match="block plans"):
        b.run(timed_item=0, timed_count=1, events=Events(16, lib=L))


def test_double_and_float_plans_share_a_repeated_batch():
    """dspfft_execute_many_repeat takes each item's buffers as float or double according to its plan (dspfft_execute_many stays f32 only)"""
    L = emul()
    h, w, c = 20, 28, 3
    f32 = ol.synth_f32(5, h * w * c).reshape(h, w, c).copy()
    f64 = ol.synth_f32(6, h * w * c).reshape(h, w, c).astype(np.float64)
    ref32, ref64 = f32.copy(), f64.copy()
    pl32 = Plan.image(h, w, c, REDFT10, lib=L)
    pl64f = Plan.image(h, w, c, REDFT10, dtype="f64", lib=L)
    pl64i = Plan.image(h, w, c, REDFT01, dtype="f64", lib=L).set_scale(1.0 / (4 * w * h))
    b = Batch([(pl64f, f64.ctypes.data, None, 1), (pl32, f32.ctypes.data, None, 2), (pl64i, f64.ctypes.data, None, 1)], lib=L)
    ev = Events(2 * (pl64f.num_passes + pl32.num_passes + pl64i.num_passes), lib=L)
    b.run_repeat(1, 0, 1, 3, ev)                      # one repeat, every pass bracketed
    assert np.abs(f64 - ref64).max() <= 1e-13        # double roundtrip
    want = ol.dct2d_interleaved(ref32.astype(np.float64), REDFT10, impl="port")
    assert np.abs(f32 - want).max() <= 2e-6 * np.abs(want).max()
    with pytest.raises(DspfftError, match="f32 plans"):
        b.run()

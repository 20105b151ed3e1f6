"""Where the small-matrix step of a lone window spends its time: the diagnostic library of tools/small_stamp.sh leaves the 100 MHz
wall clock at the phase boundaries of k_ialm_small (window 0) for every iteration; prints microseconds per phase, and the gap between
the end of one launch and the start of the next (pass + slab sum + launch latencies).  Usage: python3 tools/small_stamp.py [n] [Hc Wc]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from swiftwatcher_amd import _lib, synthetic          # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "tools", "libswk_stamp.so")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
Hc, Wc = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (212, 424)
nwin = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = _lib.Context(0)
roi = np.concatenate([synthetic.roi_window(5 + w, n, Hc, Wc, birds=12) for w in range(nwin)])
raw = ctypes.CDLL(_lib.LIB_PATH)
names = ["prologue (state, ||Z|| slabs, decision)", "zero Y/Z + Gram slab -> LDS", "||G||_F, dead rows, scaling", "Newton-Schulz loop",
         "||Z||_F check", "B = I - W/mu -> global"]
for rep in range(3):
    out = ctx.batch_run(roi, nwin, n, stages=("rpca",))
buf = np.zeros((64, 10), np.int64)
assert raw.swk_small_stamp_read(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
iters = int(out["iters"][0])
print("n = %d, %dx%d, %d window(s): %d iterations, Newton-Schulz steps in the last: %d" % (n, Hc, Wc, nwin, iters, ctx.last_eig_sweeps))
t = buf[:iters].astype(np.float64) / 100.0          # microseconds
ph = np.diff(t[:, :7], axis=1)          # 6 phases
for i, nm in enumerate(names):
    print("   %-45s %6.2f us (k = 1 .. %d mean; k = 0: %.2f)" % (nm, ph[1:, i].mean(), iters - 1, ph[0, i]))
cyc = (buf[1:iters, 9] - buf[1:iters, 8]).astype(np.float64)
print("   %-45s %6.0f shader-clock ticks = %.0f MHz against the wall clock; %.0f ticks per Newton-Schulz step" % (
    "solver loop", cyc.mean(), (cyc / ph[1:, 3]).mean(), cyc.mean() / max(ctx.last_eig_sweeps, 1)))
print("   %-45s %6.2f us" % ("kernel body, stamp 0 -> 6", (t[1:, 6] - t[1:, 0]).mean()))
gap = t[1:, 0] - t[:-1, 6]
print("   %-45s %6.2f us (pass + slab sum + three launch hops)" % ("end of step k-1 -> start of step k", gap.mean()))
print("   %-45s %6.2f us" % ("whole iteration", (t[1:, 0] - t[:-1, 0]).mean()))
ns = np.zeros((64, 8), np.int64)
assert raw.swk_ns_stamp_read(ns.ctypes.data_as(ctypes.POINTER(ctypes.c_longlong))) == 0
d = np.diff(ns[1:iters].astype(np.float64), axis=1).mean(axis=0)
print("   inside solver step 3 (shader-clock ticks, wave 0): scalars %.0f | Z Y tile + T %.0f | wave sum %.0f | barrier + residual %.0f | "
      "Y T / T Z tiles %.0f | barrier %.0f | stores + barrier %.0f | sum %.0f" % (d[0], d[1], d[2], d[3], d[4], d[5], d[6], d.sum()))

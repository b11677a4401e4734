"""Cycles per phase of k_scan_level_rt (variant library built with -DLQG_SCAN_STAMP):
LQG_HIP_LIB=variants/lib_scanstamp.so python scripts/scan_stamps.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqg_amd import _abi
from lqg_amd.tracking.delay import DelayedSubjectiveActor
lib = _abi.load()
labels = ["-", "M = I + C1 J2 (matrix core) -> LDS, rows into registers", "elim: publish column + barrier", "elim: pivot search", "elim: 1/pivot, publish rows + barrier",
          "elim: update", "X -> LDS + barrier", "A, T1, U products (matrix core) + barrier", "C, J products (matrix core) + barrier", "symmetrise + store"]
md = DelayedSubjectiveActor(T=500, device="cuda", dtype=torch.float64)
x = md.simulate(21, n=4)[..., :2].contiguous()
buf = (C.c_ulonglong * 16)()
md.log_likelihood(x)
lib.lqg_debug_scan_stamps(buf, 1)
md.log_likelihood(x)
lib.lqg_debug_scan_stamps(buf, 1)
tot = sum(buf[:10])
print("18 levels (9 of n = 39 [last window: Kalman], 9 of n = 63): total cycles of the stamped window", tot, "(100 MHz counter)")
for i, l in enumerate(labels):
    print("  %-44s %10d  %5.1f %%" % (l, buf[i], 100.0 * buf[i] / tot))

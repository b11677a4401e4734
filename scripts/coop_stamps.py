"""Cycles per stage of the cooperative forward sweep (variant library built with -DLQG_COOP_STAMP):
LQG_HIP_LIB=variants/lib_coopstamp.so python scripts/coop_stamps.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lqg_amd import _abi
from lqg_amd.tracking.delay import DelayedSubjectiveActor
lib = _abi.load()
names = ["A joint1+condF", "A kal1", "A barrier", "B joint2+cond1", "B kal2", "B barrier", "C cond2(+lists)", "C kal3", "C barrier",
         "D sig1(+ops)", "D kal4", "D barrier", "E sig2", "E kalF", "E barrier", "F kal5+barrier"]
names = ["A joint1+condF", "A kal1+barrier", "-", "B joint2+cond1", "B kal2+barrier", "-", "C cond2(+lists)", "C kal3+barrier", "-", "D sig1(+ops)"]
slots = ["A: joint1 + condF", "A: kal1", "A: barrier", "B: joint2 + cond1", "B: kal2", "B: barrier", "C: cond2 (+ F2 lists)",
         "C: kal3", "C: barrier", "D: sig1 (+ operator emission)", "D: kal4"]
# the macro stamps AFTER each piece: slot 0 = A joint, 1 = kal1 + barrier, 2 = B joint, 3 = kal2 + barrier, 4 = cond2, 5 = kal3 + barrier,
# 6 = sig1, 7 = kal4 + barrier, 8 = sig2, 9 = kalF + barrier, 10 = kal5 + barrier
labels = ["A joint1+condF", "A kal1 + barrier", "B joint2+cond1", "B kal2 + barrier", "C cond2 (+F2 lists)", "C kal3 + barrier",
          "D sig1 (+ops emission)", "D kal4 + barrier", "E sig2", "E kalF + barrier", "F kal5 + barrier"]
for dt in (torch.float32, torch.float64):
    md = DelayedSubjectiveActor(T=500, device="cuda", dtype=dt)
    x = md.simulate(21, n=4)[..., :2].contiguous()
    buf = (C.c_ulonglong * 16)()
    lib.lqg_debug_coop_stamps(buf, 1)
    md.log_likelihood(x)
    lib.lqg_debug_coop_stamps(buf, 1)
    tot = sum(buf[:11])
    print(dt, "total cycles", tot, "per step", tot / 499)
    for i, l in enumerate(labels):
        print("  %-28s %8.0f cycles/step  %5.1f %%" % (l, buf[i] / 499, 100.0 * buf[i] / tot))

"""Phase stamps of the per-window CQT kernels (tools only).  Build the instrumented library first:

    bash tools/build_variant.sh prof cqt_planes.hip -DPL_PROF
    BASIC_PITCH_AMD_LIB=$PWD/basic_pitch_amd/lib/var_prof.so python tools/experiments/cqt_prof.py

Prints, for two workgroups (blocks 0 and 131) and every wave, the s_memtime stamps relative to the wave's kernel entry in
microseconds (scaled with the 100 MHz s_memrealtime pair taken at entry and exit of the same wave)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from basic_pitch_amd import Model, _native  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = Model(max_windows=B)
lib = _native.load_library()
x = np.random.default_rng(0).uniform(-1, 1, (B, 43844)).astype(np.float32)
import torch  # noqa: E402

xd = torch.from_numpy(x).cuda()
for _ in range(30):
    out = m.predict(xd)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (2 * 2 * 16 * 16))()
rc = lib.bp_debug_pl_prof(buf)
a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 2, 16, 16).astype(np.int64)
names = {0: ["entry", "start-up", "chunk 0 in", "L1 chunks", "L1 end"] + [f"L{k}" for k in range(2, 9)],
         1: ["entry", "setup", "tasks", "barrier", "norm A", "norm B+C", "end barrier"]}
print("rc", rc)
for kern, kn in ((0, "pyramid"), (1, "filterbank")):
    for slot in (0, 1):
        print(f"== {kn}, workgroup slot {slot}: stamps in us after the wave's entry (ticks -> us via the realtime pair)")
        t = a[kern, slot]
        n = len(names[kern])
        last = n - 1
        rt = (t[:, 15] - t[:, 14]) / 100.0  # us
        ticks = (t[:, last] - t[:, 0]).astype(np.float64)
        scale = np.where(ticks > 0, rt / np.maximum(ticks, 1), 0)
        print("   wave  entry-skew(us) " + " ".join(f"{s:>11s}" for s in names[kern][1:]) + "   | realtime us, MHz")
        e0 = t[:, 0].min()
        for w in range(16):
            row = [(t[w, i] - t[w, 0]) * scale[w] for i in range(1, n)]
            mhz = ticks[w] / rt[w] if rt[w] > 0 else 0
            print(f"   {w:4d}  {(t[w, 0] - e0) * scale[w]:13.2f} " + " ".join(f"{v:11.2f}" for v in row) + f"   | {rt[w]:7.2f} {mhz:7.0f}")

# the boundary between the two launches as the device's 100 MHz clock saw it (same workgroup slot, wave 0)
for slot in (0, 1):
    pyr_in, pyr_out = a[0, slot, 0, 14], a[0, slot, 0, 15]
    fb_in, fb_out = a[1, slot, 0, 14], a[1, slot, 0, 15]
    print(f"slot {slot}: pyramid in-kernel {(pyr_out - pyr_in) / 100.0:.2f} us | exit -> filterbank entry {(fb_in - pyr_out) / 100.0:.2f} us | "
          f"filterbank in-kernel {(fb_out - fb_in) / 100.0:.2f} us | pyramid entry -> filterbank exit {(fb_out - pyr_in) / 100.0:.2f} us")

"""Note stage (conv1 7x7 stride 3 + ReLU + conv2 (7,3) + sigmoid) on random contour maps through the C ABI test hook;
saves the note map (A/B of the note kernels: run once per BP_NOTE / BASIC_PITCH_AMD_LIB setting and compare)."""
import os, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    print("max", d.max(), "n>1e-6", int((d > 1e-6).sum()), "of", d.size)
    sys.exit(0)
from stage_harness import StageRunner
rng = np.random.default_rng(12)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# sigmoid outputs: mostly small, some ridges near 1 (and exact 0 / 1 / tiny values at the map's rim)
c = rng.random((n, 172, 264), dtype=np.float32) ** 3
c[0, :3] = 0.0; c[0, -2:] = 1.0; c[1, :, :3] = 1.0; c[1, :, -4:] = 1e-7
r = StageRunner()
got = r.run("note", n, {"contour": c}, {"note": ((n, 172, 88), __import__("torch").float32)})["note"]
np.save(sys.argv[1], got)
print("saved", sys.argv[1], got.shape, float(got.mean()))

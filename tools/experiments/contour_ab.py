"""Contour stage (rim + interior conv1 + conv2) on random inputs through the C ABI test hook; saves the map (A/B of the
conv1 kernels: run once per BP_CONV1 / BASIC_PITCH_AMD_LIB setting and compare the files with --cmp)."""
import os, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    print("max", d.max(), "n>1e-6", int((d > 1e-6).sum()), "of", d.size)
    sys.exit(0)
from stage_harness import StageRunner, zp_pack
rng = np.random.default_rng(11)
n = 3
z = (rng.random((n, 172, 309), dtype=np.float32) * 2.4 - 0.8).astype(np.float32)
r = StageRunner()
got = r.run("contour", n, {"zp": zp_pack(z).view(np.int32)}, {"contour": ((n, 172, 264), __import__("torch").float32)})["contour"]
np.save(sys.argv[1], got)
print("saved", sys.argv[1], got.shape, float(got.mean()))

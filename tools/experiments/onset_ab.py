"""Onset stage on random inputs through the C ABI test hook; saves the map (A/B of the onset kernels: run once per
BP_ONSET / BASIC_PITCH_AMD_LIB setting and compare the files with --cmp)."""
import os, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    d = np.abs(a - b)
    print("max", d.max(), "n>1e-6", int((d > 1e-6).sum()), "of", d.size)
    idx = np.argwhere(d > 1e-6)
    if len(idx):
        print("windows", np.unique(idx[:, 0]), "frames", np.unique(idx[:, 1])[:40], "pixels", np.unique(idx[:, 2])[:40])
        w = np.unravel_index(d.argmax(), d.shape); print("worst at", w, a[w], b[w])
        np.set_printoptions(linewidth=200, precision=1)
        print("mean |d| by pixel (x1e6):", d.mean(axis=(0, 1)) * 1e6)
        print("mean |d| by frame (x1e6):", d.mean(axis=(0, 2)) * 1e6)
        print("signed mean:", float((a - b).mean()))
    sys.exit(0)
from stage_harness import StageRunner, zp_pack
rng = np.random.default_rng(5)
n = 3
z = (rng.random((n, 172, 309), dtype=np.float32) * 2.4 - 0.8).astype(np.float32)
note = rng.random((n, 172, 88), dtype=np.float32)
mode = os.environ.get("OM_MODE", "")
if "zf16" in mode:
    z = z.astype(np.float16).astype(np.float32)  # lo halves of the image are zero
if "note0" in mode:
    note[:] = 0
if "zsmall" in mode:
    z *= 0.01
if "zzero" in mode:
    z[:] = 0
r = StageRunner()
got = r.run("onset", n, {"zp": zp_pack(z).view(np.int32), "note": note}, {"onset": ((n, 172, 88), __import__("torch").float32)})["onset"]
np.save(sys.argv[1], got)
print("saved", sys.argv[1], got.shape, float(got.mean()))

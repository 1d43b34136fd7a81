# Stage check (tools/): the contour branch kernels alone against the fp64 oracle, fed with the oracle's own z; per-bin
# errors at the rims.  On the GPU box: python tools/stage_err_contour.py   (BP_CONV1=f16 for the all-f16 folded kernel)
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_windows
from oracle import bp_oracle as O
from stage_harness import StageRunner, zp_pack
W = O.load_weights()
x = np.concatenate([make_windows("uniform", 1, 0), make_windows("normal", 1, 1)])
r = O.forward(x, W, np.float64, intermediates=True)
run = StageRunner()
out = run.run("contour", 2, {"zp": zp_pack(r["z"].astype(np.float32))}, {"contour": ((2, 172, 264), torch.float32)})
d = np.abs(out["contour"] - r["contour"])
print("mode", os.environ.get("BP_RIM"), "max err", d.max(), "interior", d[:, :, 24:240].max(), "low rim", d[:, :, :20].max(), "high rim", d[:, :, 244:].max())
print("per-bin low rim", np.round(d[:, :, :24].max(axis=(0, 1)) * 1e5, 1))
print("per-bin high rim", np.round(d[:, :, 240:].max(axis=(0, 1)) * 1e5, 1))
print("per-bin 224..247", np.round(d[:, :, 224:248].max(axis=(0, 1)) * 1e5, 1))
print("interior 24:236", d[:, :, 24:236].max())

# Stage check (tools/): the onset branch kernel alone against the fp64 oracle (BP_ONSET=f16 for the all-f16 kernel).
# Stage check (tools/): the onset branch kernel alone against the fp64 oracle (BP_ONSET=f16 for the all-f16 kernel).
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_windows
from oracle import bp_oracle as O
from stage_harness import StageRunner, zp_pack
W = O.load_weights()
x = np.concatenate([make_windows("uniform", 1, 0), make_windows("normal", 1, 1), make_windows("tones", 1, 2)])
r = O.forward(x, W, np.float64, intermediates=True)
run = StageRunner()
n = x.shape[0]
out = run.run("onset", n, {"zp": zp_pack(r["z"].astype(np.float32)).view(np.int32), "note": r["note"].astype(np.float32)}, {"onset": ((n, 172, 88), torch.float32)})
d = np.abs(out["onset"] - r["onset"])
print("onset mode", os.environ.get("BP_ONSET"), "max err", d.max(), "per window", d.max(axis=(1, 2)))

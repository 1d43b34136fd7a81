# where does the filterbank stage differ from the oracle?  (level, filter, frame) statistics.  GPU.
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_windows
from oracle import bp_oracle as O
from stage_harness import StageRunner, pyr_pack
W = O.load_weights()
x = np.concatenate([make_windows("uniform", 2, 0), make_windows("normal", 1, 1)])
r64 = O.forward(x, W, np.float64, intermediates=True)
r32 = O.forward(x, W, np.float32, intermediates=True)
run = StageRunner()
n = len(x)
out = run.run("filterbank", n, {"audio": x, "pyr": pyr_pack(r32["levels"], run.lib)},
              {"lp": ((n, 172, 309), torch.float32), "mm": ((n, 2), torch.int32)})
mag = np.sqrt(np.maximum(10.0 ** (out["lp"].astype(np.float64) / 10.0) - 1e-10, 0))
err = np.abs(mag - r64["mag"])
print("nan count", np.isnan(out["lp"]).sum(), "max err", err.max())
for level in range(9):
    bin0 = (8 - level) * 36 - 15
    for f0, f1 in ((0, 16), (16, 32), (32, 36)):
        b0, b1 = max(0, bin0 + f0), bin0 + f1
        if b1 <= 0: continue
        e = err[:, :, b0:b1]
        w = np.unravel_index(np.argmax(e), e.shape)
        print(f"level {level} filters {f0}-{f1}: max {e.max():.2e} at window {w[0]} frame {w[1]} bin {b0 + w[2]}; per-frame-block max",
              " ".join(f"{e[:, t:t+16].max():.1e}" for t in range(0, 172, 16)))

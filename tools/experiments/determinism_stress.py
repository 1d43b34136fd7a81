"""The same batch through the whole path again and again: every output must be bit-identical to the first pass (a missing wait
behind an LDS-DMA or a barrier one item early shows up as a run-to-run difference long before it shows up as a parity error).
    python tools/experiments/determinism_stress.py [passes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from basic_pitch_amd import Model  # noqa: E402

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for B in (256, 37, 3):
    m = Model(max_windows=B)
    x = torch.from_numpy(np.random.default_rng(B).uniform(-1, 1, (B, 43844)).astype(np.float32)).cuda()
    ref = {k: v.clone() for k, v in m._predict_device(x).items()}
    torch.cuda.synchronize()
    for i in range(passes):
        out = m._predict_device(x)
        if i % 7 == 0:  # other work between passes: different timing of the next launch
            _ = (x * 1.0001).sum()
        for k in ref:
            if not torch.equal(out[k], ref[k]):
                bad += 1
                print("DIFF", B, i, k, float((out[k] - ref[k]).abs().max()))
    m.close()
    print("B", B, "passes", passes, "ok so far" if not bad else f"{bad} differences")
sys.exit(1 if bad else 0)

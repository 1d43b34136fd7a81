"""The fp8-corrections mode (BP_FLAG_FP8_CORRECTIONS; A/B library only since round 6) beside the default on given windows:
python fp8_mode_ab.py in.npz out.npz with BASIC_PITCH_AMD_LIB = the A/B library.  in.npz: x (n, 43844) float32 and,
optionally, z / note (the oracle's stage inputs).  out.npz: f16_* / fp8_* whole-path maps and stage_* maps of the fp8 mode's
contour / onset branch kernels fed z / note through the C ABI stage hook."""
import os, sys
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from basic_pitch_amd import Model
from stage_harness import StageRunner, zp_pack

d = np.load(sys.argv[1])
x = d["x"]
out = {}
for name, kw in (("f16", {}), ("fp8", {"fp8_corrections": True})):
    m = Model(max_windows=8, **kw)
    for k, v in m.predict(x).items():
        out[f"{name}_{k}"] = v
    if name == "fp8" and "z" in d.files:
        r = StageRunner(m)
        n = x.shape[0]
        zp = zp_pack(d["z"]).view(np.int32)
        out["stage_contour"] = r.run("contour", n, {"zp": zp}, {"contour": ((n, 172, 264), torch.float32)})["contour"]
        out["stage_onset"] = r.run("onset", n, {"zp": zp, "note": d["note"]}, {"onset": ((n, 172, 88), torch.float32)})["onset"]
    m.close()
np.savez(sys.argv[2], **out)
print("saved", sys.argv[2])

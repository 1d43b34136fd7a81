"""Phase clocks of the contour conv1 rim kernel (build with tools/build_variant.sh rimprof conv_contour_rim.hip -DRIM_PROF,
run with BASIC_PITCH_AMD_LIB=.../var_rimprof.so): a few steps of the bench batch; the kernel prints per sampled wave
(HW_ID, start on the constant 100 MHz clock, phase lengths in shader clocks)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from basic_pitch_amd import inference

m = inference.Model(max_windows=256)
x = (torch.rand(256, 43844, device="cuda") * 2 - 1).float()
for _ in range(3):
    out = m.predict(x)
torch.cuda.synchronize()

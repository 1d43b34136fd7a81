"""Phase clocks of the note march (build with tools/build_variant.sh n16_prof note_march16.hip -DN16_PROF, run with
BASIC_PITCH_AMD_LIB=.../var_n16_prof.so): a few steps of the bench batch; the kernel prints, for a few waves, 100 MHz
stamps relative to the wave's entry: set-up done | per share: prologue done, march done | exit."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from basic_pitch_amd import inference

m = inference.Model(max_windows=256)
x = (torch.rand(256, 43844, device="cuda") * 2 - 1).float()
for _ in range(3):
    out = m.predict(x)
torch.cuda.synchronize()

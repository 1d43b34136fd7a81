"""Decode-kernel time of the device FLAC decoder on the reference's second recording (a libFLAC stream: orders and partition
orders differ from subframe to subframe, so the lanes of a wave diverge where the synthetic corpus does not)."""
import os, sys, ctypes as C
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from basic_pitch_amd import Model
data = open(os.path.join(ROOT, "tests", "golden", "vocadito_14.flac"), "rb").read()
m = Model(max_windows=32)
print(m.flac_layout(data))
nf = C.c_int64()
for _ in range(6):
    rc = m._lib.bp_flac_decode_device(m._handle, data, len(data), None, 0, C.byref(nf))
    assert rc == 0, rc

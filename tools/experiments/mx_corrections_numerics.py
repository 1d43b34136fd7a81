"""Experiment (round 2, CPU): would the two CORRECTION products of the split-precision scheme survive on the block-scaled
fp8 matrix instruction?  Contour conv1 (the dominant layer) with hi*hi in f16 and lo_w*hi_a, hi_w*lo_a rounded to 3 / 2 / 1
mantissa bits (e4m3 has 3), against fp64.  Result: 3 bits -> 9.5e-5 on the pre-activations, 1.05e-5 on the contour map
(all three f16 products: 2.9e-8; hi*hi only: 2.5e-4) — inside the 1e-4 bar.  tools/ubench/mfma_mx.hip measured the
matching instruction mix (4 f16 + 2 MX per 64 taps) 1.82 x faster than today's 12 f16.  Not built this round:
DESIGN.md section 7."""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, torch.nn.functional as F
from oracle import bp_oracle as O
from conftest import make_windows
W = O.load_weights()
def q_m(x, mbits):
    # round to `mbits` mantissa bits (relative), no range limit: idealised block-scaled fp8 (e4m3: 3 bits)
    x = np.asarray(x, np.float64); out = np.zeros_like(x); nz = x != 0
    e = np.floor(np.log2(np.abs(x[nz]))); s = 2.0 ** (e - mbits)
    out[nz] = np.round(x[nz] / s) * s
    return out
def f16(x): return np.asarray(x, np.float64).astype(np.float16).astype(np.float64)
x = np.concatenate([make_windows("uniform", 1, 0), make_windows("normal", 1, 1), make_windows("tones", 2, 2)])
r64 = O.forward(x, W, np.float64, intermediates=True)
z = torch.from_numpy(r64["z"])                      # (n,172,309) fp64
stack = O.harmonic_stack(z)                         # (n,8,172,264)
w1 = W["contour1_w"].astype(np.float64); b1 = W["contour1_b"].astype(np.float64)
S = stack.numpy()
Sh = f16(S); Sl = S - Sh
Wh = f16(w1); Wl = w1 - Wh
def conv(wt, st):
    return F.conv2d(torch.from_numpy(st), torch.from_numpy(wt), None, padding=(1, 19)).numpy()
ref = conv(w1, S)
main = conv(Wh, Sh)
def report(name, approx):
    d = np.abs(approx - ref)
    print(f"{name:46s} conv1 pre-activation max err {d.max():.2e}  rms {np.sqrt((d**2).mean()):.2e}   (|ref| max {np.abs(ref).max():.2f})")
    return approx
report("3 products exact (hi*hi+lo*hi+hi*lo)", main + conv(Wl, Sh) + conv(Wh, Sl))
report("hi*hi only", main)
for mb in (3, 2, 1):
    report(f"hi*hi + q{mb}(lo_w)*q{mb}(hi_a) + q{mb}(hi_w)*q{mb}(lo_a)", main + conv(q_m(Wl, mb), q_m(Sh, mb)) + conv(q_m(Wh, mb), q_m(Sl, mb)))
# end to end effect on the contour map for the 3-bit case
def contour_from_c1(c1pre):
    c1 = np.maximum(c1pre + b1[None, :, None, None], 0)
    y = F.conv2d(torch.from_numpy(c1), torch.from_numpy(W["contour2_w"].astype(np.float64)), torch.from_numpy(W["contour2_b"].astype(np.float64)), padding=(2, 2))
    return torch.sigmoid(y).numpy()[:, 0]
cref = contour_from_c1(ref)
for name, ap in (("3 products", main + conv(Wl, Sh) + conv(Wh, Sl)), ("fp8 corrections (3 bits)", main + conv(q_m(Wl, 3), q_m(Sh, 3)) + conv(q_m(Wh, 3), q_m(Sl, 3))), ("hi*hi only", main)):
    print(f"contour map error, {name:28s}: {np.abs(contour_from_c1(ap) - cref).max():.2e}")

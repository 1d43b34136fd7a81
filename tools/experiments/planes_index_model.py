# Index model of csrc/cqt_planes.hip in numpy (float64, no operand split): the plane geometry, the transposed decimator
# tile with its edge masks and reflect-pad writes, and the filterbank's fragment addressing, checked against the oracle's
# pyramid / CQT on one window.  CPU only:  python tools/experiments/planes_index_model.py
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bp_oracle as O

PAD, TILE = 128, 256
W = O.load_weights()
h = W["cqt_lowpass"].astype(np.float64)
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, (1, O.AUDIO_N_SAMPLES)).astype(np.float32)
ref = O.forward(x, W, np.float64, intermediates=True)
levels = [l[0] for l in ref["levels"]]
lens = [len(l) for l in levels]
hops = [256 >> k for k in range(9)]
rlen = [((max(PAD + L + 776, 176 * hp + 256) + 63) // 64) * 64 for L, hp in zip(lens, hops)]

def split_region(sig, L, rl):
    out = np.zeros(rl)
    for idx in range(rl):
        g = idx - PAD
        g = -g if g < 0 else g
        g = 2 * (L - 1) - g if g >= L else g
        out[idx] = sig[g] if 0 <= g < L else 0.0
    return out

regions = [split_region(levels[0], lens[0], rlen[0])] + [np.full(r, np.nan) for r in rlen[1:]]
for k in range(1, 9):
    regions[k][:] = 0.0  # the memset slack
T = np.zeros((16, 288))
for u in range(16):
    for i in range(288):
        j = i - 2 * u - 1
        if 0 <= j < 256:
            T[u, i] = h[j]
for k in range(1, 9):
    L_in, L_out = lens[k - 1], lens[k]
    src, dst = regions[k - 1], regions[k]
    for tile in range((L_out + TILE - 1) // TILE):
        o0 = TILE * tile
        edge = tile == 0 or 2 * o0 + 768 > PAD + L_in
        for m in range(16):
            X = src[2 * o0 + 32 * m: 2 * o0 + 32 * m + 288].copy()
            if edge:
                idx = 2 * o0 + 32 * m + np.arange(288)
                X[(idx < PAD) | (idx >= PAD + L_in)] = 0.0
            else:
                idx = 2 * o0 + 32 * m + np.arange(288)
                assert idx.min() >= PAD and idx.max() < PAD + L_in, (k, tile, m)
            D = T @ X
            for u in range(16):
                n = o0 + 16 * m + u
                if n >= L_out:
                    continue
                dst[PAD + n] = D[u]
                if 1 <= n <= PAD:
                    dst[PAD - n] = D[u]
                if L_out - 1 - PAD <= n <= L_out - 2:
                    dst[PAD + 2 * (L_out - 1) - n] = D[u]
    got = dst[PAD:PAD + L_out]
    print(f"level {k}: max |model - oracle| = {np.abs(got - levels[k]).max():.2e}")
    want = split_region(levels[k], L_out, rlen[k])
    lo, hi = PAD - 112, min(rlen[k], PAD + L_out + 128)
    print(f"          reflect pads: max diff {np.abs(dst[lo:hi] - want[lo:hi]).max():.2e} (elements {lo}..{hi})")

# filterbank addressing: frame t of level k reads elements t*hop + tap of the region, tap in [16, 240)
re, im = W["cqt_kernel_re"].astype(np.float64), W["cqt_kernel_im"].astype(np.float64)
worst = 0.0
mag_ref = ref["mag"][0]  # (172, 309)
sl = W["cqt_sqrt_len"].astype(np.float64)
for k in range(9):
    bin0 = (8 - k) * 36 - 15
    for t in range(172):
        seg = regions[k][t * hops[k]: t * hops[k] + 256]
        for f in (0, 15, 16, 31, 32, 35):
            b = bin0 + f
            if b < 0:
                continue
            lo_t, hi_t = (16, 240) if f < 16 else (48, 208)
            r_ = np.dot(seg[lo_t:hi_t], re[f, lo_t:hi_t]) * sl[b]
            i_ = np.dot(seg[lo_t:hi_t], im[f, lo_t:hi_t]) * sl[b]
            worst = max(worst, abs(np.hypot(r_, i_) - mag_ref[t, b]))
print(f"filterbank addressing: max |mag - oracle| = {worst:.2e}")

"""Test-side FLAC *encoder* written from the format specification (RFC 9639), used to make inputs for the product's
decoder (basic_pitch_amd/csrc/flac_decode.cpp).  Deliberately exercises every decoder branch: constant / verbatim /
fixed (orders 0-4) / LPC subframes, wasted bits, Rice and Rice2 residuals with several partition orders and escaped
partitions, independent / left-side / side-right / mid-side stereo, variable last block, 8 / 16 / 24-bit samples.
Frames carry valid CRC-8 / CRC-16 and STREAMINFO the MD5 of the PCM, so the decoder's integrity checks are live.
"""
import hashlib
import struct

import numpy as np


class BitWriter:
    def __init__(self):
        self.acc, self.nbits, self.out = 0, 0, bytearray()

    def write(self, value, bits):
        if bits == 0:
            return
        self.acc = (self.acc << bits) | (int(value) & ((1 << bits) - 1))
        self.nbits += bits
        while self.nbits >= 8:
            self.nbits -= 8
            self.out.append((self.acc >> self.nbits) & 0xFF)
        self.acc &= (1 << self.nbits) - 1

    def unary(self, q):
        while q >= 32:
            self.write(0, 32)
            q -= 32
        self.write(1, q + 1)

    def align(self):
        if self.nbits:
            self.write(0, 8 - self.nbits)

    def bytes(self):
        assert self.nbits == 0
        return bytes(self.out)


def crc8(data):
    c = 0
    for b in data:
        c ^= b
        for _ in range(8):
            c = ((c << 1) ^ 0x07) & 0xFF if c & 0x80 else (c << 1) & 0xFF
    return c


def crc16(data):
    c = 0
    for b in data:
        c ^= b << 8
        for _ in range(8):
            c = ((c << 1) ^ 0x8005) & 0xFFFF if c & 0x8000 else (c << 1) & 0xFFFF
    return c


def _utf8(n):
    if n < 0x80:
        return bytes([n])
    out, lead_bits = [], 6
    while n >= (1 << lead_bits):
        out.append(0x80 | (n & 0x3F))
        n >>= 6
        lead_bits -= 1
    lead = (0xFF << (lead_bits + 1)) & 0xFF
    return bytes([lead | n] + out[::-1])


def _zigzag(r):
    return [(v << 1) if v >= 0 else ((-v) << 1) - 1 for v in r]


def _write_residual(bw, res, order, blocksize, porder, rice2, escape):
    bw.write(1 if rice2 else 0, 2)
    bw.write(porder, 4)
    pbits, esc = (5, 31) if rice2 else (4, 15)
    pos = 0
    for p in range(1 << porder):
        count = (blocksize >> porder) - (order if p == 0 else 0)
        part = res[pos : pos + count]
        pos += count
        if escape and p % 2 == 1:
            raw = max([1] + [int(v).bit_length() + 1 for v in part])
            bw.write(esc, pbits)
            bw.write(raw, 5)
            for v in part:
                bw.write(v, raw)
            continue
        u = _zigzag(part)
        mean = (sum(u) / len(u)) if u else 0
        k = max(0, min(esc - 1, int(mean).bit_length() - 1 if mean >= 1 else 0))
        bw.write(k, pbits)
        for v in u:
            bw.unary(v >> k)
            bw.write(v & ((1 << k) - 1), k)
    assert pos == len(res)


_FIXED = {0: [], 1: [1], 2: [2, -1], 3: [3, -3, 1], 4: [4, -6, 4, -1]}


def _subframe(bw, s, bps, kind, porder=0, rice2=False, escape=False, lpc_order=8):
    s = [int(v) for v in s]
    n = len(s)
    wasted = 0
    if kind != "constant" and any(s):
        while all((v >> wasted) & 1 == 0 for v in s):
            wasted += 1
    if wasted:
        s = [v >> wasted for v in s]
        bps -= wasted
    if kind == "constant":
        assert all(v == s[0] for v in s)
        code = 0
    elif kind == "verbatim":
        code = 1
    elif kind.startswith("fixed"):
        order = int(kind[5:])
        code = 8 + order
    else:
        order = lpc_order
        code = 31 + order
    bw.write(0, 1)
    bw.write(code, 6)
    if wasted:
        bw.write(1, 1)
        bw.unary(wasted - 1)
    else:
        bw.write(0, 1)
    if kind == "constant":
        bw.write(s[0], bps)
    elif kind == "verbatim":
        for v in s:
            bw.write(v, bps)
    elif kind.startswith("fixed"):
        c = _FIXED[order]
        for v in s[:order]:
            bw.write(v, bps)
        res = [s[i] - sum(c[j] * s[i - 1 - j] for j in range(order)) for i in range(order, n)]
        _write_residual(bw, res, order, n, porder, rice2, escape)
    else:
        x = np.asarray(s, dtype=np.float64)
        r = np.array([np.dot(x[: n - l], x[l:]) for l in range(order + 1)])
        r[0] = r[0] * (1 + 1e-9) + 1e-9
        a = np.zeros(order)
        err = r[0]
        for i in range(order):  # Levinson-Durbin
            k = (r[i + 1] - np.dot(a[:i], r[i:0:-1])) / err
            a[:i], a[i] = a[:i] - k * a[:i][::-1], k
            err *= 1 - k * k
        prec, shift = 12, 9
        q = np.clip(np.round(a * (1 << shift)), -(1 << (prec - 1)), (1 << (prec - 1)) - 1).astype(np.int64)
        for v in s[:order]:
            bw.write(v, bps)
        bw.write(prec - 1, 4)
        bw.write(shift, 5)
        for v in q:
            bw.write(int(v), prec)
        res = [s[i] - (sum(int(q[j]) * s[i - 1 - j] for j in range(order)) >> shift) for i in range(order, n)]
        _write_residual(bw, res, order, n, porder, rice2, escape)


def encode(pcm, sample_rate, bits, blocksize=1152, plan=None, total_in_header=True, md5_in_header=True, id3=False):
    """pcm: int array [n, channels].  plan(frame_index) -> dict(kind=..., stereo="indep"|"ls"|"sr"|"ms", porder=..,
    rice2=.., escape=.., lpc_order=1..32) chooses how each frame is coded (cycled defaults exercise everything)."""
    pcm = np.asarray(pcm, dtype=np.int64)
    n, ch = pcm.shape
    kinds = ["fixed2", "lpc", "fixed0", "fixed1", "verbatim", "fixed3", "fixed4", "lpc"]
    stereos = ["indep", "ms", "ls", "sr"]
    frames = bytearray()
    for fi, start in enumerate(range(0, n, blocksize)):
        blk = pcm[start : start + blocksize]
        bs = len(blk)
        p = dict(kind=kinds[fi % len(kinds)], stereo=stereos[fi % 4] if ch == 2 else "indep", porder=fi % 3,
                 rice2=bool(fi % 2), escape=(fi % 5 == 4))
        if plan:
            p.update(plan(fi) or {})
        while p["porder"] and (bs % (1 << p["porder"]) or (bs >> p["porder"]) < 33):
            p["porder"] -= 1
        chans = [blk[:, c] for c in range(ch)]
        widths = [bits] * ch
        ch_code = ch - 1
        if ch == 2 and p["stereo"] != "indep":
            l, r = chans
            side = l - r
            if p["stereo"] == "ls":
                chans, widths, ch_code = [l, side], [bits, bits + 1], 8
            elif p["stereo"] == "sr":
                chans, widths, ch_code = [side, r], [bits + 1, bits], 9
            else:
                chans, widths, ch_code = [(l + r) >> 1, side], [bits, bits + 1], 10
        bw = BitWriter()
        bw.write(0b11111111111110, 14)
        bw.write(0, 1)
        bw.write(0, 1)  # fixed block size stream: frame number is coded
        bs_code = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12}.get(bs)
        if bs_code is None:
            bs_code = 6 if bs <= 256 else 7
        bw.write(bs_code, 4)
        sr_code = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10,
                   96000: 11}.get(sample_rate, 0 if fi % 2 else 13 if sample_rate < 65536 else 0)
        bw.write(sr_code, 4)
        bw.write(ch_code, 4)
        bw.write({8: 1, 12: 2, 16: 4, 20: 5, 24: 6}.get(bits, 0) if fi % 3 else 0, 3)
        bw.write(0, 1)
        for b in _utf8(fi):
            bw.write(b, 8)
        if bs_code == 6:
            bw.write(bs - 1, 8)
        elif bs_code == 7:
            bw.write(bs - 1, 16)
        if sr_code == 13:
            bw.write(sample_rate, 16)
        bw.write(crc8(bw.bytes()), 8)
        for c, (x, w) in enumerate(zip(chans, widths)):
            kind = p["kind"]
            if len(set(int(v) for v in x)) == 1:
                kind = "constant"
            elif kind == "lpc" and bs <= 40:
                kind = "fixed2"
            elif kind.startswith("fixed") and bs <= int(kind[5:]):
                kind = "verbatim"
            _subframe(bw, x, w, kind, p["porder"], p["rice2"], p["escape"], lpc_order=int(p.get("lpc_order", 8)))
        bw.align()
        body = bw.bytes()
        frames += body + struct.pack(">H", crc16(body))
    bytes_per = (bits + 7) // 8
    raw = b"".join(int(v).to_bytes(bytes_per, "little", signed=True) for v in pcm.reshape(-1))
    md5 = hashlib.md5(raw).digest() if md5_in_header else b"\0" * 16
    total = n if total_in_header else 0
    si = struct.pack(">HH", blocksize, blocksize) + b"\0\0\0" + b"\0\0\0"
    packed = (sample_rate << 44) | ((ch - 1) << 41) | ((bits - 1) << 36) | total
    si += packed.to_bytes(8, "big") + md5
    out = bytearray()
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 10]) + b"\0" * 10
    out += b"fLaC" + bytes([0x00]) + len(si).to_bytes(3, "big") + si
    pad = b"\0" * 16
    out += bytes([0x81]) + len(pad).to_bytes(3, "big") + pad  # PADDING block, last
    return bytes(out + frames)

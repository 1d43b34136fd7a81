#!/usr/bin/env python
"""Registers, spills, scratch and LDS of every kernel in a built library, read from the code objects' metadata (no
recompile): python tools/kernel_resources.py [path/to/lib.so]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(lib_path, arch="gfx950"):
    d = open(lib_path, "rb").read()
    pos, out = 0, []
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    while True:
        i = d.find(magic, pos)
        if i < 0:
            return out
        (nb,) = struct.unpack_from("<Q", d, i + 24)
        p = i + 32
        for _ in range(nb):
            off, size, idl = struct.unpack_from("<QQQ", d, p)
            p += 24
            tid = d[p : p + idl].decode()
            p += idl
            if arch in tid and size > 0:
                out.append(d[i + off : i + off + size])
        pos = i + len(magic)


def kernel_resources(lib_path):
    """[{name, vgpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds}] for every kernel of the library's gfx950 code objects"""
    res = []
    for co in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("  - .agpr_count:")[1:]:
            def g(key):
                m = re.search(re.escape(key) + r":\s+(\S+)", blk)
                return m.group(1) if m else None
            name = g(".name")
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            res.append(dict(name=name, vgpr=int(g(".vgpr_count")), sgpr=int(g(".sgpr_count")),
                            vgpr_spill=int(g(".vgpr_spill_count")), sgpr_spill=int(g(".sgpr_spill_count")),
                            scratch=int(g(".private_segment_fixed_size")), lds=int(g(".group_segment_fixed_size"))))
    return res


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "basic_pitch_amd", "lib", "libbasicpitch_amd.so")
    rows = kernel_resources(lib)
    print(f"{'kernel':72s} vgpr sgpr vspill sspill scratch     lds")
    for r in sorted(rows, key=lambda r: r["name"]):
        print(f"{r['name'][:72]:72s} {r['vgpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch']:7d} {r['lds']:7d}")
    bad = [r["name"] for r in rows if r["scratch"]]
    print(f"{len(rows)} kernels, {len(bad)} with scratch")

#!/usr/bin/env python3
"""Instruction statistics of one kernel of the shipped library: extracts the gfx950 code object from libbpr1cs_hip.so (the
clang offload bundle in .hip_fatbin), disassembles the kernel whose symbol contains `pattern` and prints its register use and
an instruction histogram.   python tools/kernel_isa_stats.py [lib.so] pattern [--loop]"""
import collections
import os
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    """every gfx950 code object of the library: one clang offload bundle per translation unit (the dominant kernel is a unit of its own)"""
    data = open(lib, "rb").read()
    out, at = [], 0
    while True:
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", at)
        if i < 0:
            break
        at = i + 24
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        if n > 16:
            continue
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tl].decode(errors="replace")
            off += tl
            if "gfx950" in triple:
                out.append(data[i + o:i + o + sz])
    if not out:
        raise SystemExit("no gfx950 code object in " + lib)
    return out


def code_object(lib, pattern=None):
    """the code object that holds the kernels whose symbol contains `pattern` (None: the largest one)"""
    cos = code_objects(lib)
    if pattern:
        hit = [c for c in cos if pattern.encode() in c]
        if hit:
            return hit[0]
    return max(cos, key=len)


def kernel_hash(lib, pattern):
    """SHA-256 (hex, 16 digits) over the disassembled instruction text of the kernels whose symbol contains `pattern` - the identity of
    a kernel BUILD: bench.py compares it with the value stamped into profiles/*_pmc_hbm_traffic.txt before it quotes that
    profile's counters for the library it has loaded.  None if the tools or the kernel are missing."""
    import hashlib
    try:
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(code_object(lib, pattern))
            co = f.name
        try:
            syms = subprocess.run([LLVM + "/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
            names = sorted(set(l.split()[-1] for l in syms.split("\n") if " FUNC " in l and pattern in l))
            if not names:
                return None
            h = hashlib.sha256()
            for name in names:
                dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--disassemble-symbols=" + name, co], capture_output=True, text=True).stdout
                for l in dis.split("\n"):
                    if l.startswith("\t"):
                        h.update(l.split("//")[0].strip().encode() + b"\n")
            return h.hexdigest()[:16]
        finally:
            os.unlink(co)
    except Exception:
        return None


def main():
    if "--hash" in sys.argv:
        args = [a for a in sys.argv[1:] if not a.startswith("--")]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lib = args.pop(0) if len(args) == 2 else os.path.join(root, "bulletproofs-r1cs-gadgets_amd", "csrc", "libbpr1cs_hip.so")
        print(kernel_hash(lib, args[0]))
        return
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "bulletproofs-r1cs-gadgets_amd", "csrc", "libbpr1cs_hip.so")
    if len(args) == 2:
        lib = args.pop(0)
    pat = args[0]
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(code_object(lib, pat))
        co = f.name
    syms = subprocess.run([LLVM + "/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
    names = [l.split()[-1] for l in syms.split("\n") if " FUNC " in l and pat in l]
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for name in sorted(set(names)):
        print("==", name)
        blk = next((e for e in notes.split("\n  - ") if (".name:           " + name + "\n") in e + "\n"), "")
        print("   " + "  ".join(l.strip() for l in blk.split("\n") if any(k in l for k in (".vgpr_count", ".sgpr_count", ".private_segment_fixed_size", ".group_segment_fixed_size"))))
        dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--disassemble-symbols=" + name, co], capture_output=True, text=True).stdout
        ins = [l.split()[0] for l in dis.split("\n") if l.startswith("\t") and l.split()]
        c = collections.Counter(ins)
        print("   instructions:", len(ins), " mad64:", c["v_mad_i64_i32"] + c["v_mad_u64_u32"], " valu:", sum(v for k, v in c.items() if k.startswith("v_")),
              " scratch:", sum(v for k, v in c.items() if k.startswith("scratch")), " global:", sum(v for k, v in c.items() if k.startswith("global")),
              " ds:", sum(v for k, v in c.items() if k.startswith("ds_")))
        print("   top:", ", ".join("%s %d" % kv for kv in c.most_common(14)))
    os.unlink(co)


if __name__ == "__main__":
    main()

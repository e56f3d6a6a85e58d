#!/usr/bin/env python
"""Build-time ISA check: the split volume kernels' LDS-DMA groups write M0 in one asm statement (`glds16_m0`) and rely on it surviving until
the group's last piece (`glds16_next`) — nothing the compiler emits in between may touch M0.  hipcc cannot see that contract, so it is checked on the
generated ISA: inside every `corr_volume_split_stream` kernel the ONLY instructions that name m0 are `s_mov_b32 m0, sN` directly in front of (no-ops aside) a
`global_load_lds_dwordx4` (a group head), and the save / restore moves of the single-piece forms around one.  Run by `__graft_entry__.build()`; exits non-zero with the offending lines otherwise.

    python tools/check_isa.py [--keep out.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mac-vo_amd", "csrc")


def main() -> int:
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    with tempfile.TemporaryDirectory() as tmp:
        out = keep or os.path.join(tmp, "split.s")
        cmd = [os.environ.get("HIPCC", "hipcc"), "--offload-arch=" + os.environ.get("ARCH", "gfx950"), "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
               "-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-unroll-threshold=1000000", "--cuda-device-only", "-S",
               os.path.join(CSRC, "corr_volume_split.hip"), "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
            return 2
        lines = open(out).read().splitlines()
    kernel, bad, groups, kernels = None, [], 0, set()
    instr = re.compile(r"^\s+([a-z][a-z0-9_]+)\s+(.*)$")
    body = []
    for ln in lines:
        m = re.match(r"^(_Z\w*corr_volume_split_stream\w*):", ln)
        if m:
            kernel, body = m.group(1), []
            kernels.add(kernel)
            continue
        if kernel is None:
            continue
        if ln.startswith(".Lfunc_end"):
            # walk the kernel body
            ops = [(i, instr.match(x)) for i, x in enumerate(body)]
            ops = [(i, mm.group(1), mm.group(2).split(";")[0]) for i, mm in ops if mm]
            for k, (i, op, args) in enumerate(ops):
                if not re.search(r"\bm0\b", args):
                    continue
                if op == "s_mov_b32" and args.strip().startswith("m0,"):
                    nxt = next(((o, a) for _, o, a in ops[k + 1:k + 4] if o != "s_nop"), None)
                    prv = next(((o, a) for _, o, a in reversed(ops[max(0, k - 3):k]) if o != "s_nop"), None)
                    if nxt and nxt[0] == "global_load_lds_dwordx4":
                        groups += 1                                   # head of an LDS-DMA group (glds16_m0), or the set of a save / set / restore piece
                        continue
                    if prv and prv[0] == "global_load_lds_dwordx4":
                        continue                                      # the restore of a save / set / restore piece (glds16_s, glds16)
                elif op == "s_mov_b32" and re.fullmatch(r"s\d+, m0", args.strip()):
                    continue                                          # the save of such a piece
                bad.append(f"{kernel}: {op} {args.strip()}")
            kernel = None
            continue
        body.append(ln)
    if not kernels:
        print("check_isa: no corr_volume_split_stream kernel found in the ISA")
        return 2
    if bad:
        print("check_isa: M0 is touched outside the LDS-DMA groups:\n  " + "\n  ".join(bad[:20]))
        return 1
    print(f"check_isa OK: {len(kernels)} split kernels, {groups} M0 writes, every one an LDS-DMA group head; no other M0 user")
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python
"""Register / LDS / scratch budget of every kernel in the built library, read from the code objects' own metadata (no GPU needed).

    python tools/kernel_resources.py            # table: kernel, VGPRs, AGPRs, SGPRs, static LDS, scratch bytes, spills, max workgroup
    python tools/kernel_resources.py --check    # the budgets DESIGN.md's schedule arguments rest on; non-zero exit + the offending rows otherwise

How: `mac-vo_amd/csrc/build/*.o` (left by `make`) each carry their gfx950 code object in `.hip_fatbin`; llvm-objcopy dumps the section,
clang-offload-bundler unbundles the `hipv4-amdgcn-amd-amdhsa--gfx950` entry, llvm-readelf --notes prints the AMDGPU metadata.  Run by
`__graft_entry__.build()` (`--check`) and by `tests/test_build_checks.py`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "mac-vo_amd", "csrc", "build")
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ARCH = os.environ.get("ARCH", "gfx950")      # the Makefile's ARCH= (gfx950 is the only target this library is written for)
TARGET = "hipv4-amdgcn-amd-amdhsa--" + ARCH
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "vgpr_spill_count", "sgpr_spill_count",
          "max_flat_workgroup_size")


def demangle(names):
    try:
        # (binutils' c++filt does not know the _Float16 mangling DF16_: hand it the IEEE-half one, Dh)
        r = subprocess.run(["c++filt"], input="\n".join(n.replace("DF16_", "Dh") for n in names), capture_output=True, text=True)
        out = r.stdout.splitlines() if r.returncode == 0 and r.stdout else names
    except OSError:                                # no binutils: the mangled names still carry the kernel's identifier
        out = [re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", n) for n in names]
    clean = []
    for d in out:
        d = re.sub(r"\(anonymous namespace\)::", "", d)
        d = re.sub(r"^void ", "", d)
        d = re.sub(r"\(.*$", "", d)            # drop the argument list
        clean.append(d)
    return clean


def kernels_of(obj: str, tmp: str) -> list:
    fat, co = os.path.join(tmp, "x.fat"), os.path.join(tmp, "x.co")
    if subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj], capture_output=True).returncode != 0:
        return []
    r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        return []
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    rows, cur = [], None
    for ln in notes.splitlines():
        m = re.match(r"^\s+(- )?\.(\w+):\s+(.*)$", ln)
        if not m:
            continue
        if m.group(1) and m.group(2) in ("agpr_count", "args"):     # a new kernel entry starts with its first key
            cur = {}
            rows.append(cur)
        if cur is None:
            continue
        key, val = m.group(2), m.group(3).strip()
        if key == "name" and "name" not in cur:                       # (argument entries carry `.name` too: the kernel's comes at entry level, after them)
            cur["name"] = val
        elif key == "name":
            cur["name"] = val
        elif key in FIELDS:
            cur[key] = int(val)
    rows = [r_ for r_ in rows if "vgpr_count" in r_ and "name" in r_]
    for r_, d in zip(rows, demangle([r_["name"] for r_ in rows])):
        r_["kernel"] = d
        r_["file"] = os.path.basename(obj)[:-2]
    return rows


def collect() -> list:
    if not os.path.isdir(BUILD):
        raise SystemExit("kernel_resources: mac-vo_amd/csrc/build is missing — run make -C mac-vo_amd/csrc first")
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in sorted(os.listdir(BUILD)):
            if o.endswith(".o"):
                rows += kernels_of(os.path.join(BUILD, o), tmp)
    return rows


# (kernel-name regex, predicate on the row, what the budget is for).  A SIMD has 512 VGPR+AGPR lanes-worth of registers per lane slot and a CU 160 KB of LDS.
def regs(r):
    return r["vgpr_count"]      # gfx90a+ unified file: .vgpr_count is the wave's whole allocation (architectural + accumulation registers)


CHECKS = [
    (r"^corr_volume_split_stream<", lambda r: r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0 and regs(r) <= 512,
     "persistent split GEMM: one wave per SIMD, no scratch"),
    (r"^corr_volume_split_stream<2, true", lambda r: regs(r) <= 304,
     "f16x2 GEMM: a wave holds <= 304 of its SIMD's 512 registers — 208 stay for the kernels that run beside it (lookups 33, selector 34-117, backend front 53)"),
    (r"^cost_patch_embed_pipelined_kernel<", lambda r: r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0 and regs(r) <= 256,
     "pipelined patch embedding: 512-thread workgroup = two waves per SIMD, no spills"),
    (r"^cost_patch_embed_strip_kernel<\d+, \d+, \d+, (true|false), (true|false), true,", lambda r: r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0,
     "strip-mined patch embedding with 16-bit cells in (the Fast-mode form): no spills (the fp32-in forms at 90 / 96 x 160 spill 20-26 registers: DESIGN section 4)"),
    (r"^corr_lookup_kernel<4, 2, 16", lambda r: r["private_segment_fixed_size"] == 0 and regs(r) <= 64,
     "one-frame lookup: 8-wave workgroups fit beside a GEMM wave several times over (<= 64 registers)"),
    (r"^corr_lookup_kernel<4, 4, 32", lambda r: r["private_segment_fixed_size"] == 0 and regs(r) <= 64, "batched lookup: eight waves per SIMD"),
    (r"^convex_upsample_kernel<", lambda r: r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0 and regs(r) <= 128,
     "upsampling: every load of a wave in flight without spills, four waves per SIMD"),
    (r"^backend_front_kernel<", lambda r: r["private_segment_fixed_size"] == 0 and regs(r) <= 128, "backend front: fits beside a GEMM wave"),
    (r"^pgo_solve_kernel<\d, 4>", lambda r: r["private_segment_fixed_size"] == 0 and r["vgpr_spill_count"] == 0,
     "LM solve (4-wave form): one point per thread in registers, no spills (430-504 registers + 77 KB of LDS: it needs a CU without a GEMM workgroup)"),
    (r"^kp_nms_kernel<|^kp_finish_kernel<(1024, 16, 5|512, 16, 10)", lambda r: r["private_segment_fixed_size"] == 0,
     "selector at 640 x 480: no scratch (the 24-rows-per-thread finishing variant for larger images spills 61-74 registers)"),
]


def main() -> int:
    rows = collect()
    if not rows:
        print("kernel_resources: no kernels found (llvm tools missing?)")
        return 2
    if "--check" in sys.argv:
        bad, seen = [], set()
        for pat, ok, why in CHECKS:
            hit = [r for r in rows if re.search(pat, r["kernel"])]
            if not hit:
                bad.append(f"no kernel matches {pat!r} ({why})")
            for r in hit:
                seen.add(pat)
                if not ok(r):
                    bad.append(f"{r['kernel']}: VGPR {r['vgpr_count']} AGPR {r.get('agpr_count', 0)} scratch {r['private_segment_fixed_size']} B spills "
                               f"{r['vgpr_spill_count']} — {why}")
        if bad:
            print("kernel_resources: budget violated:\n  " + "\n  ".join(bad))
            return 1
        print(f"kernel_resources OK: {len(rows)} kernels, {len(CHECKS)} budgets hold")
        return 0
    print(f"{'file':22s} {'REGS':>4s} {'AGPR':>4s} {'SGPR':>4s} {'LDS(static)':>11s} {'scratch':>7s} {'spill':>5s} {'maxWG':>5s}  kernel")
    for r in sorted(rows, key=lambda r: (r["file"], r["kernel"])):
        print(f"{r['file']:22s} {r['vgpr_count']:4d} {r.get('agpr_count', 0):4d} {r['sgpr_count']:4d} {r.get('group_segment_fixed_size', 0):11d} "
              f"{r['private_segment_fixed_size']:7d} {r['vgpr_spill_count']:5d} {r.get('max_flat_workgroup_size', 0):5d}  {r['kernel'][:150]}")
    return 0


if __name__ == "__main__":
    sys.exit(main())

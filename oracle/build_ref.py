#!/usr/bin/env python
"""Recipe for ``oracle/_ref/pyref``: the reference's Python tree BYTE-COMPILED from the sources where they lie under ``/root/reference``.

The reference is Python: its "binary" is CPython bytecode.  This script compiles every ``.py`` of the packages the hot path's caller needs
(``Odometry/ Module/ Utility/ DataLoader/ Evaluation/``) with ``py_compile`` into sourceless ``.pyc`` files under ``oracle/_ref/pyref/`` — outputs only
there, nothing else is written; no reference SOURCE enters this repository (``oracle/_ref/`` is git-ignored, but not gpurun-ignored, so the artefact
travels to the GPU box like the built ``libmacvo_hip.so``).  ``/root/reference`` does not exist on the GPU box; with this tree

  * ``tests/refrun.py`` runs the reference's own unmodified ``Odometry/MACVO.py`` loop there (``--mode ref`` on the host cores, ``--mode hip`` with only
    the ``type:`` strings swapped to the HIP plugins), and
  * ``bench.py`` times that loop as ``cpu_baseline`` with ``kind: "reference"``.

It is test / measurement infrastructure: nothing under ``mac-vo_amd/`` imports it.  Called by ``__graft_entry__.build()`` when ``/root/reference``
exists; a no-op otherwise (the GPU box only uses the prebuilt files).

    python oracle/build_ref.py
"""
import os
import py_compile
import shutil
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "pyref")
PACKAGES = ("Odometry", "Module", "Utility", "DataLoader", "Evaluation")


def build(verbose: bool = False) -> int:
    if not os.path.isdir(os.path.join(REF, "Odometry")):
        return 0
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    n = 0
    for pkg in PACKAGES:
        for dirpath, dirnames, filenames in os.walk(os.path.join(REF, pkg)):
            dirnames[:] = [d for d in dirnames if d != "__pycache__"]
            for fn in filenames:
                if not fn.endswith(".py"):
                    continue
                src = os.path.join(dirpath, fn)
                rel = os.path.relpath(src, REF)
                dst = os.path.join(OUT, rel + "c")                       # legacy (sourceless) location: pkg/mod.pyc
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                try:
                    py_compile.compile(src, cfile=dst, dfile=rel, doraise=True,
                                       invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
                    n += 1
                except py_compile.PyCompileError as e:                   # a file of an optional subsystem that does not parse here
                    if verbose:
                        print("skipped", rel, e.msg.splitlines()[-1])
    with open(os.path.join(OUT, "README"), "w") as f:
        f.write(f"byte-compiled from {REF} by oracle/build_ref.py with python {sys.version.split()[0]}; {n} modules; not source, not tracked\n")
    return n


if __name__ == "__main__":
    print(build(verbose=True), "modules ->", OUT)

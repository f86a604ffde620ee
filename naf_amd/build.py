"""Build libnaf_hip.so (gfx950) in-tree with hipcc.  ``python -m naf_amd.build [--force]``.

hipcc cross-compiles for gfx950 without a GPU.  One object per translation unit, compiled in
parallel, then one ``hipcc -shared`` link.  The library keeps a plain DT_NEEDED on libamdhip64.so.7
and NO rpath: when loaded after ``import torch`` the loader reuses the HIP runtime torch already
mapped (same SONAME), so stream handles and device pointers are shared with torch.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OBJDIR = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libnaf_hip.so")
ARCH = "gfx950"
# MFMA results in VGPRs (no v_accvgpr_read per result register) wherever the kernel does not need the AGPR half
# of the register file; the weight-stationary 3x3 stem layer fills all 512 registers and keeps the default.
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
NO_VGPR_FORM = {"stem_conv.hip"}
# Per-file flags.  stem_conv.hip: the schedule's GroupNorm / SiLU / sums arithmetic is written as plain f32 operations on purpose
# -- a packed v_pk_*_f32 beside an MFMA stalls the matrix pipe ~16 cycles (profiles/r03_mfma_filler_prices.txt) -- and the SLP
# vectoriser would pack adjacent ones again.
EXTRA_FLAGS = {"stem_conv.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    hdrs.append(os.path.join(INCLUDE, "naf_hip.h"))
    hdrs.append(os.path.abspath(__file__))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool, hdr_mtime: float, extra) -> str:
    obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
    spath = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(spath), hdr_mtime):
        return obj
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall",
           "-Wno-unused-function", f"-I{INCLUDE}", f"-I{CSRC}", *([] if src in NO_VGPR_FORM else VGPR_FORM), *EXTRA_FLAGS.get(src, []), *extra, "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return obj


def build_library(force: bool = False, verbose: bool = False, extra_flags=()) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    hdr_mtime = _deps_mtime()
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_mtime, list(extra_flags)), srcs))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB) from {len(objs)} objects")
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv, verbose=True)

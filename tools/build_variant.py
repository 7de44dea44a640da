#!/usr/bin/env python3
"""Build a variant of libmvd_hip.so with extra -D flags for same-box A/B runs:
   python tools/build_variant.py base -DMVD_GEMM_NO_SWP   ->  mvd_amd/libmvd_hip_base.so
   python tools/build_variant.py probe -DMVD_PROBE        ->  probe build: MVD_* environment switches, a.dbg ablations and
                                                              the gemm_ring.hip experiment (force_cfg 15) are compiled in
   MVD_HIP_LIB=mvd_amd/libmvd_hip_base.so python bench.py ..."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mvd_amd import _build as B
tag, defs = sys.argv[1], sys.argv[2:]
objdir = os.path.join(B.CSRC, "build_" + tag)
os.makedirs(objdir, exist_ok=True)
def cc(src):
    o = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
    subprocess.run([B.HIPCC, *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, src), "-o", o], check=True)
    return o
with ThreadPoolExecutor(4) as ex:
    srcs = B.SOURCES + (B.PROBE_SOURCES if "-DMVD_PROBE" in defs else [])
    objs = list(ex.map(cc, srcs))
out = os.path.join(ROOT, "mvd_amd", f"libmvd_hip_{tag}.so")
subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
print(out)

#!/usr/bin/env python3
"""Disassemble the gfx950 code objects inside a built library and count instruction classes the product must not hold.

    python tools/lint_device_isa.py [mvd_amd/libmvd_hip.so]

Class checked: packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).  Round 4 traced run-to-run differences on a
shared GPU to `v_pk_fma_f32 ... op_sel:[0,1,0]`; the product is built with packed fp32 selection off
(mvd_amd/_build.py NO_PACKED_FP32) and tests/test_build_cpu.py asserts through count_packed_fp32() that none is left.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("MVD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\bv_pk_(fma|mul|add)_f32\b(.*)")


def code_objects(lib: str, tmp: str):
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    offs = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for i, o in enumerate(offs):
        part = os.path.join(tmp, f"bundle{i}.bin")
        open(part, "wb").write(blob[o:offs[i + 1] if i + 1 < len(offs) else len(blob)])
        co = os.path.join(tmp, f"dev{i}.co")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        yield co


def count_packed_fp32(lib: str):
    """-> {"code_objects": n, "instructions": total, "packed_fp32": n, "packed_fp32_low_half_select": n}"""
    res = {"code_objects": 0, "instructions": 0, "packed_fp32": 0, "packed_fp32_low_half_select": 0}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            res["code_objects"] += 1
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
            for line in dis.split("\n"):
                if "\t" not in line or line.rstrip().endswith(":"):
                    continue
                res["instructions"] += 1
                m = PK.search(line)
                if m:
                    res["packed_fp32"] += 1
                    if "op_sel:[" in m.group(2):
                        res["packed_fp32_low_half_select"] += 1
    return res


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    print(count_packed_fp32(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "mvd_amd", "libmvd_hip.so")))

#!/usr/bin/env python3
"""Build-time checks of the hand-scheduling invariants of the shipped gfx950 kernels (no GPU needed).

    python tools/lint_device_isa.py [mvd_amd/libmvd_hip.so]        # prints a JSON summary, exit code 1 on any violation

Run by ``__graft_entry__.build()`` and by ``tests/test_round3_cpu.py``.  The invariants (DESIGN.md section 0 lists them with the
reason for each) are properties of the COMPILER'S OUTPUT that the sources cannot state in C++ and that a hipcc update could
silently undo:

 I1  no packed fp32 arithmetic (v_pk_{fma,mul,add}_f32) in any shipped code object (mvd_amd/_build.py NO_PACKED_FP32: a precaution
     against an unexplained fault localised to `v_pk_fma_f32 ... op_sel:[0,1,0]`).
 I2  no scratch memory: every shipped kernel has private_segment_fixed_size 0, no VGPR spills and no `scratch_` instruction
     (a scratch access counts on vmcnt and would break every hand-counted `s_waitcnt vmcnt(N)` of the LDS-DMA rings).
 I3  conv_ws.hip: the LDS read of a weight-ring slot has RETIRED before the slot is refilled.  Every `MVD_REFILL_FENCE v[a:b]`
     marker in the compiler's assembly (an empty asm statement with a memory clobber that names the fragment registers; the
     LDS-DMA that refills the slot cannot move above it) must find the `ds_read_b128 v[a:b]` that filled those registers already
     retired by an `s_waitcnt lgkmcnt(N)` (LDS operations of a wave return in order).  Round 4's refill-before-read race.
 I4  attention.hip: an MFMA issued from an asm statement (`mfma_from`: D = A.B + C with a live C tile; the hazard recognizer
     does not see an MFMA inside an asm statement) is directly preceded, inside the statement, by its own `s_nop 1`.
 I5  a 16-byte buffer store with an SGPR soffset is followed by >= 2 wait states before anything writes its data registers
     (gfx950 hazard hipcc does not cover; `store16()` carries an `s_nop 1`).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("MVD_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


def _instructions(dis_line: str):
    """'\\tv_add_u32_e32 v1, v2, v3   // 0000: ...' -> 'v_add_u32_e32 v1, v2, v3' (None for labels / blanks)."""
    if "\t" not in dis_line or dis_line.rstrip().endswith(":"):
        return None
    return dis_line.split("//")[0].strip() or None


def disassemble(co: str):
    """-> {kernel symbol: [instruction, ...]} of one code object."""
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        ins = _instructions(line)
        if ins is not None and cur is not None:
            cur.append(ins)
    return out


def kernel_metadata(co: str):
    """-> [{name, private_segment_fixed_size, sgpr_spill_count, vgpr_spill_count, vgpr_count, agpr_count}] from the notes."""
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels, cur = [], None
    for line in txt.split("\n"):
        m = re.match(r"^\s+(?:- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if line.lstrip().startswith("- .") and key in ("agpr_count", "args"):     # first key of a kernel record
            cur = {}
            kernels.append(cur)
        if cur is not None and key in ("name", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_spill_count", "vgpr_count",
                                       "agpr_count", "sgpr_count"):
            cur[key] = val if key == "name" else int(val)
    return [k for k in kernels if "name" in k and "private_segment_fixed_size" in k]


def _regs(op: str):
    """'v[4:7]' -> ('v', 4, 7); 'v12' -> ('v', 12, 12); 'a[0:15]' -> ('a', 0, 15); anything else -> None."""
    m = re.match(r"^([va])\[(\d+):(\d+)\]$", op) or re.match(r"^([va])(\d+)()$", op)
    if not m:
        return None
    lo = int(m.group(2))
    return m.group(1), lo, int(m.group(3)) if m.group(3) else lo


def _operands(ins: str):
    parts = ins.split(None, 1)
    return (parts[0], [o.strip() for o in parts[1].split(",")]) if len(parts) == 2 else (parts[0], [])


def _writes(ins: str):
    """Vector registers an instruction writes (first operand of VALU / MFMA / loads; stores and compares write none)."""
    op, ops = _operands(ins)
    if not ops or op.startswith(("buffer_store", "global_store", "ds_write", "scratch_store", "v_cmp", "s_", "buffer_atomic",
                                 "global_atomic")) or (op.startswith("buffer_load") and "lds" in ops[-1].split()):
        return None
    if op.startswith(("v_", "ds_read", "buffer_load", "global_load", "ds_bpermute", "ds_swizzle", "scratch_load")):
        return _regs(ops[0])
    return None


def _overlap(a, b):
    return a and b and a[0] == b[0] and a[1] <= b[2] and b[1] <= a[2]


# ---------------------------------------------------------------------------------------------------------------- I1
def count_packed_fp32(lib: str):
    """-> {"code_objects": n, "instructions": total, "packed_fp32": n, "packed_fp32_low_half_select": n}"""
    res = {"code_objects": 0, "instructions": 0, "packed_fp32": 0, "packed_fp32_low_half_select": 0}
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            res["code_objects"] += 1
            for ins_list in disassemble(co).values():
                for ins in ins_list:
                    res["instructions"] += 1
                    m = PK.search(ins)
                    if m:
                        res["packed_fp32"] += 1
                        if "op_sel:[" in m.group(2):
                            res["packed_fp32_low_half_select"] += 1
    return res


# ---------------------------------------------------------------------------------------------------------------- I2, I4, I5
def check_asm_mfma(asm_text: str, min_blocks: int = 1):
    """I4 on the compiler's assembly text (asm statements are bracketed by ;;#ASMSTART / ;;#ASMEND there): every MFMA issued from
    an asm statement is directly preceded, inside the statement, by an s_nop of >= 2 wait states -> (blocks, [violations])."""
    bad, nb, inside, prev = [], 0, False, ""
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            inside, prev = True, ""
        elif line.startswith(";;#ASMEND"):
            inside = False
        elif inside and line and not line.startswith(";"):
            if line.startswith("v_mfma"):
                nb += 1
                m = re.match(r"^s_nop (\d+)", prev)
                if not m or int(m.group(1)) < 1:
                    bad.append(f"line {ln}: `{line}` inside an asm statement is preceded by `{prev}`, not by `s_nop 1`")
            prev = line
    if nb < min_blocks:
        bad.append(f"only {nb} asm-issued MFMAs found (expected >= {min_blocks}): attention.hip mfma_from changed")
    return nb, bad


def check_store_hazard(ins_list, where, horizon=12):
    """I5: buffer_store_dwordx4 <data>, <voff>, <rsrc>, s<N> offen: >= 2 wait states before a write to <data>."""
    bad = []
    for i, ins in enumerate(ins_list):
        op, ops = _operands(ins)
        if op != "buffer_store_dwordx4" or len(ops) < 4 or not re.match(r"^s\d+\b", ops[3]):
            continue
        data, states = _regs(ops[0]), 0
        for nxt in ins_list[i + 1:i + 1 + horizon]:
            if nxt.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_barrier")):
                break
            if _overlap(_writes(nxt), data):
                if states < 2:
                    bad.append(f"{where}: `{ins}` then `{nxt}` after {states} wait state(s): its data registers are rewritten too early")
                break
            m = re.match(r"^s_nop (\d+)", nxt)
            states += int(m.group(1)) + 1 if m else 1
            if states >= 2:
                break
    return bad


def check_library(lib: str, scratch_allowed=()):
    """I1, I2, I5 over every code object of a built library (or object file) -> (summary dict, [violation strings])."""
    summary = {"code_objects": 0, "kernels": 0, "instructions": 0, "packed_fp32": 0, "soffset_stores": 0}
    bad = []
    with tempfile.TemporaryDirectory() as tmp:
        for co in code_objects(lib, tmp):
            summary["code_objects"] += 1
            for k in kernel_metadata(co):
                if k["name"] in scratch_allowed:
                    continue
                if k["private_segment_fixed_size"] or k.get("vgpr_spill_count"):     # (SGPR spills go to VGPR lanes: no memory)
                    bad.append(f"{k['name']}: scratch {k['private_segment_fixed_size']} B, spills sgpr {k.get('sgpr_spill_count')} "
                               f"vgpr {k.get('vgpr_spill_count')}")
            for name, ins_list in disassemble(co).items():
                summary["kernels"] += 1
                summary["instructions"] += len(ins_list)
                for ins in ins_list:
                    if PK.search(ins):
                        summary["packed_fp32"] += 1
                        bad.append(f"{name}: packed fp32 arithmetic `{ins}`")
                    if ins.startswith("scratch_") and name not in scratch_allowed:
                        bad.append(f"{name}: scratch access `{ins}`")
                    if ins.startswith("buffer_store_dwordx4") and re.search(r", s\d+ offen", ins):
                        summary["soffset_stores"] += 1
                bad += check_store_hazard(ins_list, name)
    return summary, bad


# ---------------------------------------------------------------------------------------------------------------- I3
def device_assembly(src: str, extra_flags=()):
    """The compiler's gfx950 assembly of one source with the product's flags (comments of asm statements survive here)."""
    sys.path.insert(0, HERE)
    from mvd_amd import _build
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "dev.s")
        r = subprocess.run([_build.HIPCC, *_build.FLAGS, *extra_flags, "-S", "--cuda-device-only", src, "-o", out],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return open(out).read()


LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def check_refill_fences(asm_text: str, min_fences: int = 1):
    """I3 on assembly text -> (number of fences checked, [violations]).  Linear scan per function: LDS operations (ds_*) enter a
    FIFO, `s_waitcnt ... lgkmcnt(N)` leaves the youngest N in it (in-order return), scalar memory loads make the count unusable
    until the next lgkmcnt(0) (they return out of order) and are reported if one is in flight at a fence."""
    bad, nf = [], 0
    fifo, last_write, smem, fn = [], {}, False, "?"
    for ln, raw in enumerate(asm_text.split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$", line)
        if m and not line.startswith(".L"):
            fn, fifo, last_write, smem = m.group(1), [], {}, False
            continue
        if line.startswith("; MVD_REFILL_FENCE"):
            nf += 1
            reg = line.split()[-1]
            rid = last_write.get(reg)
            if rid is None:
                bad.append(f"{fn}:{ln}: fence names {reg} but no ds_read filled it in this function")
            elif rid in fifo:
                bad.append(f"{fn}:{ln}: the ds_read of {reg} (line {rid}) has not retired at its refill fence "
                           f"({len(fifo) - fifo.index(rid)} LDS operations in flight, no lgkmcnt wait covers it)")
            elif smem:
                bad.append(f"{fn}:{ln}: a scalar load is in flight at the fence of {reg}: lgkmcnt cannot prove the read retired")
            continue
        if not line or line.startswith((";", ".", "//")):
            continue
        op = line.split()[0]
        if op.startswith("ds_"):
            fifo.append(ln)
            if op.startswith("ds_read"):
                last_write[line.split(None, 1)[1].split(",")[0].strip()] = ln
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            smem = True
        elif op == "s_waitcnt":
            m = LGKM.search(line)
            if m:
                n = int(m.group(1))
                fifo = fifo[len(fifo) - n:] if n else []
                if n == 0:
                    smem = False
    if nf < min_fences:
        bad.append(f"only {nf} MVD_REFILL_FENCE markers found (expected >= {min_fences}): the pin is gone from the source or the output")
    return nf, bad


def check_conv_ws_fences():
    return check_refill_fences(device_assembly(os.path.join(HERE, "mvd_amd", "csrc", "conv_ws.hip")), min_fences=50)


def lint_all(lib: str):
    summary, bad = check_library(lib)
    nf, bad_f = check_conv_ws_fences()
    nb, bad_m = check_asm_mfma(device_assembly(os.path.join(HERE, "mvd_amd", "csrc", "attention.hip")))
    summary["conv_ws_refill_fences"], summary["asm_issued_mfma"] = nf, nb
    summary["violations"] = len(bad) + len(bad_f) + len(bad_m)
    return summary, bad + bad_f + bad_m


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "mvd_amd", "libmvd_hip.so")
    summary, bad = lint_all(lib)
    print(json.dumps(summary))
    for b in bad[:50]:
        print("VIOLATION:", b, file=sys.stderr)
    sys.exit(1 if bad else 0)

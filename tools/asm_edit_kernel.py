#!/usr/bin/env python3
"""Edit ONE kernel inside a hipcc device assembly file (tools/build_misc_asm_variant.sh; conv_out four-pixel diagnosis).

    asm_edit_kernel.py <file.s> <kernel-name-regex> <edit>

<edit> is either a `sed -E` script applied to the kernel's body, or `py:<file>` naming a python file with
edit(body, kernel_name, text_after) -> (body, text_after, n_rewritten); text_after holds the kernel descriptor and metadata.
"""
import importlib.util
import re
import subprocess
import sys


def main():
    path, kre, edit = sys.argv[1], sys.argv[2], sys.argv[3]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"_Z\d+" + kre + r".*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")) - 1
    name = lines[start].split(":")[0]
    body, after = "\n".join(lines[start:end + 1]), "\n".join(lines[end + 1:])
    if edit.startswith("py:"):
        spec = importlib.util.spec_from_file_location("asm_edit", edit[3:])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        new, after, n = mod.edit(body, name, after)
        print(f"{name}: {n} instructions rewritten", file=sys.stderr)
    else:
        new = subprocess.run(["sed", "-E", edit], input=body, capture_output=True, text=True, check=True).stdout.rstrip("\n")
    print(f"{name}: {end + 1 - start} -> {len(new.splitlines())} lines", file=sys.stderr)
    open(path, "w").write("\n".join(lines[:start]) + "\n" + new + "\n" + after)


if __name__ == "__main__":
    main()

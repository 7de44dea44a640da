"""Assembly edit for tools/build_misc_asm_variant.sh (conv_out four-pixel diagnosis).

Rewrites every   v_pk_fma_f32 D, A, v[b:b+1], C op_sel:[0,1,0]   (both result halves multiply by the HIGH register of src1)
into             v_mov_b32 v180, v(b+1) ; s_nop 0 ; v_pk_fma_f32 D, A, v[180:181], C op_sel_hi:[1,0,1]
(both halves multiply by the LOW register of a scratch pair), i.e. the same arithmetic without a non-default low-half select,
and grows the kernel's VGPR budget by the scratch pair.  edit(body, rest) -> (body, rest).
"""
import re

PAT = re.compile(r"^(\s*)v_pk_fma_f32 (v\[\d+:\d+\]), (v\[\d+:\d+\]), v\[(\d+):(\d+)\], (v\[\d+:\d+\]) op_sel:\[0,1,0\]\s*$")


def edit(body, name, text_after):
    out, n = [], 0
    for line in body.split("\n"):
        m = PAT.match(line)
        if not m:
            out.append(line)
            continue
        ind, d, a, _lo, hi, c = m.groups()
        out += [f"{ind}v_mov_b32_e32 v180, v{hi}", f"{ind}s_nop 0", f"{ind}v_pk_fma_f32 {d}, {a}, v[180:181], {c} op_sel_hi:[1,0,1]"]
        n += 1
    # kernel descriptor + metadata of this kernel: 178 -> 182 VGPRs, AGPR offset 180 -> 184
    sep = "\n@@BODY_END@@\n"
    text = "\n".join(out) + sep + text_after
    i = text.index(f".amdhsa_kernel {name}")
    j = text.index(".end_amdhsa_kernel", i)
    blk = text[i:j].replace(".amdhsa_next_free_vgpr 178", ".amdhsa_next_free_vgpr 182").replace(".amdhsa_accum_offset 180", ".amdhsa_accum_offset 184")
    assert blk != text[i:j]
    text = text[:i] + blk + text[j:]
    assert f".set {name}.num_vgpr, 178" in text
    text = text.replace(f".set {name}.num_vgpr, 178", f".set {name}.num_vgpr, 182")
    k = text.index(f".name:           {name}")
    k2 = text.index(".vgpr_count:", k)
    k3 = text.index("\n", k2)
    text = text[:k2] + ".vgpr_count:     182" + text[k3:]
    body, text_after = text.split(sep)
    return body, text_after, n

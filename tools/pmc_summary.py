#!/usr/bin/env python3
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into per-kernel-class HBM traffic.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads -> doubled; WRITE_SIZE is exact for 16-B stores; both are in KB."""
import collections, csv, glob, json, re, sys

CLASSES = {
    "gemm_pp_256x320_dense": r"gemm_pp_kernel<256, 2, 4, 0, false, false>", "gemm_pp_256x320_ln_dense": r"gemm_pp_kernel<256, 2, 4, 0, false, true>",
    "gemm_pp_256x320_conv3x3": r"gemm_pp_kernel<256, 2, 4, [123], false, false>",
    "gemm_pp_256x320_splitk": r"gemm_pp_kernel<256, 2, 4, \d, true, false>", "gemm_pp_256x320_geglu": r"gemm_pp_kernel<256, 4, 2, 0, false, (false|true)>",
    "gemm_sm_64x64": "gemm_sm_kernel<64, 64,", "gemm_sm_128x64": "gemm_sm_kernel<128, 64,", "gemm_sm_64x128": "gemm_sm_kernel<64, 128,",
    "gemm_sm_128x128": "gemm_sm_kernel<128, 128,", "gemm_sm_64x160": "gemm_sm_kernel<64, 160,", "gemm_sm_128x160": "gemm_sm_kernel<128, 160,",
    "gemm_sm_64x320": "gemm_sm_kernel<64, 320,",
    "gemm_xs_dense": "gemm_xs_kernel<20, false, false, false>", "gemm_xs_residual": "gemm_xs_kernel<20, false, true, false>",
    "gemm_xs_ln_dense": "gemm_xs_kernel<20, false, false, true>", "gemm_xs_geglu": "gemm_xs_kernel<20, true, false,",
    "gemm_128x160": "Cfg<128, 160, 2, 2>",
    "gemm_128x128": "Cfg<128, 128, 2, 2>", "gemm_128x64": "Cfg<128, 64, 2, 2>", "gemm_64x64": "Cfg<64, 64, 2, 2>",
    "attn_4wave": "attn_kernel<4,", "attn_8wave": "attn_kernel<8,", "attn_2wave": "attn_kernel<2,", "attn_1wave": "attn_kernel<1,",
    "groupnorm": "gn_", "layernorm": "ln_kernel",
}

def agg(path):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        d[r["Kernel_Name"]][0] += 1
        d[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return d

def main(fetch_dir, write_dir, out):
    f = agg(glob.glob(fetch_dir + "/**/*counter_collection.csv", recursive=True)[0])
    w = agg(glob.glob(write_dir + "/**/*counter_collection.csv", recursive=True)[0])
    res = {}
    for cls, pat in CLASSES.items():
        n = sum(v[0] for k, v in f.items() if re.search(pat if pat.startswith('gemm_pp') else re.escape(pat), k))
        if not n:
            continue
        fk = sum(v[1] for k, v in f.items() if re.search(pat if pat.startswith('gemm_pp') else re.escape(pat), k))
        wk = sum(v[1] for k, v in w.items() if re.search(pat if pat.startswith('gemm_pp') else re.escape(pat), k))
        nw = sum(v[0] for k, v in w.items() if re.search(pat if pat.startswith('gemm_pp') else re.escape(pat), k))
        res[cls] = {"launches": n, "fetch_bytes_per_launch": 2.0 * fk * 1024 / n, "write_bytes_per_launch": wk * 1024 / max(nw, 1),
                    "hbm_bytes_per_launch": 2.0 * fk * 1024 / n + wk * 1024 / max(nw, 1),
                    "note": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B) + WRITE_SIZE, KB -> bytes"}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_sha
    res["_meta"] = {"kernel_src_sha": kernel_source_sha(), "workload": "cfg4 cold",
                    "note": "bench.py quotes roofline.traffic from this file only while the kernel sources hash to kernel_src_sha"}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        if k.startswith("_"):
            continue
        print(f"{k:22s} launches={v['launches']:5d} fetch={v['fetch_bytes_per_launch']/1e6:9.1f} MB write={v['write_bytes_per_launch']/1e6:9.1f} MB")

if __name__ == "__main__":
    main(*sys.argv[1:4])

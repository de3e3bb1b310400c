#!/usr/bin/env python3
"""Instruction mix of the VALU-bound kernels from their gfx950 ISA: how many of a kernel's VALU instructions are 3-source VOP3
(v_bitop3_b32, v_perm_b32, v_add3_u32, v_alignbit_b32, v_lshl_or_b32 ...: half the issue rate of a VOP2 on this part, tools/mb/valu_mb.hip)
and the issue-fraction ceiling that mix allows: N / (N_full + 2 N_half).  Loop bodies are weighted by their trip counts where the
source fixes them (the cipher's middle rounds: 8 trips).  Needs hipcc (run in the build container); writes profiles/<tag>_valu_mix.json,
which bench.py quotes in roofline.
usage: python tools/valu_mix.py r06"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HALF = ("v_bitop3_b32", "v_perm_b32", "v_add3_u32", "v_alignbit_b32", "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_lshl_add_u32", "v_xad_u32",
        "v_mad_u32_u24", "v_mad_u64_u32", "v_lshl_add_u64", "v_bfe_u32", "v_bfe_i32", "v_bfi_b32", "v_cndmask_b32_e64", "v_add_lshl_u32", "v_alignbyte_b32")
QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32")


def asm_of(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "reverie_amd", "csrc"),
                           "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
    return open(out).read()


def functions(asm):
    """name -> list of (label or None, instruction) in order"""
    fns, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            fns[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        lm = re.match(r"^(\.LBB\w+):", line)
        if lm:
            cur.append((lm.group(1), None))
            continue
        im = re.match(r"^\s+([sv]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", line)
        if im:
            cur.append((None, (im.group(1), line.strip())))
    return fns


def mix(items, loop_weight):
    """loop_weight: {label: trips} -- instructions between a label and the backward branch to it count that many times"""
    # find loops: a label followed later by 's_cbranch* label'
    labels = {lab: i for i, (lab, ins) in enumerate(items) if lab}
    w = [1.0] * len(items)
    loops = []
    for i, (lab, ins) in enumerate(items):
        if ins and ins[0].startswith("s_cbranch"):
            t = ins[1].split()[-1]
            if t in labels and labels[t] < i:
                loops.append((labels[t], i))
    for k, (a, b) in enumerate(sorted(loops, key=lambda ab: ab[1] - ab[0], reverse=True)):
        trips = loop_weight[min(k, len(loop_weight) - 1)] if loop_weight else 1
        for j in range(a, b + 1):
            w[j] = max(w[j], trips)
    full = half = quarter = 0.0
    names = {}
    for (lab, ins), wt in zip(items, w):
        if not ins or not ins[0].startswith("v_"):
            continue
        op = ins[0]
        base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
        names[base] = names.get(base, 0) + wt
        if op in HALF or base in HALF:
            half += wt
        elif base in QUARTER:
            quarter += wt
        else:
            full += wt
    n = full + half + quarter
    return {"valu": n, "full_rate": full, "half_rate": half, "quarter_rate": quarter, "half_rate_share": (half + quarter) / n if n else 0,
            "issue_ceiling": n / (full + 2 * half + 4 * quarter) if n else 0, "loops": len(loops),
            "top": dict(sorted(names.items(), key=lambda kv: -kv[1])[:8])}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    out = {"note": "static gfx950 ISA of the shipped kernels, loop bodies weighted by their trip counts; issue_ceiling = N / (N_full + 2 N_half + 4 N_quarter): "
                   "the issue fraction a kernel with this mix reaches when a VALU instruction leaves every cycle the pipe allows", "kernels": {}}
    c4 = functions(asm_of(os.path.join(ROOT, "reverie_amd", "csrc", "aes_col4.hip")))
    for name, items in c4.items():
        if "k_aes_gf2_masks_col4" in name:
            # outer block loop (1 trip per CTR block: the unit) and the middle rounds' loop inside it (8 trips per block)
            out["kernels"]["rv::k_aes_gf2_masks_col4"] = mix(items, [1, 8])
    kk = functions(asm_of(os.path.join(ROOT, "reverie_amd", "csrc", "kernels.hip")))
    for name, items in kk.items():
        if "k_interp_fullILi2ELi64ELb1E" in name:
            out["kernels"]["rv::k_interp_full<2, 64, true>"] = mix(items, [1])
        if "k_b3_chunks_uni" in name and "bits" not in name and "rv::k_b3_chunks_uni" not in out["kernels"]:
            out["kernels"]["rv::k_b3_chunks_uni"] = mix(items, [1])
        if "k_b3_chunks_bits_uni" in name and "rv::k_b3_chunks_bits_uni" not in out["kernels"]:
            out["kernels"]["rv::k_b3_chunks_bits_uni"] = mix(items, [1])
    path = os.path.join(ROOT, "profiles", f"{tag}_valu_mix.json")
    json.dump(out, open(path, "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k}: {v['valu']:.0f} VALU (weighted), half-rate share {v['half_rate_share']:.2f}, issue ceiling {v['issue_ceiling']:.2f}, loops {v['loops']}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Basic blocks of one kernel in the ISA dump (make -C pathtracer_amd/csrc asm): per block the number of vector-ALU,
scalar and memory instructions, the issue cycles of its vector instructions (profiles/r02_valu_tput.txt through
isa_mix.COST) and a few opcodes that tell which part of the source it is (byte conversions: four-wide step; sqrt and
divide steps: sphere roots; mul_hi: the Philox rounds; ...).  Used to split a kernel's dynamic instruction count
between its parts together with the per-part issue tallies of the diagnostic build (make prof).

    python profiles/isa_blocks.py pathtracer_amd/csrc/pt_kernels_ring.gfx950.s '<mangled kernel name>' [first_line last_line]
"""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from isa_mix import COST, lookup  # noqa: E402

MARK = [("v_cvt_f32_ubyte", "ubyte"), ("v_sqrt_f32", "sqrt"), ("v_div_scale", "div"), ("v_mul_hi_u32", "mulhi"), ("v_rcp_f32", "rcp"),
        ("ds_write", "dsw"), ("ds_read", "dsr"), ("global_load", "gld"), ("global_store", "gst"), ("global_atomic", "gat"), ("s_cbranch", "br")]


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, len(body))
    blocks, cur = [], {"label": "entry", "line": 0, "v": 0, "s": 0, "m": 0, "cyc": 0.0, "marks": {}}
    for n, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):|^; %bb\.(\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1) or "bb." + m.group(2), "line": n + 1, "v": 0, "s": 0, "m": 0, "cyc": 0.0, "marks": {}}
            continue
        t = l.strip().split()
        if not t or t[0].startswith((";", ".")):
            continue
        op = t[0]
        if op.startswith("v_"):
            cur["v"] += 1
            cur["cyc"] += lookup(COST, op) or 4.2
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith(("ds_", "global_", "scratch_", "buffer_", "flat_")):
            cur["m"] += 1
        for pat, name in MARK:
            if op.startswith(pat):
                cur["marks"][name] = cur["marks"].get(name, 0) + 1
    blocks.append(cur)
    tv = ts = tm = 0
    tc = 0.0
    for b in blocks:
        if not (lo <= b["line"] <= hi):
            continue
        tv += b["v"]; ts += b["s"]; tm += b["m"]; tc += b["cyc"]
        print(f'{b["line"]:6d} {b["label"]:12s} valu {b["v"]:4d} salu {b["s"]:3d} mem {b["m"]:2d} cycles {b["cyc"]:7.1f}  ' + " ".join(f"{k}x{v}" for k, v in b["marks"].items()))
    print(f"total valu {tv} salu {ts} mem {tm} valu cycles {tc:.1f}")


if __name__ == "__main__":
    main()

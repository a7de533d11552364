#!/usr/bin/env python3
"""Static instruction mix of one kernel from the ISA dump (make -C pathtracer_amd/csrc asm), with the issue cost per
opcode measured by tools/mb/valu_tput.hip (profiles/r02_valu_tput.txt: cycles per wave-instruction per SIMD at 8 waves
per SIMD).  The counters of rocprofv3 give the DYNAMIC count of some classes (f32 add / mul / fma, transcendental,
conversion, 64-bit integer, 32-bit integer); everything else (moves, selects, compares, min / max, lane ops) is only
known as the remainder.  This script supplies what the counters cannot: the average issue cost of an instruction of
the 32-bit-integer class and of the remainder class, weighted by how often each opcode occurs in the kernel's code.

    python profiles/isa_mix.py pathtracer_amd/csrc/pt_kernels_ring.gfx950.s '_Z7k_pathsILb0ELb1ELi512ELb0ELi0EE' out.json
"""
import collections
import json
import re
import sys

# cycles per wave-instruction per SIMD (profiles/r02_valu_tput.txt), by opcode prefix; first match wins
COST = [
    (r"v_(rcp|sqrt|rsq|exp|log|sin|cos)_", 8.0),
    (r"v_(fma|fmac|mul|add|sub|subrev|mac|mad)_f32", 2.4),
    (r"v_pk_(fma|mul|add)_f32", 4.6),
    (r"v_cvt_", 4.2),
    (r"v_mad_u64_u32|v_mad_i64_i32|v_lshl_add_u64|v_lshlrev_b64|v_lshrrev_b64", 4.4),
    (r"v_(min|max|med)3?_", 4.3),
    (r"v_cmp_|v_cmpx_", 4.2),
    (r"v_div_(scale|fmas|fixup)_f32", 4.2),
    (r"v_cndmask_b32_e64", 4.3),
    (r"v_cndmask_b32", 2.4),
    (r"v_mov_b32|v_mov_b64|v_not_b32|v_bfrev_b32", 2.3),
    (r"v_(and|or|xor)_b32|v_bitop3_b32", 2.4),
    (r"v_(add|sub|subrev)_u32|v_(add|sub)_co_u32|v_addc_co_u32", 2.5),
    (r"v_lshrrev_b32|v_ashrrev_i32", 2.3),
    (r"v_lshlrev_b32", 4.1),
    (r"v_lshl_add_u32|v_lshl_or_b32|v_add3_u32|v_and_or_b32|v_or3_b32|v_add_lshl_u32|v_bfe_|v_bfi_|v_perm_b32|v_alignbit", 4.3),
    (r"v_mul_(lo|hi)_u32|v_mul_u32_u24|v_mad_u32_u24|v_mul_i32_i24|v_mad_i32_i24", 4.3),
    (r"v_mbcnt_", 4.2),
    (r"v_readlane_b32|v_readfirstlane_b32|v_writelane_b32", 4.2),
    (r"v_", 4.2),  # anything not listed: the common cost of three-operand encodings
]
# which dynamic counter class an opcode belongs to (as rocprofv3's SQ_INSTS_VALU_* count them)
CLASS = [
    (r"v_(rcp|sqrt|rsq|exp|log|sin|cos)_f32", "trans_f32"),
    (r"v_(fma|fmac|mad|mac)_f32|v_pk_fma_f32", "fma_f32"),
    (r"v_mul_f32|v_pk_mul_f32", "mul_f32"),
    (r"v_(add|sub|subrev)_f32|v_pk_add_f32", "add_f32"),
    (r"v_cvt_", "cvt"),
    (r"v_mad_u64_u32|v_mad_i64_i32|v_lshl_add_u64|v_lshlrev_b64|v_lshrrev_b64", "int64"),
    (r"v_(add|sub|subrev)_u32|v_(add|sub)_co_u32|v_addc_co_u32|v_(and|or|xor|not)_b32|v_bitop3_b32|v_lsh|v_ashr|v_add3_u32|v_and_or_b32|v_or3_b32|"
     r"v_add_lshl_u32|v_bfe_|v_bfi_|v_mul_(lo|hi)_u32|v_mul_u32_u24|v_mad_u32_u24|v_mul_i32_i24|v_mad_i32_i24|v_(min|max)3?_[iu]32|v_mbcnt_|v_bfrev", "int32"),
    (r"v_", "other"),
]


def lookup(table, op):
    for pat, val in table:
        if re.match(pat, op):
            return val
    return None


def main():
    path, key, out = sys.argv[1], sys.argv[2], sys.argv[3]
    ops = collections.Counter()
    inside = False
    for line in open(path):
        if not inside:
            inside = line.startswith(key) and ":" in line  # the label line of the kernel
            continue
        if line.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s+(v_[a-z0-9_]+)", line)
        if m:
            ops[m.group(1)] += 1
    by_class = collections.defaultdict(lambda: [0, 0.0])
    for op, n in ops.items():
        c = lookup(CLASS, op)
        by_class[c][0] += n
        by_class[c][1] += n * lookup(COST, op)
    res = {"kernel": key, "static_valu_instructions": sum(ops.values()),
           "classes": {c: {"static_count": n, "mean_issue_cycles": round(cy / n, 3)} for c, (n, cy) in sorted(by_class.items())},
           "opcodes": dict(ops.most_common()),
           "note": "static counts weight every instruction of the kernel's code equally; used only for the mean issue cost of the classes "
                   "whose dynamic count is known as a whole (int32) or as a remainder (other)"}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res["classes"], indent=1))


if __name__ == "__main__":
    main()

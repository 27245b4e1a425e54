#!/usr/bin/env python3
"""Static opcode histogram of a kernel's main loop, from the assembly `make -C raytracing_amd/csrc isa` writes.

    python3 tools/isa_histogram.py [file.s] [kernel-name-substring ...] > profiles/<name>.md

Counts are static (one per instruction in the text, not weighted by how often a block runs); they say which opcodes the
loop is made of, the counters in the profile summaries say how often the loop issues.  Only blocks the compiler marks as
belonging to a loop are counted, so the prologue (argument loads, table set-up) and the epilogue stay out."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = os.path.join(ROOT, "raytracing_amd", "csrc", "_build", "rt_kernels_spheres.s")

CLASSES = [
    ("fp64 arithmetic", r"v_(add|mul|fma|fmac|min|max|div_fixup|div_fmas|div_scale|rcp|rsq|sqrt|trig_preop|ldexp|frexp_mant|fract|floor|ceil|trunc|rndne)_f64"),
    ("fp32 arithmetic", r"v_(pk_)?(add|sub|subrev|mul|fma|fmac|mac|mad|min|max|rcp|rsq|sqrt|ldexp|fract|floor|ceil|trunc|rndne|exp|log|med3|min3|max3|div_fixup|div_fmas|div_scale)_(legacy_)?f32"),
    ("conversion", r"v_cvt_|v_frexp_exp"),
    ("compare", r"v_cmpx?_|v_cmp_class"),
    ("select", r"v_cndmask"),
    ("lane movement", r"v_readlane|v_writelane|v_readfirstlane|v_permlane|ds_bpermute|ds_permute|v_mov_b32_dpp|_dpp|v_mbcnt|ds_swizzle"),
    ("integer multiply", r"v_mul_(lo|hi)_[ui]32|v_mad_[ui]64_[ui]32|v_mul_[ui]32_[ui]24|v_mad_[ui]32_[ui]24|v_mul_hi"),
    ("integer add/sub", r"v_(add|sub|subrev)(_co)?(_ci)?_[ui](16|32)|v_addc|v_subb|v_lshl_add|v_add_lshl|v_add3|v_lshl_add_u64|v_mad_[ui]32_[ui]16"),
    ("shift / bit", r"v_(lshl|lshr|ashr)(rev)?_|v_(and|or|xor|not|xnor|bfe|bfi|alignbit|alignbyte|perm|bcnt|ffbh|ffbl|and_or|or3|xad|lshl_or)_"),
    ("move", r"v_mov_|v_accvgpr|v_swap|v_pk_mov"),
    ("LDS", r"ds_"),
    ("vector memory", r"global_|buffer_|flat_"),
    ("scratch (spill)", r"scratch_"),
    ("scalar memory", r"s_load|s_buffer_load|s_store|s_dcache"),
    ("wait / nop", r"s_waitcnt|s_nop|s_sleep|s_barrier"),
    ("branch", r"s_cbranch|s_branch|s_setpc|s_swappc|s_call"),
    ("scalar ALU", r"s_"),
    ("other VALU", r"v_"),
]
CLASSES = [(n, re.compile(p)) for n, p in CLASSES]
VALU = {"fp64 arithmetic", "fp32 arithmetic", "conversion", "compare", "select", "integer multiply", "integer add/sub",
        "shift / bit", "move", "other VALU"}


def classify_line(op):
    if op.endswith("_dpp") or "_dpp" in op:
        return "lane movement"
    for name, pat in CLASSES:
        if pat.match(op):
            return name
    return "other"


def kernels(path):
    """Yields (name, [(in_loop, opcode)]) per kernel in the file."""
    name, body, in_loop = None, [], False
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            name, body, in_loop = m.group(1), [], False
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            in_loop = "Loop" in line
            continue
        s = line.strip()
        if not s or s[0] in ";." or s.startswith("//"):
            continue
        body.append((in_loop, s.split()[0]))


def main():
    args = sys.argv[1:]
    path = args.pop(0) if args and args[0].endswith(".s") else DEFAULT
    wanted = args
    print(f"# Static opcode histogram of the loops in `{os.path.relpath(path, ROOT)}`\n")
    print("One count per instruction in the text of a block that belongs to a loop; not weighted by execution.\n")
    for name, body in kernels(path):
        if wanted and not any(w in name for w in wanted):
            continue
        ops = [op for in_loop, op in body if in_loop]
        if not ops:
            continue
        by_class = collections.Counter(classify_line(op) for op in ops)
        by_op = collections.Counter(ops)
        total = len(ops)
        valu = sum(n for c, n in by_class.items() if c in VALU)
        print(f"## `{name}`\n")
        print(f"{total} instructions in loops ({len(body) - total} outside); {valu} of them VALU arithmetic/logic "
              f"({by_class['fp64 arithmetic'] + by_class['fp32 arithmetic']} floating point = "
              f"{100.0 * (by_class['fp64 arithmetic'] + by_class['fp32 arithmetic']) / max(valu, 1):.0f} % of those).\n")
        print("| class | instructions | share |\n|---|---:|---:|")
        for c, n in by_class.most_common():
            print(f"| {c} | {n} | {100.0 * n / total:.1f} % |")
        print("\n| opcode | class | instructions |\n|---|---|---:|")
        for op, n in by_op.most_common(30):
            print(f"| `{op}` | {classify_line(op)} | {n} |")
        nonfp = [(op, n) for op, n in by_op.most_common() if classify_line(op) in VALU - {"fp64 arithmetic", "fp32 arithmetic"}]
        print("\nTop VALU opcodes that are not floating-point arithmetic: "
              + ", ".join(f"`{op}` {n}" for op, n in nonfp[:12]) + ".\n")


if __name__ == "__main__":
    main()

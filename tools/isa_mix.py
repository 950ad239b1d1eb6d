"""Static issue-class make-up of the shipped trace kernels, for bench.py's roofline.peak_at_mix.

The hardware counts the dynamic VALU instruction TYPES (SQ_INSTS_VALU_ADD/MUL/FMA_F32, _TRANS_F32, _INT32, _INT64, _CVT; the rest =
moves, logic, compares, selects, min / max); what it cannot see is (a) whether a float add / mul / fma carries an SGPR source (then it
issues in the base class, 4.3 cycles per wave64 instead of 2.9: profiles/r03_valu_op_rates.txt) and (b) how the integer and "other"
buckets split between fast (moves, logic, shifts, v_add_u32) and base class (min / max / compares / selects / bfe / lshl_add / 24- and
32-bit multiplies / conversions / 64-bit shifts and adds).  Those two fractions are taken from the kernel's own ISA, per counter bucket.

usage: make -C ray-tracing_amd/csrc asm && python tools/isa_mix.py  ->  profiles/isa_mix.json"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = os.path.join(ROOT, "build", "asm", "rt_context-hip-amdgcn-amd-amdhsa-gfx950.s")
KERNELS = {"bvh": "_ZN3rtk15rt_trace_kernelILb0ELb0ELb0EEEv5KArgs", "flat": "_ZN3rtk15rt_trace_kernelILb0ELb1ELb0EEEv5KArgs",
           "many": "_ZN3rtk15rt_trace_kernelILb0ELb0ELb1EEEv5KArgs"}
# Relative issue costs from profiles/r03_valu_op_rates.txt, scaled so that the cheapest class (v_mov / logic / shifts:
# 2.36 "cycles at 2.4 GHz" in that microbenchmark, whose clock sags under a pure-VALU load) is the architectural 2 cycles per wave64
# instruction that `roofline.peak` assumes: x 2.0 / 2.36.
_S = 2.0 / 2.36
CYCLES = {"fast_arith": 2.88 * _S, "base": 4.4 * _S, "fast_other": 2.36 * _S, "trans": 8.15 * _S, "packed": 5.0 * _S}

ARITH = re.compile(r"v_(add|sub|subrev|mul|fma|fmac|mac|mad|madmk|madak|fmamk|fmaak)_(legacy_)?f32")
PACKED = re.compile(r"v_pk_(add|mul|fma)_f32")
TRANS = re.compile(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)_(legacy_)?f32")
INT64 = re.compile(r"v_(lshlrev_b64|lshrrev_b64|ashrrev_i64|lshl_add_u64|mad_u64_u32|mad_i64_i32|add_co_u32|addc_co_u32)")
CVT = re.compile(r"v_cvt_")
INT32_FAST = re.compile(r"v_(add_u32|sub_u32|subrev_u32|add3_u32|lshlrev_b32|lshrrev_b32|ashrrev_i32|and_b32|or_b32|xor_b32|not_b32|xad_u32|or3_b32|and_or_b32)")
INT32_BASE = re.compile(r"v_(mul_lo_u32|mul_hi_u32|mul_u32_u24|mul_i32_i24|mad_u32_u24|mad_i32_i24|bfe_u32|bfe_i32|bfi_b32|lshl_add_u32|add_lshl_u32|lshl_or_b32|min_u32|max_u32|min_i32|max_i32|min3_|max3_|med3_i|med3_u|mbcnt|bcnt|ffbh|ffbl|alignbit|perm)")
OTHER_FAST = re.compile(r"v_(mov_b32|mov_b64|accvgpr|nop|swap)")
SGPR_SRC = re.compile(r"(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|vcc_hi|exec|m0)(?![\w])")


def classify(line):
    t = line.strip()
    op = t.split()[0]
    operands = t[len(op):].split(";")[0]
    dst_and_src = operands.split(",")
    srcs = ",".join(dst_and_src[1:])  # the destination of a VALU op is the first operand
    sgpr = bool(SGPR_SRC.search(srcs)) and not op.startswith("v_cndmask") and not op.startswith("v_readlane") and not op.startswith("v_writelane")
    if PACKED.match(op):
        return "arith", "packed"
    if TRANS.match(op):
        return "trans", "trans"
    if ARITH.match(op):
        return "arith", "base" if sgpr else "fast_arith"
    if CVT.match(op):
        return "cvt", "base"
    if INT64.match(op):
        return "int64", "base"
    if INT32_BASE.match(op):
        return "int32", "base"
    if INT32_FAST.match(op):
        return "int32", "base" if sgpr else "fast_other"
    if OTHER_FAST.match(op):
        return "other", "base" if sgpr else "fast_other"
    return "other", "base"  # compares, selects, float min / max / med3, readlane / writelane, ...


def kernel_mix(asm, name):
    i = asm.index(name + ":")
    j = asm.index(".Lfunc_end", i)
    buckets = {}
    n = 0
    for l in asm[i:j].split("\n"):
        t = l.strip()
        if not t.startswith("v_"):
            continue
        b, c = classify(t)
        buckets.setdefault(b, {}).setdefault(c, 0)
        buckets[b][c] += 1
        n += 1
    out = {"static_valu_instructions": n, "buckets": buckets}
    for b, cl in buckets.items():
        tot = sum(cl.values())
        out.setdefault("cycles_per_instruction", {})[b] = sum(CYCLES[c] * k for c, k in cl.items()) / tot
    out["static_cycles_per_instruction"] = sum(CYCLES[c] * k for cl in buckets.values() for c, k in cl.items()) / n
    return out


if __name__ == "__main__":
    asm = open(ASM).read()
    res = {"what": "static issue-class make-up per hardware counter bucket of the shipped trace kernel instantiations (tools/isa_mix.py); "
                   "cycles per wave64 instruction per SIMD from profiles/r03_valu_op_rates.txt", "class_cycles": CYCLES,
           "bucket_counters": {"arith": "SQ_INSTS_VALU_ADD_F32 + _MUL_F32 + _FMA_F32", "trans": "SQ_INSTS_VALU_TRANS_F32", "int32": "SQ_INSTS_VALU_INT32",
                               "int64": "SQ_INSTS_VALU_INT64", "cvt": "SQ_INSTS_VALU_CVT", "other": "SQ_INSTS_VALU minus the above"}}
    for k, name in KERNELS.items():
        res[k] = kernel_mix(asm, name)
        print(k, res[k]["static_valu_instructions"], {b: round(v, 2) for b, v in res[k]["cycles_per_instruction"].items()}, round(res[k]["static_cycles_per_instruction"], 3))
    json.dump(res, open(os.path.join(ROOT, "profiles", "isa_mix.json"), "w"), indent=1)

#!/usr/bin/env python3
"""Build oracle/_ref/libref.so: the REFERENCE'S OWN shader text compiled as C++.  TEST INFRASTRUCTURE ONLY.

Reads /root/reference/Assets/Scripts/Tracer/RayCommon.hlsl and RayCompute.compute where they lie, applies the
mechanical rewrites listed in REWRITES below (syntax only: no expression, statement, constant or line order is
changed), wraps the result in `namespace hlsl_ref` between oracle/ref_compat.h (the HLSL types and intrinsics,
mapped to include/rt_math.h) and oracle/ref_driver.h (buffer binding + dispatch), and pipes that translation unit
to g++ on stdin.  No reference text is written into the repository: the only output is the shared library under
oracle/_ref/ (git-ignored; it travels to the GPU box like any other built .so).  `--emit FILE` writes the
translation unit for inspection (outside the repo, e.g. /tmp).

    python oracle/make_ref.py            # libref.so      (the contract of include/rt_math.h)
    python oracle/make_ref.py --ieee     # libref_ieee.so (-DRT_MATH_IEEE, the other reading of '/', normalize, smoothstep)

tests/test_ref_pin.py compares liboracle.so with libref.so bit for bit.
"""
import argparse
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("RT_REFERENCE_ROOT", "/root/reference")
SHADER_DIR = os.path.join(REFERENCE, "Assets", "Scripts", "Tracer")
SOURCES = ("RayCommon.hlsl", "RayCompute.compute")  # RCC:3 includes RC first
OUT_DIR = os.path.join(HERE, "_ref")

CXXFLAGS = ["-O2", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=off", "-fno-unsafe-math-optimizations",
            "-fwrapv",  # HLSL int arithmetic wraps (RC:552 Frame * 719393)
            "-pthread", "-Wall", "-Wno-unknown-pragmas", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
            "-Wno-maybe-uninitialized", "-Wno-unused-function", "-Wno-pedantic"]

SWIZZLES = ("xyz", "rgb", "xz", "zy", "xy", "r", "g", "b", "_m00_m10_m20", "_m01_m11_m21")


def struct_names(text):
    return re.findall(r"^struct\s+(\w+)", text, flags=re.M)


def struct_body_lines(lines):
    """indices of lines inside `struct X { ... };` blocks (member declarations are left alone)"""
    inside, out = False, set()
    for i, ln in enumerate(lines):
        if re.match(r"^struct\s+\w+", ln):
            inside = True
        if inside:
            out.add(i)
            if re.match(r"^\};", ln):
                inside = False
    return out


def rewrite(text):
    """The complete list of edits made to the reference's text.  Each is syntactic:

     1. `#pragma kernel ...`, `#include "RayCommon.hlsl"` lines dropped (the two files are concatenated in include order)
     2. `[numthreads(8,8,1)]` attribute and the `: SV_DispatchThreadID` semantic dropped
     3. `1.#INF` -> HLSL_INF
     4. floating literals get the suffix `_h` (HLSL literals are float, C++'s would be double): `0.35` -> `0.35_h`
     5. the type name `float` -> `hfloat` (float2/3/4/4x4 keep their names)
     6. `inout T x` -> `T& x`
     7. `(T)0` -> `T{}`
     8. rvalue swizzles `.xyz .rgb .xz .zy .xy .r .g .b ._m00_m10_m20 ._m01_m11_m21` -> member calls `.xyz()` ...
     9. uniform declarations `const uint2 Resolution;` / `const bool accumulate;` lose `const` (set by the driver)
    10. struct-typed locals declared without initialiser (`ModelHitInfo result;`) get `= {}` — HLSL leaves them
        undefined, the reference reads `didHit` of such a local (RC:337,488: quirk Q6, treated as false)
    11. `int2 stats;` (RC:339, never output by the reference) -> zero-initialised + exported on scope exit
    """
    lines = text.split("\n")
    lines = [ln for ln in lines if not re.match(r"\s*#\s*(pragma|include)\b", ln)]                          # 1
    text = "\n".join(lines)
    text = re.sub(r"\[numthreads\([^)]*\)\]", "", text)                                                      # 2
    text = re.sub(r"\s*:\s*SV_DispatchThreadID", "", text)                                                   # 2
    text = text.replace("1.#INF", "HLSL_INF")                                                                # 3
    text = re.sub(r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)(?![\w.#])", r"\1_h", text)          # 4
    text = re.sub(r"\bfloat\b", "hfloat", text)                                                              # 5
    text = re.sub(r"\binout\s+(\w+)\s+(\w+)", r"\1& \2", text)                                               # 6
    text = re.sub(r"\((\w+)\)0\b", r"\1{}", text)                                                            # 7
    text = re.sub(r"\.(%s)\b(?!\s*\()" % "|".join(SWIZZLES), r".\1()", text)                                 # 8
    text = re.sub(r"^const\s+(uint2|bool)\s+(\w+);", r"\1 \2;", text, flags=re.M)                            # 9
    names = struct_names(text)
    lines = text.split("\n")
    skip = struct_body_lines(lines)
    pat = re.compile(r"^(\s+)(%s)\s+(\w+);" % "|".join(names))
    for i, ln in enumerate(lines):
        if i not in skip:
            lines[i] = pat.sub(r"\1\2 \3 = {};", ln)                                                         # 10
    text = "\n".join(lines)
    text = re.sub(r"^(\s+)int2 stats;", r"\1int2 stats = {}; RefStatsExport stats_export(stats);", text, flags=re.M)  # 11
    return text


def translation_unit():
    parts = []
    for name in SOURCES:
        with open(os.path.join(SHADER_DIR, name), "r", encoding="utf-8-sig") as f:
            parts.append("/* ---- %s ---- */\n%s" % (name, rewrite(f.read().replace("\r\n", "\n"))))
    return ('#include "%s"\nnamespace hlsl_ref {\n%s\n} /* namespace hlsl_ref */\n#include "%s"\n'
            % (os.path.join(HERE, "ref_compat.h"), "\n".join(parts), os.path.join(HERE, "ref_driver.h")))


def available():
    return all(os.path.exists(os.path.join(SHADER_DIR, s)) for s in SOURCES)


def build(ieee=False, emit=None, quiet=False):
    if not available():
        raise FileNotFoundError("reference shader sources not found under %s" % SHADER_DIR)
    tu = translation_unit()
    if emit:
        if os.path.abspath(emit).startswith(os.path.dirname(HERE) + os.sep):
            raise SystemExit("--emit inside the repository would copy reference text into it: choose a path outside")
        with open(emit, "w") as f:
            f.write(tu)
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "libref_ieee.so" if ieee else "libref.so")
    cmd = [os.environ.get("CXX", "g++")] + CXXFLAGS + (["-DRT_MATH_IEEE"] if ieee else []) + \
        ["-I", HERE, "-shared", "-o", out, "-x", "c++", "-"]
    subprocess.run(cmd, input=tu.encode(), check=True, stdout=subprocess.DEVNULL if quiet else None)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ieee", action="store_true")
    ap.add_argument("--both", action="store_true")
    ap.add_argument("--emit")
    ap.add_argument("--if-available", action="store_true", help="exit 0 quietly when /root/reference is absent (GPU box)")
    a = ap.parse_args()
    if a.if_available and not available():
        print("make_ref: %s absent — keeping whatever oracle/_ref/ holds" % SHADER_DIR)
        sys.exit(0)
    for ieee in ((False, True) if a.both else (a.ieee,)):
        print("built", build(ieee=ieee, emit=a.emit))

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
    python oracle/make_ref.py --spheres  # libref_spheres.so: the same text plus ONE declared SEMANTIC rewrite (S1 below) that revives
                                         # the reference's own RaySphere behind the call it left commented out (RC:341)

Every build also writes oracle/_ref/MANIFEST.json: sha256 of each reference source read, of this recipe and its headers, and of each
library built.  oracle/REF_EXPECTED.json (committed: hashes, not text) holds the source / recipe hashes the travelling libraries
must have been built from; tests/test_gpu_ref_pin.py and __graft_entry__.smoke() FAIL when they disagree (a stale prebuilt library
on the GPU box, which cannot rebuild it).  `--write-expected` refreshes REF_EXPECTED.json after a deliberate change.

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
    other = [ln for ln in lines if re.match(r"\s*#", ln)]
    if other:  # the text is compiled to native code and loaded: no other preprocessor directive gets through unread
        raise SystemExit("reference shader text: unexpected preprocessor directive %r — review oracle/make_ref.py" % other[0].strip())
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


# ---- S1: the one SEMANTIC rewrite (libref_spheres.so only).  The reference keeps RaySphere (RC:289-332) but its only call is the
# commented line RC:341 with a hard-coded sphere and a hard-coded material (RC:321-327).  S1 replaces that comment by a loop over a
# sphere buffer that calls the reference's own RaySphere, untouched, and substitutes the sphere's material for the hard-coded one —
# tested before the model loop with the closest-so-far carried in result.dst, as the commented call's position implies
# (SURVEY.md 8(a) S1).  These lines are the builder's; everything else of the library is the reference's text.  Written in HLSL so
# that the syntactic rewrites 1-11 apply to them like to any other line.
S1_COMMENTED_CALL = "    //result = RaySphere(worldRay.origin, worldRay.dir, float3(0, 1.8, 0), 1);\n"
S1_HOOK = """    for (int sphereIndex = 0; sphereIndex < sphereCount; sphereIndex++)
    {
        Sphere sphere = Spheres[sphereIndex];
        ModelHitInfo sphereHit = RaySphere(worldRay.pos, worldRay.dir, sphere.centre, sphere.radius);
        if (sphereHit.didHit && sphereHit.dst < result.dst)
        {
            result = sphereHit;
            result.material = sphere.material;
        }
    }
"""
S1_DECL_BEFORE = "ModelHitInfo CalculateRayCollision(Ray worldRay, bool forceDontCullBack)\n"
S1_DECL = """struct Sphere
{
    float3 centre;
    float radius;
    RayTracingMaterial material;
};
StructuredBuffer<Sphere> Spheres;
int sphereCount;

"""


def apply_s1(text):
    if text.count(S1_COMMENTED_CALL) != 1 or text.count(S1_DECL_BEFORE) != 1:
        raise SystemExit("RayCommon.hlsl: the commented RaySphere call (RC:341) was not found exactly once — the reference changed, review S1")
    text = text.replace(S1_DECL_BEFORE, S1_DECL + S1_DECL_BEFORE)
    return text.replace(S1_COMMENTED_CALL, S1_HOOK)


def translation_unit(spheres=False):
    parts = []
    for name in SOURCES:
        with open(os.path.join(SHADER_DIR, name), "r", encoding="utf-8-sig") as f:
            text = f.read().replace("\r\n", "\n")
            if spheres and name == "RayCommon.hlsl":
                text = apply_s1(text)
            parts.append("/* ---- %s ---- */\n%s" % (name, rewrite(text)))
    return ('#include "%s"\nnamespace hlsl_ref {\n%s\n} /* namespace hlsl_ref */\n#include "%s"\n'
            % (os.path.join(HERE, "ref_compat.h"), "\n".join(parts), os.path.join(HERE, "ref_driver.h")))


# ---------------------------------------------------------------------------------------------------------------------
# MANIFEST.json / REF_EXPECTED.json: what the libraries under oracle/_ref/ were built from
import hashlib
import json

RECIPE_FILES = ("make_ref.py", "ref_compat.h", "ref_driver.h", "ref_bvh_compat.h", "ref_bvh_driver.h", "../include/rt_math.h", "../include/rt_abi.h")
MANIFEST = os.path.join(OUT_DIR, "MANIFEST.json")
EXPECTED = os.path.join(HERE, "REF_EXPECTED.json")


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def input_hashes():
    """sha256 of every input of the libraries: the reference sources where they lie, the recipe and its headers"""
    out = {"reference": {}, "recipe": {}}
    for name in SOURCES:
        out["reference"][name] = _sha(os.path.join(SHADER_DIR, name))
    if bvh_available():
        out["reference"]["BVH.cs"] = _sha(BVH_SOURCE)
    for name in RECIPE_FILES:
        out["recipe"][os.path.basename(name)] = _sha(os.path.join(HERE, name))
    return out


def recipe_hashes_here():
    """the recipe half of input_hashes(): computable anywhere (the GPU box has the recipe, not the reference)"""
    return {os.path.basename(name): _sha(os.path.join(HERE, name)) for name in RECIPE_FILES}


def write_manifest(built):
    """called after every build: merge the libraries just built into MANIFEST.json under the inputs they were built from"""
    man = {}
    if os.path.exists(MANIFEST):
        try:
            man = json.load(open(MANIFEST))
        except Exception:
            man = {}
    inputs = input_hashes()
    if man.get("inputs") != inputs:  # other inputs than last time: whatever else lies in _ref/ is no longer vouched for
        man = {"inputs": inputs, "libraries": {}}
    for path in built:
        man["libraries"][os.path.basename(path)] = _sha(path)
    with open(MANIFEST, "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)


def check_manifest(libraries):
    """None when oracle/_ref/<libraries> are what MANIFEST.json says AND MANIFEST.json's inputs are REF_EXPECTED.json's AND the
    recipe files in this tree are the ones they were built with; else a sentence saying what is stale."""
    if not os.path.exists(EXPECTED):
        return "oracle/REF_EXPECTED.json is missing"
    if not os.path.exists(MANIFEST):
        return "oracle/_ref/MANIFEST.json is missing: rebuild with `python oracle/make_ref.py --all` where /root/reference exists"
    man, exp = json.load(open(MANIFEST)), json.load(open(EXPECTED))
    if man.get("inputs") != exp.get("inputs"):
        diff = [k for sec in ("reference", "recipe") for k in set(man.get("inputs", {}).get(sec, {})) | set(exp.get("inputs", {}).get(sec, {}))
                if man.get("inputs", {}).get(sec, {}).get(k) != exp.get("inputs", {}).get(sec, {}).get(k)]
        return "oracle/_ref was built from other inputs than oracle/REF_EXPECTED.json names: " + ", ".join(sorted(diff))
    here = recipe_hashes_here()
    stale = sorted(k for k, v in here.items() if exp["inputs"]["recipe"].get(k) != v)
    if stale:
        return "recipe files changed since oracle/_ref was built (rebuild, then `make_ref.py --write-expected`): " + ", ".join(stale)
    for name in libraries:
        path = os.path.join(OUT_DIR, name)
        if not os.path.exists(path):
            return "oracle/_ref/%s is missing" % name
        if man.get("libraries", {}).get(name) != _sha(path):
            return "oracle/_ref/%s is not the file MANIFEST.json describes" % name
    return None


# ---------------------------------------------------------------------------------------------------------------------
# The reference's BVH builder (C#): Assets/Scripts/Types/BVH.cs -> oracle/_ref/libref_bvh.so
BVH_SOURCE = os.path.join(REFERENCE, "Assets", "Scripts", "Types", "BVH.cs")
BVH_TYPES = ("Node", "BVHTriangle", "Triangle", "NodeList", "BuildStats")   # nested types, in dependency order
BVH_CUT = (r"public \(bool hit, float dst, Vector3 pos, bool backface, Vector3 normal\) Search\(",   # BVH:321-375  CPU-side traversal:
           r"public static \(bool hit, float dst\) RayBoundingBox\(",                                # BVH:377-401  not on the build path
           r"static \(bool hit, float dst, bool backface, Vector3 normal\) RayTriangle\(",           # BVH:403-424  (and C# tuples / Stack<T>)
           r"public override string ToString\(\)")                                                 # BVH:555-575  interpolated strings


def _block(text, header_regex):
    """(start, end) of the declaration whose header matches: from the start of the header's line to its closing brace"""
    m = re.search(header_regex, text)
    if not m:
        raise SystemExit("BVH.cs: %r not found — the reference changed, review oracle/make_ref.py" % header_regex)
    start = text.rfind("\n", 0, m.start()) + 1
    i = text.index("{", m.end())
    depth = 0
    while True:
        if text[i] == "{":
            depth += 1
        elif text[i] == "}":
            depth -= 1
            if depth == 0:
                return start, i + 1
        i += 1


def rewrite_bvh(text):
    """The complete list of edits made to BVH.cs.  Syntax only: no expression, statement, constant or statement order changes.

     B1. `using ...;` lines and the `[System.Serializable]` attribute dropped; `namespace A.B` -> `namespace A::B`
     B2. the members that are not on the build path are cut whole: Search, RayBoundingBox, RayTriangle (CPU-side traversal, BVH:321-424)
         and BuildStats.ToString (BVH:555-575)
     B3. the nested types Node, BVHTriangle, Triangle, NodeList, BuildStats are moved in front of the fields that use them (C++ wants
         complete member types) and every type's closing brace gets its `;`
     B4. `public`, `readonly`, `private`, `override` dropped; `class` -> `struct`; `enum` -> `enum class`
     B5. `T[]` -> `CsArray<T>`; `new T[n]` -> `CsArray<T>(n)`
     B6. `new T(args)` -> `T(args)`; target-typed `T x = new(args);` -> `T x = T(args);`; `x = new();` -> `x = {};`
     B7. `ref T x = ref e;` -> `T& x = e;`
     B8. `var` -> `auto`; `System.Diagnostics.Stopwatch.StartNew()` -> `Stopwatch::StartNew()`
     B9. `float.MaxValue / MinValue / PositiveInfinity`, `int.MaxValue` -> CS_FLOAT_MAX / CS_FLOAT_MIN / CS_FLOAT_POSITIVE_INFINITY / CS_INT_MAX
    B10. the literal `1f` -> `1.0f`
    B11. tuples: return type `(int axis, float pos, float cost)` -> `std::tuple<int, float, float>`, `return (a, b, c);` ->
         `return std::make_tuple(a, b, c);`, `(int a, float b, float c) = f(...)` -> `auto [a, b, c] = f(...)`
    B12. switch expressions `x switch { 0 => a, 1 => b, _ => c }` -> `(x == 0 ? a : x == 1 ? b : c)`
    B13. static member access `Mathf.` `Math.` `Quality.` -> `::`; `Array.Resize(ref a, n)` -> `Array::Resize(a, n)`
    B14. `this.` -> `this->`
    B15. the expression-bodied property `int NodeCount => Index;` -> a member function, its use gets `()`
    B16. `int` / `float` fields and locals declared without initialiser get `{}` (C# zero-initialises fields; locals are assigned
         before use, so it changes nothing for them)
    B17. BuildStats' field `Quality` is renamed `Quality_` (a C++ member cannot take the name of its type)
    B18. types that declare a constructor get `T() = default;` (C# structs always have the parameterless one; `new()` uses it)
    """
    text = text.replace("\r\n", "\n").replace("\t", "    ")
    text = "\n".join(ln for ln in text.split("\n") if not re.match(r"\s*using\s+[\w.]+;", ln))           # B1
    text = text.replace("[System.Serializable]", "")                                                       # B1
    text = re.sub(r"namespace\s+(\w+)\.(\w+)", r"namespace \1::\2", text)                                     # B1
    for pat in BVH_CUT:                                                                                    # B2
        a, b = _block(text, pat)
        text = text[:a] + text[b:]
    text = re.sub(r"\n\s*// ---- Traversal ---\s*\n", "\n", text)
    moved = []
    for name in BVH_TYPES:                                                                                 # B3
        a, b = _block(text, r"public (?:readonly )?(?:struct|class) %s\b" % name)
        blk = text[a:b]
        text = text[:a] + text[b:]
        if re.search(r"\b%s\(" % name, blk):                                                              # B18
            i = blk.index("{") + 1
            blk = blk[:i] + "\n            %s() = default;" % name + blk[i:]
        if name == "BuildStats":                                                                           # B17
            blk = blk.replace("public Quality Quality;", "public Quality Quality_;").replace("this.Quality = quality;", "this.Quality_ = quality;")
        moved.append(blk + ";")
    a, b = _block(text, r"public enum Quality\b")
    text = text[:b] + ";\n\n" + "\n\n".join(moved) + "\n" + text[b:]
    a, b = _block(text, r"public class BVH\b")
    text = text[:b] + ";" + text[b:]
    text = re.sub(r"\b(public|readonly|private|override)\s+", "", text)                                    # B4
    text = re.sub(r"\bclass\b", "struct", text)
    text = re.sub(r"\benum\b", "enum class", text)
    text = re.sub(r"\bnew (\w+)\[([^\]]+)\]", r"CsArray<\1>(\2)", text)                                    # B5
    text = re.sub(r"\b(\w+)\[\]", r"CsArray<\1>", text)
    text = re.sub(r"\b(\w+) (\w+) = new\(", r"\1 \2 = \1(", text)                                          # B6
    text = re.sub(r"= new\(\);", "= {};", text)
    text = re.sub(r"\bnew (\w+)\(", r"\1(", text)
    text = re.sub(r"\bref (\w+) (\w+) = ref ", r"\1& \2 = ", text)                                         # B7
    text = text.replace("System.Diagnostics.Stopwatch.StartNew()", "Stopwatch::StartNew()")               # B8
    text = re.sub(r"\bvar\b", "auto", text)
    for cs, cpp in (("float.MaxValue", "CS_FLOAT_MAX"), ("float.MinValue", "CS_FLOAT_MIN"),                # B9
                    ("float.PositiveInfinity", "CS_FLOAT_POSITIVE_INFINITY"), ("int.MaxValue", "CS_INT_MAX")):
        text = text.replace(cs, cpp)
    text = re.sub(r"(?<![\w.])(\d+)f\b", r"\1.0f", text)                                                   # B10
    text = re.sub(r"\(int axis, float pos, float cost\) (\w+)\(", r"std::tuple<int, float, float> \1(", text)   # B11
    text = re.sub(r"\breturn \(([^;]*,[^;]*,[^;]*)\);", r"return std::make_tuple(\1);", text)
    text = re.sub(r"\(int (\w+), float (\w+), float (\w+)\) = ", r"auto [\1, \2, \3] = ", text)
    text = re.sub(r"(\w+) switch\s*\{\s*0 => ([^,]+),\s*1 => ([^,]+),\s*_ => ([^}]+?)\s*\};",                 # B12
                  r"(\1 == 0 ? \2 : \1 == 1 ? \3 : \4);", text)
    text = re.sub(r"\b(Mathf|Math|Quality)\.", r"\1::", text)                                              # B13
    text = re.sub(r"\bArray\.Resize\(ref ", "Array::Resize(", text)
    text = re.sub(r"\bthis\.", "this->", text)                                                             # B14
    text = re.sub(r"\bint NodeCount => Index;", "int NodeCount() const { return Index; }", text)           # B15
    text = re.sub(r"\.NodeCount\b(?!\s*\()", ".NodeCount()", text)
    text = re.sub(r"^(\s*)(int|float) (\w+);", r"\1\2 \3{};", text, flags=re.M)                              # B16
    if re.search(r"=>|\bswitch\s*\{|\bnew\b|\bref\b|\$\"", text):
        raise SystemExit("BVH.cs: a C# construct survived the rewrites — the reference changed, review oracle/make_ref.py")
    return text


def bvh_translation_unit():
    with open(BVH_SOURCE, "r", encoding="utf-8-sig") as f:
        body = rewrite_bvh(f.read())
    return ('#include "%s"\n/* ---- BVH.cs ---- */\n%s\n#include "%s"\n'
            % (os.path.join(HERE, "ref_bvh_compat.h"), body, os.path.join(HERE, "ref_bvh_driver.h")))


def bvh_available():
    return os.path.exists(BVH_SOURCE)


def build_bvh(emit=None, quiet=False):
    if not bvh_available():
        raise FileNotFoundError("reference BVH.cs not found: %s" % BVH_SOURCE)
    tu = bvh_translation_unit()
    if emit:
        if os.path.abspath(emit).startswith(os.path.dirname(HERE) + os.sep):
            raise SystemExit("--emit inside the repository would copy reference text into it: choose a path outside")
        with open(emit, "w") as f:
            f.write(tu)
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "libref_bvh.so")
    cmd = [os.environ.get("CXX", "g++")] + CXXFLAGS + ["-I", HERE, "-shared", "-o", out, "-x", "c++", "-"]
    subprocess.run(cmd, input=tu.encode(), check=True, stdout=subprocess.DEVNULL if quiet else None)
    write_manifest([out])
    return out


def available():
    return all(os.path.exists(os.path.join(SHADER_DIR, s)) for s in SOURCES)


def build(ieee=False, emit=None, quiet=False, spheres=False):
    if not available():
        raise FileNotFoundError("reference shader sources not found under %s" % SHADER_DIR)
    tu = translation_unit(spheres=spheres)
    if emit:
        if os.path.abspath(emit).startswith(os.path.dirname(HERE) + os.sep):
            raise SystemExit("--emit inside the repository would copy reference text into it: choose a path outside")
        with open(emit, "w") as f:
            f.write(tu)
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, "libref_spheres.so" if spheres else "libref_ieee.so" if ieee else "libref.so")
    cmd = [os.environ.get("CXX", "g++")] + CXXFLAGS + (["-DRT_MATH_IEEE"] if ieee else []) + (["-DREF_SPHERES"] if spheres else []) + \
        ["-I", HERE, "-shared", "-o", out, "-x", "c++", "-"]
    subprocess.run(cmd, input=tu.encode(), check=True, stdout=subprocess.DEVNULL if quiet else None)
    write_manifest([out])
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ieee", action="store_true")
    ap.add_argument("--both", action="store_true")
    ap.add_argument("--emit")
    ap.add_argument("--bvh", action="store_true", help="build oracle/_ref/libref_bvh.so (BVH.cs) instead of the shader library")
    ap.add_argument("--spheres", action="store_true", help="build oracle/_ref/libref_spheres.so (the shader text + the declared semantic rewrite S1)")
    ap.add_argument("--all", action="store_true", help="libref.so, libref_ieee.so, libref_spheres.so, libref_bvh.so")
    ap.add_argument("--write-expected", action="store_true", help="record the present source / recipe hashes in oracle/REF_EXPECTED.json")
    ap.add_argument("--check", action="store_true", help="exit 1 with a sentence when oracle/_ref is stale against REF_EXPECTED.json")
    ap.add_argument("--if-available", action="store_true", help="exit 0 quietly when /root/reference is absent (GPU box)")
    a = ap.parse_args()
    if a.if_available and not available():
        print("make_ref: %s absent — keeping whatever oracle/_ref/ holds" % SHADER_DIR)
        sys.exit(0)
    if a.check:
        why = check_manifest(["libref.so", "libref_ieee.so", "libref_spheres.so", "libref_bvh.so"])
        print(why or "oracle/_ref is what oracle/REF_EXPECTED.json names")
        sys.exit(1 if why else 0)
    if a.write_expected:
        with open(EXPECTED, "w") as f:
            json.dump({"what": "sha256 of the inputs oracle/_ref/*.so must have been built from (reference sources as they lie under "
                               "/root/reference, this recipe and its headers); hashes, not text — see make_ref.py",
                       "inputs": input_hashes()}, f, indent=1, sort_keys=True)
        print("wrote", EXPECTED)
        sys.exit(0)
    if a.bvh:
        print("built", build_bvh(emit=a.emit))
        sys.exit(0)
    if a.spheres:
        print("built", build(spheres=True, emit=a.emit))
        sys.exit(0)
    both = a.both or a.all
    for ieee in ((False, True) if both else (a.ieee,)):
        print("built", build(ieee=ieee, emit=a.emit))
    if a.all:
        print("built", build(spheres=True))
    if both and bvh_available():
        print("built", build_bvh())

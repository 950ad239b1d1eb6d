"""host/dotnet/RayTraceNative.cs cannot be compiled here (no .NET in the image); keep it honest by
checking its struct declarations field-by-field against include/rt_abi.h, and its DllImports
against the header's function list."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def c_structs():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \1;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, rest = decl.split(None, 1)
            for item in rest.split(","):
                item = item.strip()
                am = re.match(r"(\w+)\[(\d+)\]", item)
                fields.append((ctype, am.group(1), int(am.group(2))) if am else (ctype, item, 1))
        out[m.group(1)] = fields
    return out


def cs_structs():
    text = open(os.path.join(ROOT, "host", "dotnet", "RayTraceNative.cs")).read()
    out = {}
    for m in re.finditer(r"public struct (\w+)\s*\{(.*?)\n    \}", text, flags=re.S):
        fields = []
        for fm in re.finditer(r"public (\w+) (\w+);", m.group(2)):
            t, n = fm.group(1), fm.group(2)
            # blittable vector types = that many consecutive floats of the C struct
            fields.append(("float", n, VECTORS[t]) if t in VECTORS else (t, n, 1))
        out[m.group(1)] = fields
    return out


VECTORS = {"Float3": 3, "Float4": 4, "Float4x4": 16}


def test_csharp_structs_are_blittable():
    """No reference-type fields (arrays, strings) and no MarshalAs: the runtime can pin RtModel[] etc. and pass the
    memory as it is, like the reference's own RayTracingMaterial / MeshInfo (Color / Matrix4x4 fields)."""
    text = open(os.path.join(ROOT, "host", "dotnet", "RayTraceNative.cs")).read()
    body = text[:text.index("public static class RayTraceNative")]
    assert "MarshalAs" not in body and "[]" not in re.sub(r"//.*", "", body)
    for t, n in VECTORS.items():
        m = re.search(r"public struct %s\b(.*?)\n    \}" % t, text, flags=re.S) or re.search(r"public struct %s \{(.*?)\}\s*\n" % t, text, flags=re.S)
        assert m, t


TYPE_MAP = {"float": "float", "int32_t": "int", "uint32_t": "uint", "uint64_t": "ulong", "double": "double",
            "RtMaterial": "RtMaterial"}


def test_csharp_structs_match_header():
    c, cs = c_structs(), cs_structs()
    for name in ("RtMaterial", "RtModel", "RtTriangle", "RtBVHNode", "RtSphere", "RtParams", "RtCounters"):
        assert name in c and name in cs, name
        assert len(c[name]) == len(cs[name]), name
        for (ct, cn, cl), (st, sn, sl) in zip(c[name], cs[name]):
            assert TYPE_MAP[ct] == st and cn == sn and cl == sl, (name, cn, sn)


def test_csharp_imports_exist_in_header():
    text = open(os.path.join(ROOT, "host", "dotnet", "RayTraceNative.cs")).read()
    header = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    imports = re.findall(r"extern \w+ (rt_\w+)\(", text)
    assert len(imports) >= 20
    for name in imports:
        assert re.search(r"\b%s\s*\(" % name, header), name


def _split_params(s):
    s = s.strip()
    if not s or s == "void":
        return []
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([<":
            depth += 1
        elif ch in ")]>":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    out.append(cur.strip())
    return out


def test_csharp_imports_have_the_headers_parameter_counts_and_the_host_uses_only_imports():
    """The P/Invoke declarations cannot be compiled here: every DllImport must at least take as many parameters as the C function it
    binds (a missing `int n_spheres` would corrupt the stack at run time), return int / void / IntPtr / double as the header does, and
    Program.cs may only call functions RayTraceNative declares."""
    cs = open(os.path.join(ROOT, "host", "dotnet", "RayTraceNative.cs")).read()
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rt_abi.h")).read(), flags=re.S)
    c_funcs = {m.group(2): (m.group(1).strip(), _split_params(m.group(3)))
               for m in re.finditer(r"^\s*((?:const\s+)?[A-Za-z_][\w\s\*]*?)\b(rt_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", header, flags=re.M | re.S)}
    imports = {m.group(2): (m.group(1), _split_params(m.group(3)))
               for m in re.finditer(r"public static extern (\w+) (rt_\w+)\(([^;]*?)\);", cs, flags=re.S)}
    assert len(imports) >= 30
    ret = {"int": "int", "void": "void", "double": "double"}
    for name, (rtype, params) in imports.items():
        assert name in c_funcs, name
        ctype, cparams = c_funcs[name]
        assert len(params) == len(cparams), (name, params, cparams)
        if "*" in ctype:
            assert rtype == "IntPtr", (name, ctype, rtype)
        else:
            assert ret.get(ctype.replace("const ", "").strip()) == rtype, (name, ctype, rtype)
    missing = set(c_funcs) - set(imports)
    assert all(n.startswith("rt_debug_") for n in missing), sorted(missing)   # every entry point of the header is bound (test hooks aside)
    prog = open(os.path.join(ROOT, "host", "dotnet", "Program.cs")).read()
    for name in set(re.findall(r"RayTraceNative\.(rt_\w+)\(", prog)):
        assert name in imports, name

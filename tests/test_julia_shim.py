"""Mechanical check of julia/RayAMD.jl against include/ray_amd.h (no Julia in the image: the shim cannot be
executed, so every `ccall((:sym, LIB), Ret, (types...), args...)` is parsed and compared with the C prototype:
symbol declared, return type, arity of the type tuple AND of the argument list, and the width class of every
parameter (pointer / int32 / int64 / double).  The C-mirroring structs are checked against the sizes
tests/test_abi.py pins for the ctypes mirror."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _split_top(s):
    """split at top-level commas (parentheses, brackets and braces nest; strings are skipped)"""
    out, depth, cur, i = [], 0, "", 0
    while i < len(s):
        ch = s[i]
        if ch == '"':
            j = s.index('"', i + 1)
            cur += s[i:j + 1]
            i = j + 1
            continue
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def _balanced(s, start):
    """s[start] == '(' -> index just past its matching ')'"""
    depth = 0
    for i in range(start, len(s)):
        if s[i] == "(":
            depth += 1
        elif s[i] == ")":
            depth -= 1
            if depth == 0:
                return i + 1
    raise ValueError("unbalanced")


def julia_ccalls():
    src = open(os.path.join(ROOT, "julia", "RayAMD.jl")).read()
    src = "\n".join(l.split("#")[0] if not l.lstrip().startswith('"') else l for l in src.splitlines())
    calls = []
    for m in re.finditer(r"ccall\(", src):
        end = _balanced(src, m.end() - 1)
        args = _split_top(src[m.end():end - 1])
        sym = re.match(r"\(:(\w+),\s*LIB\)", args[0]).group(1)
        types = _split_top(args[2].strip()[1:-1]) if args[2].strip() != "()" else []
        calls.append((sym, args[1], [t for t in types if t], args[3:]))
    return calls


def c_prototypes():
    src = open(os.path.join(ROOT, "include", "ray_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char \*)\s*(ray_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        params = [p.strip() for p in m.group(3).replace("\n", " ").split(",")]
        if params == ["void"]:
            params = []
        protos[m.group(2)] = (m.group(1), params)
    return protos


def c_class(param):
    if "*" in param or "[" in param:
        return "ptr"
    base = param.replace("const", "").split()
    t = base[0] if len(base) > 1 else base[-1]
    return {"double": "f64", "int64_t": "i64", "int32_t": "i32", "int": "i32"}[t]


def jl_class(t):
    if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
        return "ptr"
    return {"Cdouble": "f64", "Float64": "f64", "Int64": "i64", "Cint": "i32", "Int32": "i32"}[t]


def test_every_ccall_matches_its_prototype():
    protos = c_prototypes()
    calls = julia_ccalls()
    assert len(calls) >= 40
    for sym, ret, types, args in calls:
        assert sym in protos, f"{sym} is not declared in include/ray_amd.h"
        cret, cparams = protos[sym]
        assert ret == ("Cstring" if cret.startswith("const char") else "Cint"), (sym, ret)
        assert len(types) == len(cparams), f"{sym}: {len(types)} Julia types for {len(cparams)} C parameters"
        assert len(args) == len(cparams), f"{sym}: {len(args)} arguments passed for {len(cparams)} C parameters"
        for k, (jt, cp) in enumerate(zip(types, cparams)):
            assert jl_class(jt) == c_class(cp), f"{sym} parameter {k}: Julia {jt} vs C '{cp}'"


def test_shim_covers_the_boundary():
    """every reference entry point of SURVEY.md 8(b) + the NEXT rows has a binding"""
    used = {c[0] for c in julia_ccalls()}
    for sym in ("ray_shoot_to_exit", "ray_shoot_paths", "ray_trace_fixed_time", "ray_trace_paths", "ray_forward_operator",
                "ray_frechet_export_csr", "ray_frechet_dofs", "ray_frechet_matvec", "ray_frechet_matvec_t",
                "ray_frechet_normal_matvec", "ray_frechet_normal_diag", "ray_frechet_fidelity_update", "ray_classify",
                "ray_misfit", "ray_medium_create_grid", "ray_medium_create_sinprod", "ray_medium_create_radial",
                "ray_group_create", "ray_group_shoot_to_exit", "ray_group_trace_fixed_time", "ray_group_forward_operator",
                "ray_group_frechet_export_csr", "ray_group_set_observation",
                "ray_group_misfit",
                "ray_ctx_set_arith"):
        assert sym in used, sym


def test_struct_mirrors_have_the_c_layout():
    src = open(os.path.join(ROOT, "julia", "RayAMD.jl")).read()
    size = {"Int32": 4, "Float64": 8, "Int64": 8}

    def jl_sizeof(name):
        body = re.search(r"struct %s\b.*?\n(.*?)\nend" % name, src, flags=re.S).group(1)
        fields = [f.split("::")[1].strip() for f in re.sub(r"#.*", "", body).replace("\n", ";").split(";") if "::" in f]
        off = 0
        for t in fields:
            off = (off + size[t] - 1) // size[t] * size[t] + size[t]
        return (off + 7) // 8 * 8
    assert jl_sizeof("CObstacle") == 32 and jl_sizeof("CShootParams") == 48

"""The Julia ccall layer (raycore.jl_amd/julia/RaycoreMI355X.jl) cannot be executed here (no julia in the image), so it is pinned
mechanically instead: every `ccall((:rc_x, LIB), Ret, (T...), ...)` in it is parsed and compared with the prototype of rc_x in
include/raycore_mi355x.h -- same arity, pointers where the C side has pointers, and the same scalar width and signedness
everywhere else -- and every entry point the header declares must be bound (so the binding cannot silently rot when the ABI grows)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "raycore_mi355x.h")
JULIA = os.path.join(ROOT, "raycore.jl_amd", "julia", "RaycoreMI355X.jl")

C_SCALARS = {"int": "i32", "uint32_t": "u32", "uint64_t": "u64", "int64_t": "i64", "float": "f32", "size_t": "u64", "double": "f64"}
JL_SCALARS = {"Cint": "i32", "Int32": "i32", "UInt32": "u32", "Cuint": "u32", "UInt64": "u64", "Int64": "i64", "Cfloat": "f32", "Float32": "f32",
              "Csize_t": "u64", "Cdouble": "f64", "Float64": "f64"}


def c_prototypes():
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"\b(int|const char\s*\*)\s+(rc_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "[" in a:
                    kinds.append("ptr")
                else:
                    ty = a.replace("const ", "").split()[0]
                    kinds.append(C_SCALARS[ty])
        protos[name] = ("ptr" if "*" in ret else "i32", kinds)
    return protos


def split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def julia_ccalls():
    text = open(JULIA).read()
    text = re.sub(r"#[^\n]*", "", text)
    calls = []
    for m in re.finditer(r"ccall\(\(\s*:?(\w+)\s*,\s*(\w+)\s*\)\s*,\s*(\w+)\s*,\s*\(", text):
        name, libname, ret = m.group(1), m.group(2), m.group(3)
        # the type tuple: balanced parentheses from the '(' that ends the match
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        types = split_top_level(text[m.end():i - 1])
        calls.append((name, libname, ret, types))
    return calls


def jl_kind(t):
    t = t.strip()
    if t.startswith(("Ptr{", "Ref{")) or t in ("Cstring", "Ptr"):
        return "ptr"
    return JL_SCALARS[t]


def test_every_ccall_matches_its_c_prototype():
    protos = c_prototypes()
    assert len(protos) >= 58 and "rc_view_factors_device" in protos and len(protos["rc_view_factors_device"][1]) == 13
    calls = [c for c in julia_ccalls() if c[1] == "LIB"]
    assert len(calls) >= 45
    for name, _, ret, types in calls:
        if name == "f":  # `ccall((f, LIB), ...)` with f chosen between two symbols of identical signature (rc_trace_closest / rc_trace_any)
            continue
        assert name in protos, f"{name} is not declared in include/raycore_mi355x.h"
        c_ret, c_kinds = protos[name]
        assert ("ptr" if ret == "Cstring" else JL_SCALARS[ret]) == c_ret, f"{name}: return type {ret}"
        kinds = [jl_kind(t) for t in types]
        assert kinds == c_kinds, f"{name}: Julia ({', '.join(types)}) vs C {c_kinds}"


def test_symbol_variable_ccalls_have_both_signatures_right():
    """`f = any ? :rc_trace_any : :rc_trace_closest` style calls: the type tuple must fit every symbol the variable can take."""
    protos = c_prototypes()
    text = open(JULIA).read()
    calls = [c for c in julia_ccalls() if c[0] == "f"]
    assert calls
    groups = re.findall(r"f = any \? :(\w+) : :(\w+)", text)
    assert groups
    sigs = {tuple(jl_kind(t) for t in c[3]) for c in calls}
    for a, b in groups:
        assert protos[a] == protos[b]
        assert tuple(protos[a][1]) in sigs, (a, protos[a][1], sigs)


def test_every_entry_point_is_bound():
    protos = c_prototypes()
    text = open(JULIA).read()
    bound = set(re.findall(r":(rc_[a-z0-9_]+)", text))
    missing = sorted(set(protos) - bound)
    assert not missing, f"declared in the header but not bound in RaycoreMI355X.jl: {missing}"

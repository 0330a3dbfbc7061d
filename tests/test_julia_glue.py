"""Mechanical check of julia/INSHip.jl against include/ins_hip.h (Julia is not installed here, so the glue cannot be executed):

  * every `ccall((:sym, lib), Ret, (ArgTypes...), args...)` names a symbol the header declares, with the header's arity, a compatible Julia
    type for every C parameter and return type, and as many actual arguments as declared types;
  * the `GridDesc` struct mirrors `ins_grid_desc_t` field by field;
  * every reference function the glue claims to cover (the operator list of SURVEY.md §8b) has a method in the file, no comment stands in
    for a method, and every non-Base helper the file calls is defined in it or qualified with its module."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ins_hip.h")
GLUE = os.path.join(ROOT, "julia", "INSHip.jl")


def c_prototypes():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*|void)\s+(ins_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        params = [] if args in ("void", "") else [a.strip() for a in args.split(",")]
        protos[name] = (ret, params)
    return protos


def c_kind(p):
    """Classify a C parameter declaration."""
    t = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", p).strip() if not p.endswith("*") else p  # drop the parameter name
    t = t.replace("const ", "").replace(" const", "").replace(" ", "")
    table = {
        "int": "int", "int32_t": "int", "double": "double", "int64_t": "int64", "void*": "ptr", "char*": "cstring",
        "double*": "ptr_f64", "double**": "ptr_ptr", "int32_t*": "ptr_i32", "int64_t*": "ptr_i64", "int*": "ptr_i32", "float*": "ptr_f32",
    }
    if t in table:
        return table[t]
    if re.fullmatch(r"ins_[a-z0-9_]+_t\*\*", t):
        return "handle_out"
    if re.fullmatch(r"ins_[a-z0-9_]+_t\*", t):
        return "ptr"
    raise AssertionError(f"unclassified C parameter: {p!r} -> {t!r}")


JULIA_OK = {
    "int": {"Cint", "Int32"},
    "double": {"Cdouble", "Float64"},
    "int64": {"Int64", "Clonglong"},
    "ptr": {"Ptr{Cvoid}"},
    "cstring": {"Cstring"},
    "ptr_f64": {"Ptr{Float64}", "Ref{Float64}"},
    "ptr_ptr": {"Ptr{Ptr{Float64}}", "Ptr{Ptr{Cvoid}}"},
    "ptr_i32": {"Ptr{Int32}", "Ptr{Cint}"},
    "ptr_i64": {"Ptr{Int64}", "Ref{Int64}"},
    "handle_out": {"Ref{Ptr{Cvoid}}", "Ptr{Ptr{Cvoid}}"},
}


def split_top(s):
    """Split on commas at bracket depth 0."""
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def ccalls():
    src = open(GLUE).read()
    calls = []
    for m in re.finditer(r"ccall\(\(:([a-z0-9_]+), lib\),", src):
        i = m.end()
        depth, j = 1, i
        while depth:
            ch = src[j]
            depth += ch in "([{"
            depth -= ch in ")]}"
            j += 1
        body = src[i : j - 1]
        parts = split_top(body)
        ret, types = parts[0], parts[1]
        assert types.startswith("(") and types.endswith(")"), (m.group(1), types)
        tlist = split_top(types[1:-1])
        calls.append((m.group(1), ret, tlist, parts[2:], src.count("\n", 0, m.start()) + 1))
    return calls


def test_every_ccall_matches_the_header():
    protos = c_prototypes()
    calls = ccalls()
    assert len(calls) >= 45
    for name, ret, types, args, line in calls:
        assert name in protos, f"INSHip.jl:{line}: {name} is not declared in ins_hip.h"
        cret, cparams = protos[name]
        assert ret == {"int": "Cint", "const char*": "Cstring"}[cret], f"INSHip.jl:{line}: {name} returns {cret}, bound as {ret}"
        assert len(types) == len(cparams), f"INSHip.jl:{line}: {name} takes {len(cparams)} arguments, bound with {len(types)}"
        assert len(args) == len(types), f"INSHip.jl:{line}: {name}: {len(types)} argument types but {len(args)} arguments"
        for k, (jt, cp) in enumerate(zip(types, cparams)):
            kind = c_kind(cp)
            if name == "ins_grid_create" and k == 0:
                assert jt == "Ref{GridDesc}"
                continue
            assert jt in JULIA_OK[kind], f"INSHip.jl:{line}: {name} argument {k + 1} ({cp}) bound as {jt}"


def test_griddesc_mirrors_the_c_struct():
    hdr = open(HEADER).read()
    body = re.search(r"typedef struct ins_grid_desc \{(.*?)\} ins_grid_desc_t;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    cfields = []
    for decl in filter(None, (d.strip() for d in body.split(";"))):
        m = re.fullmatch(r"(const double\*|int32_t|double)\s+([A-Za-z0-9_]+)((?:\[\d+\])*)", decl)
        assert m, decl
        count = 1
        for d in re.findall(r"\[(\d+)\]", m.group(3)):
            count *= int(d)
        cfields.append((m.group(2), {"const double*": "Ptr{Float64}", "int32_t": "Int32", "double": "Float64"}[m.group(1)], count))
    src = open(GLUE).read()
    jbody = re.search(r"struct GridDesc\n(.*?)\nend", src, flags=re.S).group(1)
    jfields = []
    for ln in jbody.splitlines():
        ln = ln.split("#")[0].strip()
        if not ln:
            continue
        name, typ = ln.split("::")
        m = re.fullmatch(r"NTuple\{(\d+),(.+)\}", typ)
        jfields.append((name, m.group(2), int(m.group(1))) if m else (name, typ, 1))
    assert jfields == cfields


def test_tempdesc_mirrors_the_c_struct():
    hdr = open(HEADER).read()
    body = re.search(r"typedef struct ins_temperature_desc \{(.*?)\} ins_temperature_desc_t;", hdr, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    cfields = []
    for decl in filter(None, (d.strip() for d in body.split(";"))):
        m = re.fullmatch(r"(int32_t|double)\s+([A-Za-z0-9_]+)(?:\[(\d+)\])?", decl)
        assert m, decl
        cfields.append((m.group(2), {"int32_t": "Int32", "double": "Float64"}[m.group(1)], int(m.group(3) or 1)))
    src = open(GLUE).read()
    jbody = re.search(r"struct TempDesc\n(.*?)\nend", src, flags=re.S).group(1)
    jfields = []
    for ln in jbody.splitlines():
        name, typ = ln.strip().split("::")
        m = re.fullmatch(r"NTuple\{(\d+),(.+)\}", typ)
        jfields.append((name, m.group(2), int(m.group(1))) if m else (name, typ, 1))
    assert jfields == cfields


COVERED = [  # "methods divergence!, pressuregradient!, applypressure!, convection!, diffusion!, convectiondiffusion!, momentum!,
    # laplacian!, scalewithvolume!, apply_bc_u!, apply_bc_p!, project!, psolver_spectral, psolver_cg, timestep!" + what §8f added
    "divergence!", "pressuregradient!", "applypressure!", "convection!", "diffusion!", "convectiondiffusion!", "momentum!", "laplacian!",
    "scalewithvolume!", "apply_bc_u!", "apply_bc_p!", "apply_bc_temp!", "project!", "poisson!", "psolver_spectral", "psolver_cg", "psolver_direct",
    "timestep!", "ode_method_cache", "get_cfl_timestep!", "kinetic_energy!", "total_kinetic_energy", "vorticity!", "interpolate_u_p!",
    "interpolate_ω_p!", "Qfield!", "Dfield!", "eig2field!", "dissipation_from_strain!", "convection_diffusion_temp!", "dissipation!", "gravity!",
    "smagorinsky_closure", "laplacian_1d",
]


def test_every_covered_function_has_a_method():
    src = open(GLUE).read()
    code = "\n".join(ln.split("#")[0] if not ln.lstrip().startswith("#") else "" for ln in src.splitlines())
    for f in COVERED:
        pat = r"(?m)^(?:function\s+)?" + re.escape(f) + r"\("
        assert re.search(pat, code), f"no method of {f} in julia/INSHip.jl"
    imported = re.search(r"import IncompressibleNavierStokes:(.*?)\n\n", src, flags=re.S).group(1)
    for f in COVERED:
        if f in ("laplacian_1d",):
            continue
        assert re.search(r"(?<![A-Za-z_!])" + re.escape(f) + r"(?![A-Za-z_!])", imported), f"{f} is extended but not imported"
    assert "follow the same" not in src and "three-line pattern" not in src  # no comment standing in for methods
    # constant Dirichlet data reaches the descriptor (a lid-driven cavity is not a no-slip box)
    assert "bcconst(bcs[β][s], α)" in src and "ntuple(_ -> 0.0, 18)" not in src
    # every unqualified helper that is called is defined in the file
    for helper in ("laplacian_1d", "bc_planes", "isclosure", "bcconst", "bccode", "grid_handle", "handle", "native!", "check", "stream"):
        assert re.search(r"(?m)^(?:function\s+)?" + re.escape(helper) + r"\(", code), f"{helper} is called but not defined"

"""Row b (the drop-in boundary) pinned by test instead of by reading: the reference's own Python and C++ are PARSED (never imported,
nothing of them is copied) and every name it imports from `torch_bnb_fp4_ext`, every call site's positional arity and every
parameter type of the seven bound functions is checked against the signatures pybind11 prints for this build's extension.

/root/reference exists in the build container only - it never travels to the GPU box - so these tests skip where it is absent."""
import ast
import os
import re

import pytest
import torch  # noqa: F401

import torch_bnb_fp4 as pkg

REF = "/root/reference"
REF_INIT = os.path.join(REF, "torch_bnb_fp4", "__init__.py")
REF_CPP = os.path.join(REF, "csrc", "torch_fp4.cpp")
needs_reference = pytest.mark.skipif(not (os.path.exists(REF_INIT) and os.path.exists(REF_CPP)), reason="the reference checkout is not on this machine")

# C++ parameter type of the reference's bound functions -> the type pybind11 prints for the same position
CPP_TO_PYBIND = {"torch::Tensor": "torch.Tensor", "int": "typing.SupportsInt", "ScalarTypeEnum": "torch_bnb_fp4_ext.ScalarType",
                 "std::vector<uint32_t>": "collections.abc.Sequence[typing.SupportsInt]"}


def pybind_signature(fn):
    """('name', [type of arg0, ...], return type) from the first docstring line pybind11 generates."""
    head = fn.__doc__.splitlines()[0]
    m = re.match(r"(\w+)\((.*)\) -> (.+)$", head)
    assert m, head
    args = []
    depth, cur = 0, ""
    for ch in m.group(2):  # split on top-level commas (Sequence[...] holds none today, but stay safe)
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur)
    types = []
    for a in args:
        name, _, ty = a.strip().partition(":")
        assert re.fullmatch(r"arg\d+", name.strip()), f"{head}: the reference binds positionally (no py::arg names): {a!r}"
        types.append(ty.strip())
    return m.group(1), types, m.group(3).strip()


def reference_ext_imports():
    """{local alias: extension attribute} for every `from torch_bnb_fp4_ext import X as Y` of the reference package (:11-18)."""
    tree = ast.parse(open(REF_INIT).read())
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ImportFrom) and node.module == "torch_bnb_fp4_ext":
            for a in node.names:
                out[a.asname or a.name] = a.name
        elif isinstance(node, ast.Import):
            assert all(a.name != "torch_bnb_fp4_ext" for a in node.names), "module-style import: extend this test"
    return tree, out


@needs_reference
def test_every_extension_name_the_reference_imports_exists_here():
    _, imports = reference_ext_imports()
    assert set(imports.values()) == {"ScalarType", "dequantize_fp4", "gemv_fp4", "qlinear", "qlinear_bias", "dequantize_fp4_codebook",
                                     "qlinear_codebook", "qlinear_codebook_bias"}, imports  # reference __init__.py:11-18
    for attr in imports.values():
        assert hasattr(pkg.ext, attr), attr
    # the enum members the reference's Python enum re-exports (:27-29) and their C++ numbering (torch_fp4.cpp:22-26)
    tree, _ = reference_ext_imports()
    members = {n.attr for n in ast.walk(tree) if isinstance(n, ast.Attribute) and isinstance(n.value, ast.Name) and n.value.id == "ScalarType_"}
    assert members == {"bfloat16", "float16", "float32"}
    assert {k: int(v) for k, v in pkg.ext.ScalarType.__members__.items()} == {"float16": 0, "float32": 1, "bfloat16": 2}


@needs_reference
def test_every_reference_call_site_fits_the_bound_signature():
    tree, imports = reference_ext_imports()
    calls = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name) and n.func.id in imports
             and imports[n.func.id] != "ScalarType"]
    seen = {}
    for c in calls:
        attr = imports[c.func.id]
        _, types, ret = pybind_signature(getattr(pkg.ext, attr))
        assert not c.keywords, f"reference line {c.lineno}: keyword arguments cannot reach a positional pybind binding"
        assert not any(isinstance(a, ast.Starred) for a in c.args), c.lineno
        assert len(c.args) == len(types), f"reference line {c.lineno}: {attr} called with {len(c.args)} arguments, bound with {len(types)}"
        assert ret == "torch.Tensor"
        seen.setdefault(attr, []).append(c.lineno)
    # the wrappers (:119,159,208,255,295,330) and the qlinear family (:507,516,539,549); every function is called somewhere
    assert set(seen) == set(imports.values()) - {"ScalarType"}, seen
    assert sum(len(v) for v in seen.values()) == 10, seen


def reference_cpp_bindings():
    """{python name: [C++ parameter types]} from PYBIND11_MODULE's m.def lines and the definitions they point at."""
    text = open(REF_CPP).read()
    text = re.sub(r"//[^\n]*", "", text)
    defs = dict(re.findall(r'm\.def\(\s*"(\w+)"\s*,\s*&(\w+)', text))
    enum_values = re.findall(r'\.value\(\s*"(\w+)"', text)
    out = {}
    for py_name, cpp_name in defs.items():
        m = re.search(r"torch::Tensor\s+" + cpp_name + r"\s*\(([^)]*)\)\s*\{", text)
        assert m, cpp_name
        params = [p.strip() for p in m.group(1).split(",") if p.strip()]
        out[py_name] = [p.rsplit(None, 1)[0].replace("const ", "").replace("&", "").strip() for p in params]
    return out, enum_values


@needs_reference
def test_the_seven_bound_functions_have_the_reference_parameter_types_in_order():
    bindings, enum_values = reference_cpp_bindings()
    assert sorted(bindings) == sorted(["dequantize_fp4", "dequantize_fp4_codebook", "gemv_fp4", "qlinear", "qlinear_bias", "qlinear_codebook",
                                       "qlinear_codebook_bias"])  # torch_fp4.cpp:132-138
    assert sorted(enum_values) == ["bfloat16", "float16", "float32"]  # :127-129
    for name, cpp_types in bindings.items():
        _, types, ret = pybind_signature(getattr(pkg.ext, name))
        want = [CPP_TO_PYBIND[t] for t in cpp_types]
        assert types == want, f"{name}: bound as {types}, the reference declares {cpp_types}"
        assert ret == "torch.Tensor"


def test_pybind_signature_parser_on_this_build():
    """Runs everywhere (no reference needed): the parser itself, and the arities SURVEY 8b lists, against this build's extension."""
    want = {"dequantize_fp4": 6, "dequantize_fp4_codebook": 8, "gemv_fp4": 7, "qlinear": 6, "qlinear_bias": 7, "qlinear_codebook": 7,
            "qlinear_codebook_bias": 8}
    for name, arity in want.items():
        got_name, types, ret = pybind_signature(getattr(pkg.ext, name))
        assert (got_name, len(types), ret) == (name, arity, "torch.Tensor")
    assert pybind_signature(pkg.ext.gemv_fp4)[1][-1] == CPP_TO_PYBIND["std::vector<uint32_t>"]


# ---- counter evidence is tied to the source it profiled (bench.committed_traffic / tools/source_digest.py) ---------------------------

def _fake_tree(tmp_path):
    import shutil
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(repo, "tools"))
    import source_digest

    root = tmp_path / "tree"
    for rel in source_digest.file_digests(repo):
        dst = root / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(repo, rel), dst)
    (root / "profiles").mkdir()
    return root, source_digest


def test_traffic_figure_is_fresh_while_the_kernel_source_is_unchanged_and_stale_after_an_edit(tmp_path):
    import json

    import bench

    root, sd = _fake_tree(tmp_path)
    rec = {"_method": "test", "_source_sha256": sd.file_digests(str(root)),
           "dequant_tiles_kernel<2, 4, true>": {"traffic_bytes": 43041742}, "gemv16_regx_kernel<2, 4, 1, 2, 4>": {"traffic_bytes": 9593610}}
    (root / "profiles" / "r05_traffic.json").write_text(json.dumps(rec))
    for prefix in ("dequant_tiles_kernel<2,", "gemv16_regx_kernel<2,"):
        traffic, src, stale, why = bench.committed_traffic(prefix, repo=str(root))
        assert traffic and src.startswith("profiles/r05_traffic.json:") and stale is False and why is None
    # edit the GEMV's source: its figure goes stale, the dequant's (other files) stays fresh
    gemv = root / "torch-bnb-fp4_amd" / "csrc" / "gemv_fp4.hip"
    gemv.write_text(gemv.read_text() + "\n// edited\n")
    traffic, _, stale, why = bench.committed_traffic("gemv16_regx_kernel<2,", repo=str(root))
    assert traffic == 9593610 and stale is True and "gemv_fp4.hip" in why  # the number is kept, and marked
    assert bench.committed_traffic("dequant_tiles_kernel<2,", repo=str(root))[2] is False
    # a shared header or the build recipe (compiler flags) touches every kernel
    hdr = root / "torch-bnb-fp4_amd" / "csrc" / "fp4_common.h"
    hdr.write_text(hdr.read_text() + "\n// edited\n")
    assert bench.committed_traffic("dequant_tiles_kernel<2,", repo=str(root))[2] is True


def test_a_profile_without_digests_is_reported_stale(tmp_path):
    import json

    import bench

    root, _ = _fake_tree(tmp_path)
    (root / "profiles" / "r04_traffic.json").write_text(json.dumps({"_method": "old", "dequant_tiles_kernel<2, 4, true>": {"traffic_bytes": 1}}))
    traffic, _, stale, why = bench.committed_traffic("dequant_tiles_kernel<2,", repo=str(root))
    assert traffic == 1 and stale is True and "no source digests" in why


# ---- the Python surface (functions, classes, methods, parameter names, defaults) ----------------------------------------------------

def _ref_params(fn: ast.FunctionDef):
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args if x.arg not in ("self", "cls")]
    defaults = {}
    pos = [x.arg for x in a.posonlyargs + a.args]
    for name, d in zip(pos[len(pos) - len(a.defaults):], a.defaults):
        try:
            defaults[name] = ast.literal_eval(d)
        except (ValueError, SyntaxError):
            defaults[name] = ast.unparse(d)  # an expression (torch.float16, torch.device(...)): compared as text below
    return names, defaults


def _our_params(obj):
    import inspect

    sig = inspect.signature(obj)
    ps = [p for p in sig.parameters.values() if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) and p.name not in ("self", "cls")]
    return [p.name for p in ps], {p.name: p.default for p in ps if p.default is not p.empty}


def _same_default(ref, ours) -> bool:
    if isinstance(ref, str) and not isinstance(ours, str):  # an expression on the reference side, kept as text
        text = ref.replace(" ", "")
        if text.startswith("torch.device("):
            return isinstance(ours, torch.device)  # (the reference picks cuda-if-available at import time; so do we)
        # torch.float16, ScalarType.bfloat16.value, ...: evaluated against THIS package's names (a two-name namespace, reference text
        # is only ever an attribute chain here - anything else fails the comparison rather than being executed blindly)
        if not re.fullmatch(r"(torch|ScalarType)(\.\w+)+", text):
            return False
        obj = {"torch": torch, "ScalarType": pkg.ScalarType}[text.split(".")[0]]
        for part in text.split(".")[1:]:
            obj = getattr(obj, part)
        return obj == ours
    return ref == ours


@needs_reference
def test_python_surface_is_a_superset_of_the_reference_package():
    """Every module-level function and every method of ScalarType / QuantData / TorchFP4Linear in the reference's
    torch_bnb_fp4/__init__.py exists here under the same name, takes the same leading positional parameters BY NAME (keyword calls keep
    working) and has the same default values; extra trailing parameters with defaults are allowed (this package's opt-in extensions)."""
    import inspect

    tree = ast.parse(open(REF_INIT).read())
    checked = 0
    for node in tree.body:
        targets = []
        if isinstance(node, ast.FunctionDef):
            targets.append((node.name, node, getattr(pkg, node.name, None)))
        elif isinstance(node, ast.ClassDef):
            cls = getattr(pkg, node.name, None)
            assert cls is not None, f"class {node.name} is missing"
            for m in node.body:
                if isinstance(m, ast.FunctionDef):
                    targets.append((f"{node.name}.{m.name}", m, inspect.getattr_static(cls, m.name, None)))
        for name, ref_fn, ours in targets:
            assert ours is not None, f"{name} is missing"
            if isinstance(ours, property):
                continue  # (ScalarType.torch_dtype: a property on both sides)
            if isinstance(ours, (classmethod, staticmethod)):
                ours = ours.__func__
            r_names, r_defaults = _ref_params(ref_fn)
            o_names, o_defaults = _our_params(ours)
            assert o_names[:len(r_names)] == r_names, f"{name}: reference parameters {r_names}, ours {o_names}"
            assert all(n in o_defaults for n in o_names[len(r_names):]), f"{name}: an extra parameter without a default breaks reference callers"
            for pname, rd in r_defaults.items():
                assert pname in o_defaults and _same_default(rd, o_defaults[pname]), f"{name}({pname}=...): reference default {rd!r}, ours {o_defaults.get(pname)!r}"
            checked += 1
    assert checked >= 25, checked

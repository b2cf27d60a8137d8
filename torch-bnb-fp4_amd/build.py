"""In-tree build of the two native artefacts (gfx950 only):

* ``torch_bnb_fp4/lib/libtorch_bnb_fp4_hip.so`` - the HIP kernels behind the C ABI
  (``include/torch_bnb_fp4_hip.h``); no torch dependency.
* ``torch_bnb_fp4_ext.so`` - the pybind/torch host layer (``csrc/torch_ext.cpp``), linked against
  the first with an ``$ORIGIN`` rpath.

Both are rebuilt only when a source is newer than the artefact.  hipcc cross-compiles without a
GPU, so this runs in CI as well as on the GPU box.  Usage: ``python build.py [--force] [--no-ext]``.
"""
from __future__ import annotations

import os
import shlex
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(REPO, "include")
LIB_DIR = os.path.join(HERE, "torch_bnb_fp4", "lib")
HIP_LIB = os.path.join(LIB_DIR, "libtorch_bnb_fp4_hip.so")
EXT_LIB = os.path.join(HERE, "torch_bnb_fp4_ext.so")
OBJ_DIR = os.path.join(REPO, "build_tmp", "obj")

HIP_SOURCES = ["capi.hip", "dequant_fp4.hip", "gemv_fp4.hip", "gemm_small_fp4.hip", "gemm_wide_fp4.hip", "gemm_splitk_fp4.hip", "quantize_fp4.hip", "allreduce_oneshot.hip"]
HIP_HEADERS = ["fp4_common.h", "gemv_common.h", os.path.join(INCLUDE, "torch_bnb_fp4_hip.h")]
ARCH = "gfx950"
# Kernel arguments are preloaded into SGPRs at wave launch instead of being fetched with s_load at the top of the
# kernel: for kernels this short that is measurable (4096x4096 bf16: GEMV 4.16 -> 3.91 us, dequant 7.44 -> 7.22 us).
HIPCC_FLAGS = ["-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _hipcc() -> str:
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cand = os.path.join(rocm, "bin", "hipcc")
    return cand if os.path.exists(cand) else "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd) -> None:
    print("+", " ".join(shlex.quote(c) for c in cmd), flush=True)
    subprocess.check_call(cmd)


def parse_resource_usage(text: str):
    """hipcc -Rpass-analysis=kernel-resource-usage remarks -> {kernel: {"vgprs", "sgprs", "scratch", "vgpr_spill", "sgpr_spill",
    "occupancy", "lds"}} (the last record of a kernel wins: remarks repeat per offload pass)."""
    import re

    fields = {"VGPRs": "vgprs", "TotalSGPRs": "sgprs", "ScratchSize [bytes/lane]": "scratch", "VGPRs Spill": "vgpr_spill",
              "SGPRs Spill": "sgpr_spill", "Occupancy [waves/SIMD]": "occupancy", "LDS Size [bytes/block]": "lds"}
    out, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+?): (\d+) \[-Rpass-analysis", line)
        if m and cur is not None and m.group(1) in fields:
            cur[fields[m.group(1)]] = int(m.group(2))
    return out


def check_no_spills(usage, source: str) -> None:
    """A shipped kernel that spills runs from scratch memory: refuse to build it (an unreachable instantiation that
    spills is dead weight one dispatch typo away from running - delete it instead)."""
    bad = {k: v for k, v in usage.items() if v.get("vgpr_spill", 0) or v.get("sgpr_spill", 0) or v.get("scratch", 0)}
    if bad:
        lines = [f"  {k}: {v}" for k, v in sorted(bad.items())]
        raise RuntimeError(f"{source}: {len(bad)} kernel(s) spill registers / use scratch:\n" + "\n".join(lines))


def _compile_one(src: str, obj: str, extra) -> dict:
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", "-ffp-contract=off", "-fvisibility=hidden",
           "-Wall", "-Wextra", "-Rpass-analysis=kernel-resource-usage", *HIPCC_FLAGS, *extra, f"-I{INCLUDE}", f"-I{CSRC}", src,
           "-o", obj]
    print("+", " ".join(shlex.quote(c) for c in cmd), flush=True)
    p = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    diag = "\n".join(l for l in p.stderr.splitlines() if "-Rpass-analysis" not in l and not l.lstrip().startswith(("|", "^"))
                     and not __import__("re").match(r"\s*\d+ \|", l))
    if diag.strip():
        print(diag, file=sys.stderr, flush=True)
    if p.returncode != 0:
        raise subprocess.CalledProcessError(p.returncode, cmd)
    usage = parse_resource_usage(p.stderr)
    check_no_spills(usage, os.path.basename(src))
    return usage


def build_hip_lib(force: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in HIP_SOURCES]
    deps = srcs + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HIP_HEADERS] + [os.path.abspath(__file__)]
    if force or _stale(HIP_LIB, deps):
        import json
        from concurrent.futures import ThreadPoolExecutor

        os.makedirs(LIB_DIR, exist_ok=True)
        os.makedirs(OBJ_DIR, exist_ok=True)
        extra = shlex.split(os.environ.get("FP4_EXTRA_HIPCC_FLAGS", ""))  # experiments only
        objs = [os.path.join(OBJ_DIR, os.path.basename(s) + ".o") for s in srcs]
        # one translation unit per source, compiled side by side; every kernel's register / scratch use is checked
        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as pool:
            usages = list(pool.map(lambda so: _compile_one(so[0], so[1], extra), zip(srcs, objs)))
        _run([_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fvisibility=hidden", *objs, "-o", HIP_LIB])
        merged = {k: v for u in usages for k, v in u.items()}
        with open(os.path.join(LIB_DIR, "kernel_resources.json"), "w") as f:
            json.dump(merged, f, indent=0, sort_keys=True)
        print(f"{len(merged)} kernels, none spills", flush=True)
    return HIP_LIB


def build_torch_ext(force: bool = False) -> str:
    src = os.path.join(CSRC, "torch_ext.cpp")
    deps = [src, os.path.join(INCLUDE, "torch_bnb_fp4_hip.h"), HIP_LIB, os.path.abspath(__file__)]
    if not (force or _stale(EXT_LIB, deps)):
        return EXT_LIB
    import torch
    from torch.utils import cpp_extension as ce

    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", f"-I{INCLUDE}",
                                                     f"-I{os.environ.get('ROCM_PATH', '/opt/rocm')}/include"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    _run([_hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=torch_bnb_fp4_ext",
          "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-DUSE_ROCM", "-Wno-unused-parameter",
          *inc, src, "-o", EXT_LIB, f"-L{torch_lib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch", "-ltorch_hip",
          "-ltorch_python", "-lhipblaslt", "-lamdhip64", f"-L{LIB_DIR}", "-ltorch_bnb_fp4_hip", "-Wl,-rpath,$ORIGIN/torch_bnb_fp4/lib",
          f"-Wl,-rpath,{torch_lib}"])
    return EXT_LIB


PROBE_SRC = os.path.join(REPO, "tools", "stream_probe.hip")
PROBE_LIB = os.path.join(REPO, "tools", "libfp4_stream_probe.so")


def build_stream_probe(force: bool = False) -> str:
    """bench.py's measuring stick (what this box streams with the dequant kernel's access geometry and no arithmetic): not part of
    the product, not behind its ABI - but a HIP artefact the bench loads, so it is built with everything else and travels with it."""
    if force or _stale(PROBE_LIB, [PROBE_SRC, os.path.abspath(__file__)]):
        _run([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-shared", "-fPIC", *HIPCC_FLAGS, PROBE_SRC, "-o", PROBE_LIB])
    return PROBE_LIB


def build_all(force: bool = False, ext: bool = True, probe: bool = True):
    """The product's two artefacts, then (probe=True: the test session, __graft_entry__.build(), this file run as a script) bench.py's
    measuring stick.  The probe is not part of the product: packaging (setup.py) leaves it out, and a missing source or a failed
    compile of it never fails a build - bench.py reports its figures as unavailable instead."""
    out = [build_hip_lib(force)]
    if ext:
        out.append(build_torch_ext(force))
    if probe:
        if not os.path.exists(PROBE_SRC):
            print(f"note: {os.path.relpath(PROBE_SRC, REPO)} is absent; bench.py will run without its box-stream figures", file=sys.stderr, flush=True)
        else:
            try:
                out.append(build_stream_probe(force))
            except (subprocess.CalledProcessError, OSError) as exc:
                print(f"warning: the stream probe (bench-only) did not build: {exc}; bench.py will run without its box-stream figures",
                      file=sys.stderr, flush=True)
    return out


if __name__ == "__main__":
    for path in build_all(force="--force" in sys.argv, ext="--no-ext" not in sys.argv):
        print("built", os.path.relpath(path, REPO))

"""Ties counter evidence to the source it profiled: SHA-256 of every file a kernel's machine code depends on.

tools/pmc_traffic.py writes `file_digests()` into profiles/rNN_traffic.json (`_source_sha256`); bench.py compares the digests of the
files behind the kernel it quotes (`sources_of`) with the tree it runs from and marks the figure `traffic_stale` on a mismatch, so a
kernel edit can never ship with an earlier round's counters presented as its own."""
import glob
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join("torch-bnb-fp4_amd", "csrc")
COMMON = [os.path.join(CSRC, "fp4_common.h"), os.path.join("torch-bnb-fp4_amd", "build.py")]  # code tables / conversions; compiler flags
# kernel-name prefix (as it appears in profiles/rNN_traffic.json) -> the files its code object is compiled from
KERNEL_SOURCES = {
    "dequant_": [os.path.join(CSRC, "dequant_fp4.hip")] + COMMON,
    "gemv": [os.path.join(CSRC, "gemv_fp4.hip"), os.path.join(CSRC, "gemv_common.h")] + COMMON,
    "quantize_": [os.path.join(CSRC, "quantize_fp4.hip")] + COMMON,
    # (longest prefix first: dict order is lookup order)
    "gemm16_wide": [os.path.join(CSRC, "gemm_wide_fp4.hip"), os.path.join(CSRC, "gemv_common.h")] + COMMON,
    "gemm16_xstat": [os.path.join(CSRC, "gemm_splitk_fp4.hip"), os.path.join(CSRC, "gemv_common.h")] + COMMON,
    "splitk_reduce": [os.path.join(CSRC, "gemm_splitk_fp4.hip"), os.path.join(CSRC, "gemv_common.h")] + COMMON,
    "gemm16_": [os.path.join(CSRC, "gemm_small_fp4.hip"), os.path.join(CSRC, "gemv_common.h")] + COMMON,
}


def file_digests(repo: str = REPO) -> dict:
    """{path relative to the repository: sha256 hex} for every kernel source, header and the build recipe."""
    files = sorted(glob.glob(os.path.join(repo, CSRC, "*.hip")) + glob.glob(os.path.join(repo, CSRC, "*.h")))
    files.append(os.path.join(repo, "torch-bnb-fp4_amd", "build.py"))
    return {os.path.relpath(f, repo): hashlib.sha256(open(f, "rb").read()).hexdigest() for f in files}


def sources_of(kernel: str):
    for prefix, files in KERNEL_SOURCES.items():
        if kernel.startswith(prefix):
            return files
    return None  # unknown kernel: every file counts


def stale(recorded, kernel: str, repo: str = REPO):
    """(is_stale, reason) for a traffic record taken when the sources had the `recorded` digests (None: the profile predates them)."""
    if not isinstance(recorded, dict) or not recorded:
        return True, "the profile carries no source digests (taken before round 5): it cannot be tied to this tree"
    now = file_digests(repo)
    files = sources_of(kernel) or sorted(now)
    changed = [f for f in files if recorded.get(f) != now.get(f)]
    if changed:
        return True, "changed since the profile was taken: " + ", ".join(changed)
    return False, None

"""The C-ABI library loads and exports exactly what include/torch_bnb_fp4_hip.h declares; argument
validation (which returns before any HIP call) behaves as documented.  No GPU needed."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import hipabi
from oracle import fp4_oracle as o


def test_header_symbols_are_exported():
    declared = hipabi.declared_symbols()
    assert set(declared) >= {"fp4_hip_abi_version", "fp4_hip_last_error", "fp4_hip_code_table", "fp4_hip_dequantize_blockwise",
                             "fp4_hip_gemv", "fp4_hip_gemv_partial", "fp4_hip_gemm_small", "fp4_hip_quantize_blockwise", "fp4_hip_set_variant"}
    l = hipabi.lib()
    for name in declared:
        assert hasattr(l, name), name
    # and nothing else leaks out of the library (built with -fvisibility=hidden)
    nm = subprocess.run(["nm", "-D", "--defined-only", hipabi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    assert exported == set(declared), exported ^ set(declared)


def test_abi_version_and_tables():
    l = hipabi.lib()
    header = open(hipabi.HEADER).read()
    assert l.fp4_hip_abi_version() == int(header.split('#define FP4_HIP_ABI_VERSION')[1].split()[0]) == 7
    for which, tab in ((hipabi.TABLE_CODEBOOK, o.CODEBOOK_TABLE), (hipabi.TABLE_TREE, o.TREE_TABLE)):
        out = np.zeros(16, np.float32)
        assert l.fp4_hip_code_table(which, out.ctypes.data_as(ctypes.c_void_p)) == hipabi.OK
        assert (out.view(np.uint32) == tab.view(np.uint32)).all()
    assert l.fp4_hip_code_table(7, None) == hipabi.ERR_INVALID


def test_argument_validation_without_gpu():
    l = hipabi.lib()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    assert l.fp4_hip_dequantize_blockwise(one, one, one, 64, -1, hipabi.F16, 0, 0, None) == hipabi.ERR_INVALID
    assert l.fp4_hip_dequantize_blockwise(one, one, one, 63, 128, hipabi.F16, 0, 0, None) == hipabi.ERR_INVALID
    assert "blocksize" in hipabi.last_error()
    assert l.fp4_hip_dequantize_blockwise(one, one, one, 64, 128, hipabi.F16, 5, 0, None) == hipabi.ERR_INVALID
    assert l.fp4_hip_dequantize_blockwise(None, one, one, 64, 128, hipabi.F16, 0, 0, None) == hipabi.ERR_INVALID
    assert l.fp4_hip_dequantize_blockwise(one, one, one, 64, 128, 9, 0, 0, None) == hipabi.ERR_UNSUPPORTED
    assert l.fp4_hip_dequantize_blockwise(None, None, None, 64, 0, hipabi.F16, 0, 0, None) == hipabi.OK  # empty input
    assert l.fp4_hip_dequantize_blockwise(one, one, one, 64, 128, hipabi.F16, 0, 7, None) == hipabi.ERR_INVALID  # bad flags
    assert l.fp4_hip_gemv(one, one, one, None, one, 4, 63, 64, hipabi.BF16, None) == hipabi.ERR_INVALID  # odd K
    assert l.fp4_hip_gemv(one, one, one, None, one, 4, 64, 64, 9, None) == hipabi.ERR_UNSUPPORTED
    assert l.fp4_hip_gemv(None, None, None, None, None, 0, 64, 64, hipabi.BF16, None) == hipabi.OK  # M == 0
    assert l.fp4_hip_quantize_blockwise(one, hipabi.F16, one, one, 64, 48, None) == hipabi.ERR_UNSUPPORTED
    # fused epilogues and the one-shot all-reduce: argument errors are reported before anything touches a GPU
    l.fp4_hip_gemv_fused.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    assert l.fp4_hip_gemv_fused(one, one, one, None, None, one, 4, 64, 64, hipabi.BF16, 9, None) == hipabi.ERR_INVALID  # unknown epilogue
    assert l.fp4_hip_gemv_fused(one, one, one, None, None, one, 5, 64, 64, hipabi.BF16, 1, None) == hipabi.ERR_INVALID  # odd rows, gate|up
    assert "even row count" in hipabi.last_error()
    l.fp4_hip_comm_bytes.restype = ctypes.c_int64
    l.fp4_hip_comm_bytes.argtypes = [ctypes.c_int, ctypes.c_int64]
    assert l.fp4_hip_comm_bytes(8, 16384) == 256 + 2 * 8 * 16384 * 8 and l.fp4_hip_comm_bytes(17, 1) == -1 and l.fp4_hip_comm_bytes(2, 0) == -1
    l.fp4_hip_allreduce_oneshot.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    bufs = (ctypes.c_void_p * 2)(16, 16)
    assert l.fp4_hip_allreduce_oneshot(one, bufs, 2, 2, 64, 1024, None, None, one, hipabi.BF16, 0, None) == hipabi.ERR_INVALID  # rank >= world
    assert l.fp4_hip_allreduce_oneshot(one, bufs, 0, 2, 2048, 1024, None, None, one, hipabi.BF16, 0, None) == hipabi.ERR_INVALID  # M > capacity
    assert l.fp4_hip_allreduce_oneshot(one, bufs, 0, 2, 64, 1024, None, None, one, 9, 0, None) == hipabi.ERR_UNSUPPORTED
    assert l.fp4_hip_allreduce_oneshot(None, None, 0, 2, 0, 1024, None, None, None, hipabi.BF16, 0, None) == hipabi.OK  # nothing to do
    assert l.fp4_hip_set_variant(b"nope", 1) == hipabi.ERR_INVALID
    assert l.fp4_hip_set_variant(b"gemv", -1) == hipabi.OK


def test_build_refuses_kernels_that_spill():
    """build.py parses hipcc's kernel-resource-usage remarks and fails the build on any spill / scratch use."""
    import importlib.util
    import json

    PKG_ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "torch-bnb-fp4_amd")
    spec = importlib.util.spec_from_file_location("fp4_build", os.path.join(PKG_ROOT, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    remark = ("a.hip:1:1: remark: Function Name: k_good [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:1:1: remark:     VGPRs: 64 [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:1:1: remark:     ScratchSize [bytes/lane]: 0 [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:1:1: remark:     VGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:9:1: remark: Function Name: k_bad [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:9:1: remark:     VGPRs: 128 [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:9:1: remark:     ScratchSize [bytes/lane]: 100 [-Rpass-analysis=kernel-resource-usage]\n"
              "a.hip:9:1: remark:     VGPRs Spill: 25 [-Rpass-analysis=kernel-resource-usage]\n")
    usage = mod.parse_resource_usage(remark)
    assert usage == {"k_good": {"vgprs": 64, "scratch": 0, "vgpr_spill": 0}, "k_bad": {"vgprs": 128, "scratch": 100, "vgpr_spill": 25}}
    mod.check_no_spills({"k_good": usage["k_good"]}, "a.hip")
    with pytest.raises(RuntimeError, match="k_bad"):
        mod.check_no_spills(usage, "a.hip")
    # the library that was actually built: every shipped kernel is listed and none spills
    shipped = json.load(open(os.path.join(PKG_ROOT, "torch_bnb_fp4", "lib", "kernel_resources.json")))
    assert len(shipped) > 100 and all(not (v.get("vgpr_spill") or v.get("sgpr_spill") or v.get("scratch")) for v in shipped.values())

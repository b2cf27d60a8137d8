"""The reference-side binding INTEGRATION.md documents (section 3: the three functions csrc/torch_fp4.cpp:5-17 of the reference
declares, as thin shims over the C ABI; section 4: one more for the fused epilogue) is extracted from the document and COMPILED,
so that the documented code cannot rot: same torch headers the reference's own file includes, this repo's C header, `hipcc -c`
(no link against reference objects, no GPU).  The shims' signatures are pinned against the prototypes the reference's pybind file
expects (types only - those are the interface facts a drop-in must match)."""
import os
import re
import subprocess
import sys
import sysconfig

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpp_blocks(section_title: str):
    text = open(os.path.join(REPO, "INTEGRATION.md")).read()
    start = text.index(section_title)
    nxt = re.search(r"^## ", text[start + 3:], re.M)
    body = text[start:start + 3 + nxt.start()] if nxt else text[start:]
    return re.findall(r"```cpp\n(.*?)```", body, re.S)


def test_documented_shims_compile_against_the_c_header(tmp_path):
    import torch
    from torch.utils import cpp_extension as ce

    sec3, sec4 = cpp_blocks("## 3. Bind the C ABI directly"), cpp_blocks("## 4. Beyond the reference's surface")
    assert len(sec3) == 1 and len(sec4) >= 1
    for fn in ("dequantize_blockwise_fp4", "dequantize_blockwise_codebook_fp4", "gemv_4bit_inference"):
        assert fn in sec3[0]
    src = tmp_path / "shim.cpp"
    src.write_text(
        "#include <torch/extension.h>\n#include <type_traits>\n#include <vector>\n"
        + sec3[0] + "\n" + sec4[0] + "\n"
        # what the reference's csrc/torch_fp4.cpp:5-17 declares and then calls: the shims must have exactly these types
        + "static_assert(std::is_same_v<decltype(&dequantize_blockwise_fp4), void (*)(torch::Tensor, torch::Tensor, int, int, int, int, torch::Tensor)>);\n"
        + "static_assert(std::is_same_v<decltype(&dequantize_blockwise_codebook_fp4), torch::Tensor (*)(torch::Tensor, torch::Tensor, torch::Tensor, int, int, int, int, torch::ScalarType)>);\n"
        + "static_assert(std::is_same_v<decltype(&gemv_4bit_inference), torch::Tensor (*)(torch::Tensor, torch::Tensor, torch::Tensor, torch::Tensor, int, torch::ScalarType, std::vector<uint32_t>)>);\n")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    hipcc = os.path.join(rocm, "bin", "hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    inc = [f"-I{p}" for p in ce.include_paths()] + [f"-I{sysconfig.get_paths()['include']}", f"-I{os.path.join(REPO, 'include')}",
                                                     f"-I{rocm}/include"]
    cmd = [hipcc, "-std=c++17", "-O0", "-fPIC", "-c", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-unused-parameter", *inc, str(src), "-o",
           str(tmp_path / "shim.o")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    # every C-ABI entry point the shims call is one the header declares (and the library exports: tests/test_abi.py)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import hipabi

    declared = set(hipabi.declared_symbols())
    used = set(re.findall(r"\b(fp4_hip_\w+)\s*\(", sec3[0] + sec4[0]))
    assert used and used <= declared, used - declared

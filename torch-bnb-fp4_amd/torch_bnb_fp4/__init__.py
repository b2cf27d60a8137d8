"""torch_bnb_fp4 for AMD Instinct MI355X (gfx950): FP4 Linear inference over bitsandbytes-format weights.

Same public surface as aredden/torch-bnb-fp4's ``torch_bnb_fp4`` package (reference
torch_bnb_fp4/__init__.py), backed by hand-written CDNA4 HIP kernels:

* functional ops      - :mod:`torch_bnb_fp4.functional`   (reference :87-337)
* dispatcher          - :class:`QuantData`                (reference :340-618)
* nn.Module shell     - :class:`TorchFP4Linear`           (reference :621-714)
* model surgery       - :func:`recursively_replace_with_fp4_linear` and helpers (reference :717-922)

bitsandbytes is optional: :mod:`torch_bnb_fp4.nn` provides attribute-compatible ``LinearFP4`` /
``Params4bit`` / ``QuantState`` and the quantiser runs on the GPU through this package.
"""
from ._ext import HIP_LIBRARY_PATH, ext
from .dtypes import ScalarType
from .functional import (
    dequantize_fp4,
    dequantize_fp4_codebook_invoke,
    dequantize_fp4_codebook_invoke_qtype,
    dequantize_fp4_qtype,
    gemm_4bit_inference,
    gemm_4bit_inference_qtype,
    quantize_fp4,
)
from .comm import OneShotAllReduce
from .fused import FusedFP4Linear
from .graphs import GraphedStep
from .linear import TorchFP4Linear
from .nn import Linear4bit, LinearFP4, Params4bit, QuantState
from .quant_data import QuantData
from .serialization import fp4_linear_from_bnb_state, fp4_linear_to_bnb_state, load_fp4_layers, save_fp4_model
from .surgery import (
    FusedGatedMLP,
    check_if_name_contained_in_list,
    fuse_gated_mlps,
    recursively_replace_with_fp4_linear,
    set_small_batch_fused,
    swap_linear_with_bnb_linear,
    todevice_if_necessary,
)

__all__ = [
    "ScalarType",
    "dequantize_fp4",
    "dequantize_fp4_codebook_invoke_qtype",
    "dequantize_fp4_codebook_invoke",
    "gemm_4bit_inference",
    "gemm_4bit_inference_qtype",
    "dequantize_fp4_qtype",
    "quantize_fp4",
    "QuantData",
    "TorchFP4Linear",
    "swap_linear_with_bnb_linear",
    "check_if_name_contained_in_list",
    "todevice_if_necessary",
    "recursively_replace_with_fp4_linear",
    "LinearFP4",
    "Linear4bit",
    "Params4bit",
    "QuantState",
    "fp4_linear_to_bnb_state",
    "fp4_linear_from_bnb_state",
    "save_fp4_model",
    "load_fp4_layers",
    "set_small_batch_fused",
    "fuse_gated_mlps",
    "FusedGatedMLP",
    "FusedFP4Linear",
    "OneShotAllReduce",
    "GraphedStep",
]
__version__ = "0.1.0"

import numpy as np
import torch

NPDT = {torch.float16: "float16", torch.float32: "float32", torch.bfloat16: "bfloat16"}


def dev():
    return torch.device("cuda", 0)


def to_dev(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def bits(t: torch.Tensor) -> np.ndarray:
    """Raw bit patterns of a float tensor as an integer numpy array."""
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.float32:
        return t.view(torch.int32).numpy().view(np.uint32)
    return t.view(torch.int16).numpy().view(np.uint16)


def np_bits(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint16)


def torch_values(x: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """float array -> device tensor of ``dtype`` (rounded by torch, RNE)."""
    return torch.from_numpy(np.asarray(x, np.float32)).to(dtype).to(dev())

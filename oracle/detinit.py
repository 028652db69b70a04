"""Deterministic, name-keyed parameter values (TEST INFRASTRUCTURE).

Golden fixtures are made by loading these values into the reference's modules
(tools/gen_golden.py); the oracle and the HIP path regenerate the same values
from the parameter name alone, so no weight file has to be committed.
"""
import zlib

import numpy as np
import torch


def det_tensor(name, shape, dtype=torch.float32):
    """N(0,1)-based values keyed by `name`: conv/linear weights ~ N(0, 1/fan_in),
    1-D tensors ~ 0.1*N(0,1) (+1 for *norm*.weight)."""
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(name.encode("utf-8"))))
    a = rng.standard_normal(size=tuple(shape)).astype(np.float64)
    if len(shape) > 1:
        fan_in = int(np.prod(shape[1:]))
        a *= (1.0 / fan_in) ** 0.5
        if "attn_layer" in name:
            a *= 4.0  # make the gates depart from sigmoid(0)
    else:
        a *= 0.1
        if "norm" in name and name.endswith("weight"):
            a += 1.0
    return torch.from_numpy(a).to(dtype)


def det_state_dict(shapes, dtype=torch.float32):
    """shapes: mapping name -> shape (e.g. from module.state_dict())."""
    return {k: det_tensor(k, tuple(v), dtype) for k, v in shapes.items()}


def det_input(name, shape, kind="randn", dtype=torch.float32):
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(("input:" + name).encode("utf-8"))))
    if kind == "rand":
        a = rng.random(size=tuple(shape))
    elif kind == "randint2":
        a = rng.integers(0, 2, size=tuple(shape)).astype(np.float64)
    else:
        a = rng.standard_normal(size=tuple(shape))
    return torch.from_numpy(np.asarray(a, dtype=np.float64)).to(dtype)

"""Deterministic synthetic parameters / inputs shared by the fixture generator and the parity tests.

A fixture stores only ``{key: shape}`` and a seed; both sides (reference in the build container,
oracle and HIP product in the tests) regenerate bit-identical CPU tensors from them, so multi-MB
state dicts never have to be committed.  Values are drawn per key from a generator seeded by
``crc32(key) ^ seed`` (order independent) and scaled by a rule on the key's suffix so activations stay
O(1) and *every* parameter is non-trivial (e.g. MSDeformAttn's sampling-offset / attention-weight
Linears, which the reference zero-initialises, get real weights here).
"""
import zlib

import torch


def synth_tensor(key: str, shape, seed: int = 0, dtype=torch.float32):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "running_var":
        x = x.abs() * 0.2 + 0.8
    elif leaf == "running_mean":
        x = x * 0.1
    elif leaf == "relative_position_index":
        raise ValueError("integer buffers are not synthesised")
    elif leaf == "bias" or leaf == "in_proj_bias":
        x = x * 0.05
    elif leaf == "weight" and len(shape) == 1:          # norm scales (LN / GN / FrozenBN)
        x = 1.0 + 0.1 * x
    elif len(shape) >= 2:                               # linear / conv weights, embeddings, tables
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        x = x * (1.0 / max(fan_in, 1)) ** 0.5
        if "sampling_offsets" in key:
            x = x * 2.0                                  # spread the sampling points over a few pixels
    else:
        x = x * 0.1
    return x.to(dtype)


def synth_state_dict(shapes: dict, seed: int = 0, dtype=torch.float32):
    return {k: synth_tensor(k, s, seed, dtype) for k, s in shapes.items()}


def shapes_of(module_or_state):
    sd = module_or_state.state_dict() if hasattr(module_or_state, "state_dict") else module_or_state
    return {k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}


def rand(key: str, shape, seed: int = 0, scale: float = 1.0, uniform: bool = False, dtype=torch.float32):
    """Seeded input tensor (same keyed-generator scheme as the parameters)."""
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(("in:" + key).encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    shape = tuple(int(s) for s in shape)
    x = torch.rand(shape, generator=g) if uniform else torch.randn(shape, generator=g)
    return (x * scale).to(dtype)


from ocpg_amd.util.synthetic import synthetic_targets  # noqa: E402,F401  (single source: the package)

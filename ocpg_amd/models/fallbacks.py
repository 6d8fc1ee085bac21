"""Accounting of every place where the product leaves its hand-written HIP kernels for a library path ON A GPU.

The HIP kernels serve the shapes of the BASELINE configurations (head_dim 32, <= 32 text keys, ...).  Other shapes run a
library kernel (torch SDPA): correct, but not the native path -- so it must never happen silently.  `note()` counts the
event, warns once per (site, reason) and raises under OCPG_STRICT_HIP=1 (the -m gpu parity tests of the head_dim-32 fixtures
and bench.py's config-#2 run set it / report the counts).  CPU tensors (host-logic unit tests) are not counted.
"""
import os
import warnings

COUNTS = {}


def strict():
    return os.environ.get("OCPG_STRICT_HIP") == "1"


def note(site, reason, tensor):
    if not tensor.is_cuda:
        return
    key = f"{site}: {reason}"
    first = key not in COUNTS
    COUNTS[key] = COUNTS.get(key, 0) + 1
    if strict():
        raise RuntimeError(f"OCPG_STRICT_HIP=1: {key} (library fallback instead of the HIP kernel)")
    if first:
        warnings.warn(f"ocpg_amd: {key} -- running the library path, not the HIP kernel", RuntimeWarning, stacklevel=3)


def snapshot():
    return dict(COUNTS)


def reset():
    COUNTS.clear()

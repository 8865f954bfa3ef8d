"""Fixture (de)serialisation shared by the generator and the tests.

A fixture is one ``.npz`` file of plain arrays (loaded with ``allow_pickle=False``).
bf16 / fp16 tensors are stored as their raw 16-bit patterns under a suffixed key
(``name__bf16`` / ``name__f16``) so no precision is lost; scalars/config live in a
JSON string under ``__meta__``.
"""
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def to_arrays(tensors):
    out = {}
    for k, v in tensors.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().contiguous()
            if v.dtype == torch.bfloat16:
                out[k + "__bf16"] = v.view(torch.int16).numpy().view(np.uint16)
            elif v.dtype == torch.float16:
                out[k + "__f16"] = v.view(torch.int16).numpy().view(np.uint16)
            elif v.dtype == torch.bool:
                out[k + "__bool"] = v.to(torch.uint8).numpy()
            else:
                out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    return out


def save(name, tensors, meta=None):
    arrs = to_arrays(tensors)
    arrs["__meta__"] = np.frombuffer(json.dumps(meta or {}).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)


def load(name):
    """Return (dict of torch tensors, meta dict)."""
    z = np.load(os.path.join(HERE, name + ".npz"), allow_pickle=False)
    out, meta = {}, {}
    for k in z.files:
        a = z[k]
        if k == "__meta__":
            meta = json.loads(a.tobytes().decode())
        elif k.endswith("__bf16"):
            out[k[:-6]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        elif k.endswith("__f16"):
            out[k[:-5]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.float16)
        elif k.endswith("__bool"):
            out[k[:-6]] = torch.from_numpy(a.copy()).to(torch.bool)
        else:
            out[k] = torch.from_numpy(a.copy())
    return out, meta


def exists(name):
    return os.path.exists(os.path.join(HERE, name + ".npz"))

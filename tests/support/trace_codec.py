"""TEST-ONLY: (de)serialisation of recorded engine calls (oracle/make_driver_trace.py -> tests/golden/G28_driver_trace.npz):
nested lists / tuples / dicts of scalars and arrays become a JSON-able index whose arrays live under generated keys of one npz."""
import numpy as np


def encode(obj, arrs):
    if obj is None or isinstance(obj, (bool, str)):
        return {"t": "v", "v": obj}
    if isinstance(obj, (int, np.integer)):
        return {"t": "i", "v": int(obj)}
    if isinstance(obj, (float, np.floating)):
        key = "a%05d" % len(arrs)
        arrs[key] = np.array(float(obj))
        return {"t": "f", "k": key}
    if isinstance(obj, np.ndarray):
        key = "a%05d" % len(arrs)
        arrs[key] = obj
        return {"t": "a", "k": key}
    if isinstance(obj, (list, tuple)):
        return {"t": "l" if isinstance(obj, list) else "u", "v": [encode(o, arrs) for o in obj]}
    if isinstance(obj, dict):
        return {"t": "d", "v": {str(k): encode(v, arrs) for k, v in obj.items()}}
    raise TypeError("cannot record %r" % type(obj))


def decode(node, arrs):
    t = node["t"]
    if t in ("v", "i"):
        return node["v"]
    if t == "f":
        return float(arrs[node["k"]])
    if t == "a":
        return arrs[node["k"]]
    if t in ("l", "u"):
        out = [decode(o, arrs) for o in node["v"]]
        return out if t == "l" else tuple(out)
    if t == "d":
        return {k: decode(v, arrs) for k, v in node["v"].items()}
    raise ValueError(t)

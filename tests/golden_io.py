"""Flatten / restore nested scene dicts to flat npz keys ("0/graph/pre/3/u") for the golden fixtures."""
import numpy as np
import torch


def flatten(obj, prefix="", out=None):
    out = {} if out is None else out
    if isinstance(obj, dict):
        for k, v in obj.items():
            flatten(v, "%s%s/" % (prefix, k), out)
    elif isinstance(obj, (list, tuple)):
        for i, v in enumerate(obj):
            flatten(v, "%s%d/" % (prefix, i), out)
    else:
        if torch.is_tensor(obj):
            obj = obj.detach().cpu().numpy()
        out[prefix[:-1]] = np.asarray(obj)
    return out


def unflatten(flat, prefix):
    """Inverse of flatten for keys under `prefix/`; integer-named levels become lists."""
    tree = {}
    for key, val in flat.items():
        if not key.startswith(prefix + "/"):
            continue
        node = tree
        parts = key[len(prefix) + 1:].split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val

    def fix(n):
        if not isinstance(n, dict):
            return n
        n = {k: fix(v) for k, v in n.items()}
        if n and all(k.isdigit() for k in n):
            return [n[str(i)] for i in range(len(n))]
        return n

    return fix(tree)


def load_scenes(flat, prefix="scenes"):
    """Scene list as numpy trees; `num_nodes` back to int."""
    scenes = unflatten(flat, prefix)
    for s in scenes:
        s["graph"]["num_nodes"] = int(s["graph"]["num_nodes"])
    return scenes

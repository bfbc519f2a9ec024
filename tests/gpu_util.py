"""Helpers shared by the -m gpu parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def report(name, **kv):
    """Append a json line to gpurun_out/parity.jsonl so one GPU call leaves a full record."""
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity.jsonl"), "a") as f:
            kv["name"] = name
            f.write(json.dumps({k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kv.items()}) + "\n")
    except Exception:
        pass

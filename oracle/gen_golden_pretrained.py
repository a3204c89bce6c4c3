"""Fixture from the reference's own pretrained brains (build container only; data in, data out -- no reference source).

The real reference's brain classes are constructed with `load_model=<pretrained/All/.../brain_gene_N.pt>`
(ReinLife/Models/DQN.py:60-63, PERD3QN.py:72-79, PPO.py:49-52) and asked for their forward outputs on observation rows of
tests/golden/models.npz.  Recorded per brain: the state dict (key names, shapes, float32 values -- the `.pt` file's content as
plain arrays), the scalar `parameters_*.json` the reference's Saver wrote next to it, and the reference's outputs.
tests/test_hip_saver_gpu.py rebuilds a `.pt` from the arrays, loads it through the product's `load_model=` and must
reproduce the outputs through the HIP forward (1e-5).

    python oracle/gen_golden_pretrained.py     ->  tests/golden/pretrained.npz
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness as rh  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pretrained.npz")
PRE = os.path.join(rh.REFERENCE_ROOT, "pretrained", "All")
CASES = {"DQN": ("DQN", "brain_gene_0"), "PERD3QN": ("PERD3QN", "brain_gene_3"), "D3QN": ("D3QN", "brain_gene_1"), "PPO": ("PPO", "brain_gene_4")}
N_ROWS = 96


def main():
    ref = rh.load_reference()
    torch = ref.torch
    torch.set_num_threads(1)
    obs = np.load(os.path.join(os.path.dirname(OUT), "models.npz"))["obs"][:N_ROWS].astype(np.float32)
    out = {"obs": obs}
    meta = {}
    for kind, (folder, stem) in CASES.items():
        path = os.path.join(PRE, folder, stem + ".pt")
        if kind == "DQN":
            b = ref.DQN(load_model=path, training=False)
            net = b.agent
            fwd = lambda x: net(x)  # noqa: E731
        elif kind in ("D3QN", "PERD3QN"):
            b = getattr(ref, kind)(load_model=path, training=False)
            net = b.eval_net
            fwd = lambda x: torch.cat([net(r[None]) for r in x])  # noqa: E731  (advantage.mean() is a whole-tensor mean: batch 1)
        else:
            b = ref.PPO(load_model=path)
            net = b.model
            fwd = lambda x: net.pi(x, softmax_dim=1)  # noqa: E731
        sd = net.state_dict()
        with torch.no_grad():
            y = fwd(torch.from_numpy(obs)).numpy().astype(np.float32)
        out[kind + "_out"] = y
        out[kind + "_weights"] = np.concatenate([v.numpy().astype(np.float32).reshape(-1) for v in sd.values()])
        meta[kind] = {"keys": [[k, list(v.shape)] for k, v in sd.items()], "file": "pretrained/All/%s/%s.pt" % (folder, stem),
                      # scalar attributes only (the json also holds the class docstring: source text, not data -- dropped)
                      "parameters": {k: v for k, v in json.load(open(os.path.join(PRE, folder, stem.replace("brain", "parameters") + ".json"))).items()
                                     if not k.startswith("__")}}
        print(kind, y.shape, out[kind + "_weights"].shape, float(np.abs(y).max()))
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    main()

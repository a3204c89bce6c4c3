"""GPU: SURVEY 8f-3 end to end -- Saver / load_model wire format through the HIP forward (saver.py:58-97, entities.py:224-242,
Models/*.py load_model branches)."""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

import golden_io

pytestmark = pytest.mark.gpu
NET_ATTR = {"DQN": "agent", "D3QN": "eval_net", "PERD3QN": "eval_net", "PPO": "model"}


def _set_flat(net, flat):
    import torch
    sd, o = OrderedDict(), 0
    for k, v in net.state_dict().items():
        n = v.numel()
        sd[k] = torch.from_numpy(np.asarray(flat[o:o + n], np.float32).reshape(tuple(v.shape)).copy())
        o += n
    assert o == len(flat)
    net.load_state_dict(sd)


def test_saved_brains_reload_through_load_model_and_reproduce_the_reference_outputs(tmp_path, monkeypatch):
    """Brains carrying the golden weights -> Saver.save (reference layout) -> fresh brains built with load_model=<that file> ->
    rl_policy_pack_weights -> rl_policy_forward == the reference's torch outputs recorded in models.npz (1e-5), for all four
    kinds; and the same through Environment.act()'s batched path for the greedy kinds."""
    import torch
    from reinlife_amd import Models
    from reinlife_amd.Helpers.saver import SavedAgent, Saver
    monkeypatch.chdir(tmp_path)
    m = np.load(golden_io.GOLDEN_DIR + "/models.npz")
    ctor = {"DQN": lambda **k: Models.DQN(training=False, **k), "D3QN": lambda **k: Models.D3QN(training=False, **k),
            "PERD3QN": lambda **k: Models.PERD3QN(training=False, **k), "PPO": lambda **k: Models.PPO(**k)}
    brains = []
    for name in ("DQN", "D3QN", "PERD3QN", "PPO"):
        b = ctor[name]()
        _set_flat(getattr(b, NET_ATTR[name]), m[name + "_weights"])
        brains.append(b)
    exp = Saver("experiments").save([SavedAgent(g, b) for g, b in enumerate(brains)], True, {"Avg Number of Populations": []}, {"Width": 30})
    for g, name in enumerate(("DQN", "D3QN", "PERD3QN", "PPO")):
        path = os.path.join(exp, name, "brain_gene_%d.pt" % g)
        fresh = ctor[name](load_model=path)
        assert np.array_equal(fresh.state_dict_flat(), m[name + "_weights"])
        out = fresh.forward_batch(m["obs"]).cpu().numpy()
        np.testing.assert_allclose(out, m[name + "_out"], rtol=0, atol=1e-5, err_msg=name)
        if name != "PPO":   # greedy get_action on single states == the reference's recorded greedy action where its top two are apart
            srt = np.sort(m[name + "_out"], axis=1)
            for r in np.nonzero(srt[:, -1] - srt[:, -2] > 1e-4)[0][:25]:
                assert fresh.get_action(m["obs"][r], 0) == int(m[name + "_greedy"][r]), (name, r)
        torch.cuda.synchronize()


def test_reference_pretrained_brains_load_and_forward_through_hip(tmp_path):
    """tests/golden/pretrained.npz = the reference's own pretrained/All/* brains as arrays + the outputs the REAL reference gave
    after load_model= (oracle/gen_golden_pretrained.py).  A .pt rebuilt from the arrays must load through the product's
    load_model= (same key names / shapes) and give those outputs through the HIP forward: 1e-5 of the output scale (trained Q
    values reach ~30, where one float32 ulp is already 4e-6)."""
    import torch
    from reinlife_amd import Models
    p = np.load(golden_io.GOLDEN_DIR + "/pretrained.npz")
    meta = json.loads(bytes(p["meta"]).decode())
    ctor = {"DQN": lambda f: Models.DQN(load_model=f, training=False), "D3QN": lambda f: Models.D3QN(load_model=f, training=False),
            "PERD3QN": lambda f: Models.PERD3QN(load_model=f, training=False), "PPO": lambda f: Models.PPO(load_model=f)}
    for name in ("DQN", "D3QN", "PERD3QN", "PPO"):
        sd, o = OrderedDict(), 0
        for key, shape in meta[name]["keys"]:
            n = int(np.prod(shape))
            sd[key] = torch.from_numpy(p[name + "_weights"][o:o + n].reshape(shape).copy())
            o += n
        f = str(tmp_path / (name + ".pt"))
        torch.save(sd, f)
        b = ctor[name](f)
        want = p[name + "_out"]
        scale = max(1.0, float(np.abs(want).max()))
        out = b.forward_batch(p["obs"]).cpu().numpy()
        np.testing.assert_allclose(out / scale, want / scale, rtol=0, atol=1e-5, err_msg=name)
        # the scalar attributes the reference's Saver wrote next to the brain are the ones this build's brains expose
        for k in meta[name]["parameters"]:
            if k == "one":   # a constant of the reference's BasicBrain (Models/utils.py:7): not an attribute here, the Saver writes it
                continue
            assert hasattr(b, k), (name, k)

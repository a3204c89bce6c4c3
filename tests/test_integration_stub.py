"""The reference-side ctypes stub printed in INTEGRATION.md section 2 is EXECUTED here (extracted from the document), so
the document cannot drift from include/reinlife_hip.h: struct layouts (all 13 rl_step_out pointers), argument orders and
one rl_step / rl_update through it, compared with the oracle."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):]
    m = re.search(r"```python\n(# ReinLife/World/hip_backend\.py.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section 2 lost its stub"
    return m.group(1)


def test_stub_structs_match_the_header():
    """CPU: field names and order of every struct in the stub == include/reinlife_hip.h (parsed), without loading the library."""
    src = _stub_source()
    hdr = open(os.path.join(ROOT, "include", "reinlife_hip.h")).read()

    def header_fields(struct):
        end = hdr.index("} %s;" % struct)
        body = hdr[hdr.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                names.append(re.findall(r"[A-Za-z_][A-Za-z_0-9]*", part)[-1])
        return names

    ns = {}
    code = src.replace('lib = C.CDLL(os.environ.get("REINLIFE_HIP_LIB", "libreinlife_hip.so"))', "lib = None").replace(
        "lib.rl_last_error.restype = C.c_char_p", "")
    exec(compile(code, "INTEGRATION.md#stub", "exec"), ns)
    assert [n for n, _ in ns["Config"]._fields_] == header_fields("rl_config")
    assert [n for n, _ in ns["State"]._fields_] == header_fields("rl_state")
    assert [n for n, _ in ns["StepOut"]._fields_] == header_fields("rl_step_out") and len(ns["StepOut"]._fields_) == 13
    assert [n for n, _ in ns["UpdateOut"]._fields_] == header_fields("rl_update_out")


@pytest.mark.gpu
def test_stub_runs_step_and_update_on_the_gpu(monkeypatch):
    from oracle import oracle as orc
    from reinlife_amd import build
    monkeypatch.setenv("REINLIFE_HIP_LIB", build.LIB_PATH)
    ns = {}
    exec(compile(_stub_source(), "INTEGRATION.md#stub", "exec"), ns)
    ow = orc.OracleWorlds(n_worlds=1, n_brains=3, seed=12)
    ow.reset_synthetic(100)
    snap = ow.world(0)
    hw = ns["HipWorld"](30, 30, 100, 3, seed=12)
    hw.load(snap["cell_type"], {k: snap[k] for k in ("i", "j", "health", "age", "max_age", "gene", "brain", "uid", "flags", "action", "fitness")})
    n = hw.n_agents()
    assert np.array_equal(hw.state[:n].cpu().numpy(), ow.observe()[0, :n])
    rng = np.random.RandomState(0)
    for t in range(20):
        acts = np.zeros((1, ow.cap), np.int8)
        acts[0, :n] = rng.randint(0, 8, size=n)
        hw.step(acts[0, :n]); ow.step(acts)
        n1 = hw.n_agents()
        assert n1 == int(ow.s["n_agents"][0])
        assert np.array_equal(hw.reward[0, :n1].cpu().numpy(), ow.reward[0, :n1])
        assert np.array_equal(hw.done[0, :n1].cpu().numpy(), ow.done[0, :n1])
        assert np.array_equal(hw.src[0, :n1].cpu().numpy(), ow.src1[0, :n1])
        assert np.array_equal(hw.state_prime[:n1].cpu().numpy(), ow.obs1[0, :n1])
        hw.update_env(); ow.update()
        n = hw.n_agents()
        assert n == int(ow.s["n_agents"][0])
        assert np.array_equal(hw.s["cell_type"][0].cpu().numpy(), ow.s["cell_type"][0])
        assert np.array_equal(hw.s["a_health"][0, :n].cpu().numpy(), ow.s["a_health"][0, :n])
        assert np.array_equal(hw.state[:n].cpu().numpy(), ow.obs2[0, :n])

"""A short run of the differential fuzzer (tools/fuzz.py) inside the GPU suite: random small
forward / backward / decode problems and LFA-shaped forwards against the oracle.  Longer
campaigns are run by hand."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz():
    spec = importlib.util.spec_from_file_location("ganet_fuzz", os.path.join(ROOT, "tools", "fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [101, 102])
def test_short_fuzz_campaign(seed, capsys):
    fz = _fuzz()
    rng = np.random.default_rng(seed)
    bad = fz.fuzz_forward(rng, 150) + fz.fuzz_backward(rng, 60) + fz.fuzz_decode(rng, 150) + \
        fz.fuzz_lfa(rng, 12)
    out = capsys.readouterr().out
    assert bad == 0, out

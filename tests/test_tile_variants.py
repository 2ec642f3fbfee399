"""The tuning knobs that select alternative tile shapes must all give reference-parity results (GPU; each setting runs in
its own process because the library reads the environment once)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SNIPPET = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r)
from onepose_amd import GATsSuperGlue, synthetic
from oracle import gatsspg_oracle as orc
hp = dict(orc.DEFAULT_HPARAMS, match_threshold=0.0)
sd = synthetic.make_state_dict(0)
data = synthetic.make_inputs(b=2, n1=150, n2=300, num_leaf=8, seed=11)
m = GATsSuperGlue(hp).eval(); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.cuda()
conf, m0, m1, s0, s1 = m.forward_batched({k: torch.from_numpy(v).cuda() for k, v in data.items()})
_, ref, inter = orc.forward(sd, data, hp, return_intermediates=True)
err = float(np.abs(conf.cpu().numpy() - ref).max())
ok = err < 1e-4 and (m0.cpu().numpy() == inter["batched"]["matches0"]).all() and (m1.cpu().numpy() == inter["batched"]["matches1"]).all()
print("PARITY", ok, err)
sys.exit(0 if ok else 1)
""" % ROOT


@pytest.mark.parametrize("env", [{"GATSSPG_MLP0_TILE": "0"}, {"GATSSPG_MLP0_TILE": "1"}, {"GATSSPG_MLP0_TILE": "2"},
                                 {"GATSSPG_MLP0_TILE": "3"}, {"GATSSPG_MLP0_TILE": "4"}, {"GATSSPG_MLP3_TILE": "1"}, {"GATSSPG_MLP3_TILE": "2"},
                                 {"GATSSPG_GATS_TILE": "8"}, {"GATSSPG_LDS_SHAPING": "1"}, {"GATSSPG_VSTORE": "0"},
                                 {"GATSSPG_QKV_TILE": "0"}, {"GATSSPG_MLP3_TILE": "3"}, {"GATSSPG_MLP3_TILE": "4"}, {"GATSSPG_SCORE_TILE": "0"},
                                 {"GATSSPG_PREC": "bf16x3"}, {"GATSSPG_MLP3_PREC": "bf16x3"}])
def test_tile_variant_parity(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", SNIPPET], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, f"{env}: {r.stdout[-500:]} {r.stderr[-800:]}"
    assert "PARITY True" in r.stdout

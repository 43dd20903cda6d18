"""dynamics.pth compatibility, checked in the BUILD container against the live reference: the file is
13 MB (3.27 M fp32 parameters incl. the saved_* copies), too large to commit as a fixture, so this test writes one with
the reference's own `MOBODYEnsembleDynamics.save` (mobody_dynamics.py:1158-1161) into tmp_path and

  (1) loads it through the mirror's `.load()` (torch.load(weights_only=True)) -> every tensor of the mirror's
      state_dict equals the reference's, key for key (saved_weight/saved_bias, za_de_*, max/min_logvar(_latent), elites);
  (2) has the mirror `.save()` it again and loads THAT with the reference's own `.load()` -> the reference's
      forward_trg on the reloaded model reproduces the committed G2 means.

Skipped where /root/reference does not exist (the GPU box); needs no GPU."""
import os
import sys
import types

import numpy as np
import pytest
import torch

import golden_util as gu

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")


def _reference():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    stub = types.ModuleType("algo.mb_utils.logger")
    stub.Logger = object
    # make sure `algo` resolves to the reference here, whatever was imported before
    saved = {k: v for k, v in sys.modules.items() if k == "algo" or k.startswith("algo.")}
    for k in saved:
        del sys.modules[k]
    sys.modules["algo.mb_utils.logger"] = stub
    from algo.dynamics.mobody_module import MOBODYModule
    from algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from algo.mb_utils.terminal_funs import get_termination_fn
    return MOBODYModule, MOBODYEnsembleDynamics, get_termination_fn, saved


def test_dynamics_checkpoint_round_trip_with_the_reference(tmp_path, monkeypatch):
    g = gu.load("g234_dynamics_walker")
    S, A = int(g["S"]), int(g["A"])
    p = gu.dyn_params_for(g)
    RefModule, RefDyn, ref_term, saved = _reference()
    try:
        cfg = dict(mopo=0, latent_reward=0, encoder_loss_coef=1, domain_loss_coef=0.0, cycle_loss_coef=0.3)
        rm = RefModule(S, A, 256, 7, 5, device="cpu", config=dict(cfg))
        sd = rm.state_dict()
        for k, v in p.items():
            sd[k] = torch.from_numpy(v)
        rm.load_state_dict(sd)
        rm.set_elites([6, 2, 3, 5, 4])
        rd = RefDyn(dict(cfg), rm, None, None, ref_term("walker2d-medium-v2"), penalty_coef=0.1)
        d1 = tmp_path / "ref_written"; d1.mkdir()
        rd.save(str(d1))
        assert sorted(os.listdir(d1)) == ["dynamics.pth", "mu.npy", "std.npy"]

        from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
        from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
        from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn
        mcfg = gu.policy_cfg(S, A)
        mm = MOBODYModule(S, A, 256, 7, 5, device="cpu", config=mcfg)
        md = MOBODYEnsembleDynamics(mcfg, mm, None, None, get_termination_fn("walker2d-medium-v2"), penalty_coef=0.1)
        md.load(str(d1))                                                       # (1) reference-written -> mirror
        want, got = rm.state_dict(), mm.state_dict()
        assert sorted(want) == sorted(got)
        for k in want:
            assert got[k].dtype == want[k].dtype and torch.equal(got[k].cpu(), want[k]), k
        assert [int(x) for x in mm.elites.tolist()] == [6, 2, 3, 5, 4]

        d2 = tmp_path / "mirror_written"; d2.mkdir()
        md.save(str(d2))                                                       # (2) mirror-written -> reference
        rm2 = RefModule(S, A, 256, 7, 5, device="cpu", config=dict(cfg))
        rd2 = RefDyn(dict(cfg), rm2, None, None, ref_term("walker2d-medium-v2"), penalty_coef=0.1)
        o_to = torch.Tensor.to
        monkeypatch.setattr(torch.Tensor, "to", lambda t, *a, **k: o_to(t, *[("cpu" if x == "cuda" else x) for x in a], **k))
        rd2.load(str(d2))                                                      # load_scaler hard-codes .to('cuda') (:152-153)
        monkeypatch.undo()
        rm2.inference()
        with torch.no_grad():
            mean, _, _ = rm2.forward_trg(torch.from_numpy(g["obs"]), torch.from_numpy(g["act"]))
        np.testing.assert_array_equal(mean.numpy(), g["mean_trg"])
    finally:
        for k in [k for k in sys.modules if k == "algo" or k.startswith("algo.")]:
            del sys.modules[k]
        sys.modules.update(saved)

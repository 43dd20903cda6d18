"""Input / output side of the path: target-data ingestion and the normalised score, against
values the reference itself produced (fixtures g14 via make_golden.g14, g15 via tools/extract_ref_scores.py)."""
import json
import os

import numpy as np
import pytest

import golden_util as gu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_transitions_from_arrays_vs_reference(tag):
    from mobody_amd.dataset.call_dataset import transitions_from_arrays
    g = gu.load("g14_ingest")
    ds = {k[len(tag) + 4:]: v for k, v in g.items() if k.startswith(tag + "_in_")}
    out = transitions_from_arrays(ds)
    for k, v in out.items():
        want = g[f"{tag}_out_{k}"]
        assert v.shape == want.shape and v.dtype == want.dtype, k
        assert np.array_equal(v, want), k


def test_dataset_path_and_missing_h5py():
    from mobody_amd.dataset import call_dataset as cd
    assert cd.dataset_path("walker2d-friction", 2.0, "medium", root="/d") == "/d/mujoco/walker2d_friction_2.0_medium.hdf5"
    assert cd.dataset_path("antmaze-small-empty", "easy", root="/d") == "/d/antmaze/antmaze_small_empty_easy.hdf5"
    with pytest.raises(NotImplementedError):
        cd.domain_of("reacher_x")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            cd.call_tar_dataset("walker2d-friction", 2.0, "medium")


def test_normalized_score_vs_reference():
    from mobody_amd.envs.infos import get_normalized_score
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "g15_scores.json"))):
        assert get_normalized_score(c["score"], c["env"]) == c["normalized"]
    with pytest.raises(KeyError):
        get_normalized_score(1.0, "no-such-task")

"""Normalised score -- mirror of the reference's `envs/infos.py:253-255`.

`REF_MIN_SCORE` / `REF_MAX_SCORE` are the benchmark's reference returns per task (data, kept in `ref_scores.json`,
extracted from the reference's tables by tools/extract_ref_scores.py); an unknown task raises KeyError like the reference."""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_scores.json")) as _f:
    _T = json.load(_f)
REF_MIN_SCORE, REF_MAX_SCORE = _T["min"], _T["max"]


def get_normalized_score(score, env_name):
    ref_min_score = REF_MIN_SCORE[env_name]
    ref_max_score = REF_MAX_SCORE[env_name]
    return (score - ref_min_score) / (ref_max_score - ref_min_score) * 100

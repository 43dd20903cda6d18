"""Benchmark reference scores as DATA: the two constant tables of the reference's envs/infos.py (random / expert returns
per task, the D4RL-style normalisation constants) dumped to JSON for the product (`mobody_amd/envs/ref_scores.json`),
plus a few evaluations of the reference's own get_normalized_score as a test fixture (tests/golden/g15_scores.json).
Run in the build container only:  python tools/extract_ref_scores.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
from envs import infos  # noqa: E402  (pure-Python constants, no third-party imports)

pkg = os.path.join(ROOT, "mobody-model-based-off-dynamics-offline-reinforcement-learning_amd", "envs", "ref_scores.json")
json.dump(dict(min=infos.REF_MIN_SCORE, max=infos.REF_MAX_SCORE), open(pkg, "w"), indent=0, sort_keys=True)
cases = []
names = sorted(infos.REF_MAX_SCORE)
for i, name in enumerate(names[::7]):
    for score in (0.0, 1234.5 - 100.0 * i, -3.25):
        cases.append(dict(env=name, score=score, normalized=infos.get_normalized_score(score, name)))
json.dump(cases, open(os.path.join(ROOT, "tests", "golden", "g15_scores.json"), "w"), indent=0)
print(len(names), "tasks,", len(cases), "fixture cases")

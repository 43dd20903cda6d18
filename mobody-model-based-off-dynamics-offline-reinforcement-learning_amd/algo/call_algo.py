"""Policy plugin registry -- host-side mirror of `algo/call_algo.py:3-27`.

`call_algo(algo_name, config, mode, device, **kwargs)` lower-cases the name, looks the class up
and returns `cls(config, device)`; `mode` and kwargs are accepted and ignored exactly as the
reference does.  Unknown names raise KeyError like the reference's dict lookup.  The baseline
algorithms (dara, bosa, iql, td3_bc, igdf) are outside the accelerated path.
"""
from .offline_offline.mobody import MOBODY

_BASELINES = ("dara", "bosa", "iql", "td3_bc", "igdf")


def call_algo(algo_name, config, mode, device, **kwargs):
    algo_name = algo_name.lower()
    algo_to_call = {"mobody": MOBODY}
    if algo_name in _BASELINES:
        raise NotImplementedError(f"'{algo_name}' is a reference baseline outside the MI355X-accelerated path; "
                                  "only 'mobody' is built here")
    algo = algo_to_call[algo_name]          # KeyError for unknown names, as in the reference
    return algo(config, device)

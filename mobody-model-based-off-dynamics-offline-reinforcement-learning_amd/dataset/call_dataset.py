"""Target-domain data ingestion -- mirror of the reference's `dataset/call_dataset.py:21-109`.

`transitions_from_arrays` is the array-level transformation `call_tar_dataset` applies to an ODRL HDF5 file: row i
becomes the transition (observations[i], actions[i], rewards[i], observations[i+1], terminals[i]) for i < N-1.  The
reference computes episode ends (the `timeouts` field, or a step counter against `_max_episode_steps`) but never skips a
row on them, so transitions that straddle an episode boundary are kept; that behaviour is reproduced (the counters have
no effect on the outputs and are omitted).  Rewards stored as [N, 1] are flattened (`rewards[i][0]`, :88-91).
The result feeds `ReplayBuffer.convert_D4RL` (train_mobody.py:553-557), after which the data lives in HBM.

`call_tar_dataset(tar_env_name, shift_scale, quality)` keeps the reference's signature and file naming
(`<dir>/<domain>/<env>_<shift>_<quality>.hdf5`); it needs `h5py`, which this image lacks -> ImportError with that
message rather than a silent fallback."""
import os

import numpy as np


def transitions_from_arrays(dataset):
    obs = np.asarray(dataset["observations"])
    n = np.asarray(dataset["rewards"]).shape[0]
    rew = np.asarray(dataset["rewards"])
    if rew.ndim > 1:
        rew = rew.reshape(n, -1)[:, 0]
    return {
        "observations": obs[:n - 1].astype(np.float32),
        "actions": np.asarray(dataset["actions"])[:n - 1].astype(np.float32),
        "next_observations": obs[1:n].astype(np.float32),
        "rewards": rew[:n - 1].astype(np.float32),
        "terminals": np.asarray(dataset["terminals"])[:n - 1].astype(bool),
    }


def domain_of(tar_env_name):
    """call_dataset.py:25-46."""
    if any(name in tar_env_name for name in ("halfcheetah", "hopper", "walker2d")) or tar_env_name.split("_")[0] == "ant":
        return "mujoco"
    if any(name in tar_env_name for name in ("pen", "door", "relocate", "hammer")):
        return "adroit"
    if "antmaze" in tar_env_name:
        return "antmaze"
    raise NotImplementedError


def dataset_path(tar_env_name, shift_scale, quality="random", root=None):
    """call_dataset.py:22-51: `-` -> `_`, `<root>/<domain>/<env>_<shift>[_<quality>].hdf5`."""
    tar_env_name = tar_env_name.replace("-", "_")
    domain = domain_of(tar_env_name)
    root = root or os.path.dirname(os.path.abspath(__file__))
    tail = f"{tar_env_name}_{shift_scale}.hdf5" if domain == "antmaze" else f"{tar_env_name}_{shift_scale}_{quality}.hdf5"
    return os.path.join(root, domain, tail)


def transitions_from_hdf5(path):
    """Every dataset of an ODRL HDF5 file (call_dataset.py:53-66, the get_keys walk) through transitions_from_arrays."""
    try:
        import h5py
    except ImportError as exc:
        raise ImportError("ODRL HDF5 files need h5py; feed transitions_from_arrays() with the file's arrays (e.g. from an "
                          ".npz copy) instead") from exc
    data = {}
    with h5py.File(path, "r") as f:
        def visit(name, item):
            if isinstance(item, h5py.Dataset):
                try:
                    data[name] = item[:]
                except ValueError:
                    data[name] = item[()]
        f.visititems(visit)
    return transitions_from_arrays(data)


def call_tar_dataset(tar_env_name, shift_scale, quality="random", root=None):
    return transitions_from_hdf5(dataset_path(tar_env_name, shift_scale, quality, root))

"""Synthetic MuJoCo-shaped data and randomly initialised networks: there are no datasets,
simulators or checkpoints on the GPU box, so benchmarks and the CLI's --synthetic mode use these."""
import numpy as np
import torch

SHAPES = {  # env prefix -> (state_dim, action_dim, termination task name)
    "walker2d": (17, 6, "walker2d-medium-v2"), "halfcheetah": (17, 6, "halfcheetah-medium-v2"),
    "hopper": (11, 3, "hopper-medium-v2"), "ant": (111, 8, "ant-medium-v2"), "pen": (45, 24, "pen-human-v1"),
}


def env_shape(env):
    for k, v in SHAPES.items():
        if env.split("-")[0].split("_")[0] == k:
            return v
    raise KeyError(f"no synthetic shape for env '{env}'")


def alive_mean(task, S):
    mu = np.zeros(S, np.float32)
    if "walker2d" in task:
        mu[0] = 1.25
    elif "hopper" in task:
        mu[0] = 1.25
    elif "ant" in task:
        mu[0] = 0.6
    elif "pen" in task:
        mu[26] = 0.2
    return mu


def fill_buffer(rb, rows, task, seed):
    """state,next_state ~ N(mu_env, 0.1^2), action ~ U(-1,1), reward ~ N(0,1), not_done = 1 except 0.1 % zeros."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    S, A = rb.state_dim, rb.action_dim
    mu = torch.from_numpy(alive_mean(task, S))
    ds = dict(observations=(mu + 0.1 * torch.randn(rows, S, generator=g)).numpy(),
              actions=(torch.rand(rows, A, generator=g) * 2 - 1).numpy(),
              next_observations=(mu + 0.1 * torch.randn(rows, S, generator=g)).numpy(),
              rewards=torch.randn(rows, generator=g).numpy(),
              terminals=(torch.rand(rows, generator=g) < 0.001).numpy())
    rb.convert_D4RL(ds)
    return rb


def alive_dynamics(model, task):
    """Shift the random-init transition head so imagined next states sit inside the task's alive box
    (otherwise every synthetic rollout terminates at step 1 and the multi-step path is never exercised)."""
    sd = model.state_dict()
    mu = torch.from_numpy(alive_mean(task, model.obs_dim)).to(sd["transition3.bias"].device)
    sd["transition3.bias"] = sd["transition3.bias"] + mu.view(1, 1, -1)
    model.load_state_dict(sd)
    return model

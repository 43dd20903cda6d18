"""CLI driver -- mirror of the reference's `train_mobody.py` for mode 3 (offline-offline = MOBODY).

Accepts the reference's flags with the same names, types and defaults (train_mobody.py:210-307),
merges configuration in the same precedence (yaml <- --params JSON <- CLI-derived keys, :407-416,
:470-531) and runs the same sequence: seed -> build policy via call_algo -> build buffers -> build
dynamics -> attach -> `policy.train(...)` loop (:927-928).

Simulators (gym / mujoco-py / d4rl) and datasets are not part of this build: `--synthetic 1` (the
default when gym/d4rl are not importable) fills the replay buffers with synthetic MuJoCo-shaped
transitions and random-initialised networks and skips simulator evaluation; `--synthetic 0` trains on real
offline datasets without the simulators: `--src_data` (an .npz in d4rl.qlearning_dataset's layout) and `--tar_data` (the
ODRL target file: .npz of its raw arrays, or the .hdf5 itself when h5py is importable; default: the reference's
`dataset/<domain>/<env>_<shift>_<quality>.hdf5` path), ingested exactly as train_mobody.py:548-557 does.  The ensemble
dynamics follows the reference's load-or-train branches (`build_dynamics`, train_mobody.py:817-877): load
`--dynamics_path` / the default `pretrained_dynamics/<env>/srcdatatype-...` tree when present and `--train_dynamics 0`,
otherwise `MOBODYEnsembleDynamics.train` on the two buffers, then save in the reference's directory scheme.  A random
"alive" ensemble is kept only under an explicit `--synthetic 1` with nothing to load.
"""
import argparse
import json
import os
import random
import time

import numpy as np
import torch
import yaml

# (flag, default, type or None for strings / store_true marker)
_FLAGS = [
    ("--dir", "./logs", None), ("--policy", "SAC", None), ("--env", "halfcheetah-friction", None),
    ("--srctype", "medium", None), ("--tartype", "medium", None), ("--shift_level", 0.1, None),
    ("--mode", 3, int), ("--seed", 0, int), ("--save-model", False, "store_true"),
    ("--tar_env_interact_interval", 10, int), ("--max_step", int(1e6), int), ("--params", None, None),
    ("--purely_model_based", 0, int), ("--num_envs", 32, int), ("--eval_freq", 1000, int), ("--dara_eta", 0, float),
    ("--only_use_trg_transition", 5e4, int), ("--transition_update_freq", 2500, int),
    ("--transition_update_start", 5e4, int), ("--trg_rollout_batch_size", 5e4, int), ("--trg_rollout_length", 1, int),
    ("--src_rollout_batch_size", 5e4, int), ("--src_rollout_length", 1, int), ("--model_based_training_steps", 100, int),
    ("--fake_batch_scale", 0.5, float), ("--dynamics_lr", 1e-3, float), ("--encoder_loss_coef", 1, float),
    ("--domain_loss_coef", 0.0, float), ("--cycle_loss_coef", 0.3, float), ("--bc_coef", 1.0, float),
    ("--q_weighted", 1, int), ("--advantage", 0, int), ("--scale_q", 1, int), ("--mobile", 0, int),
    ("--relu_reward", 0, int), ("--penalize_fake", 0, int), ("--gaussian_dynamics", 0, int),
    ("--sep_reward_dynamics", 0, int), ("--vae", 0, int), ("--sas_reward", 0, int), ("--inverse", 0, int),
    ("--inverse_sep_reward_loss", 0, int), ("--latent_reward", 0, int), ("--filter_bad_rollout", 1, int),
    ("--train_together", 0, int), ("--encode_sa", 0, int), ("--vae_a", 0, int), ("--train_with_src_threshold", 1, float),
    ("--env_filter", 10, float), ("--use_src_sa_to_get_target_next_state", 1, int), ("--env_penalty_coef", 0.1, float),
    ("--lcb_penalty_coef", 0, float), ("--rollout_length", 1, int), ("--rollout_from_src", 0, int),
    ("--rollout_from_src_length", 2, int), ("--trg_ratio", 1, float), ("--src_ratio", 1, float),
    ("--dynamics_path", None, str), ("--train_dynamics", 0, int), ("--out_dir_remark", "", str),
    ("--penalty_type", "par", str), ("--penalty_coef", 0.1, float), ("--representation_noise", 0, float),
    ("--group", None, str), ("--wandb", 1, int), ("--cat", 0, int), ("--no_vae", 0, int), ("--trg_only", 0, int),
    ("--mopo", 0, int),
]


def build_parser():
    p = argparse.ArgumentParser()
    for flag, default, typ in _FLAGS:
        if typ == "store_true":
            p.add_argument(flag, action="store_true")
        elif typ is None:
            p.add_argument(flag, default=default)
        else:
            p.add_argument(flag, default=default, type=typ)
    # additions of this build (not in the reference)
    p.add_argument("--synthetic", default=None, type=int, help="1: synthetic buffers/networks (no gym/d4rl needed)")
    p.add_argument("--rng", default="numpy", choices=["numpy", "device"], help="index/elite RNG: reference NumPy stream or device Philox")
    p.add_argument("--src_rows", default=int(1e6), type=int)
    p.add_argument("--tar_rows", default=5000, type=int)
    p.add_argument("--log_every", default=1000, type=int)
    p.add_argument("--dynamics_max_epochs", default=None, type=int, help="cap on pre-training epochs (reference: until early stopping)")
    p.add_argument("--scalars", default=1, type=int, help="1: write the writer.add_scalar stream to <outdir>/tb/scalars.csv")
    p.add_argument("--src_data", default=None, help="--synthetic 0: source transitions (.npz in d4rl.qlearning_dataset layout)")
    p.add_argument("--tar_data", default=None, help="--synthetic 0: target ODRL dataset (.npz of its raw arrays, or the .hdf5 itself "
                                                     "when h5py is importable); default: the reference's dataset/<domain>/... path")
    return p


class ScalarLog:
    """The `writer` the reference hands to policy.train / dynamics.train (a tensorboard SummaryWriter,
    train_mobody.py:455-458): same add_scalar(tag, value, global_step) surface, rows appended to a CSV file (tensorboard is
    not part of this image).  Values may be device tensors; they are read when the row is written."""

    def __init__(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        self.path = path
        self.f = open(path, "w")
        self.f.write("tag,step,value\n")

    def add_scalar(self, tag, scalar_value, global_step=None, *_, **__):
        v = float(scalar_value.item() if hasattr(scalar_value, "item") else scalar_value)
        self.f.write(f"{tag},{-1 if global_step is None else int(global_step)},{v!r}\n")

    def flush(self):
        self.f.flush()

    def close(self):
        self.f.close()


def load_datasets(args):
    """The two offline datasets of the real-data mode without simulators: the source transitions the reference takes from
    d4rl.qlearning_dataset (train_mobody.py:548-552) come from an .npz file with the same keys, the target transitions from
    the ODRL file through dataset/call_dataset.py's transformation (:553-557)."""
    from mobody_amd.dataset import call_dataset
    if args.src_data is None:
        raise NotImplementedError("--synthetic 0 without simulators needs --src_data FILE.npz (observations, actions, "
                                  "next_observations, rewards, terminals: d4rl.qlearning_dataset's keys); d4rl itself is not in this image")
    with np.load(args.src_data) as z:
        src = {k: z[k] for k in ("observations", "actions", "next_observations", "rewards", "terminals")}
    if args.tar_data is not None and args.tar_data.endswith(".npz"):
        with np.load(args.tar_data) as z:
            tar = call_dataset.transitions_from_arrays({k: z[k] for k in z.files})
    elif args.tar_data is not None:
        tar = call_dataset.transitions_from_hdf5(args.tar_data)
    else:
        tar = call_dataset.call_tar_dataset(args.env, args.shift_level, args.tartype)
    return src, tar


def domain_of(env):
    """train_mobody.py:314-321."""
    if "halfcheetah" in env or "hopper" in env or "walker2d" in env or env.split("-")[0] == "ant":
        return "mujoco"
    if "pen" in env or "relocate" in env or "door" in env or "hammer" in env:
        return "adroit"
    if "antmaze" in env:
        return "antmaze"
    raise NotImplementedError


def load_yaml(domain, policy, env_base):
    """config/<domain>/<policy>/<env>.yaml next to the script (train_mobody.py:410-411); the four
    mujoco MOBODY files of the reference are identical except eval_freq, so a built-in copy of the
    hyper-parameters is used when no config tree is present."""
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, "config", domain, policy, env_base + ".yaml")
    if os.path.exists(path):
        with open(path, "r", encoding="utf-8") as f:
            return yaml.safe_load(f)
    if policy != "mobody" or domain != "mujoco":
        raise FileNotFoundError(path)             # same failure as the reference for configs it does not ship
    return dict(alpha=0.2, batch_size=128, actor_lr=0.0003, critic_lr=0.0003, gamma=0.99, state_dim=17, action_dim=3,
                hidden_sizes=256, max_action=1, gaussian_noise_std=1.0, eta=0.1, temperature_opt=False, tau=0.005,
                update_interval=2, expl_noise=0.2, eval_episode=10, eval_freq=2500, start_steps=5000, max_step=500000,
                device="cuda", save_freq=5000, lam=0.7, temp=3.0, weight=2.5)


def build_config(args, state_dim, action_dim, max_action):
    """yaml <- --params <- CLI keys (train_mobody.py:407-416, 470-531); env dims override the yaml's."""
    domain = domain_of(args.env)
    config = load_yaml(domain, args.policy.lower(), args.env.split("-")[0])
    if args.params is not None:
        config.update(json.loads(args.params))
    shift = args.shift_level
    if domain == "mujoco" and shift not in ("easy", "medium", "hard"):
        shift = float(shift)
    a = args
    config.update({
        "env_name": a.env, "state_dim": state_dim, "action_dim": action_dim, "max_action": max_action,
        "tar_env_interact_interval": int(a.tar_env_interact_interval), "max_step": int(a.max_step), "shift_level": shift,
        "dara_eta": a.dara_eta, "only_use_trg_transition": a.only_use_trg_transition,
        "transition_update_freq": a.transition_update_freq, "transition_update_start": a.transition_update_start,
        "trg_rollout_batch_size": int(a.trg_rollout_batch_size), "trg_rollout_length": int(a.trg_rollout_length),
        "src_rollout_batch_size": int(a.src_rollout_batch_size), "src_rollout_length": int(a.src_rollout_length),
        "model_based_training_steps": a.model_based_training_steps, "fake_batch_scale": a.fake_batch_scale,
        "env_penalty_coef": a.env_penalty_coef, "lcb_penalty_coef": a.lcb_penalty_coef,
        "encoder_loss_coef": a.encoder_loss_coef, "domain_loss_coef": a.domain_loss_coef,
        "cycle_loss_coef": a.cycle_loss_coef,
        "use_src_sa_to_get_target_next_state": a.use_src_sa_to_get_target_next_state,
        "penalty_type": a.penalty_type, "penalty_coef": a.penalty_coef, "rollout_length": a.rollout_length,
        "penalize_fake": a.penalize_fake, "representation_noise": a.representation_noise, "eval_freq": a.eval_freq,
        "bc_coef": a.bc_coef, "rollout_from_src": a.rollout_from_src, "rollout_from_src_length": a.rollout_from_src_length,
        "env_filter": a.env_filter, "trg_ratio": a.trg_ratio, "src_ratio": a.src_ratio, "q_weighted": a.q_weighted,
        "advantage": a.advantage, "scale_Q": a.scale_q, "inverse_sep_reward_loss": a.inverse_sep_reward_loss,
        "latent_reward": a.latent_reward, "filter_bad_rollout": a.filter_bad_rollout, "train_together": a.train_together,
        "train_with_src_threshold": a.train_with_src_threshold, "no_vae": a.no_vae, "trg_only": a.trg_only,
        "mopo": a.mopo,
    })
    config["rng"], config["seed"] = a.rng, a.seed
    return config


def dynamics_save_path(args, root):
    """`<root>/<env>/srcdatatype-<src>-tardatatype-<tar>-<shift>` (train_mobody.py:822-823, 843-844)."""
    return os.path.join(root, args.env, f"srcdatatype-{args.srctype}-tardatatype-{args.tartype}-{args.shift_level}")


def build_dynamics(args, dynamics, model, src_rb, tar_rb, writer, task, explicit_synthetic):
    """The reference's load-or-train logic for the ensemble dynamics (train_mobody.py:817-877), branch for branch:

    * `--dynamics_path P --train_dynamics 0`: load `P/<env>/srcdatatype-...` when that directory exists, otherwise train on
      the two buffers and save there (:819-840);
    * otherwise the default tree `pretrained_dynamics/<env>/srcdatatype-...` is tried: loaded when it exists and
      `--train_dynamics 0` (a failing load falls through to training, :848-864), else the model is trained
      (`dynamics.train(src_all, trg_all, writer=writer, buffer=[src, tar])`) and saved -- under `--dynamics_path` when
      one is given, under the default tree when not (:865-877).

    The one addition of this build: with an EXPLICIT `--synthetic 1`, no `--dynamics_path` and nothing in the default tree,
    `--train_dynamics 0` keeps a random-initialised ensemble whose transition head is shifted into the task's alive box (benchmarks and smoke runs on
    synthetic buffers; there is nothing to learn from them).  Returns 'loaded' / 'trained' / 'random'."""
    from mobody_amd import synthetic

    def train_and_save(save_path):
        if explicit_synthetic:
            synthetic.alive_dynamics(model, task)          # synthetic buffers only: start inside the alive box
        dynamics.optim = type("Opt", (), {"param_groups": [{"lr": args.dynamics_lr}]})()
        dynamics.train(src_rb.sample_all(), tar_rb.sample_all(), writer=writer, buffer=[src_rb, tar_rb],
                       max_epochs=args.dynamics_max_epochs)
        if args.dynamics_path is not None:                                     # :831-836, 855-860, 867-872
            save_path = dynamics_save_path(args, args.dynamics_path)
        os.makedirs(save_path, exist_ok=True)
        dynamics.save(save_path)
        print(f"----------dynamics trained and saved to {save_path}----------")
        return "trained"

    if args.dynamics_path is not None and args.train_dynamics == 0:            # :819-840
        save_path = dynamics_save_path(args, args.dynamics_path)
        if os.path.exists(save_path):
            dynamics.load(save_path)
            print("----------pretrained dynamics loaded----------")
            return "loaded"
        return train_and_save(save_path)
    save_path = dynamics_save_path(args, "pretrained_dynamics")                # :842-846
    if os.path.exists(save_path) and args.train_dynamics == 0:
        try:
            dynamics.load(save_path)
            print("----------pretrained dynamics loaded----------")
            return "loaded"
        except Exception:                                                      # the reference's bare `except:` (:851)
            return train_and_save(save_path)
    if explicit_synthetic and args.train_dynamics == 0:
        synthetic.alive_dynamics(model, task)
        print("--synthetic 1: nothing to load, random-initialised ensemble dynamics (pass --train_dynamics 1 to pre-train it)")
        return "random"
    return train_and_save(save_path)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if "_" in args.env:
        args.env = args.env.replace("_", "-")
    if args.mode != 3:
        raise NotImplementedError("only mode 3 (offline-offline, MOBODY) is built on the MI355X path")
    from mobody_amd import synthetic
    from mobody_amd.algo import utils
    from mobody_amd.algo.call_algo import call_algo
    from mobody_amd.algo.dynamics.mobody_dynamics import MOBODYEnsembleDynamics
    from mobody_amd.algo.dynamics.mobody_module import MOBODYModule
    from mobody_amd.algo.mb_utils.terminal_funs import get_termination_fn

    synthetic_mode = args.synthetic
    if synthetic_mode is None:
        try:
            import d4rl  # noqa: F401
            import gym  # noqa: F401
            synthetic_mode = 0
        except Exception:
            synthetic_mode = 1
    state_dim, action_dim, task = synthetic.env_shape(args.env)
    src_ds = tar_ds = None
    if not synthetic_mode:
        src_ds, tar_ds = load_datasets(args)
        if src_ds["observations"].shape[1] != state_dim or src_ds["actions"].shape[1] != action_dim:
            raise ValueError(f"--src_data has shapes {src_ds['observations'].shape[1]}/{src_ds['actions'].shape[1]}, "
                             f"env {args.env} expects {state_dim}/{action_dim}")
    max_action = 1.0
    torch.manual_seed(args.seed); np.random.seed(args.seed); random.seed(args.seed)
    torch.cuda.manual_seed_all(args.seed)
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    config = build_config(args, state_dim, action_dim, max_action)
    env_penalty_coef = 0.0 if args.mobile == 1 else args.env_penalty_coef
    print("-" * 60 + f"\nPolicy: {args.policy}, Env: {args.env}, Seed: {args.seed}\n" + "-" * 60)

    terminal_fn = get_termination_fn(task)
    policy = call_algo(args.policy, config, args.mode, device, terminal_fn=terminal_fn)
    src_rb = utils.ReplayBuffer(state_dim, action_dim, device, rng=args.rng, seed=args.seed + 1)
    tar_rb = utils.ReplayBuffer(state_dim, action_dim, device, rng=args.rng, seed=args.seed + 2)
    if synthetic_mode:
        synthetic.fill_buffer(src_rb, args.src_rows, task, args.seed)
        synthetic.fill_buffer(tar_rb, args.tar_rows, task, args.seed + 100)
    else:                                             # train_mobody.py:548-557: both buffers adopt their datasets
        src_rb.convert_D4RL(src_ds)
        tar_rb.convert_D4RL(tar_ds)
        print(f"datasets: {src_rb.size} source / {tar_rb.size} target transitions")

    model = MOBODYModule(obs_dim=state_dim, action_dim=action_dim, hidden_dims=256, num_ensemble=7, num_elites=5,
                         weight_decays=[2.5e-5, 5e-5, 7.5e-5, 7.5e-5, 1e-4], device=device,
                         reward_relu=args.relu_reward, config=config)
    dynamics = MOBODYEnsembleDynamics(config, model, None, None, terminal_fn, penalty_coef=env_penalty_coef,
                                      rng=args.rng, seed=args.seed + 3)
    outdir = f"{args.dir}/{args.policy}/{args.env}-srcdatatype-{args.srctype}-tardatatype-{args.tartype}-{args.shift_level}/r{args.seed}{args.out_dir_remark}"
    writer = ScalarLog(f"{outdir}/tb/scalars.csv") if args.scalars else None          # train_mobody.py:455-458
    build_dynamics(args, dynamics, model, src_rb, tar_rb, writer, task, explicit_synthetic=args.synthetic == 1)
    config.update({"dynamics": dynamics})
    policy.dynamics = dynamics

    if args.save_model:
        os.makedirs(f"{outdir}/models", exist_ok=True)
    start = time.time()
    for t in range(int(config["max_step"])):
        policy.train(src_rb, tar_rb, config["batch_size"], writer, None)
        if (t + 1) % args.log_every == 0:
            q, pi, bc = policy.losses()
            dt = time.time() - start
            print(f"step {t + 1}: q_loss {q:.4f} pi_loss {pi:.4f} bc_loss {bc:.4f}  {args.log_every / dt:.1f} grad-steps/s")
            start = time.time()
        if (t + 1) % config["eval_freq"] == 0:
            # The reference's evaluation block (train_mobody.py:928-975) rolls the policy out in the simulators; of it, what
            # does not need a simulator is kept on the same cadence: the transition model's error on TARGET transitions
            # (eval_policy_batch's obs-RMSE / reward-MSE, :100-133, here on a target-buffer batch) and the model checkpoint.
            if writer is not None:
                s_, a_, s2_, r_, _ = tar_rb.sample(min(1000, tar_rb.size))
                err = dynamics.model_error(s_, a_, s2_, r_)
                writer.add_scalar("test/model error next_obs", err["obs_mse"], global_step=t + 1)
                writer.add_scalar("test/model error reward", err["reward_mse"], global_step=t + 1)
                writer.flush()
            if args.save_model:
                policy.save(f"{outdir}/models/model")
    if writer is not None:
        writer.close()
    torch.cuda.synchronize()
    return policy


if __name__ == "__main__":
    main()

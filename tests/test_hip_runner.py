"""GPU: the training / evaluation drivers on top of the C-ABI -- OnPolicyRunner.learn with checkpoints in
rsl_rl's file layout, resume through task_registry (the path scripts/play.py takes), TorchScript export of
the trained actor against lg_ppo_act_inference, and the play loop itself."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(task, n, extra=()):
    from legged_gym_dev_amd.utils import get_args
    a = get_args(["--task", task, "--num_envs", str(n), "--headless", *extra])
    a.sim_device = a.rl_device = "cuda:0"
    return a


def test_learn_save_resume_export_play(tmp_path, monkeypatch):
    import legged_gym_dev_amd
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.rl import checkpoint as ck
    monkeypatch.setattr(legged_gym_dev_amd, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    import legged_gym_dev_amd.utils  # noqa: F401
    monkeypatch.setattr(sys.modules["legged_gym_dev_amd.utils.task_registry"], "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    args = _args("anymal_c_flat", 64, ("--max_iterations", "3"))
    env, env_cfg = task_registry.make_env(name=args.task, args=args)
    env_cfg2, train_cfg = task_registry.get_cfgs(args.task)
    train_cfg.runner.save_interval = 2
    train_cfg.policy.actor_hidden_dims = [64, 32]
    train_cfg.policy.critic_hidden_dims = [64, 32]
    log_root = os.path.join(str(tmp_path), "logs", train_cfg.runner.experiment_name)
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args, train_cfg=train_cfg, log_root=log_root)
    runner.learn(num_learning_iterations=3, init_at_random_ep_len=True)
    files = sorted(os.listdir(runner.log_dir))
    assert "model_0.pt" in files and "model_2.pt" in files and "model_3.pt" in files, files
    d = torch.load(os.path.join(runner.log_dir, "model_3.pt"), map_location="cpu", weights_only=True)
    assert set(d) == {"model_state_dict", "optimizer_state_dict", "iter", "infos"} and d["iter"] == 3
    assert list(d["model_state_dict"])[:3] == ["std", "actor.0.weight", "actor.0.bias"]
    # the reference side can consume it: ActorCritic.load_state_dict + torch Adam.load_state_dict
    from oracle import ppo_torch
    ac = ppo_torch.ActorCritic(48, 48, 12, [64, 32], [64, 32], "elu", 1.0)
    ac.load_state_dict(d["model_state_dict"])
    torch.optim.Adam(ac.parameters()).load_state_dict(d["optimizer_state_dict"])
    want_sd = {k: v.cpu().clone() for k, v in runner.ppo.state_dict().items()}
    want_opt = runner.ppo.optimizer_state_dict()
    obs = env.get_observations().clone()
    mean_hip = runner.alg.actor_critic.actor(obs).cpu()
    # TorchScript export == HIP inference (fp32 MFMA vs torch CPU: rtol 1e-4)
    path = ck.export_policy_as_jit(runner.alg.actor_critic, str(tmp_path / "exported"))
    mod = torch.jit.load(path)
    np.testing.assert_allclose(mod(obs.cpu()).detach().numpy(), mean_hip.numpy(), rtol=1e-4, atol=1e-5)
    env.close(); runner.ppo.close()
    # resume the way play.py does
    args2 = _args("anymal_c_flat", 8)
    env_cfg, train_cfg = task_registry.get_cfgs(args2.task)
    env_cfg.env.num_envs = 8
    env2, _ = task_registry.make_env(name=args2.task, args=args2, env_cfg=env_cfg)
    train_cfg.runner.resume = True
    train_cfg.policy.actor_hidden_dims = [64, 32]
    train_cfg.policy.critic_hidden_dims = [64, 32]
    runner2, _ = task_registry.make_alg_runner(env=env2, name=args2.task, args=args2, train_cfg=train_cfg, log_root=log_root)
    for k, v in runner2.ppo.state_dict().items():
        assert torch.equal(v.cpu(), want_sd[k]), k
    got_opt = runner2.ppo.optimizer_state_dict()
    assert got_opt["param_groups"][0]["lr"] == want_opt["param_groups"][0]["lr"]
    for i in want_opt["state"]:
        assert torch.equal(got_opt["state"][i]["exp_avg_sq"], want_opt["state"][i]["exp_avg_sq"])
        assert float(got_opt["state"][i]["step"]) == float(want_opt["state"][i]["step"]) == 60.0     # 3 iterations x 20 steps
    assert runner2.current_learning_iteration == 3
    policy = runner2.get_inference_policy(device=env2.device)
    o = env2.get_observations()
    for _ in range(5):
        o, _, r, dn, info = env2.step(policy(o.detach()))
    assert torch.isfinite(o).all() and torch.isfinite(r).all()
    env2.close(); runner2.ppo.close()


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough"])
def test_play_script_runs(task, tmp_path, monkeypatch):
    import legged_gym_dev_amd
    from legged_gym_dev_amd.envs import task_registry
    monkeypatch.setattr(legged_gym_dev_amd, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    monkeypatch.setattr(sys.modules["legged_gym_dev_amd.utils.task_registry"], "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    args = _args(task, 16)
    env, _ = task_registry.make_env(name=args.task, args=args)
    _, shared = task_registry.get_cfgs(args.task)       # registered cfg objects are shared and mutated in place (as in the reference)
    shared.runner.resume = False
    shared.policy.actor_hidden_dims = [128, 64, 32]
    shared.policy.critic_hidden_dims = [128, 64, 32]
    runner, train_cfg = task_registry.make_alg_runner(env=env, name=args.task, args=args)
    runner.learn(num_learning_iterations=1)
    env.close(); runner.ppo.close()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "legged_gym_dev_amd", "scripts"))
    import importlib
    play = importlib.import_module("play")
    monkeypatch.setattr(play, "LEGGED_GYM_ROOT_DIR", str(tmp_path))
    monkeypatch.chdir(tmp_path)
    rec = play.play(_args(task, 1), num_steps=120, out_mat=str(tmp_path / "play_data.mat"))
    assert rec["pos"].shape == (120, 3) and np.isfinite(rec["torque"]).all() and np.abs(rec["action"]).sum() > 0
    import scipy.io
    m = scipy.io.loadmat(str(tmp_path / "play_data.mat"))
    assert set(("cmd", "action", "pos", "quat", "dof", "vel", "omega", "ddof", "torque")) <= set(m)
    assert os.path.exists(tmp_path / "logs" / train_cfg.runner.experiment_name / "exported" / "policies" / "policy_1.pt")


def test_trajectory_dataset_rollout():
    """SURVEY.md 8(f) f3, the dataset side: the reference's rollout loop (deep_tube_learning/data_collection_trajectory.py:97-183)
    on the HIP trajectory env -- records at the ROM rate, shapes and bookkeeping as the tube-learning stage expects them."""
    import copy
    from legged_gym_dev_amd.envs import task_registry
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "legged_gym_dev_amd", "scripts"))
    import collect_trajectory_data as ctd
    args = _args("anymal_c_flat_trajectory", 64)
    env_cfg, train_cfg = task_registry.get_cfgs(args.task)
    env_cfg = copy.deepcopy(env_cfg)
    env_cfg.env.num_envs = 64
    env, _ = task_registry.make_env(name=args.task, args=args, env_cfg=env_cfg)
    try:
        A = env.num_actions
        policy = lambda obs: torch.zeros(obs.shape[0], A, device=obs.device)      # stand still: few falls
        recs = ctd.collect(env, policy, epochs=2, episode_length_s=1.5, save_debugging_data=True)
        assert len(recs) == 2
        r = recs[-1]
        T = int(1.5 / env.rom.dt)
        assert r["z"].shape == (64, T + 1, 2) and r["v"].shape == (64, T, 2) and r["pz_x"].shape == (64, T + 1, 2)
        assert r["done"].shape == (64, T) and r["x"].shape == (64, T + 1, 7 + A + 6 + A)
        assert np.isfinite(r["z"]).all() and np.isfinite(r["x"]).all()
        assert np.abs(r["v"]).max() <= 0.35 + 1e-6                                # ROM input bounds (rom.v_max)
        np.testing.assert_array_equal(r["pz_x"], r["x"][:, :, :2])                 # proj_z = base xy
        # between two records every ROM made exactly one step: the recorded ROM state moves by at most rom_dt * v_max per axis
        live = ~r["done"]
        dz = np.abs(np.diff(r["z"], axis=1))[live]
        assert dz.max() <= 0.35 * env.rom.dt * 1.5 + 1e-5, dz.max()
        # a terminated env restarts with zero tracking error
        if r["done"].any():
            i, t = np.argwhere(r["done"])[0]
            np.testing.assert_array_equal(r["z"][i, t + 1], r["pz_x"][i, t + 1])
    finally:
        env.close()


@pytest.mark.parametrize("case", [("anymal_c_flat", 1000, [512, 256, 128]), ("cassie", 777, [512, 256, 128]), ("anymal_c_flat", 7, [64, 32]),
                                  ("anymal_c_flat_trajectory", 1234, [512, 256, 128]), ("anymal_c_rough_trajectory", 600, [512, 256, 128]),
                                  ("anymal_c_rough", 2049, [96, 40, 24])])
def test_sizes_off_every_tile_grid(case):
    """Env counts that are a multiple of no tile (control loop: 16 / 32 envs per workgroup; post-step: 16; one-launch act: 32
    rows; minibatch rows off the 128-row GEMM and 64-row head grids; fewer envs than action dimensions) through the product
    runner for two PPO iterations: finite state and parameters, no physics fault, parameters moved (tools/size_sweep.py)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import size_sweep
    size_sweep.run(*case)


@pytest.mark.parametrize("task", ["anymal_c_flat", "anymal_c_rough", "anymal_c_flat_trajectory"])
def test_fused_rollout_epilogue_equals_separate_launches(task):
    """OnPolicyRunner.rollout with the env attached to the learner (lg_ppo_attach_env: the step's single-workgroup epilogue and
    process_env_step ride on the next act's launch) against the same rollout with every piece as a launch of its own: rollout
    storage (observations, actions, values, log-probs, rewards incl. the time-out bootstrap on the reference's stale mask, dones),
    the runner's episode bookkeeping, the env's extras / logging sums and its state must come out bit for bit the same.  Episode
    clocks are scattered so that time-outs (and steps without any reset, where the stale mask is what counts) both occur."""
    import copy
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.rl.runner import OnPolicyRunner
    from legged_gym_dev_amd.utils.helpers import class_to_dict
    outs = []
    for fuse in (True, False):
        args = _args(task, 256)
        env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(task))
        env_cfg.env.num_envs = 256
        if env_cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            env_cfg.terrain.num_rows, env_cfg.terrain.num_cols, env_cfg.terrain.border_size = 4, 8, 5
            env_cfg.terrain.max_init_terrain_level = 3
        train_cfg.policy.actor_hidden_dims = train_cfg.policy.critic_hidden_dims = [128, 64, 32]
        env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
        torch.manual_seed(7)
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
        runner.fuse_epilogue = fuse
        try:
            assert runner.ppo.lib.lg_ppo_debug_get_fused_act(runner.ppo.ctx) == 1
            ep = torch.arange(256, device="cuda:0") * 4 % 1001
            ep[:6] = torch.tensor([1000, 1001, 999, 990, 985, 980], device="cuda:0")
            env.episode_length_buf = ep
            rec = {}
            for it in range(2):
                runner.rollout()
                torch.cuda.synchronize()
                for k in ("obs", "actions", "values", "log_prob", "rewards", "dones", "mu", "returns", "advantages", "cur_reward_sum",
                          "cur_episode_len", "ep_stats", "ep_ring", "ep_ring_count"):
                    rec[f"{it}_{k}"] = runner.ppo.t[k].clone()
                for k in ("extras_episode", "extras_time_outs", "extras_episode_acc", "n_reset", "root_states", "dof_state", "episode_length",
                          "episode_sums", "obs", "rew", "reset"):
                    rec[f"{it}_env_{k}"] = env.core.t[k].clone()
                runner.ppo._call("end_update")                   # rewind the storage cursor (no update: same policy in both runs)
            assert int(rec["1_ep_ring_count"]) > 5 and bool(rec["1_dones"].any())
            outs.append(rec)
        finally:
            env.close()
            runner.ppo.close()
    a, b = outs
    for k in a:
        if k.endswith("ep_ring"):            # slot order inside one step follows the atomics; compare as multisets
            assert torch.equal(a[k].sort(dim=1).values, b[k].sort(dim=1).values), k
        elif k.endswith("ep_stats") or k.endswith("adv_partial"):
            torch.testing.assert_close(a[k], b[k], rtol=1e-5, atol=1e-5, msg=k)
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("case", [("anymal_c_flat", [512, 256, 128], "elu"), ("anymal_c_rough", [128, 64, 32], "elu"),
                                  ("anymal_c_flat", [96, 64, 32], "tanh")])
def test_deterministic_mode_reproduces_training_bit_for_bit(case):
    """lg_ppo_set_deterministic: every cross-workgroup sum of the update (weight-gradient slices, bias column sums, head row sums,
    KL / loss sums, gradient norm, advantage moments) accumulated as fixed-point integers.  Two fresh runs of the product runner
    from one seed then hold identical parameters, Adam moments, learning rate, rollout storage and env state after three PPO
    iterations -- through the k_head_net (128 wide), k_head_fused (32 wide) and k_loss (non-ELU) head paths -- while the
    default float-atomic mode lands within the atomics' noise of it after the first iteration."""
    import copy
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.rl.runner import OnPolicyRunner
    from legged_gym_dev_amd.utils.helpers import class_to_dict
    task, hidden, act = case

    def run(det, iters):
        args = _args(task, 256)
        env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(task))
        env_cfg.env.num_envs = 256
        if env_cfg.terrain.mesh_type in ("heightfield", "trimesh"):
            env_cfg.terrain.num_rows, env_cfg.terrain.num_cols, env_cfg.terrain.border_size = 4, 8, 5
            env_cfg.terrain.max_init_terrain_level = 3
        train_cfg.policy.actor_hidden_dims = train_cfg.policy.critic_hidden_dims = list(hidden)
        train_cfg.policy.activation = act
        env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
        torch.manual_seed(11)
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
        try:
            runner.ppo.set_deterministic(det)
            first = None
            for it in range(iters):
                runner.learn(1, init_at_random_ep_len=False)
                torch.cuda.synchronize()
                if it == 0:
                    first = runner.ppo.t["params"].clone()
            out = {k: runner.ppo.t[k].clone() for k in ("params", "adam_m", "adam_v", "stats", "obs", "actions", "values", "returns", "advantages")}
            out.update({"env_" + k: env.core.t[k].clone() for k in ("root_states", "dof_state", "obs", "episode_length", "fault_total")})
            out["first"] = first
            return out
        finally:
            env.close()
            runner.ppo.close()

    a, b = run(True, 3), run(True, 3)
    assert int(a["env_fault_total"][0]) == 0 and bool(torch.isfinite(a["params"]).all())
    assert not torch.equal(a["first"], a["params"])                       # it trained
    for k in a:
        assert torch.equal(a[k], b[k]), k
    c = run(False, 1)                                                      # float atomics: same numbers up to their order noise
    n = a["first"].numel() - 2
    rel = float((c["first"][:n] - a["first"][:n]).norm() / a["first"][:n].norm())
    assert rel < 1e-5, rel


def test_deferred_log_blocks_equal_the_synchronous_ones(monkeypatch, capsys):
    """OnPolicyRunner.learn prints iteration k's block while the GPU works on iteration k + 1 (values from one packed asynchronous copy,
    phase times from HIP events); LG_LOG_SYNC=1 keeps rsl_rl's synchronise - measure - print order.  In deterministic mode the two must
    print the same blocks, line for line, apart from the three timing lines -- same losses, learning rate, std, episode statistics,
    per-term episode sums, step totals -- one block per iteration, in order."""
    import copy
    from legged_gym_dev_amd.envs import task_registry
    from legged_gym_dev_amd.rl.runner import OnPolicyRunner
    from legged_gym_dev_amd.utils.helpers import class_to_dict
    monkeypatch.setenv("LG_DETERMINISTIC", "1")

    def blocks(sync):
        monkeypatch.setenv("LG_LOG_SYNC", "1" if sync else "0")
        task = "anymal_c_rough"
        args = _args(task, 256)
        env_cfg, train_cfg = (copy.deepcopy(c) for c in task_registry.get_cfgs(task))
        env_cfg.env.num_envs = 256
        env_cfg.terrain.num_rows, env_cfg.terrain.num_cols, env_cfg.terrain.border_size = 4, 8, 5
        env_cfg.terrain.max_init_terrain_level = 3
        train_cfg.policy.actor_hidden_dims = train_cfg.policy.critic_hidden_dims = [64, 32]
        env, _ = task_registry.make_env(name=task, args=args, env_cfg=env_cfg)
        torch.manual_seed(5)
        runner = OnPolicyRunner(env, class_to_dict(train_cfg), None, device="cuda:0")
        capsys.readouterr()
        try:
            runner.learn(4, init_at_random_ep_len=True)
            runner.learn(1)
        finally:
            env.close()
            runner.ppo.close()
        out = capsys.readouterr().out
        keep = [ln for ln in out.splitlines() if not any(k in ln for k in ("Computation:", "Iteration time:", "Total time:"))]
        assert sum("Learning iteration" in ln for ln in keep) == 5
        return keep, runner.tot_timesteps

    a, na = blocks(False)
    b, nb = blocks(True)
    assert na == nb == 5 * 24 * 256
    assert a == b
    assert any("Mean reward:" in ln for ln in a) and any("Mean episode terrain_level:" in ln for ln in a)
    assert [ln.strip() for ln in a if "Learning iteration" in ln] == ["Learning iteration 0/4", "Learning iteration 1/4", "Learning iteration 2/4",
                                                                        "Learning iteration 3/4", "Learning iteration 4/5"]

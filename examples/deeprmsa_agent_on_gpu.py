#!/usr/bin/env python3
"""An agent on the same GPU as the envs — the loop the reference's examples/stable_baselines3/DeepRMSA.ipynb:272-302 runs through
SB3 (Monitor -> DummyVecEnv -> PPO), without anything crossing PCIe and without the host in the rollout: DeepRMSA-v0
observations, rewards and dones are torch views of the batch's own device arrays, the policy network writes its actions into the
batch's action array, and one `step(None, fetch=False)` is one launch of the step kernel for every env.

Round 6: the whole rollout — T steps of (observation cast, MLP forward, sampling, action store, env step, rollout buffers) — is
captured ONCE in a `torch.cuda.CUDAGraph` on the batch's own HIP stream (`env.torch_stream()`; `step(None, fetch=False)` neither
synchronises nor allocates, its flag word is read lazily by `env.check()`) and replayed per update: the host launches one graph
per rollout instead of ~15 kernels per step.  Round 5's eager loop reached 2.8e7 env-steps/s at 65 536 envs, bound by torch's
per-kernel launch cost.

The policy is a small MLP trained with a plain policy-gradient update over the stored rollout (reward-to-go, batch-mean baseline,
entropy bonus; log-probabilities recomputed with gradients from the stored observations and actions, as PPO does); the point of the
example is the data path, not the learning algorithm — the SAP-FF heuristic's acceptance on the same traffic is printed beside it.

    python examples/deeprmsa_agent_on_gpu.py [num_envs] [updates] [--eager]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import torch  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
EAGER = "--eager" in sys.argv
B = int(args[0]) if len(args) > 0 else 4096
UPDATES = int(args[1]) if len(args) > 1 else 60
T = 32  # steps per rollout
kw = dict(topology="nsfnet_chen", mean_service_holding_time=7.5, mean_service_inter_arrival_time=1 / 12.0, j=1,
          episode_length=50, num_spectrum_resources=100)

# the heuristic on the same traffic, entirely on the device
ref = orl.make("DeepRMSA-v0", num_envs=B, seeds=1, **kw)
ref.run("SAP", 50 * 20)
processed, accepted = ref.totals()
print("SAP-FF heuristic: %.4f of the requests accepted" % (accepted / processed))
ref.close()

env = orl.make("DeepRMSA-v0", num_envs=B, seeds=1, **kw)
dev = "cuda:%d" % env.device_id
obs, rew, done, act = (env.device_tensor(n) for n in ("obs", "reward", "done", "actions"))  # views of the batch's arrays
n_actions = env.k_paths * env.j + (1 if env.allow_rejection else 0)
net = torch.nn.Sequential(torch.nn.Linear(env.obs_dim, 128), torch.nn.ELU(), torch.nn.Linear(128, 128), torch.nn.ELU(),
                          torch.nn.Linear(128, n_actions)).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-4)
obs_buf = torch.zeros((T, B, env.obs_dim), device=dev)
act_buf = torch.zeros((T, B), dtype=torch.long, device=dev)
rew_buf = torch.zeros((T, B), device=dev)
env.reset()
env.observation()  # the kernels keep `obs` current from here on
stream = env.torch_stream()


def policy_part(t):
    """observation -> action of step t, stored for the update (no gradients: the update recomputes the log-probabilities)"""
    with torch.no_grad():
        x = obs.float()
        obs_buf[t].copy_(x)
        logits = net(x)
        u = torch.rand_like(logits).clamp_(1e-7, 1 - 1e-7)
        a = (logits - torch.log(-torch.log(u))).argmax(dim=1)  # Gumbel-max = a sample of Categorical(logits)
        act_buf[t].copy_(a)
        act[:, 0] = a.int()


def env_part(t):
    env.step(None, auto_reset=True, fetch=False)  # one launch; reward / done / obs are rewritten in place
    rew_buf[t].copy_(rew)


def rollout():
    for t in range(T):
        policy_part(t)
        env_part(t)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g


with torch.cuda.stream(stream):  # (libraries' workspaces are set up outside the capture)
    rollout()
torch.cuda.synchronize()
g_roll = None
if not EAGER:
    g_roll = capture(rollout)
    g_net = capture(lambda: [policy_part(t) for t in range(T)])
    g_env = capture(lambda: [env_part(t) for t in range(T)])


def timed(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.cuda.stream(stream):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n


def update():
    logits = net(obs_buf.view(T * B, -1))
    logp_all = torch.log_softmax(logits, dim=1)
    logp = logp_all.gather(1, act_buf.view(-1, 1)).view(T, B)
    entropy = -(logp_all.exp() * logp_all).sum(dim=1).mean()
    ret = torch.zeros(B, device=dev)
    rets = []
    for t in reversed(range(T)):  # reward-to-go, discounted
        ret = rew_buf[t] + 0.95 * ret
        rets.append(ret)
    rets = torch.stack(rets[::-1])
    adv = rets - rets.mean(dim=1, keepdim=True)
    loss = -(logp * adv).mean() - 0.01 * entropy
    opt.zero_grad()
    loss.backward()
    opt.step()


torch.cuda.synchronize()
t0 = time.time()
steps = 0
with torch.cuda.stream(stream):  # everything below is queued on the stream the step kernel runs on
    for u_ in range(UPDATES):
        if g_roll is not None:
            g_roll.replay()
        else:
            rollout()
        steps += T * B
        update()
        if u_ % 10 == 9 or u_ == UPDATES - 1:
            mean_r = rew_buf.mean().item()  # (.item() waits) +1 accepted, -1 blocked (deeprmsa_env.py:123-124)
            print("update %3d: accepted %.4f of the requests of its rollout, %.2f M env-steps/s incl. the network and the update"
                  % (u_ + 1, 0.5 + 0.5 * mean_r, steps / (time.time() - t0) / 1e6))
torch.cuda.synchronize()
if g_roll is not None:
    # where a rollout's time goes (graphs of the two halves on their own; the env half steps on whatever actions are in the array)
    t_roll, t_net, t_env = timed(g_roll.replay), timed(g_net.replay), timed(g_env.replay)
    t_upd = timed(update, 3)
    print("rollout of %d steps x %d envs: %.2f ms = %.1f M env-steps/s (network forward + sampling %.2f ms = %.0f %%, env steps %.2f ms = %.0f %%); "
          "update %.2f ms" % (T, B, 1e3 * t_roll, T * B / t_roll / 1e6, 1e3 * t_net, 100 * t_net / t_roll, 1e3 * t_env, 100 * t_env / t_roll,
                              1e3 * t_upd))
env.check()
env.close()

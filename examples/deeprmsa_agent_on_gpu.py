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

`--halves`: the envs as TWO batches of num_envs / 2, each with its own stream and its own captured rollout, replayed side by side:
the step kernel of one half (bound by memory latency, half of the vector ALU idle) overlaps the network's GEMMs of the other.

    python examples/deeprmsa_agent_on_gpu.py [num_envs] [updates] [--eager] [--halves]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import torch  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
EAGER = "--eager" in sys.argv
HALVES = "--halves" in sys.argv and not EAGER
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

# one batch, or two halves with a stream each (env i of the second half is env B/2 + i of the whole: seed 1 + B/2 + i)
parts = [(0, B)] if not HALVES else [(0, B // 2), (B // 2, B - B // 2)]
envs = [orl.make("DeepRMSA-v0", num_envs=n, seeds=1 + lo, **kw) for lo, n in parts]
env = envs[0]
dev = "cuda:%d" % env.device_id
n_actions = env.k_paths * env.j + (1 if env.allow_rejection else 0)
net = torch.nn.Sequential(torch.nn.Linear(env.obs_dim, 128), torch.nn.ELU(), torch.nn.Linear(128, 128), torch.nn.ELU(),
                          torch.nn.Linear(128, n_actions)).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=3e-4)
obs_buf = torch.zeros((T, B, env.obs_dim), device=dev)  # the rollout of all envs; each part writes its columns
act_buf = torch.zeros((T, B), dtype=torch.long, device=dev)
rew_buf = torch.zeros((T, B), device=dev)


class Part:
    """one batch of envs: views of its device arrays, its columns of the rollout buffers, its stream and its captured rollout"""

    def __init__(self, e, lo, n):
        self.env, self.lo, self.hi = e, lo, lo + n
        self.obs, self.rew, self.done, self.act = (e.device_tensor(k) for k in ("obs", "reward", "done", "actions"))
        e.reset()
        e.observation()  # the kernels keep `obs` current from here on
        self.stream = e.torch_stream()
        self.graph = None

    def policy_part(self, t):
        """observation -> action of step t, stored for the update (no gradients: the update recomputes the log-probabilities)"""
        with torch.no_grad():
            x = self.obs.float()
            obs_buf[t, self.lo:self.hi].copy_(x)
            logits = net(x)
            u = torch.rand_like(logits).clamp_(1e-7, 1 - 1e-7)
            a = (logits - torch.log(-torch.log(u))).argmax(dim=1)  # Gumbel-max = a sample of Categorical(logits)
            act_buf[t, self.lo:self.hi].copy_(a)
            self.act[:, 0] = a.int()

    def env_part(self, t):
        self.env.step(None, auto_reset=True, fetch=False)  # one launch; reward / done / obs are rewritten in place
        rew_buf[t, self.lo:self.hi].copy_(self.rew)

    def rollout(self):
        for t in range(T):
            self.policy_part(t)
            self.env_part(t)

    def capture(self, fn):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=self.stream):
            fn()
        return g


P = [Part(e, lo, n) for e, (lo, n) in zip(envs, parts)]
for p in P:
    with torch.cuda.stream(p.stream):  # (libraries' workspaces are set up outside the capture)
        p.rollout()
torch.cuda.synchronize()
if not EAGER:
    for p in P:
        p.graph = p.capture(p.rollout)
        p.g_net = p.capture(lambda: [p.policy_part(t) for t in range(T)])
        p.g_env = p.capture(lambda: [p.env_part(t) for t in range(T)])
main = torch.cuda.current_stream()


def rollouts(which="graph"):
    """the parts' rollouts side by side, each on its own stream; the caller's stream continues when all are done"""
    for p in P:
        p.stream.wait_stream(main)
        with torch.cuda.stream(p.stream):
            if EAGER:
                p.rollout()
            else:
                getattr(p, which).replay()
    for p in P:
        main.wait_stream(p.stream)


def timed(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.time()
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
for u_ in range(UPDATES):
    rollouts()
    steps += T * B
    update()
    if u_ % 10 == 9 or u_ == UPDATES - 1:
        mean_r = rew_buf.mean().item()  # (.item() waits) +1 accepted, -1 blocked (deeprmsa_env.py:123-124)
        print("update %3d: accepted %.4f of the requests of its rollout, %.2f M env-steps/s incl. the network and the update"
              % (u_ + 1, 0.5 + 0.5 * mean_r, steps / (time.time() - t0) / 1e6))
torch.cuda.synchronize()
if not EAGER:
    # where a rollout's time goes (graphs of the two halves of a step on their own; the env half steps on whatever actions are in the array)
    t_roll, t_net, t_env = timed(rollouts), timed(lambda: rollouts("g_net")), timed(lambda: rollouts("g_env"))
    t_upd = timed(update, 3)
    print("rollout of %d steps x %d envs%s: %.2f ms = %.1f M env-steps/s (network forward + sampling alone %.2f ms, env steps alone %.2f ms); "
          "update %.2f ms" % (T, B, " as two halves side by side" if HALVES else "", 1e3 * t_roll, T * B / t_roll / 1e6, 1e3 * t_net, 1e3 * t_env,
                              1e3 * t_upd))
for e in envs:
    e.check()
    e.close()

#!/usr/bin/env python3
"""An agent on the same GPU as the envs — the loop the reference's examples/stable_baselines3/DeepRMSA.ipynb runs through SB3,
without anything crossing PCIe: DeepRMSA-v0 observations, rewards and dones are read as torch tensors over the batch's own
device arrays, the policy network writes its actions into the batch's action array, and one `step(None, fetch=False)` is one
launch of the step kernel for every env.  The network runs on the batch's own HIP stream (`env.torch_stream()`), so the step
kernel and the network's kernels are ordered by the stream: the host only queues work and never waits inside a rollout.

The policy is a small MLP trained with a plain policy-gradient update (reward-to-go over a short rollout, batch-mean baseline,
entropy bonus); the point of the example is the data path, not the learning algorithm — the SAP-FF heuristic's acceptance on
the same traffic is printed beside it.

    python examples/deeprmsa_agent_on_gpu.py [num_envs] [updates]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import torch  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
UPDATES = int(sys.argv[2]) if len(sys.argv) > 2 else 60
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
env.reset()
env.observation()  # the kernels keep `obs` current from here on
torch.cuda.synchronize()
t0 = time.time()
steps = 0
with torch.cuda.stream(env.torch_stream()):  # everything below is queued on the stream the step kernel runs on
    for u in range(UPDATES):
        logps, rewards, entropies = [], [], []
        for t in range(T):
            dist = torch.distributions.Categorical(logits=net(obs.float()))
            a = dist.sample()
            act[:, 0] = a.int()
            env.step(None, auto_reset=True, fetch=False)  # one launch; reward / done / obs are rewritten in place
            logps.append(dist.log_prob(a))
            entropies.append(dist.entropy())
            rewards.append(rew.float().clone())
            steps += B
        ret = torch.zeros(B, device=dev)
        loss = 0.0
        for t in reversed(range(T)):  # reward-to-go, discounted
            ret = rewards[t] + 0.95 * ret
            loss = loss - (logps[t] * (ret - ret.mean())).mean() - 0.01 * entropies[t].mean()
        opt.zero_grad()
        (loss / T).backward()
        opt.step()
        if u % 10 == 9 or u == UPDATES - 1:
            mean_r = torch.stack(rewards).mean().item()  # (.item() waits) +1 accepted, -1 blocked (deeprmsa_env.py:123-124)
            print("update %3d: accepted %.4f of the requests of its rollout, %.2f M env-steps/s incl. the network and the update"
                  % (u + 1, 0.5 + 0.5 * mean_r, steps / (time.time() - t0) / 1e6))
env.check()
env.close()

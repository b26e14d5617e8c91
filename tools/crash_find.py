import os, sys, subprocess
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
code = '''
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_gpu_parity as T
import optical_rl_gym_amd as orl
from bench import WORKLOADS
v, masks, workload, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
for k, val in T.IMPL_ENV[v].items():
    if val is None: os.environ.pop(k, None)
    else: os.environ[k] = val
if masks != "-": os.environ["ORL_ITEM_MASKS"] = masks
fam, topo, kw, policy = WORKLOADS[workload]
kw = dict(kw, episode_length=70)
env = orl.make(fam, topology=topo, num_envs=batch, seeds=[5 + 11 * i for i in range(batch)], **kw)
env.run(policy, 500)
print("ok", v, masks, workload, int(env.lib.orl_batch_debug_persist_form(env._h)), int(env.lib.orl_batch_debug_persist_spec(env._h)))
'''
open("/tmp/one.py", "w").write(code)
for workload, batch in (("cfg2", "4096"), ("cfg4n", "1024")):
    for v, m in (("wave64", "-"), ("split2", "1"), ("split2", "2"), ("persist", "1"), ("persist", "2"), ("persist_global", "1"), ("persist_lds", "1")):
        p = subprocess.run([sys.executable, "/tmp/one.py", v, m, workload, batch], capture_output=True, text=True)
        print(workload, v, m, "rc", p.returncode, p.stdout.strip()[-80:], p.stderr.strip()[-200:].replace("\n", " | "), flush=True)

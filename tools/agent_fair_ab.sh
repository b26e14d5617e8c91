mkdir -p gpurun_out/fair
for f in 0 7 5 9 0 7; do
  ORL_AGENT_FAIR=$f python3 tools/agent_loop_rate.py cfg2 65536 2>/dev/null | tail -1 > gpurun_out/fair/agent_f$f.json
  python3 -c "
import json
d=json.load(open('gpurun_out/fair/agent_f$f.json'))
print('agent fair=$f', ' '.join('%s %.1f' % (k, v['us_per_step']) for k, v in d.items() if isinstance(v, dict)))"
done

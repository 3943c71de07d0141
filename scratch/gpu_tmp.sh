R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --cpu-fp16-steps 0 --side-legs "" 2>&1 | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); x=d['xcd_replicas']
print(d['value'], x['tokens_per_s'], x['frac'], x['parity']['sequence_0_ids_equal_single_sequence_engine'], x.get('prefill_then_decode'), x['one_per_xcd']['frac'])"

# round 6, call 5: full GPU suite, then the bench in the driver's form
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err; echo bench rc=$?; wc -c $O/bench_line.json; cp bench_detail.json $O/
python3 -c "
import json; d=json.load(open('$O/bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['other_configs'].items(): print(k, {a:b for a,b in v.items() if a!='workload'})"

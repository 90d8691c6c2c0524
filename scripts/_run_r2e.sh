cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02e
HPV_BENCH_ONE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 400 --warmup 40 2> gpurun_out/r02e/bench2.err | tail -1 > gpurun_out/r02e/bench2.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02e/bench2.json').read())
for k,v in d.items(): print(k, ':', json.dumps(v)[:260])
PY
tail -5 gpurun_out/r02e/bench2.err
HPV_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 400 --warmup 40 --no-cpu-baseline --no-residual-roofline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FORCE_DIST 1 rank:', d['value'], d['config'].get('exchange'), d.get('ms_per_step'))"

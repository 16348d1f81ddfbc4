mkdir -p gpurun_out/final1
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final1/bench.json 2> gpurun_out/final1/bench.err
tail -c 600 gpurun_out/final1/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final1/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"])
for s in d.get("secondary", []):
    print({k: s.get(k) for k in ("metric", "value", "unit", "ms_per_step", "config")})
PY

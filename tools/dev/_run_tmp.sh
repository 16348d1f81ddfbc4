mkdir -p gpurun_out/final4
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > gpurun_out/final4/bench.json 2> gpurun_out/final4/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final4/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("metric", "value", "ms_per_step")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
for s in d.get("secondary", []):
    print(s["metric"][:72], "|", s["value"], s.get("ms_per_step"), s["config"].get("batch"), s["config"].get("optimizer", ""))
PY

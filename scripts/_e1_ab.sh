set -x
O=gpurun_out/r04_v; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_step or rollout or forward or agent" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 200 python bench.py --steps 10 --warmup 3 > $O/bench_words.log 2>&1; tail -c 400 $O/bench_words.log | head -c 0
python - <<'PY'
import json
for n in ("words",):
    l=[x for x in open(f"gpurun_out/r04_v/bench_{n}.log") if x.startswith("{")][-1]
    j=json.loads(l); print(n, j["value"], j.get("gpu_reference_semantics_E1"))
PY
PH_ACT_HOST_WAIT=stream timeout 200 python bench.py --steps 10 --warmup 3 > $O/bench_stream.log 2>&1
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r04_v/bench_stream.log") if x.startswith("{")][-1]
j=json.loads(l); print("stream", j["value"], j.get("gpu_reference_semantics_E1"))
PY

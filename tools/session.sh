#!/bin/bash
# round 5, session 3: whole GPU suite + the default bench line (live counter passes, forward_other_configs) with its wall time
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
( time timeout 2400 python -m pytest tests -x -q -m gpu ) > gpurun_out/s3/pytest_gpu.txt 2>&1
tail -6 gpurun_out/s3/pytest_gpu.txt
( time python bench.py ) > gpurun_out/s3/bench.json 2> gpurun_out/s3/bench.err
tail -4 gpurun_out/s3/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/s3/bench.json").read().strip().splitlines()[0])
print({k: d[k] for k in ("value", "ms_per_step")})
r = d["roofline"]; print(r["kernel"], r["frac"], r["traffic"], r.get("traffic_detail"))
print(d["token_cluster"]["cfg2"])
for k, v in d["forward_other_configs"].items():
    print(k, v["ms_per_step"], v["clips_per_s"], v["whole_step_frac_of_f16_mfma_peak"], v["roofline"]["kernel"], v["roofline"]["frac"])
PY

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_spectral_gpu.py tests/test_cluster_gpu.py -x -q 2>&1 | tail -3
python - <<'PY'
import torch, bench, json
dev = torch.device("cuda:0")
for name in ("cfg5 ViT-B/16", "cfg6 ViT-B/16 ActivityNet (per GPU)"):
    sh = bench.CLUSTER_SHAPES[name]
    print(name, json.dumps(bench.cluster_bench(sh, dev, iters=10)), flush=True)
    print(name, "spectral", json.dumps(bench.spectral_cluster_bench(sh, dev)), flush=True)
PY

#!/bin/bash
bash tools/gpu_pmc.sh r02_band > gpurun_out/pmc_r02_band.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/pmc_r02_band_summary.json"))
for k,v in d.items():
    if "spmv" in k: print(k, {c: round(x["mean"],1) for c,x in v.items()})
PY

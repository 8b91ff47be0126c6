#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest_gpu.log | cut -c1-300

#!/bin/bash
mkdir -p gpurun_out
if [ -n "$1" ] && [ -e "$1" ]; then T="$1"; shift; else T=tests; fi
timeout 1500 python -m pytest $T -m gpu -q "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -30 gpurun_out/pytest_gpu.log | cut -c1-300

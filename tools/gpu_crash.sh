#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/crash_each.txt
for T in $(python -m pytest tests/test_gpu_parity.py --collect-only -q -k "not isai" 2>/dev/null | grep "::" | sed 's/\[.*//' | sort -u); do
  MALLOC_CHECK_=3 timeout 300 python -m pytest "$T" -m gpu -q -x > gpurun_out/crash_one.log 2>&1; rc=$?
  echo "$rc $T $(grep -m1 -i "free()\|malloc\|corrupt\|double" gpurun_out/crash_one.log | cut -c1-100)" >> gpurun_out/crash_each.txt
done
cat gpurun_out/crash_each.txt

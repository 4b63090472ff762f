#!/bin/bash
# attention kernels: parity tests, then per-geometry timings
mkdir -p gpurun_out/attn
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | tail -25 > gpurun_out/attn/tests.log
timeout 300 python scripts/attn_bench.py > gpurun_out/attn/bench.log 2>&1
cat gpurun_out/attn/tests.log gpurun_out/attn/bench.log

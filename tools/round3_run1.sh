#!/bin/bash
# GPU run 1 of round 3: builder (doubling) tests, the new at-scale parity shapes, the 8.6 Gbp repeat-rich build timed
set -u
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_build.py -x -q -m gpu -s 2>&1 | tail -25 > gpurun_out/r3a/build_tests.log
timeout 420 python tools/build_scale.py 2048 4194304 /tmp/cf_scale_2r repeat > gpurun_out/r3a/build_2r_8.6Gbp.log 2>&1
echo "rc=$?" >> gpurun_out/r3a/build_2r_8.6Gbp.log
rm -rf /tmp/cf_scale_2r
timeout 600 python -m pytest tests/test_gpu_scale.py -x -q -m gpu -k "other_kernel_forms" 2>&1 | tail -25 > gpurun_out/r3a/scale_tests.log
timeout 120 python -m pytest tests/test_async_abi.py -x -q -m gpu -k "misuse" 2>&1 | tail -5 > gpurun_out/r3a/abi.log
tail -5 gpurun_out/r3a/*.log

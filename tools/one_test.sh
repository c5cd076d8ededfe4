#!/bin/bash
# usage: tools/one_test.sh VARIANT PYTEST_K  -> runs one parity test against a variant library
export LRBMS_HIP_LIB=$GRAFT_REPO_ROOT/pylrbms_amd/_variants/$1.so
timeout -k 5 120 python -m pytest tests/test_parity_gpu.py -q -x -k "$2" > gpurun_out/one_$1.log 2>&1
echo "$1: rc=$? $(tail -1 gpurun_out/one_$1.log | cut -c1-100)"

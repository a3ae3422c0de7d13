#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03m
export CVC_EXTRA_HIPCC_FLAGS="-DCVC_TS"
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
python tools/ts_probe.py > gpurun_out/r03m/ts.log 2>&1
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
cat gpurun_out/r03m/ts.log

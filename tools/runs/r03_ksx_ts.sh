#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r03o
export CVC_EXTRA_HIPCC_FLAGS="-DCVC_TS"
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
python tools/bench_ksx.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03o/ksx_ts.log
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
cat gpurun_out/r03o/ksx_ts.log

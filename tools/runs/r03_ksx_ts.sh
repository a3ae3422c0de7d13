#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# whatever happens below, the in-tree library is rebuilt without ablation flags on exit (build_hip.py also stamps the flags of a
# build and refuses to call a .so built with other flags up to date)
trap 'CVC_EXTRA_HIPCC_FLAGS= python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1' EXIT
mkdir -p gpurun_out/r03o
export CVC_EXTRA_HIPCC_FLAGS="-DCVC_TS"
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
python tools/bench_ksx.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r03o/ksx_ts.log
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
cat gpurun_out/r03o/ksx_ts.log

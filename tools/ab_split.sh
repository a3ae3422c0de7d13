#!/bin/bash
# A/B builds of the packed GEMM (split vs fp32, ablations) on the GPU box: each argument is a hipcc flag set
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for flags in "$@"; do
  export CVC_EXTRA_HIPCC_FLAGS="$flags"
  python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
  echo "== flags='$flags'"
  python tools/bench_split.py 2>/dev/null | cut -c1-260
done
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1

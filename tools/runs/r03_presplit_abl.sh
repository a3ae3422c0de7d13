#!/bin/bash
# Gate GEMM ablation: what would activations that arrive pre-split (3 x bf16 fragments written by their producers) buy?
#   PABL=5: no VALU split of the activations, same bytes;  PABL=6: no VALU split, 1.5 x the activation bytes (L2 hits)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
# whatever happens below, the in-tree library is rebuilt without ablation flags on exit (build_hip.py also stamps the flags of a
# build and refuses to call a .so built with other flags up to date)
trap 'CVC_EXTRA_HIPCC_FLAGS= python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1' EXIT
mkdir -p gpurun_out/r03k
for abl in 0 5 6; do
  if [ "$abl" = "0" ]; then export CVC_EXTRA_HIPCC_FLAGS=""; else export CVC_EXTRA_HIPCC_FLAGS="-DCVC_PABL=$abl"; fi
  python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
  echo "== PABL=$abl" >> gpurun_out/r03k/abl.log
  python tools/bench_packed_m.py >> gpurun_out/r03k/abl.log 2>&1
done
export CVC_EXTRA_HIPCC_FLAGS=""
python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
cat gpurun_out/r03k/abl.log

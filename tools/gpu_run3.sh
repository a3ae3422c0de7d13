export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
for v in "-DCVC_TILE_NO_XCD=1" ""; do
  CVC_EXTRA_HIPCC_FLAGS="$v" python cyclical-visual-captioning_amd/build_hip.py --force > /dev/null 2>&1
  echo "== flags $v" | tee -a gpurun_out/r02c/tile_xcd.log
  timeout 300 python tools/bench_tile.py 2>/dev/null | tee -a gpurun_out/r02c/tile_xcd.log
  timeout 300 python tools/bench_tile.py 2>/dev/null | tee -a gpurun_out/r02c/tile_xcd.log
done
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile" 2>&1 | tail -3

set -x
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r02a/gputest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/gputest.log
tail -15 gpurun_out/r02a/gputest.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02a/bench_20.json 2> gpurun_out/r02a/bench_20.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r02a/bench_300.json 2> gpurun_out/r02a/bench_300.err
timeout 600 python bench.py --beam 5 --steps 20 --warmup 2 > gpurun_out/r02a/bench_beam5.json 2> gpurun_out/r02a/bench_beam5.err
timeout 900 python bench.py --mode train --config cfg3 --steps 20 --warmup 3 > gpurun_out/r02a/bench_train.json 2> gpurun_out/r02a/bench_train.err
cut -c1-600 gpurun_out/r02a/bench_20.json; cut -c1-300 gpurun_out/r02a/bench_300.json; cut -c1-400 gpurun_out/r02a/bench_beam5.json; cut -c1-1500 gpurun_out/r02a/bench_train.json
tail -3 gpurun_out/r02a/*.err

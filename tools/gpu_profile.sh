# Collects the round's measured evidence into gpurun_out/prof/ (copy the summaries into profiles/ afterwards):
#   PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --kernel-trace only) of EVERY benched workload -> profiles/traffic/*.json
#   (tools/collect_traffic.py), the matrix-pipe-busy counters, bench lines, rocprofv3 kernel-trace summaries of the same commands,
#   the decode-step timeline.
#   usage (GPU box): bash tools/gpu_profile.sh r06 [pmc|bench|all]
TAG=${1:-r06}
WHAT=${2:-all}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O $R/profiles/traffic
pmc_pair() {   # name, config, beam, mode, bench args...
  local name=$1 cfg=$2 beam=$3 mode=$4; shift 4
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f -- python3 $R/bench.py "$@" > $O/pmc_${name}_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w -- python3 $R/bench.py "$@" > $O/pmc_${name}_write.log 2>&1
  FDB=$(find $O/pmc_f -name "*.db" | head -1); WDB=$(find $O/pmc_w -name "*.db" | head -1)
  python3 $R/tools/collect_traffic.py $FDB $WDB --config $cfg --beam $beam --mode $mode --outdir $R/profiles/traffic --md $O/${TAG}_pmc_traffic.md > $O/pmc_${name}.md 2>&1
  rm -rf $O/pmc_f $O/pmc_w
}
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  rm -f $O/${TAG}_pmc_traffic.md
  pmc_pair cfg2_greedy cfg2 1 decode --no-graph --steps 4 --warmup 1 --no-cpu-baseline --no-secondary
  pmc_pair cfg3_beam5 cfg3 5 decode --config cfg3 --beam 5 --no-graph --steps 3 --warmup 1 --no-cpu-baseline
  pmc_pair cfg5_greedy cfg5 1 decode --config cfg5 --no-graph --steps 3 --warmup 1 --no-cpu-baseline
  pmc_pair cfg5_beam5 cfg5 5 decode --config cfg5 --beam 5 --no-graph --steps 2 --warmup 1 --no-cpu-baseline
  pmc_pair cfg3_train cfg3 1 train --mode train --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-train-graph
  pmc_pair cfg4_train cfg4 1 train --mode train --config cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-train-graph
  pmc_pair cfg2_encoder cfg2 1 encoder --mode encoder --steps 2 --warmup 1 --encoder-forward-only
  # matrix-pipe busy per launch role (tools/rocpd_mfma_busy.py): greedy decode, beam decode, the training step
  rm -f $O/${TAG}_pmc_mfma_busy.md
  mfma_pass() {   # config, beam, mode, bench args...
    local cfg=$1 beam=$2 mode=$3; shift 3
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_m -o m -- python3 $R/bench.py "$@" > $O/pmc_mfma_${cfg}_${mode}_b${beam}.log 2>&1
    python3 $R/tools/rocpd_mfma_busy.py $(find $O/pmc_m -name "*.db" | head -1) --config $cfg --beam $beam --mode $mode >> $O/${TAG}_pmc_mfma_busy.md 2>&1
    echo >> $O/${TAG}_pmc_mfma_busy.md
    rm -rf $O/pmc_m
  }
  mfma_pass cfg2 1 decode --no-graph --steps 4 --warmup 1 --no-cpu-baseline --no-secondary
  mfma_pass cfg3 5 decode --config cfg3 --beam 5 --no-graph --steps 3 --warmup 1 --no-cpu-baseline
  mfma_pass cfg3 1 train --mode train --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline --no-train-graph
  cat $O/${TAG}_pmc_traffic.md
  # the per-workload traffic files travel back through gpurun_out/ (copy them into profiles/traffic/ and commit)
  rm -rf $O/traffic; cp -r $R/profiles/traffic $O/traffic
fi
if [ "$WHAT" = "bench" ] || [ "$WHAT" = "all" ]; then
  # the bench lines read this build's traffic (bench.py refuses a collection whose kernel-source hash is not the build's)
  python3 $R/bench.py > $O/bench_${TAG}_greedy.json 2> $O/bench_greedy.err
  python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${TAG}_greedy_20steps.json 2>> $O/bench_greedy.err
  python3 $R/bench.py --config cfg3 --beam 5 --steps 50 --warmup 3 > $O/bench_${TAG}_beam5.json 2> $O/bench_beam.err
  python3 $R/bench.py --config cfg5 --steps 50 --warmup 3 --no-cpu-baseline > $O/bench_${TAG}_cfg5_greedy.json 2> $O/bench_cfg5.err
  python3 $R/bench.py --config cfg5 --beam 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_${TAG}_cfg5_beam5.json 2>> $O/bench_cfg5.err
  python3 $R/bench.py --mode train --config cfg3 --steps 30 --warmup 3 > $O/bench_${TAG}_train.json 2> $O/bench_train.err
  python3 $R/bench.py --mode train --config cfg4 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_${TAG}_train_cfg4.json 2>> $O/bench_train.err
  python3 $R/bench.py --mode encoder --steps 10 --warmup 2 > $O/bench_${TAG}_encoder.json 2> $O/bench_encoder.err
  for what in greedy beam5 train encoder; do
    case $what in
      greedy) ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-secondary";;
      beam5) ARGS="--config cfg3 --beam 5 --steps 6 --warmup 2 --no-cpu-baseline";;
      train) ARGS="--mode train --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline --no-train-graph";;
      encoder) ARGS="--mode encoder --steps 3 --warmup 1";;
    esac
    rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/kt_$what.log 2>&1
    DB=$(find $O/kt -name "*.db" | head -1)
    python3 $R/tools/rocpd_summary.py $DB > $O/${TAG}_${what}_kernel_stats.md 2>&1
    rm -rf $O/kt
  done
  # decode-step timeline (per-position duration + idle gap) from a graph-replay kernel trace
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/kt_timeline.log 2>&1
  DB=$(find $O/kt -name "*.db" | head -1)
  python3 $R/tools/rocpd_step_timeline.py $DB 7 > $O/${TAG}_greedy_step_timeline.md 2>&1
  rm -rf $O/kt
  # the N-rank entry path at N = 1 (bench.py starts its rank processes itself): decode line and config 4's per-GPU training step
  python3 $R/bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-secondary > $O/bench_${TAG}_spawn1_greedy.json 2> $O/bench_spawn.err
  # (the N-rank path runs the gradient exchange whatever N is: here the per-bucket RS + AG on a ONE-rank RCCL communicator, inside the captured step)
  python3 $R/bench.py --gpus 1 --spawn --mode train --config cfg4 --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_${TAG}_spawn1_train_cfg4.json 2>> $O/bench_spawn.err
  python3 $R/bench.py --mode train --config cfg4 --always-exchange --steps 30 --warmup 3 --no-cpu-baseline > $O/bench_${TAG}_train_cfg4_exchange.json 2>> $O/bench_train.err
  # round 6: the N-rank entry path of the DEFAULT command (decode + config 4's training step run by every rank), the reference's
  # real flow end to end (raw features through the encoder), the VALU issue rates behind the score pass's cost model
  python3 $R/bench.py --gpus 1 --spawn --steps 20 --warmup 5 > $O/bench_${TAG}_spawn1_default.json 2>> $O/bench_spawn.err
  for c in cfg2 refdefault; do
    python3 $R/bench.py --mode e2e-train --config $c --steps 10 --warmup 2 > $O/bench_${TAG}_e2e_train_$c.json 2>> $O/bench_e2e.err
    python3 $R/bench.py --mode e2e-eval --config $c --steps 10 --warmup 2 > $O/bench_${TAG}_e2e_eval_$c.json 2>> $O/bench_e2e.err
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result $R/tools/valu_rates.hip -o /tmp/valu_rates 2>/dev/null && /tmp/valu_rates > $O/${TAG}_valu_rates.log
  head -c 600 $O/bench_${TAG}_greedy.json; echo
  for f in beam5 cfg5_greedy cfg5_beam5 train train_cfg4; do python3 -c "
import json,sys
d=json.load(open('$O/bench_${TAG}_$f.json')); r=d.get('roofline') or {}; print('$f', d['value'], d['ms_per_step'], r.get('kernel'), r.get('frac'), r.get('traffic'))"; done
fi
ls $O | head -80

# Collects the round's measured evidence into gpurun_out/prof/ (copy the summaries into profiles/ afterwards):
#   bench lines (greedy default, driver-style 20 steps, beam 5, cfg5 greedy, train, encoder), rocprofv3 kernel-trace summaries of the same
#   commands, and the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, --kernel-trace only) that tools/collect_traffic.py
#   turns into profiles/traffic.json.
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f -- python3 $R/bench.py --no-graph --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w -- python3 $R/bench.py --no-graph --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
FDB=$(find $O/pmc_f -name "*.db" | head -1); WDB=$(find $O/pmc_w -name "*.db" | head -1)
python3 $R/tools/collect_traffic.py $FDB $WDB --config cfg2 --beam 1 --out $O/traffic.json --md $O/${TAG}_pmc_traffic.md
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_m -o m -- python3 $R/bench.py --no-graph --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_mfma.log 2>&1
MDB=$(find $O/pmc_m -name "*.db" | head -1)
python3 $R/tools/rocpd_pmc.py $MDB > $O/${TAG}_pmc_mfma_busy.md 2>&1
rm -rf $O/pmc_f $O/pmc_w $O/pmc_m
# the same two passes over the training step (its table averages every dispatch of an entry point's kernel)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_tf -o f -- python3 $R/bench.py --mode train --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_train_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_tw -o w -- python3 $R/bench.py --mode train --config cfg3 --steps 2 --warmup 1 --no-cpu-baseline > $O/pmc_train_write.log 2>&1
FDB=$(find $O/pmc_tf -name "*.db" | head -1); WDB=$(find $O/pmc_tw -name "*.db" | head -1)
python3 $R/tools/collect_traffic.py $FDB $WDB --config cfg3 --beam 1 --mode train --out $O/traffic_train.json --md $O/${TAG}_pmc_traffic_train.md
rm -rf $O/pmc_tf $O/pmc_tw
cp $O/traffic_train.json $R/profiles/traffic_train.json
# the bench lines below read this build's traffic (bench.py refuses a traffic.json whose kernel-source hash is not the build's)
cp $O/traffic.json $R/profiles/traffic.json
python3 $R/bench.py > $O/bench_${TAG}_greedy.json 2> $O/bench_greedy.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_${TAG}_greedy_20steps.json 2>> $O/bench_greedy.err
python3 $R/bench.py --beam 5 --steps 50 --warmup 3 > $O/bench_${TAG}_beam5.json 2> $O/bench_beam.err
python3 $R/bench.py --config cfg5 --steps 50 --warmup 3 --no-cpu-baseline > $O/bench_${TAG}_cfg5_greedy.json 2> $O/bench_cfg5.err
python3 $R/bench.py --config cfg5 --beam 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_${TAG}_cfg5_beam5.json 2>> $O/bench_cfg5.err
python3 $R/bench.py --mode train --config cfg3 --steps 30 --warmup 3 > $O/bench_${TAG}_train.json 2> $O/bench_train.err
python3 $R/bench.py --mode encoder --steps 10 --warmup 2 > $O/bench_${TAG}_encoder.json 2> $O/bench_encoder.err
for what in greedy beam5 train encoder; do
  case $what in
    greedy) ARGS="--steps 20 --warmup 3 --no-cpu-baseline";;
    beam5) ARGS="--beam 5 --steps 6 --warmup 2 --no-cpu-baseline";;
    train) ARGS="--mode train --config cfg3 --steps 8 --warmup 2 --no-cpu-baseline";;
    encoder) ARGS="--mode encoder --steps 3 --warmup 1";;
  esac
  rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py $ARGS > $O/kt_$what.log 2>&1
  DB=$(find $O/kt -name "*.db" | head -1)
  python3 $R/tools/rocpd_summary.py $DB > $O/${TAG}_${what}_kernel_stats.md 2>&1
  rm -rf $O/kt
done
# ---- round 3 additions
# decode-step timeline (per-position duration + idle gap) from a graph-replay kernel trace
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/kt_timeline.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
python3 $R/tools/rocpd_step_timeline.py $DB 7 > $O/${TAG}_greedy_step_timeline.md 2>&1
rm -rf $O/kt
# L2-side counters of the gate GEMM forms (full K / exchange finish / K-split + finishing launch), standalone launches
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_READ_sum -d $O/pmc_l2a -o a -- python3 $R/tools/bench_ksx.py > $O/pmc_l2a.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2b -o b -- python3 $R/tools/bench_ksx.py > $O/pmc_l2b.log 2>&1
{ echo "## TCP_TCC_READ_REQ_sum / TCC_READ_sum (requests per dispatch), tools/bench_ksx.py"; python3 $R/tools/rocpd_pmc.py $(find $O/pmc_l2a -name "*.db" | head -1);
  echo; echo "## TCC_HIT_sum / TCC_MISS_sum"; python3 $R/tools/rocpd_pmc.py $(find $O/pmc_l2b -name "*.db" | head -1); } > $O/${TAG}_pmc_l2_gate_gemm.md 2>&1
rm -rf $O/pmc_l2a $O/pmc_l2b
python3 $R/tools/bench_ksx.py > $O/${TAG}_gate_gemm_forms.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/l2_ingress.hip -o /tmp/l2i && /tmp/l2i > $O/${TAG}_l2_ingress.log 2>&1
# the N-rank entry path at N = 1 (bench.py starts its rank processes itself): decode line and config 4's per-GPU training step
python3 $R/bench.py --gpus 1 --spawn --steps 20 --warmup 5 --no-secondary > $O/bench_${TAG}_spawn1_greedy.json 2> $O/bench_spawn.err
python3 $R/bench.py --gpus 1 --spawn --mode train --config cfg4 --steps 30 --warmup 3 > $O/bench_${TAG}_spawn1_train_cfg4.json 2>> $O/bench_spawn.err
ls -la $O | head -60
head -c 700 $O/bench_${TAG}_greedy.json; echo
for f in beam5 cfg5_greedy cfg5_beam5 train; do python3 -c "
import json,sys
d=json.load(open('$O/bench_${TAG}_$f.json')); print('$f', d['value'], d['ms_per_step'], (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('frac'))"; done
cat $O/${TAG}_pmc_traffic.md

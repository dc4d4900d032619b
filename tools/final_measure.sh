set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fin; mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-upload-pass > $O/bench_under_rocprof.json 2> $O/kt.err || exit 2
B="python3 $R/bench.py --steps 2 --warmup 0 --inflight 1 --no-cpu-baseline --no-upload-pass"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- $B > $O/pmc1.json 2> $O/pmc1.err || exit 3
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc2 -- $B > $O/pmc2.json 2> $O/pmc2.err || exit 4
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc3 -- $B > $O/pmc3.json 2> $O/pmc3.err || exit 5
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- $B > $O/pmc4.json 2> $O/pmc4.err || exit 6
cd $R
python tools/pmc_valu.py ksw_extd2_wave_kernel $O/pmc_valu.json $O/pmc1 $O/pmc2 --bench-log $O/pmc1.json
python tools/pmc_traffic.py ksw_extd2_wave_kernel $O/pmc_traffic.json $O/pmc3 $O/pmc4 --bench-log $O/pmc3.json
find $O -name "*_counter_collection.csv" -size +5M -delete
ls $O

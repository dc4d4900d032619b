# round-3 measurements on the GPU box (two gpurun calls: `bash tools/final_measure3.sh a` and `... b`); results under gpurun_out/fin3
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fin3; mkdir -p $O
export TMPDIR=/tmp
if [ "$1" = a ]; then
cd $R && python bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-upload-pass --no-latency-modes > $O/bench_under_rocprof.json 2> $O/kt.err || exit 2
echo "kernel trace done"
B="python3 $R/bench.py --steps 2 --warmup 0 --inflight 1 --no-cpu-baseline --no-upload-pass --no-latency-modes"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc1 -- $B > $O/pmc1.json 2> $O/pmc1.err || exit 3
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc2 -- $B > $O/pmc2.json 2> $O/pmc2.err || exit 4
echo "pmc valu done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc3 -- $B > $O/pmc3.json 2> $O/pmc3.err || exit 5
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- $B > $O/pmc4.json 2> $O/pmc4.err || exit 6
cd $R
python tools/pmc_valu.py ksw_extd2_wave_kernel $O/pmc_valu.json $O/pmc1 $O/pmc2 --bench-log $O/pmc1.json
python tools/pmc_traffic.py ksw_extd2_wave_kernel $O/pmc_traffic.json $O/pmc3 $O/pmc4 --bench-log $O/pmc3.json
find $O -name "*_counter_collection.csv" -size +5M -delete
elif [ "$1" = b ]; then
cd $R
python tools/bench_variant.py --kind sr --ref-mbp 3088 --steps 40 --inflight 4 > $O/sr.json 2> $O/sr.err || exit 1
python tools/bench_variant.py --kind ont --ref-mbp 3088 --steps 9 --inflight 3 > $O/ont.json 2> $O/ont.err || exit 2
echo "variants done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sr -- python3 $R/tools/bench_variant.py --kind sr --ref-mbp 3088 --steps 20 --inflight 4 > $O/sr_rocprof.json 2> $O/kt_sr.err || exit 3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ont -- python3 $R/tools/bench_variant.py --kind ont --ref-mbp 3088 --steps 6 --inflight 3 > $O/ont_rocprof.json 2> $O/kt_ont.err || exit 4
echo "variant traces done"
cd $R
python tools/synth_fastq.py /tmp/hf 3088 163840 hifi noref && python tools/map_file.py --preset hifi synth:3088 /tmp/hf/reads.fq -o /dev/null -K 76789488 --inflight 3 --reader-threads 2 > $O/file_hifi.json 2> $O/file_hifi.err || exit 5
rm -f /tmp/hf/reads.fq
python tools/synth_fastq.py /tmp/mf 3088 8000000 sr noref && python tools/map_file.py --preset sr synth:3088 /tmp/mf/reads.fq -o /dev/null --inflight 4 --reader-threads 4 > $O/file_sr.json 2> $O/file_sr.err || exit 6
rm -f /tmp/mf/reads.fq
python tools/bench_extz2.py > $O/extz2.json 2> $O/extz2.err
find $O -name "*_kernel_trace.csv" -size +20M -delete
fi
ls $O
# part c (after the skewed-pipeline DP kernel for short reads): `bash tools/final_measure3.sh c`
if [ "$1" = c ]; then
cd $R
python tools/bench_variant.py --kind sr --ref-mbp 3088 --steps 64 --inflight 8 > $O/sr8.json 2> $O/sr8.err || exit 1
python tools/bench_variant.py --kind sr --ref-mbp 3088 --steps 48 --inflight 4 > $O/sr4.json 2>> $O/sr8.err || exit 1
GDIET_SR_PIPE=0 python tools/bench_variant.py --kind sr --ref-mbp 3088 --steps 48 --inflight 8 > $O/sr8_nopipe.json 2>> $O/sr8.err || exit 1
python tools/bench_extz2.py > $O/extz2_c.json 2> $O/extz2_c.err || exit 2
python tools/bench_extz2.py 400000 d > $O/extd2_400k.json 2>> $O/extz2_c.err || exit 2
GDIET_SR_PIPE=0 python tools/bench_extz2.py 400000 d > $O/extd2_400k_nopipe.json 2>> $O/extz2_c.err || exit 2
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_sr8 -- python3 $R/tools/bench_variant.py --kind sr --ref-mbp 3088 --steps 32 --inflight 8 > $O/sr8_rocprof.json 2> $O/kt_sr8.err || exit 3
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmcp1 -- python3 $R/tools/bench_extz2.py 400000 d > $O/pmcp1.log 2>&1 || exit 4
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmcp2 -- python3 $R/tools/bench_extz2.py 400000 d > $O/pmcp2.log 2>&1 || exit 4
cd $R
python tools/pmc_valu.py ksw_extd2_pipe_kernel $O/pmc_pipe.json $O/pmcp1 $O/pmcp2 > /dev/null
python tools/synth_fastq.py /tmp/mf 3088 8000000 sr noref && python tools/map_file.py --preset sr synth:3088 /tmp/mf/reads.fq -o /dev/null --inflight 8 --reader-threads 4 > $O/file_sr8.json 2> $O/file_sr8.err
rm -f /tmp/mf/reads.fq
find $O -name "*_kernel_trace.csv" -size +20M -delete
find $O -name "*_counter_collection.csv" -size +5M -delete
ls $O
fi

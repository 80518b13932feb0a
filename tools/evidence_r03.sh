# Round-3 evidence: the bench line, its rocprofv3 kernel statistics and three counter passes.  Run on the GPU box:
#   gpurun -- 'bash tools/evidence_r03.sh'     then copy gpurun_out/r03_* into profiles/.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py --cpu-baseline 0 --traffic off > gpurun_out/r03_bench_under_rocprof.json 2> gpurun_out/r03_bench_under_rocprof.err
ARGS="--cpu-baseline 0 --graph 0 --steps 3 --warmup 1 --profile-steps 1 --traffic off"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/pmc_fetch.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py $ARGS > gpurun_out/pmc_write.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py $ARGS > gpurun_out/pmc_mfma.txt 2>&1
STATS=$(find gpurun_out/prof_final -name "*kernel_stats.csv" | head -1)
cp $STATS gpurun_out/r03_bench_kernel_stats.csv
# the same with every vocabulary launch on the whole chip: the launch durations the roofline is priced on
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dw0 -o bench -- python3 bench.py --cpu-baseline 0 --traffic off --dw-beside 0 > gpurun_out/r03_bench_dw0_under_rocprof.json 2> gpurun_out/r03_bench_dw0_under_rocprof.err
cp $(find gpurun_out/prof_dw0 -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_kernel_stats_dw0.csv
rm -rf gpurun_out/prof_dw0
python tools/pmc_report.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma $STATS gpurun_out/r03_pmc_traffic.json > /dev/null
TRACE=$(find gpurun_out/prof_final -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $TRACE > gpurun_out/r03_step_timeline.txt
rm -rf gpurun_out/prof_final gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma
for c in movie book mutual; do timeout -k 10 300 python bench.py --config $c --cpu-baseline 0 > gpurun_out/r03_bench_$c.json 2> gpurun_out/r03_bench_$c.err; done
timeout -k 10 300 python bench.py --cpu-baseline 0 --traffic off --rehearse-dp 1 > gpurun_out/r03_bench_dp_rehearsal.json 2> gpurun_out/r03_bench_dp_rehearsal.err
timeout -k 10 300 python tools/bench_engine.py > gpurun_out/r03_bench_engine.txt 2>&1
PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip_stamps.so timeout -k 10 200 python tools/vocab_stamps.py > gpurun_out/r03_vocab_stamps.txt 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.txt 2>&1
tail -c 400 gpurun_out/r03_bench.json; tail -1 gpurun_out/r03_smoke.txt

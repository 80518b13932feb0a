set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py > gpurun_out/r01_bench.json.log 2>gpurun_out/r01_bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- python3 bench.py --cpu-baseline 0 > gpurun_out/r01_bench_under_rocprof.json.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --cpu-baseline 0 --graph 0 --steps 3 --warmup 1 --profile-steps 1 > gpurun_out/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --cpu-baseline 0 --graph 0 --steps 3 --warmup 1 --profile-steps 1 > gpurun_out/pmc_write.log 2>&1
tail -1 gpurun_out/r01_bench.json.log | cut -c1-200
timeout -k 10 300 python bench.py --cpu-baseline 0 --latent gauss > gpurun_out/r01_bench_gauss.json.log 2>&1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1
tail -1 gpurun_out/r01_bench_gauss.json.log | cut -c1-200
tail -1 gpurun_out/smoke.log

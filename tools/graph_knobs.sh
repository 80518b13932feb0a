# Experiment: the HIP graph executor's knobs against the step time (one bench line each).  gpurun -- 'bash tools/graph_knobs.sh'
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-baseline 0 --traffic off --profile-steps 1"
o=gpurun_out/graph_knobs.txt; : > $o
run() { echo "== $1" >> $o; env $1 timeout -k 10 200 $B 2>> gpurun_out/graph_knobs.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> $o; }
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run PEVAE_EAGER_FLUSH=0
mkdir -p gpurun_out/dot && cd gpurun_out/dot && DEBUG_HIP_GRAPH_DOT_PRINT=1 timeout -k 10 200 python ../../bench.py --cpu-baseline 0 --traffic off --profile-steps 1 --steps 2 --warmup 1 > ../dot_run.txt 2>&1; ls -la >> ../graph_knobs.txt; cd ../..
cat $o

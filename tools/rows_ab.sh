# A/B of the Block kernels: 16-row slabs (PEVAE_BLOCK_ROWS=16) against 4-row slabs, one bench line each
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-baseline 0 --traffic off --profile-steps 1"
o=gpurun_out/rows_ab.txt; : > $o
run() { echo "== $1" >> $o; env $1 timeout -k 10 200 $B 2>> gpurun_out/rows_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_vae_nll'])" >> $o; }
run PEVAE_BLOCK_ROWS=16
run PEVAE_BLOCK_ROWS=4
run PEVAE_BLOCK_ROWS_WIDE=1
cat $o

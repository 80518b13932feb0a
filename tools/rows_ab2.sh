# A/B of two builds of the library (PEVAE_HIP_LIB), one bench line each, twice
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-baseline 0 --traffic off --profile-steps 1"
o=gpurun_out/rows_ab2.txt; : > $o
run() { echo "== $1" >> $o; env $1 timeout -k 10 200 $B 2>> gpurun_out/rows_ab2.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_vae_nll'])" >> $o; }
for i in 1 2; do
run PEVAE_HIP_LIB=pevae_amd/lib/libpevae_hip.so
run PEVAE_HIP_LIB=$1
done
cat $o

# A/B: decoder input hoisted beside the latent stack, zero_grad fill beside the head of the forward pass
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --cpu-baseline 0 --traffic off --profile-steps 1"
o=gpurun_out/hoist_ab.txt; : > $o
run() { echo "== $1 $2" >> $o; env $1 $2 timeout -k 10 200 $B 2>> gpurun_out/hoist_ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['final_vae_nll'])" >> $o; }
for i in 1 2; do
run PEVAE_CONS_BESIDE=0 A=1
run PEVAE_CONS_BESIDE=1 A=1
done
cat $o

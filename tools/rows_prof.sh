# per-kernel durations of the Block kernels inside the bench step (rocprofv3 kernel trace), for the env given as $1
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export $1
rm -rf gpurun_out/prof_rows
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_rows -o bench -- python3 bench.py --cpu-baseline 0 --traffic off --profile-steps 1 --steps 50 > gpurun_out/rows_prof.json 2> gpurun_out/rows_prof.err
python - <<'P'
import csv,glob
f=glob.glob('gpurun_out/prof_rows/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'block' in n or 'linear_f32_group' in n or 'dls_level' in n:
        print("%-70s calls %5s avg %8.1f us min %8.1f max %8.1f"%(n.replace('(anonymous namespace)::','')[:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
P
python tools/step_timeline.py $(find gpurun_out/prof_rows -name "*kernel_trace.csv" | head -1) > gpurun_out/rows_timeline.txt
rm -rf gpurun_out/prof_rows

cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-legs --no-latency --no-farm --parity-jobs 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/rocpd_summary.py /tmp/prof_b/b_results.db 2>&1 | head -24 | cut -c1-170 > gpurun_out/r03_run20_bench_kernels.txt
python tools/rocpd_timeline.py /tmp/prof_b/b_results.db 2>&1 > gpurun_out/r03_run20_timeline_full.txt
python - <<'PY'
import re
rows=[l.split(None,4) for l in open('gpurun_out/r03_run20_timeline_full.txt') if not l.startswith('#')]
# find the last occurrence of the loop kernel (non-PROF) and print the 60 dispatches before it
idx=[i for i,r in enumerate(rows) if 'icp_sweep_kernel<1024, 8, true, true, false, true' in r[4]]
end=idx[2] if len(idx)>2 else idx[-1]
start=idx[1]+1 if len(idx)>2 else max(0,end-60)
t0=float(rows[start][0])
out=[]
for r in rows[start:end+1]:
    out.append("%9.1f %9.1f q%s %s"%(float(r[0])-t0,float(r[1])-t0,r[2],r[4].strip()[:70]))
open('gpurun_out/r03_run20_timeline.txt','w').write("# one timed step: start_us end_us queue kernel (relative to the step's first dispatch)\n"+"\n".join(out)+"\n")
print("\n".join(out))
PY

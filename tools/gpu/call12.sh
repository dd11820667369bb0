cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 120 rocprofv3 --kernel-trace --pmc $c -d /tmp/pe_$c -o e -- python $GRAFT_REPO_ROOT/tools/extract_times.py 256 > /tmp/pe_$c.log 2>&1
echo "== $c rc=$?"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py /tmp/pe_$c/e_results.db 2>&1 | grep -E "extract_|counter|avg_us" | cut -c1-160
done > $GRAFT_REPO_ROOT/gpurun_out/r03_extract_pmc.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r03_extract_pmc.txt; tail -2 /tmp/pe_FETCH_SIZE.log

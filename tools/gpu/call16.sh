cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in 5151 6262; do timeout 600 python tools/oracle_soak.py --big 2000 --small 8000 --seed $s > gpurun_out/r03_oracle_soak_seed$s.json 2> gpurun_out/r03_oracle_soak_seed$s.err; echo "rc=$?"; cat gpurun_out/r03_oracle_soak_seed$s.json; tail -3 gpurun_out/r03_oracle_soak_seed$s.err; done
timeout 600 python tools/icp_soak.py > gpurun_out/r03_icp_soak.txt 2>&1; tail -5 gpurun_out/r03_icp_soak.txt

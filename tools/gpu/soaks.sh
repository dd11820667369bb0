#!/bin/bash
# end-of-round soaks on the final kernels: oracle (2 seeds), pipeline, extraction (both scan paths), sweep-vs-brute-force
tag=${1:-r05}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 1717 2828; do
  timeout 600 python tools/oracle_soak.py --big 2000 --small 8000 --seed $seed > gpurun_out/${tag}_oracle_soak_seed$seed.json 2> gpurun_out/oracle_soak_$seed.err
  echo "oracle soak seed $seed rc=$?"; python -c "import json;d=json.load(open('gpurun_out/${tag}_oracle_soak_seed$seed.json'));print({k:d[k] for k in ('scan_matches','status_mismatch','iteration_mismatch','pose_mismatch_f64','ill_conditioned','max_pose_diff_f64_p2p','max_pose_diff_f64_p2plane')})"
done
{ timeout 300 python tools/pipeline_soak.py --seconds 90 2>&1 | tail -2
  timeout 200 python tools/extract_soak.py --seconds 60 2>&1 | tail -2
  timeout 300 python tools/icp_soak.py --seconds 90 2>&1 | tail -2
  timeout 200 python tools/default_flow_soak.py --seconds 60 2>&1 | tail -2; } > gpurun_out/${tag}_soaks.txt 2>&1
cat gpurun_out/${tag}_soaks.txt

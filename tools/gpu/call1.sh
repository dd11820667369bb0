cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 60 pip download opencv-python-headless -d /tmp/w 2>&1 | tail -8; echo "---"; timeout 60 pip install --user opencv-python-headless 2>&1 | tail -5; echo "--- import:"; python -c "import cv2; print(cv2.__version__)" 2>&1 | tail -1) > gpurun_out/r03_pip_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_icp.py -q -m gpu --timeout 300 -x 2>&1 | tail -40 > gpurun_out/r03_t_icp.txt
timeout 600 python -m pytest tests/test_gpu_configs.py -q -m gpu --timeout 400 2>&1 | tail -40 > gpurun_out/r03_t_cfg.txt
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err
timeout 300 python tools/stage_times.py --batch 512 --icp-variants 0 > gpurun_out/r03_stage1.txt 2>&1
tail -3 gpurun_out/r03_t_icp.txt gpurun_out/r03_t_cfg.txt; tail -c 600 gpurun_out/r03_bench1.err; head -c 300 gpurun_out/r03_bench1.json

cd $GRAFT_REPO_ROOT
for c in 1024 2048 4096 512; do echo "chunk=$c"; SFE_EXTRACT_CHUNK=$c python tools/extract_times.py 4096 2>&1 | tail -1; done
for c in 1024 4096; do echo "bench chunk=$c"; SFE_EXTRACT_CHUNK=$c python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-legs --no-latency --no-farm --parity-jobs 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"; done

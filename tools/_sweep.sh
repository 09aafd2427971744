cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_train_graph.py -x -q 2>&1 | tail -5
for c in 32 96 256; do for b in 8 32; do CBD_DW_MAX_CHUNKS=$c python tools/train_bench.py --batch $b 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chunks $c batch $b', d['ms_per_step'])"; done; done

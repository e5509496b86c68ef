cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -x -q -k small_calls 2>&1 | tail -12

set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_persist_robustness_gpu.py tests/test_component.py tests/test_nnet.py -m gpu -x -q 2>&1 | tail -8

cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "predict_many or note_events" 2>&1 | tail -5

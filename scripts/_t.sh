timeout 900 python -m pytest tests/test_gpu_scan.py -x -q 2>&1 | tail -1
python scripts/delay12_time.py 2>&1 | grep "LQG_SCAN=default"
python scripts/small_batch.py 2>/dev/null | grep -v amdgpu | head -30

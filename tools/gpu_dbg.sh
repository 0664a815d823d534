cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_gpu_baseline_geometry.py -x -q -m gpu -s -k "11b_shipped_shape" 2>&1 | tail -8

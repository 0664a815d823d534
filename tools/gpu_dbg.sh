cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_mmdit.py tests/test_gpu_fp8.py -x -q -m gpu -k "hipgraph or fp8 or auto" 2>&1 | tail -6
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "auto_bound or rownorm" 2>&1 | tail -3

cd "$(dirname "$0")/.."
timeout 300 python tools/_dbg_gelu.py 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_overlap.py tests/test_gpu_seqpar_1gpu.py tests/test_gpu_stdit_shapes.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm" 2>&1 | tail -3

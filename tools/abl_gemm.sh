python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "planes" 2>&1 | tail -5
ALIVE_GEMM_VARIANT=1 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "planes" 2>&1 | tail -3
for v in 0 1; do echo "== variant $v"; ALIVE_GEMM_VARIANT=$v python tools/stamp_gemm.py; ALIVE_GEMM_VARIANT=$v python tools/bench_gemm_planes.py | grep "gelu\|pw2\|pe out" ; done

set -u
export STEP_TIMEOUT=900
tools/gpu_steps.sh r3c3 \
  "python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_config4.py -m gpu -q -x" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=19552 python tools/sweep.py --windows 1048576,131072,4194304 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_BUCKET_ORDER=1 SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=19552 python tools/sweep.py --windows 1048576,131072,4194304 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=13024 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=8192 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 8 --iters 10" \
  "SPMV_BUCKET_ORDER=1 SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=8192 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 8 --iters 10" \
  "python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; tail -c 300 gpurun_out/r3_bench2.err"

set -u
export STEP_TIMEOUT=900
tools/gpu_steps.sh r3c1 \
  "python -m pytest tests -m gpu -q -x --durations=8" \
  "python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; tail -c 600 gpurun_out/r3_bench1.err" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=20448 python tools/sweep.py --windows 1048576 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=10208 SPMV_LDS_MIN=163584 python tools/sweep.py --windows 1048576 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=10208 python tools/sweep.py --windows 1048576 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=13632 SPMV_LDS_MIN=163584 python tools/sweep.py --windows 1048576 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=16384 SPMV_LDS_MIN=163584 python tools/sweep.py --windows 1048576 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10"

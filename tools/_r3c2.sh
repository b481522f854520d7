set -u
export STEP_TIMEOUT=900
RO=536870912
tools/gpu_steps.sh r3c2 \
  "python -m pytest tests -m gpu -q --durations=8" \
  "python tools/sweep.py --rows 1000000 --k 16 --family banded --windows 0 --hll-kernels 1 --csr-kernels 2,4 --waves 4,8 --variants 0,$RO --flush 536870912 --iters 30" \
  "python tools/sweep.py --rows 1000000 --k 16 --family banded --windows 0 --hll-kernels 1 --csr-kernels 2,4 --waves 4,8 --variants 0 --flush 0 --iters 30" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=19552 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=13024 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=9792 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10" \
  "SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=9792 SPMV_LDS_MIN=163584 python tools/sweep.py --windows 1048576,131072 --hll-kernels 4 --csr-kernels '' --waves 16,8 --iters 10"

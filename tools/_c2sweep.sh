for t in 512 1024 2048 4096; do
  echo "== chain tile $t"
  SPMV_PANEL_SCHED=chain SPMV_TILE_ROWS=$t python tools/sweep.py --rows 1000000 --k 16 --family banded --windows 0 --hll-kernels "" --csr-kernels 5 --waves 4,8 --variants 0 --iters 30 --flush 536870912 | grep "ms  "
done
echo "== hll direct"
python tools/sweep.py --rows 1000000 --k 16 --family banded --windows 0 --hll-kernels 1,2 --csr-kernels 2,4 --waves 4,8 --variants 0 --iters 30 --flush 536870912 | grep "ms  "

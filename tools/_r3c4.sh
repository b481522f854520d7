set -u
export STEP_TIMEOUT=900
tools/gpu_steps.sh r3c4 \
  "python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x" \
  "python tools/sweep.py --rows 1000000 --k 16 --family banded --windows 0 --hll-kernels 1 --csr-kernels 4 --waves 4 --variants 256,128,288,160,536871168,536871040 --flush 536870912 --iters 30" \
  "python tools/sweep.py --rows 1000000 --k 16 --family banded --windows 0 --hll-kernels 1 --csr-kernels 4 --waves 4 --variants 256,128 --flush 1073741824 --iters 30" \
  "python tools/sweep.py --rows 1000000 --k 32 --family random --windows 2048,65536 --hll-kernels 1 --csr-kernels 4 --waves 4 --variants 256,128 --flush 536870912 --iters 20" \
  "python tools/sweep.py --rows 1500000 --k 27 --family stencil --windows 0 --hll-kernels 1 --csr-kernels 4 --waves 4 --variants 256,128 --flush 536870912 --iters 20" \
  "python bench.py --config 2 --steps 30 --no-cpu-baseline > gpurun_out/r3_bench_c2.json 2> gpurun_out/r3_bench_c2.err; tail -c 300 gpurun_out/r3_bench_c2.err; cat gpurun_out/r3_bench_c2.json"

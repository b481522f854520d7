import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spmv_scpa_amd as S
p = os.path.join(tempfile.gettempdir(), "spmv_kkt160.mtx")
if not os.path.exists(p):
    subprocess.run([os.path.join(S.ROOT, "spmv_scpa_amd", "bin", "gen_kkt_mtx"), "160", p], check=True, capture_output=True)
A = S.io_load_csr_cached(p)
dA = S.CsrDevice.upload(A)
dH = dA.to_hll(True)
N, M = A.contents.N, A.contents.M
d_x, d_y = S.DevBuffer(N * 8), S.DevBuffer(M * 8)
S.dev_fill_synth(d_x.ptr, N, 7)
for k in (1, 2):
    for v in (0, 1):
        ms = float(np.median(dH.time(k, d_x.ptr, d_y.ptr, warmup=2, iters=10, variant=v)))
        print("kkt160 hll k%d variant %d (0 = balanced XCD ranges, 1 = hardware order): %.4f ms  %.1f%%" % (k, v, ms, 100 * dH.algorithmic_bytes / (ms * 1e6) / 8000), flush=True)

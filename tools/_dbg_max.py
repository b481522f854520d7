import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spmv_scpa_amd as S
for M in (67_108_832, 67_108_863):
    dA = S.CsrDevice.generate(S.SYNTH_RANDOM, M, M, 32, 1 << 20, 0, 42)
    d_x, d_y = S.DevBuffer(M * 8), S.DevBuffer(M * 8)
    print(M, dA.NZ, hex(d_x.ptr), hex(d_y.ptr), flush=True)
    S.dev_fill_synth(d_x.ptr, M, 7)
    for k in (4, 2, 0, 1, 3):
        o = S._opts(0, 0, 0)
        rc = S._lib.spmv_csr_launch(dA.h, k, __import__("ctypes").byref(o), d_x.ptr, d_y.ptr, None)
        S.stream_sync()
        print("  kernel", k, "rc", rc, flush=True)
    dA.release(); d_x.free(); d_y.free()

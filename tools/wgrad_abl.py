import os, sys, ctypes, torch
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
B = 8192
shapes = [(16, 32, 32, 9), (8, 64, 64, 9), (4, 128, 128, 9), (16, 16, 32, 1)]
for n in sys.argv[1:]:
    L = ctypes.CDLL(os.path.join(root, "contextflow_amd/build/abl/libwg%s.so" % n))
    L.cf_wgrad_ws_bytes.restype = ctypes.c_int64
    vp = ctypes.c_void_p
    L.cf_wgrad.argtypes = [vp, vp, vp, vp, vp] + [ctypes.c_int] * 6 + [vp]
    for (H, MR, NR, taps) in shapes:
        A = torch.randn(B, MR, H * H, device="cuda"); Bm = torch.randn(B, NR, H * H, device="cuda")
        gw = torch.empty(taps, MR, NR, device="cuda"); gb = torch.empty(MR, device="cuda")
        ws = torch.empty(L.cf_wgrad_ws_bytes(B, MR, NR, H, H, taps), device="cuda", dtype=torch.uint8)
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: L.cf_wgrad(A.data_ptr(), Bm.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), B, MR, NR, H, H, taps, st)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        print("variant %s H=%2d MR=%3d NR=%3d taps=%d: %7.1f us" % (n, H, MR, NR, taps, e0.elapsed_time(e1) * 100))

import os, sys, ctypes, torch
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
B = 8192
L = ctypes.CDLL(os.path.join(root, "contextflow_amd/build/abl/libwg5.so"))
L.cf_wgrad_ws_bytes.restype = ctypes.c_int64
vp = ctypes.c_void_p
L.cf_wgrad.argtypes = [vp, vp, vp, vp, vp] + [ctypes.c_int] * 6 + [vp]
names = ["barrier1", "lds_write", "barrier2", "gload_issue", "mfma_loop", "backedge"]
for (H, MR, NR, taps) in [(16, 32, 32, 9), (8, 64, 64, 9), (4, 128, 128, 9), (16, 16, 32, 1)]:
    A = torch.randn(B, MR, H * H, device="cuda"); Bm = torch.randn(B, NR, H * H, device="cuda")
    gw = torch.empty(taps, MR, NR, device="cuda"); gb = torch.empty(MR, device="cuda")
    nb = L.cf_wgrad_ws_bytes(B, MR, NR, H, H, taps)
    ws = torch.zeros(nb + 4096, device="cuda", dtype=torch.uint8)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.cf_wgrad(A.data_ptr(), Bm.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), B, MR, NR, H, H, taps, st)
    torch.cuda.synchronize()
    f = ws.view(torch.float32).cpu()
    base = nb // 4 + 64
    KC = max(H * H, 64); mt = (MR + 31) // 32; nch = B // (KC // (H * H)); S = min(512 // mt, nch); per = nch / S
    for w in range(4):
        t = f[base + w * 8: base + w * 8 + 6].tolist()
        print("H=%d MR=%d taps=%d wave %d (chunks/WG %.0f): " % (H, MR, taps, w, per) + "  ".join("%s %.0f" % (n, v / per) for n, v in zip(names, t)) + "  | total/chunk %.0f" % (sum(t) / per))

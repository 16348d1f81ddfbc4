#!/usr/bin/env python3
"""Per-phase cycles of k_flow_step_bwd (workgroup 0) from the probe build of tools/dev/make_bwd_ticks.py."""
import ctypes, os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = ctypes.CDLL(os.path.join(root, "contextflow_amd/build/abl/libcf_ticks.so"))
names = open(os.path.join(root, "contextflow_amd/build/abl/ticks_names.txt")).read().split()
L.cf_flow_step_ws_bytes.restype = ctypes.c_int64
L.cf_flow_step_bwd_ws_bytes.restype = ctypes.c_int64
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = "cuda"
P = lambda t: ctypes.c_void_p(t.data_ptr())
I = ctypes.c_int
for (C, H) in [(16, 16), (32, 8), (64, 4)]:
    HID, HALF, HW = 2 * C, C // 2, H * H
    g = torch.Generator().manual_seed(C)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
    Wm = (torch.linalg.qr(torch.randn(C, C, generator=g))[0]).contiguous().to(dev)
    t, logs = r(C), r(C)
    w1, b1, w2, b2, w3, b3 = r(HID, HALF), r(HID), r(HID, HID, 3, 3), r(HID), r(C, HID), r(C)
    ws = torch.empty(L.cf_flow_step_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    wsb = torch.empty(L.cf_flow_step_bwd_ws_bytes(C, H, H), device=dev, dtype=torch.uint8)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert L.cf_flow_step_prepare(P(Wm), P(t), P(logs), P(w1), P(b1), P(w2), P(b2), P(w3), P(b3), P(ws), I(C), I(H), I(H), st) == 0
    assert L.cf_flow_step_bwd_prepare(P(Wm), P(logs), P(w1), P(w2), P(w3), P(wsb), I(C), I(H), I(H), st) == 0
    x, gz, gld = r(B, C, H, H), r(B, C, H, H), r(B)
    L.cf_flow_step_tape_aux_bytes.restype = ctypes.c_int64
    new = lambda rows: torch.empty(B, rows, HW, device=dev)
    gx = torch.empty(B, C, H, H, device=dev)
    z, ld = torch.empty_like(x), torch.zeros(B, device=dev)
    y0, h1, h2 = new(HALF), new(HID), new(HID)
    aux = torch.empty(L.cf_flow_step_tape_aux_bytes(I(B), I(C), I(H), I(H)), device=dev, dtype=torch.uint8)
    assert L.cf_flow_step_fwd_taped(P(x), P(z), P(ld), P(ws), P(y0), P(h1), P(h2), P(aux), I(B), I(C), I(H), I(H),
                                    ctypes.c_int64(C * HW), I(0), st) == 0
    bufs = [new(C), new(HID), new(HID), new(C)]
    run = lambda: L.cf_flow_step_bwd_taped(P(gz), P(gld), P(wsb), P(aux), P(gx), *[P(b) for b in bufs], I(B), I(C), I(H), I(H), I(0), st)
    for _ in range(2):
        assert run() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    tk = gx.flatten()[:48].cpu().view(4, 12)[:, :len(names)]
    tot = tk.sum(1)
    print("C=%d %dx%d B=%d: %.1f us/launch; workgroup 0 total %.0f cycles" % (C, H, H, B, e0.elapsed_time(e1) * 200, tot[0]))
    for w in (0, 3):
        print("   wave %d: " % w + "  ".join("%s %.0f (%.0f%%)" % (n, v, 100 * v / tot[w]) for n, v in zip(names, tk[w].tolist())))

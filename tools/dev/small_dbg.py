import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import contextflow_amd as cfa
from contextflow_amd.layers import _hip
L = cfa.layers
lib = _hip.lib()
fn = lib.cf_flow_step_fwd_debug
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
dev = "cuda:0"
C, H, W, B = int(sys.argv[1]), 16, 16, 3
torch.manual_seed(0)
conv, act, cpl = L.Conv1x1((C, H, W)).to(dev), L.ActNorm((C, H, W)).to(dev), L.Coupling(C, (3, 3), (1, 1)).to(dev)
x = torch.randn(B, C, H, W, device=dev)
ws = torch.empty(lib.cf_flow_step_ws_bytes(C, H, W), device=dev, dtype=torch.uint8)
f, pp = _hip.f32, _hip.p
c1, c2, c3 = cpl.NN[0], cpl.NN[2], cpl.NN[4]
_hip.call("cf_flow_step_prepare", pp(f(conv.NN.detach())), pp(f(act.NN_t.detach())), pp(f(act.NN_logs.detach())),
          pp(f(c1.weight.detach())), pp(f(c1.bias.detach())), pp(f(c2.weight.detach())), pp(f(c2.bias.detach())),
          pp(f(c3.weight.detach())), pp(f(c3.bias.detach())), pp(ws), C, H, W, _hip.stream())
out = []
for flags in (0, 3 << 16):
    z = torch.zeros_like(x); ldj = torch.zeros(B, device=dev)
    _hip.check(fn(pp(x), pp(z), pp(ldj), pp(ws), B, C, H, W, C * H * W, 0, None, flags, _hip.stream()))
    torch.cuda.synchronize()
    out.append((z.cpu(), ldj.cpu()))
dz = (out[0][0] - out[1][0]).abs()
print("ldj", out[0][1], out[1][1])
print("per-channel max err", dz.amax((0, 2, 3)))
print("per-row max err (sample 0, worst channel)", dz[0].amax(0).amax(1))
print("per-col max err", dz[0].amax(0).amax(0))

# ---- verify the packed operands of the 16x16x4 phases (C = 8)
if C == 8:
    import numpy as np
    wsf = ws.view(torch.float32).cpu().numpy()
    HALF, HID = 4, 16
    OFF_SB0, OFF_SB3, OFF_SA0, OFF_SA3, OFF_SA1, OFF_SA2 = 5764, 5780, 5796, 6052, 6308, 6564
    w1 = c1.weight.detach().cpu().reshape(HID, HALF).numpy(); w2 = c2.weight.detach().cpu().numpy().reshape(-1)
    w3 = c3.weight.detach().cpu().reshape(C, HID).numpy()
    def chan(p):
        idx = 2 * (p >> 2) + (p & 1)
        return ((HALF if (p & 2) else 0) + idx) if idx < HALF else -1
    SA1 = np.zeros(256, np.float32); SA2 = np.zeros(9 * 256, np.float32); SA3 = np.zeros(256, np.float32)
    for e in range(256):
        jj, ln, gg = e & 3, (e >> 2) & 63, e >> 8
        row, k = ln & 15, 4 * (4 * gg + jj) + (ln >> 4)
        SA1[e] = w1[row, k] if k < HALF else 0
        ch = chan(ln & 15)
        SA3[e] = w3[ch, k] if (ch >= 0 and k < HID) else 0
    for e in range(9 * 256):
        jj, ln, tap = e & 3, (e >> 2) & 63, e >> 8
        SA2[e] = w2[((ln & 15) * HID + 4 * jj + (ln >> 4)) * 9 + tap]
    print("SA1 err", np.abs(wsf[OFF_SA1:OFF_SA1 + 256] - SA1).max(), "SA2 err", np.abs(wsf[OFF_SA2:OFF_SA2 + 2304] - SA2).max(),
          "SA3 err", np.abs(wsf[OFF_SA3:OFF_SA3 + 256] - SA3).max(), "ws floats", wsf.size)
    print("SB3", wsf[OFF_SB3:OFF_SB3 + 16], c3.bias.detach().cpu().numpy())

"""TEST INFRASTRUCTURE (like the rest of oracle/): CPU restatement of the Winograd F(2x2,3x3) form in which the HIP step
kernels evaluate the reflect-padded 3x3 convolution of the coupling nets (contextflow/layers/coupling.py:27; kernel side:
contextflow_amd/csrc/cf_step_common.h: winograd_phase2).  fp32 arithmetic in the kernel's order of operations: weights
transformed in fp64 and rounded once (k_step_pack), input transform B^T d B by additions, per-position channel contraction
with fp32 accumulation, output transform A^T M A by additions.  Pinned by tests/test_oracle_golden.py against the direct
convolution and, through the flow oracle, against the reference's end-to-end fixtures."""
import torch
import torch.nn.functional as F

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd3x3_reflect(h, w, b):
    """h (B, Ci, H, W) fp32, w (Co, Ci, 3, 3), b (Co,) -> conv2d(reflect_pad(h, 1), w) + b as (B, Co, H, W); H, W even."""
    B, Ci, H, W = h.shape
    U = torch.einsum("xa,oiab,yb->xyoi", G, w.double(), G).float()                 # (4, 4, Co, Ci), rounded once
    d = F.pad(h, (1, 1, 1, 1), mode="reflect").unfold(2, 4, 2).unfold(3, 4, 2)     # (B, Ci, H/2, W/2, 4, 4) input patches
    V = torch.einsum("xa,ncijab->ncijxb", BT, d)
    V = torch.einsum("ncijxb,yb->ncijxy", V, BT)
    M = torch.einsum("xyoc,ncijxy->noijxy", U, V)
    Y = torch.einsum("px,noijxy->noijpy", AT, M)
    Y = torch.einsum("noijpy,qy->noijpq", Y, AT)                                   # (B, Co, H/2, W/2, 2, 2)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, H, W) + b.view(1, -1, 1, 1)


# F(4x4, 3x3) (Lavin & Gray): 36 multiplications per 16 outputs (2.25 / output against 4 for F(2x2, 3x3) and 9 direct), but
# transform entries up to 8 (output) / 5 (input) instead of +-1: restated here ONLY to measure what fp32 would cost in bits/dim
# (tests/dev_winograd_numerics.py f4) before anyone writes a kernel for it - see DESIGN.md section 8.
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                  dtype=torch.float64)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                    [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)


def winograd3x3_reflect_f4(h, w, b):
    """The same convolution in the F(4x4, 3x3) form, fp32 (weights transformed in fp64, rounded once); H, W multiples of 4."""
    B, Ci, H, W = h.shape
    U = torch.einsum("xa,oiab,yb->xyoi", G4, w.double(), G4).float()               # (6, 6, Co, Ci)
    d = F.pad(h, (1, 1, 1, 1), mode="reflect").unfold(2, 6, 4).unfold(3, 6, 4)     # (B, Ci, H/4, W/4, 6, 6)
    V = torch.einsum("xa,ncijab->ncijxb", BT4, d)
    V = torch.einsum("ncijxb,yb->ncijxy", V, BT4)
    M = torch.einsum("xyoc,ncijxy->noijxy", U, V)
    Y = torch.einsum("px,noijxy->noijpy", AT4, M)
    Y = torch.einsum("noijpy,qy->noijpq", Y, AT4)                                  # (B, Co, H/4, W/4, 4, 4)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, H, W) + b.view(1, -1, 1, 1)


def coupling_net_winograd_f4(x0, p, prefix, pad, min_hw=8):
    """coupling_net with the 3x3 in the F(4x4, 3x3) form on images of at least min_hw x min_hw (else F(2x2, 3x3))."""
    h = F.relu(F.conv2d(x0, p[prefix + "NN.0.weight"], p[prefix + "NN.0.bias"]))
    if h.dtype == torch.float32 and tuple(pad) == (1, 1) and h.shape[2] % 4 == 0 and h.shape[3] % 4 == 0 and h.shape[2] >= min_hw:
        h = F.relu(winograd3x3_reflect_f4(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    elif h.dtype == torch.float32 and tuple(pad) == (1, 1) and h.shape[2] % 2 == 0 and h.shape[3] % 2 == 0:
        h = F.relu(winograd3x3_reflect(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    else:
        if pad[0] or pad[1]:
            h = F.pad(h, (pad[1], pad[1], pad[0], pad[0]), mode="reflect")
        h = F.relu(F.conv2d(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    return F.conv2d(h, p[prefix + "NN.4.weight"], p[prefix + "NN.4.bias"])


def coupling_net_winograd(x0, p, prefix, pad):
    """oracle.flow_oracle.coupling_net with the 3x3 in Winograd form (fp32 inputs, (1, 1) padding; otherwise the direct form)."""
    h = F.relu(F.conv2d(x0, p[prefix + "NN.0.weight"], p[prefix + "NN.0.bias"]))
    if h.dtype == torch.float32 and tuple(pad) == (1, 1) and h.shape[2] % 2 == 0 and h.shape[3] % 2 == 0:
        h = F.relu(winograd3x3_reflect(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    else:
        if pad[0] or pad[1]:
            h = F.pad(h, (pad[1], pad[1], pad[0], pad[0]), mode="reflect")
        h = F.relu(F.conv2d(h, p[prefix + "NN.2.weight"], p[prefix + "NN.2.bias"]))
    return F.conv2d(h, p[prefix + "NN.4.weight"], p[prefix + "NN.4.bias"])


# Weight gradient of the same convolution in the transposed form F(3x3, 2x2) (contextflow_amd/csrc/cf_wgrad.hip, k_wgrad
# with WINO): the 2x2 tiles of the upstream gradient play the filter, the 4x4 reflect-padded input patches the data, and
# the sum over tiles and samples runs in the Winograd domain (16 positions) before ONE output transform.  The kernel uses
# the unscaled G' = 2G rows and folds the factors 1/2 into the output transform; this restatement keeps them in G.
G32 = torch.tensor([[1, 0], [.5, .5], [.5, -.5], [0, 1]], dtype=torch.float32)
AT32 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, -1]], dtype=torch.float32)


def winograd3x3_wgrad_reflect(g, h):
    """g (B, Co, H, W) upstream gradient of conv2d(reflect_pad(h, 1), w), h (B, Ci, H, W) -> dL/dw as (Co, Ci, 3, 3); fp32."""
    dy = g.unfold(2, 2, 2).unfold(3, 2, 2)                                          # (B, Co, H/2, W/2, 2, 2)
    d = F.pad(h, (1, 1, 1, 1), mode="reflect").unfold(2, 4, 2).unfold(3, 4, 2)      # (B, Ci, H/2, W/2, 4, 4)
    Gy = torch.einsum("xa,noijab,yb->noijxy", G32, dy, G32)
    V = torch.einsum("xa,ncijab,yb->ncijxy", BT, d, BT)
    M = torch.einsum("noijxy,ncijxy->ocxy", Gy, V)                                  # summed over samples and tiles
    return torch.einsum("px,ocxy,qy->ocpq", AT32, M, AT32)

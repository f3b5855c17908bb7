"""Lane-level emulation of the planned split-bf16 weight-gradient path (design check, CPU only)."""
import numpy as np, torch

def bf16_bits(x):  # RNE bf16 of fp32 array -> uint16 bits
    t = torch.from_numpy(np.asarray(x, np.float32)).to(torch.bfloat16)
    return t.view(torch.int16).numpy().astype(np.uint16)

def f32_from_hi(bits16):
    return (bits16.astype(np.uint32) << 16).view(np.float32)

def pack_bf16(a, b):  # low = bf16(a), high = bf16(b)
    return bf16_bits(a).astype(np.uint32) | (bf16_bits(b).astype(np.uint32) << 16)

LD = 20
def store_rows_bf2(S, x, T):
    """S: uint32 [64*LD]; x: [T][4][64 lanes] values: lane (j,g) reg (t,r) = feature 16t+4g+r of point j."""
    lanes = np.arange(64); j = lanes & 15; g = lanes >> 4
    for t in range(T):
        for r0 in (0, 2):
            v0, v1 = x[t][r0], x[t][r0 + 1]
            ph = pack_bf16(v0, v1)
            res0 = (v0 - f32_from_hi((ph & 0xffff).astype(np.uint16))).astype(np.float32)
            res1 = (v1 - (ph & 0xffff0000).view(np.float32)).astype(np.float32)
            pm = pack_bf16(res0, res1)
            qh, qm = ph[lanes ^ 1], pm[lanes ^ 1]   # DPP quad_perm [1,0,3,2]
            even = (j & 1) == 0
            wh = np.where(even, (ph & 0xffff) | (qh << 16), (qh >> 16) | (ph & 0xffff0000)).astype(np.uint32)
            wm = np.where(even, (pm & 0xffff) | (qm << 16), (qm >> 16) | (pm & 0xffff0000)).astype(np.uint32)
            row = 16 * t + 4 * g + r0 + np.where(even, 0, 1)
            S[row * LD + (j >> 1)] = wh
            S[row * LD + 8 + (j >> 1)] = wm

def read_op(S_areas, A0, row_base, lane_i, g, piece):
    """u4 (as 8 floats) of lane (i, g): area A0 + (g>>1), words 4*(g&1).. of the h (piece 0) / m (piece 1) part."""
    S = S_areas[A0 + (g >> 1)]
    base = (row_base + lane_i) * LD + 8 * piece + 4 * (g & 1)
    w = S[base:base + 4]
    out = np.empty(8, np.float32)
    out[0::2] = f32_from_hi((w & 0xffff).astype(np.uint16))
    out[1::2] = (w & 0xffff0000).view(np.float32)
    return out

rs = np.random.RandomState(0)
n_areas, T_d, T_x = 8, 4, 4
dY = rs.standard_normal((n_areas, 16, 64)).astype(np.float32) * np.exp(rs.uniform(-6, 2, (n_areas, 16, 64))).astype(np.float32)  # [area][point][neuron]
X = rs.standard_normal((n_areas, 16, 64)).astype(np.float32)
Sd = [np.zeros(64 * LD, np.uint32) for _ in range(n_areas)]
Sx = [np.zeros(64 * LD, np.uint32) for _ in range(n_areas)]
lanes = np.arange(64); jj = lanes & 15; gg = lanes >> 4
for a in range(n_areas):
    xd = [[dY[a][jj, 16 * t + 4 * gg + r] for r in range(4)] for t in range(T_d)]
    xx = [[X[a][jj, 16 * t + 4 * gg + r] for r in range(4)] for t in range(T_x)]
    store_rows_bf2(Sd[a], xd, T_d); store_rows_bf2(Sx[a], xx, T_x)
# coop_dw_bf2 for output tile (n, m): C[4g+r][j] per lane
def coop(n, m):
    C = np.zeros((16, 16), np.float64)
    db = np.zeros(16, np.float64)
    for A0 in range(0, n_areas, 2):
        for g in range(4):
            for i in range(16):
                ah, am = read_op(Sd, A0, 16 * n, i, g, 0), read_op(Sd, A0, 16 * n, i, g, 1)
                db[i] += float(ah.sum()) + float(am.sum())
                for j in range(16):
                    bh, bm = read_op(Sx, A0, 16 * m, j, g, 0), read_op(Sx, A0, 16 * m, j, g, 1)
                    C[i, j] += float(am @ bh) + float(ah @ bm) + float(ah @ bh)
    return C, db
worst = 0.0
for n in range(4):
    for m in range(4):
        C, db = coop(n, m)
        ref = sum(dY[a][:, 16 * n:16 * n + 16].astype(np.float64).T @ X[a][:, 16 * m:16 * m + 16].astype(np.float64) for a in range(n_areas))
        worst = max(worst, float(np.abs(C - ref).max() / np.abs(ref).max()))
        dref = sum(dY[a][:, 16 * n:16 * n + 16].astype(np.float64).sum(0) for a in range(n_areas))
        assert np.abs(db - dref).max() <= 1e-4 * np.abs(dref).max(), (n, m)
print("max |err| / max |dW| over the 16 tiles:", worst)
assert worst < 1e-4
print("layout OK")

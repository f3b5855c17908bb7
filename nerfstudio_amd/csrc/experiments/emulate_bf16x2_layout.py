#!/usr/bin/env python3
"""Lane-level emulation (numpy, CPU) of field_mlp_bwd_bf16x2.patch: the scratch layout store_rows_bf2 writes, the operands
load_op2 builds from it (two ds_read_b128 + v_perm_b32 de-interleave) and the v_mfma_f32_16x16x32_bf16 products of
coop_dw_bf2, for 8 areas (waves) of 16 points: the result must be dY^T X to the two-piece precision, and the bias sums the
sums of dY. Run it after touching any index in the patch (it restates the patch's index arithmetic, it does not parse it)."""
import numpy as np
import torch

LD = 20  # kScratchLd


def bf16_bits(x):
    t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16)  # RNE, as v_cvt_pk_bf16_f32
    return t.view(torch.int16).numpy().astype(np.uint16)


def widen(bits16):
    return (bits16.astype(np.uint32) << 16).view(np.float32)


def store_rows_bf2(S, x, T):
    """x[t][r][lane]: lane (j, g) register (t, r) = feature 16t + 4g + r of point j -> S[feature][point] = h | m << 16."""
    lanes = np.arange(64)
    j, g = lanes & 15, lanes >> 4
    for t in range(T):
        for r in range(4):
            v = x[t][r].astype(np.float32)
            h = bf16_bits(v)
            m = bf16_bits((v - widen(h)).astype(np.float32))
            S[(16 * t + 4 * g + r) * LD + j] = h.astype(np.uint32) | (m.astype(np.uint32) << 16)


def perm(src0, src1, sel):
    """v_perm_b32: result byte i = byte sel[i] of the 8-byte value src0:src1 (0-3 -> src1, 4-7 -> src0)."""
    both = (np.uint64(src0) << np.uint64(32)) | np.uint64(src1)
    out = 0
    for i in range(4):
        out |= int((both >> np.uint64(8 * ((sel >> (8 * i)) & 0xff))) & np.uint64(0xff)) << (8 * i)
    return np.uint32(out)


def load_op2(S, base):
    w = S[base:base + 8]
    h = [perm(w[2 * k + 1], w[2 * k], 0x05040100) for k in range(4)]
    m = [perm(w[2 * k + 1], w[2 * k], 0x07060302) for k in range(4)]

    def slots(words):  # 8 bf16 K-slots of the packed operand, slot 2k = low half of word k
        out = np.empty(8, np.float32)
        for k, wd in enumerate(words):
            out[2 * k] = widen(np.array([wd & 0xffff], np.uint16))[0]
            out[2 * k + 1] = np.array([wd & 0xffff0000], np.uint32).view(np.float32)[0]
        return out

    return slots(h), slots(m)


def main():
    rs = np.random.RandomState(0)
    areas = 8
    dY = (rs.standard_normal((areas, 16, 64)) * np.exp(rs.uniform(-6, 2, (areas, 16, 64)))).astype(np.float32)  # [area][point][neuron]
    X = rs.standard_normal((areas, 16, 64)).astype(np.float32)
    Sd = [np.zeros(64 * LD, np.uint32) for _ in range(areas)]
    Sx = [np.zeros(64 * LD, np.uint32) for _ in range(areas)]
    lanes = np.arange(64)
    jj, gg = lanes & 15, lanes >> 4
    for a in range(areas):
        store_rows_bf2(Sd[a], [[dY[a][jj, 16 * t + 4 * gg + r] for r in range(4)] for t in range(4)], 4)
        store_rows_bf2(Sx[a], [[X[a][jj, 16 * t + 4 * gg + r] for r in range(4)] for t in range(4)], 4)
    worst = 0.0
    for n in range(4):
        for m in range(4):
            C = np.zeros((16, 16))  # C[row of A][column of B]; lane (j, g) register r holds C[4g + r][j]
            db = np.zeros(16)
            for pair in range(0, areas, 2):
                for g in range(4):
                    area, half = pair + (g >> 1), 8 * (g & 1)
                    A = [load_op2(Sd[area], (16 * n + i) * LD + half) for i in range(16)]
                    B = [load_op2(Sx[area], (16 * m + j) * LD + half) for j in range(16)]
                    for i in range(16):
                        db[i] += float(A[i][0].sum()) + float(A[i][1].sum())
                        for j in range(16):
                            C[i, j] += float(A[i][0] @ B[j][1]) + float(A[i][1] @ B[j][0]) + float(A[i][0] @ B[j][0])
            ref = sum(dY[a][:, 16 * n:16 * n + 16].astype(np.float64).T @ X[a][:, 16 * m:16 * m + 16].astype(np.float64)
                      for a in range(areas))
            worst = max(worst, float(np.abs(C - ref).max() / np.abs(ref).max()))
            dref = sum(dY[a][:, 16 * n:16 * n + 16].astype(np.float64).sum(0) for a in range(areas))
            assert np.abs(db - dref).max() <= 1e-4 * np.abs(dref).max(), (n, m)
    print(f"max |err| / max |dW| over the 16 output tiles (128 points): {worst:.2e}")
    assert worst < 1e-4
    print("layout OK")


if __name__ == "__main__":
    main()

"""Lane-level simulation (numpy, CPU) of the address arithmetic of onepose_amd/csrc/gemm_split_glds.h: the LDS-DMA piece ->
LDS image mapping (swizzle on the global side), the fragment read offsets and the 32x32x16 MFMA operand / accumulator layouts.
It reproduces C = A * B for one workgroup tile from the kernel's own index formulas; run it after changing any of them.

    python tests/studies/sim_split_glds_layout.py
"""
import numpy as np


def simulate(BM=128, WM=2, WN=4, PA=2, K=64, seed=0):
    rs = np.random.RandomState(seed)
    TM, BN, WAVES = BM // WM // 32, 32 * WN, WM * WN
    KT = K // 32
    M_total = 512                          # rows of the whole operator; the tile starts at row0
    row0, col0, ldb = 128, 64, 640
    A = rs.standard_normal((PA, M_total, K)).astype(np.float32)       # planes (any values)
    B = rs.standard_normal((K, ldb)).astype(np.float32)
    # slab-major planes: (m, k) at ((k / 32) * M + m) * 32 + k % 32
    planes = np.zeros((PA, KT * M_total * 32), np.float32)
    for p in range(PA):
        for kt in range(KT):
            planes[p, kt * M_total * 32:(kt + 1) * M_total * 32] = A[p][:, kt * 32:(kt + 1) * 32].reshape(-1)
    A_PLANE = BM * 64 // 2                  # LDS image in 2-byte elements
    A_EL = PA * A_PLANE
    B_EL = 32 * BN                          # floats
    NA, NB = PA * BM * 64 // 1024, 32 * BN * 4 // 1024
    G = (NA + NB) // WAVES
    RPP = BM // 16
    LPR, KPP = BN // 4, 64 // (BN // 4)
    C = np.zeros((BM, BN), np.float64)
    for kt in range(KT):
        ldsA = np.full(A_EL, np.nan, np.float32)      # element index = byte / 2
        ldsB = np.full(B_EL, np.nan, np.float32)      # element index = (byte - A_BYTES) / 4
        for wave in range(WAVES):
            for j in range(G):
                q = j * WAVES + wave
                for lane in range(64):
                    if q < NA:
                        plane, qi = q // RPP, q % RPP
                        a_lane = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16)     # bytes
                        src_el = (kt * M_total + row0) * 32 + (qi * 1024 + a_lane) // 2         # a_pl(kt, plane) + bytes / 2
                        dst_el = (plane * BM * 64 + qi * 1024 + lane * 16) // 2
                        ldsA[dst_el:dst_el + 8] = planes[plane, src_el:src_el + 8]
                    else:
                        qb = q - NA
                        b_lane = ((lane // LPR) * ldb + (lane % LPR) * 4)                        # floats
                        src = (kt * 32) * ldb + col0 + qb * KPP * ldb + b_lane
                        dst = (qb * 1024 + lane * 16) // 4
                        ldsB[dst:dst + 4] = B.reshape(-1)[src:src + 4]
        assert not np.isnan(ldsA).any() and not np.isnan(ldsB).any(), "LDS image not fully written"
        for wave in range(WAVES):
            wm, wn = wave // WN, wave % WN
            for P in range(2):
                # fragments: a[tm][plane][lane] = 8 values, b[lane] = 8 values
                for tm in range(TM):
                    Afrag = np.zeros((PA, 64, 8), np.float32)
                    Bfrag = np.zeros((64, 8), np.float32)
                    for lane in range(64):
                        half, l31 = lane >> 5, lane & 31
                        row = (wm * TM + tm) * 32 + l31
                        x = (row >> 2) & 3
                        a_off = row * 64 + (((2 * P + half) ^ x) * 16)
                        for pl in range(PA):
                            e = (pl * BM * 64 + a_off) // 2
                            Afrag[pl, lane] = ldsA[e:e + 8]
                        b_off = ((8 * half) * BN + wn * 32 + l31) * 4
                        for jj in range(8):
                            Bfrag[lane, jj] = ldsB[(b_off + (P * 16 + jj) * BN * 4) // 4]
                    # MFMA 32x32x16: A lane (row l31, half) holds k = 8 half + j; B lane (col l31, half) likewise
                    a_tile = np.zeros((PA, 32, 16))
                    b_tile = np.zeros((16, 32))
                    for lane in range(64):
                        half, l31 = lane >> 5, lane & 31
                        a_tile[:, l31, 8 * half:8 * half + 8] = Afrag[:, lane]
                        b_tile[8 * half:8 * half + 8, l31] = Bfrag[lane]
                    r0 = (wm * TM + tm) * 32
                    C[r0:r0 + 32, wn * 32:wn * 32 + 32] += a_tile.sum(0) @ b_tile       # all planes against B (term structure aside)
    ref = A.astype(np.float64).sum(0)[row0:row0 + BM] @ B[:, col0:col0 + BN].astype(np.float64)
    err = np.abs(C - ref).max()
    return err


def simulate_f32(BM=128, WM=4, WN=2, K=64, lda=512, seed=0):
    """MODE 0: the A operand is the row-major fp32 matrix itself: 128-byte LDS rows, chunk c of row r at c ^ ((r >> 1) & 7)."""
    rs = np.random.RandomState(seed)
    TM, BN, WAVES = BM // WM // 32, 32 * WN, WM * WN
    KT = K // 32
    row0, col0, ldb = 128, 64, 640
    A = rs.standard_normal((512, lda)).astype(np.float32)
    B = rs.standard_normal((K, ldb)).astype(np.float32)
    Af = A.reshape(-1)
    NA, NB = BM * 128 // 1024, 32 * BN * 4 // 1024
    G = (NA + NB) // WAVES
    CPR, RPQ = 8, 8
    LPR, KPP = BN // 4, 64 // (BN // 4)
    C = np.zeros((BM, BN), np.float64)
    for kt in range(KT):
        ldsA = np.full(BM * 32, np.nan, np.float32)   # float index = byte / 4
        ldsB = np.full(32 * BN, np.nan, np.float32)
        for wave in range(WAVES):
            for j in range(G):
                q = j * WAVES + wave
                for lane in range(64):
                    if q < NA:
                        qi = q
                        a_lane = (lane // CPR) * (lda * 4) + (((lane % CPR) ^ ((lane >> 4) & 3)) * 16)
                        flip = (qi & 1) * 64
                        src_b = (row0 * lda + kt * 32) * 4 + qi * RPQ * lda * 4 + (a_lane ^ flip)
                        dst_b = qi * 1024 + lane * 16
                        ldsA[dst_b // 4:dst_b // 4 + 4] = Af[src_b // 4:src_b // 4 + 4]
                    else:
                        qb = q - NA
                        b_lane = ((lane // LPR) * ldb + (lane % LPR) * 4)
                        src = (kt * 32) * ldb + col0 + qb * KPP * ldb + b_lane
                        dst = (qb * 1024 + lane * 16) // 4
                        ldsB[dst:dst + 4] = B.reshape(-1)[src:src + 4]
        assert not np.isnan(ldsA).any() and not np.isnan(ldsB).any()
        for wave in range(WAVES):
            wm, wn = wave // WN, wave % WN
            for P in range(2):
                for tm in range(TM):
                    a_tile = np.zeros((32, 16)); b_tile = np.zeros((16, 32))
                    for lane in range(64):
                        half, l31 = lane >> 5, lane & 31
                        row = (wm * TM + tm) * 32 + l31
                        x = (row >> 1) & 7
                        vals = []
                        for e in range(2):
                            off = row * 128 + ((((2 * P + half) * 2 + e) ^ x) * 16)
                            vals += list(ldsA[off // 4:off // 4 + 4])
                        a_tile[l31, 8 * half:8 * half + 8] = vals
                        b_off = ((8 * half) * BN + wn * 32 + l31) * 4
                        b_tile[8 * half:8 * half + 8, l31] = [ldsB[(b_off + (P * 16 + jj) * BN * 4) // 4] for jj in range(8)]
                    r0 = (wm * TM + tm) * 32
                    C[r0:r0 + 32, wn * 32:wn * 32 + 32] += a_tile @ b_tile
    ref = A[row0:row0 + BM, :K].astype(np.float64) @ B[:, col0:col0 + BN].astype(np.float64)
    return np.abs(C - ref).max()


def simulate_bt_swap(BM=128, WM=2, WN=2, PA=2, K=64, seed=0):
    """Round 5: the B operand given TRANSPOSED in memory (B^T [column][k] fp32, row stride ldb: mlp3_sp reading mlp0_sp's U^T) through 128-byte
    LDS rows (chunk c of row r at c ^ ((r >> 1) & 7), SP_OPT_BT), and the swapped MFMA operands (SP_OPT_SWAP): the accumulators hold the
    transposed block -- lane = row of the A operand, register r = column mfma_row(r, half) -- and leave as U^T[col][row] straight from them."""
    rs = np.random.RandomState(seed)
    TM, BN, WAVES = BM // WM // 32, 32 * WN, WM * WN
    KT = K // 32
    M_total, row0, col0, ldb = 512, 128, 64, 512          # B^T has K <= ldb floats per column row
    A = rs.standard_normal((PA, M_total, K)).astype(np.float32)
    NCOL = 256
    Bt = rs.standard_normal((NCOL, ldb)).astype(np.float32)   # B^T[column][k]
    planes = np.zeros((PA, KT * M_total * 32), np.float32)
    for p in range(PA):
        for kt in range(KT):
            planes[p, kt * M_total * 32:(kt + 1) * M_total * 32] = A[p][:, kt * 32:(kt + 1) * 32].reshape(-1)
    NA, NB = PA * BM * 64 // 1024, BN * 128 // 1024
    G = (NA + NB) // WAVES
    RPP = BM // 16
    Btf = Bt.reshape(-1)
    UT = np.full((BN, BM), np.nan)                          # what the transposed direct stores write: U^T[point][channel]
    acc = {}                                                # (wave, tm) -> [64 lanes][16 regs]
    for kt in range(KT):
        ldsA = np.full(PA * BM * 32, np.nan, np.float32)
        ldsB = np.full(BN * 32, np.nan, np.float32)         # float index = (byte - A_BYTES) / 4
        for wave in range(WAVES):
            for j in range(G):
                q = j * WAVES + wave
                for lane in range(64):
                    if q < NA:
                        plane, qi = q // RPP, q % RPP
                        a_lane = (lane >> 2) * 64 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16)
                        src_el = (kt * M_total + row0) * 32 + (qi * 1024 + a_lane) // 2
                        dst_el = (plane * BM * 64 + qi * 1024 + lane * 16) // 2
                        ldsA[dst_el:dst_el + 8] = planes[plane, src_el:src_el + 8]
                    else:
                        qb = q - NA
                        b_lane = (lane >> 3) * ldb * 4 + (((lane & 7) ^ ((lane >> 4) & 3)) * 16)          # bytes
                        src_b = (col0 * ldb + kt * 32) * 4 + qb * 8 * ldb * 4 + (b_lane ^ ((qb & 1) * 64))   # b_slab(kt) = &Bt[col0][32 kt]
                        dst_b = qb * 1024 + lane * 16
                        ldsB[dst_b // 4:dst_b // 4 + 4] = Btf[src_b // 4:src_b // 4 + 4]
        assert not np.isnan(ldsA).any() and not np.isnan(ldsB).any(), "LDS image not fully written"
        for wave in range(WAVES):
            wm, wn = wave // WN, wave % WN
            for P in range(2):
                b_tile = np.zeros((16, 32))                 # [k][column l31]
                for lane in range(64):
                    half, l31 = lane >> 5, lane & 31
                    row = wn * 32 + l31
                    vals = []
                    for e in range(2):
                        off = row * 128 + ((((2 * P + half) * 2 + e) ^ ((row >> 1) & 7)) * 16)
                        vals += list(ldsB[off // 4:off // 4 + 4])
                    b_tile[8 * half:8 * half + 8, l31] = vals
                for tm in range(TM):
                    a_tile = np.zeros((PA, 32, 16))
                    for lane in range(64):
                        half, l31 = lane >> 5, lane & 31
                        row = (wm * TM + tm) * 32 + l31
                        a_off = row * 64 + (((2 * P + half) ^ ((row >> 2) & 3)) * 16)
                        for pl in range(PA):
                            e = (pl * BM * 64 + a_off) // 2
                            a_tile[pl, l31, 8 * half:8 * half + 8] = ldsA[e:e + 8]
                    # swapped instruction: D'[i][j] = sum_k B[k][i] A[j][k]: i = column (point), j = row (channel);
                    # accumulator register r of lane (l31 = j, half) holds D'[mfma_row(r, half)][j]
                    d = b_tile.T @ a_tile.sum(0).T          # [32 points][32 channels]
                    regs = acc.setdefault((wave, tm), np.zeros((64, 16)))
                    for lane in range(64):
                        half, l31 = lane >> 5, lane & 31
                        for r in range(16):
                            regs[lane, r] += d[(r & 3) + 8 * (r >> 2) + 4 * half, l31]
    for (wave, tm), regs in acc.items():
        wm, wn = wave // WN, wave % WN
        for lane in range(64):
            half, l31 = lane >> 5, lane & 31
            ch = (wm * TM + tm) * 32 + l31
            for r in range(16):
                UT[wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, ch] = regs[lane, r]     # Ut[(c0 + pt0 + mfma_row) * 512 + rt * BM + ch]
    ref = (A.astype(np.float64).sum(0)[row0:row0 + BM, :] @ Bt[col0:col0 + BN, :K].astype(np.float64).T).T    # [point][channel]
    return np.abs(UT - ref).max()


if __name__ == "__main__":
    for cfg in (dict(WN=2), dict(WN=4), dict(WN=2, PA=3)):
        e = simulate_bt_swap(**cfg)
        print("transposed B operand + swapped MFMA operands", cfg, "max |U^T - (A B)^T| =", e)
        assert e < 1e-9
    for cfg in (dict(lda=512), dict(lda=256), dict(lda=512, WM=2, WN=4)):
        e = simulate_f32(**cfg)
        print("fp32 mode", cfg, "max |C - A B| =", e)
        assert e < 1e-9
    for cfg in (dict(BM=128, WM=2, WN=4, PA=2), dict(BM=128, WM=2, WN=2, PA=2), dict(BM=128, WM=2, WN=4, PA=3), dict(BM=128, WM=2, WN=2, PA=3)):
        e = simulate(**cfg)
        print(cfg, "max |C - A B| =", e)
        assert e < 1e-9
    print("layout OK")

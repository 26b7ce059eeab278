"""Brute-force bank-conflict check of the A-fragment ds_read_b128 reads from the 18-pixel-wide LDS halo (64-byte rows):
for every tap, k-group, lane half and 16-lane group of the instruction, count the extra LDS cycles under a chunk swizzle.
(R >> 2) & 3 is what tapgemm_halo_kernel uses (a 2-way conflict on every A read); ((R >> 1) + R // 18) & 3 is conflict free."""
groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]


def conflicts(swz):
    tot = n = 0
    for wm in range(4):
        for i in range(2):
            for dh in (-1, 0, 1):
                for dw in (-1, 0, 1):
                    for kk in range(2):
                        for h in range(2):
                            for g in groups:
                                units = {}
                                for l31 in g:
                                    R = (4 * wm + 2 * i + (l31 >> 4) + 1 + dh) * 18 + (l31 & 15) + 1 + dw
                                    u = (R * 4 + ((2 * kk + h) ^ swz(R))) % 16          # 16-byte unit inside the 256-byte bank row
                                    units[u] = units.get(u, 0) + 1
                                tot += max(units.values()) - 1
                                n += 1
    return tot, n


def conflicts_f32(fn):
    """fp32 kernel on v_mfma_f32_16x16x4: lane l reads pixel l & 15 of one patch row, chunk fn(l >> 4, R)."""
    g64 = groups + [[l + 32 for l in g] for g in groups]
    tot = n = 0
    for prow in range(8):
        for dh in (-1, 0, 1):
            for dw in (-1, 0, 1):
                for g in g64:
                    units = {}
                    for l in g:
                        R = (prow + 1 + dh) * 18 + (l & 15) + 1 + dw
                        u = (R * 4 + fn(l >> 4, R)) % 16
                        units[u] = units.get(u, 0) + 1
                    tot += max(units.values()) - 1
                    n += 1
    return tot, n


if __name__ == "__main__":
    print("fp32 16x16x4, q ^ ((R >> 2) & 3)  : extra cycles / group reads =", conflicts_f32(lambda q, R: q ^ ((R >> 2) & 3)))
    print("fp32 16x16x4, (q + (R >> 1)) & 3  : extra cycles / group reads =", conflicts_f32(lambda q, R: (q + (R >> 1)) & 3))
    print("(R >> 2) & 3             : extra cycles / group reads =", conflicts(lambda R: (R >> 2) & 3))
    print("((R >> 1) + R // 18) & 3 : extra cycles / group reads =", conflicts(lambda R: ((R >> 1) + R // 18) & 3))

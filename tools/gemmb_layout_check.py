#!/usr/bin/env python
"""Index arithmetic of the fused dense backward (kgcn_amd/csrc/gemmb.hip) checked on the CPU: the LDS image of a 32-row stage
of d pre-activation pieces is written once (8-byte row writes) and read two ways -- 16-byte A-operand rows for dX and
ds_read_b64_tr_b16 transpose reads for the B operand of dW.  This script emulates the address functions the kernel uses
(the SAME lane-base + immediate decomposition), the transpose read's lane exchange, and the LDS bank rules of
MI355X_MICROARCH.md (lane groups per instruction, 64 banks of 4 bytes) and asserts: the image is injective, both reads deliver
exactly the MFMA fragments, and all three access patterns are bank-conflict free.  tests/test_host_logic.py runs it."""
import numpy as np

PLANE = 20480                       # bytes of one piece plane: 8 row quads x 2,560


def addr(r, f):
    """byte address of element (row r of the stage, column f) inside a piece plane"""
    R, a = r >> 2, r & 3
    return R * 2560 + a * 64 + (f >> 5) * 320 + ((((f & 31) >> 3) ^ (R & 3)) << 4) + (f & 7) * 2


def write_addr(wave, i, lane):
    """staging: wave `wave` (0..3) owns rows 8 wave + i; lane = columns 4 lane .. 4 lane + 3 (8 bytes per piece)"""
    p, a = i >> 2, i & 3
    seg, lane_c, off8 = lane >> 3, (lane >> 1) & 3, 8 * (lane & 1)
    base_p = (2 * wave + p) * 2560 + seg * 320 + ((lane_c ^ ((2 * wave + p) & 3)) << 4) + off8
    return base_p + a * 64


def a_read_addr(lane, ks):
    """dX A operand (v_mfma_f32_32x32x16_f16): lane (li, hi) = row li, k = 16 ks + 8 hi + 0..7 -> 16 bytes"""
    li, hi = lane & 31, lane >> 5
    R, a = li >> 2, li & 3
    base = R * 2560 + a * 64 + ((((2 if ks & 1 else 0) + hi) ^ (R & 3)) << 4)       # two lane bases: ks even / odd
    return base + (ks >> 1) * 320


def tr_read_addr(lane, q, t, jt):
    """dW B operand through ds_read_b64_tr_b16: what lane `lane` ADDRESSES (8 bytes) for k-step q, half t, column tile jt"""
    g16, i = lane >> 4, lane & 15
    hi, half16, a = g16 >> 1, g16 & 1, i >> 2
    base_t = 2 * hi * 2560 + a * 64 + ((((2 * half16) + ((i & 3) >> 1)) ^ (2 * hi + t)) << 4) + 8 * (i & 1)   # two lane bases: t
    return base_t + (4 * q + t) * 2560 + jt * 320


def tr_read(mem16, addrs):
    """ds_read_b64_tr_b16: within 16 lanes, lane l element j receives element l & 3 of what lane 4 j + ((l >> 2) & 3) addressed"""
    out = np.zeros((64, 4), mem16.dtype)
    for l in range(64):
        g = l & ~15
        for j in range(4):
            src = g + 4 * j + ((l >> 2) & 3)
            out[l, j] = mem16[addrs[src] // 2 + (l & 3)]
    return out


def banks_ok(addrs, nbytes, groups, modulo):
    """every lane group touches each bank at most once (identical addresses broadcast)"""
    worst = 1
    for grp in groups:
        seen = {}
        for l in grp:
            for b in range(0, nbytes, 4):
                bank = ((addrs[l] + b) // 4) % modulo
                seen.setdefault(bank, set()).add((addrs[l] + b) // 4)
        worst = max(worst, max(len(v) for v in seen.values()))
    return worst


def main():
    # 1. injective, inside the plane
    A = np.array([[addr(r, f) for f in range(256)] for r in range(32)])
    assert A.max() + 2 <= PLANE and len(np.unique(A)) == A.size
    mem = np.zeros(PLANE // 2, np.int32) - 1
    val = lambda r, f: r * 256 + f
    # 2. staging writes land where addr() says
    W = {}
    for wave in range(4):
        for i in range(8):
            was = [write_addr(wave, i, lane) for lane in range(64)]
            W[(wave, i)] = was
            for lane in range(64):
                for e in range(4):
                    assert was[lane] + 2 * e == addr(8 * wave + i, 4 * lane + e)
                    mem[(was[lane] + 2 * e) // 2] = val(8 * wave + i, 4 * lane + e)
    assert (mem[A // 2] >= 0).all()
    # 3. A operand rows
    for ks in range(16):
        ad = [a_read_addr(l, ks) for l in range(64)]
        for l in range(64):
            li, hi = l & 31, l >> 5
            for e in range(8):
                assert mem[ad[l] // 2 + e] == val(li, 16 * ks + 8 * hi + e)
        g128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
        g128 += [[x + 32 for x in g] for g in g128]
        assert banks_ok(ad, 16, g128, 64) == 1, ("A read", ks)
    # 4. B operand through the transpose read
    for q in range(2):
        for jt in range(8):
            frag = np.zeros((64, 8), np.int32)
            for t in range(2):
                ad = [tr_read_addr(l, q, t, jt) for l in range(64)]
                assert all(a % 8 == 0 for a in ad)
                frag[:, 4 * t:4 * t + 4] = tr_read(mem, ad)
                assert banks_ok(ad, 8, [list(range(32)), list(range(32, 64))], 64) == 1, ("tr read", q, t, jt)
            for l in range(64):
                li, hi = l & 31, l >> 5
                for e in range(8):
                    assert frag[l, e] == val(16 * q + 8 * hi + e, 32 * jt + li), (q, jt, l, e)
    # 5. the 8-byte staging writes: ds_write_b64 = 4 groups of 16 contiguous lanes, 32 banks
    for (wave, i), was in W.items():
        assert banks_ok(was, 8, [list(range(g, g + 16)) for g in range(0, 64, 16)], 32) == 1, ("write", wave, i)
    print("gemmb LDS image: injective, A rows and transpose-read B fragments exact, writes / b128 reads / tr reads conflict-free")


if __name__ == "__main__":
    main()

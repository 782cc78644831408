"""Bank-conflict model of k_describe's LDS instructions (my-slam_amd/csrc/orbx_describe.hip), from the banking rules of
/opt/skills/guides/MI355X_MICROARCH.md section LDS: per instruction kind, the lane groups that are serviced together and the bank of a
byte address.  The task tables are rebuilt exactly as orbx_upload_constants() builds them; the rBRIEF gathers are sampled over
angles.  Prints, per phase, the LDS-array cycles of one wave (one keypoint) and how many of them are conflict cycles -- the split
that SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE give only as one total per kernel.
usage: python tools/lds_model_describe.py [raw_stride] [p_stride_dwords] [bl_stride]"""
import re
import sys

import numpy as np

RAW = int(sys.argv[1]) if len(sys.argv) > 1 else 48
PST = int(sys.argv[2]) if len(sys.argv) > 2 else 40
BLS = int(sys.argv[3]) if len(sys.argv) > 3 else 40


def pattern():
    txt = open(__file__.replace("tools/lds_model_describe.py", "my-slam_amd/csrc/orb_pattern_data.h")).read()
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    return np.array([int(v) for v in re.findall(r"-?\d+", body)], np.int32)[:1024]


def groups(kind):
    if kind in ("b32", "u8", "b16", "wb32"):
        return [list(range(0, 32)), list(range(32, 64))], 32
    if kind == "b64":
        return [list(range(0, 32)), list(range(32, 64))], 64
    if kind == "r2b64":          # each of the two accesses: 4 x 16 contiguous lanes, bank mod 32
        return [list(range(g * 16, g * 16 + 16)) for g in range(4)], 32
    if kind == "w128":           # 8 x 8 contiguous
        return [list(range(g * 8, g * 8 + 8)) for g in range(8)], 32
    raise ValueError(kind)


def cost(kind, addr, nbytes):
    """addr[lane] = byte address or None (inactive).  Returns (array cycles, conflict cycles)."""
    gs, mod = groups(kind)
    cyc = conf = 0
    for g in gs:
        banks = {}
        active = False
        for ln in g:
            a = addr[ln]
            if a is None:
                continue
            active = True
            for dw in range(a // 4, (a + nbytes - 1) // 4 + 1):
                banks.setdefault(dw % mod, set()).add(dw)
        if not active:
            continue
        m = max(len(s) for s in banks.values())
        cyc += m
        conf += m - 1
    return cyc, conf


def main():
    pat = pattern()
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    # c_mom_tab
    mom = []
    for t in range(31 * 9):
        vr, dw = t // 9, t % 9 + 1
        v = vr - 15
        um = umax[abs(v)]
        lo, hi, c0 = 21 - um, 21 + um, 4 * dw
        nlo, nhi = min(max(lo - c0, 0), 4), min(max(c0 + 3 - hi, 0), 4)
        if nlo >= 4 or nhi >= 4 or nlo + nhi >= 4:
            continue
        mom.append((6 + vr) * RAW + c0)
    # need[][]
    maxr2 = max(int(pat[2 * i]) ** 2 + int(pat[2 * i + 1]) ** 2 for i in range(512))
    R = np.sqrt(maxr2) + 0.72
    need = [[(i - 18) ** 2 + (j - 18) ** 2 <= R * R for j in range(37)] for i in range(37)]
    rt, ct = [], []
    for rp in range(21, -1, -1):
        for gq in range(10):
            ok = any(need[i][col] for row in (2 * rp, 2 * rp + 1) for col in range(4 * gq, min(4 * gq + 4, 37))
                     for i in range(max(row - 6, 0), min(row, 36) + 1))
            if ok:
                rt.append((2 * rp * RAW + 4 * gq, rp * PST + 4 * gq))
    for q in range(19):
        for cp in range(19):
            ok = any(need[i][j] for i in range(2 * q, min(2 * q + 2, 37)) for j in range(2 * cp, min(2 * cp + 2, 37)))
            if ok:
                ct.append((q * PST + 2 * cp, 2 * BLS * q + 2 * cp))
    tot = {}

    def add(phase, c):
        a, b = tot.get(phase, (0, 0))
        tot[phase] = (a + c[0], b + c[1])

    # tile store: 5 rows x 12 lanes (11 active), nine passes of 5 rows
    for p in range(9):
        addr = [None] * 64
        for lane in range(60):
            rr, d = lane // 12, lane % 12
            if d < 11 and (p < 8 or rr < 3):
                addr[lane] = (rr + 5 * p) * RAW + 4 * d
        add("tile store (ds_write_b32 x9)", cost("wb32", addr, 4))
    # moments
    for it in range(4):
        addr = [mom[it * 64 + l] if it * 64 + l < len(mom) else None for l in range(64)]
        add("IC_Angle reads (ds_read_b32 x4)", cost("b32", addr, 4))
    # row pass
    for it in range(3):
        t = [rt[it * 64 + l] if it * 64 + l < len(rt) else None for l in range(64)]
        for k in range(6):
            addr = [None if e is None else e[0] + (k % 3) * 4 + (k // 3) * RAW for e in t]
            add("row pass reads (6 dwords per task)", cost("b32", addr, 4))
        for k in range(4):          # the 16-byte store is emitted as two ds_write2_b32 = four dword stores
            addr = [None if e is None else e[1] * 4 + 4 * k for e in t]
            add("row pass stores (4 dwords per task)", cost("wb32", addr, 4))
    # column pass
    for it in range(5):
        t = [ct[it * 64 + l] if it * 64 + l < len(ct) else None for l in range(64)]
        for j in range(4):
            addr = [None if e is None else (e[0] + j * PST) * 4 for e in t]
            add("column pass reads (ds_read2_b64: 4 x 8 B per task)", cost("r2b64", addr, 8))
        for k in range(2):
            addr = [None if e is None else e[1] + k * BLS for e in t]
            add("column pass stores (ds_write_b16 x2 per task)", cost("b16", addr, 2))
    # gathers, averaged over angles
    acc = np.zeros(2)
    angles = np.linspace(0, 2 * np.pi, 180, endpoint=False)
    for th in angles:
        a, b = np.float32(np.cos(th)), np.float32(np.sin(th))
        for j in range(4):
            for s in range(2):
                addr = []
                for lane in range(64):
                    bit = j * 64 + lane
                    px, py = np.float32(pat[4 * bit + 2 * s]), np.float32(pat[4 * bit + 2 * s + 1])
                    r = int(np.rint(px * b + py * a)); q = int(np.rint(px * a - py * b))
                    addr.append((r + 18) * BLS + (q + 18))
                acc += np.array(cost("u8", addr, 1))
    acc /= len(angles)
    tot["rBRIEF gathers (ds_read_u8 x8)"] = (acc[0], acc[1])
    tc = sum(v[0] for v in tot.values()); tf = sum(v[1] for v in tot.values())
    print("strides: raw %d B, P %d dwords, bl %d B" % (RAW, PST, BLS))
    for k, v in tot.items():
        print("%-58s %7.1f cycles, %6.1f of them conflicts" % (k, v[0], v[1]))
    print("%-58s %7.1f cycles, %6.1f conflicts (%.0f %%)" % ("one wave", tc, tf, 100 * tf / tc))


if __name__ == "__main__":
    main()

"""Round 5: LDS bank-conflict model of cost_patch_embed_kernel<60,80> (round-4 layout) per the lane groups of
/opt/skills/guides/MI355X_MICROARCH.md §LDS.  Prints LDS-array cycles per slice and per phase, conflict-free vs modelled."""
import numpy as np

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 = G128 + [[l + 32 for l in g] for g in G128]
G32 = [list(range(32)), list(range(32, 64))]
G64W = [list(range(16 * i, 16 * i + 16)) for i in range(4)]          # ds_write_b64: contiguous 16-lane groups

def cycles(addrs, width, groups, mod):
    """addrs: 64 byte addresses (None = inactive); width bytes per lane.  returns LDS-array cycles"""
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for d in range(width // 4 if width >= 4 else 1):
                dw = a // 4 + d
                banks.setdefault(dw % mod, set()).add(dw)
        tot += max((len(v) for v in banks.values()), default=0) if banks else 0
    return tot

H2, W2 = 60, 80
HP, WP = 64, 80
H1, W1, H2o, W2o, H3, W3 = 32, 40, 16, 20, 8, 10
IN_PITCH = WP + 8
O1_ROWS, O1_COLS, O2_ROWS, O2_COLS = H1 + 4, W1 + 4, H2o + 4, W2o + 4
def pad_xh(lo, r):
    x = lo
    while x % 8 != r: x += 1
    return x
O1_XH, O2_XH = pad_xh(O1_COLS // 2, 2), pad_xh(O2_COLS // 2, 5)
O1_PLANE, O2_PLANE = O1_ROWS * O1_XH * 16, O2_ROWS * O2_XH * 16 + 64
IN0_BYTES = (HP + 6) * IN_PITCH * 2
OFF_O1 = IN0_BYTES
OFF_O2 = OFF_O1 + 4 * O1_PLANE
O2_BYTES = 8 * O2_PLANE
print("O1_XH", O1_XH, "O2_XH", O2_XH, "O1_PLANE", O1_PLANE, "O2_PLANE", O2_PLANE, "IN0", IN0_BYTES, "OFF_O1 % 256", OFF_O1 % 256, "OFF_O2 % 256", OFF_O2 % 256)
def o1_cell(c, row, col): return (c * 2 + (col & 1)) * O1_PLANE + (row * O1_XH + (col >> 1)) * 16
def o2_cell(c, row, col): return (c * 2 + (col & 1)) * O2_PLANE + (row * O2_XH + (col >> 1)) * 16

lanes = np.arange(64)
res = {}
# staging: thread t float4 q -> 2 dword stores at ((y+2)*PITCH + x + 2)*2
ideal = act = 0
for wave in range(4):
    for i in range(5):
        for e in range(2):
            ad = []
            for l in range(64):
                q = wave * 64 + l + 256 * i
                if q >= H2 * W2 // 4: ad.append(None); continue
                y, x = q // (W2 // 4), 4 * (q % (W2 // 4))
                ad.append(((y + 2) * IN_PITCH + x + 2) * 2 + 4 * e)
            if all(a is None for a in ad): continue
            act += max(cycles(ad, 4, G32, 32), 2); ideal += 2
res["stage ds_write_b32"] = (ideal, act)
# conv1 reads: 4 x ds_read_b32 per k-step (2), 20 tiles per wave
ideal = act = 0
for wave in range(4):
    for m in range(4):
        for r5 in range(5):
            tile = wave + 4 * r5 + 20 * m
            for s in range(2):
                for d in range(4):
                    ad = []
                    for l in range(64):
                        n16, g4 = l & 15, l >> 4
                        p = tile * 16 + n16; oy, ox = p // W1, p % W1
                        ad.append(((2 * oy + g4 + 4 * s) * IN_PITCH + 2 * ox) * 2 + 4 * d)
                    act += cycles(ad, 4, G32, 32); ideal += 2
res["conv1 ds_read_b32 x8/tile"] = (ideal, act)
# conv1 epilogue: 4 x ds_write_b16
ideal = act = 0
for wave in range(4):
    for m in range(4):
        for r5 in range(5):
            tile = wave + 4 * r5 + 20 * m
            for e in range(4):
                ad = []
                for l in range(64):
                    n16, g4 = l & 15, l >> 4
                    pp = tile * 16 + 4 * g4; y, x = pp // W1, pp % W1
                    ad.append(OFF_O1 + (n16 >> 3) * 2 * O1_PLANE + (n16 & 7) * 2 + o1_cell(0, y + 2, x + 2) + (e & 1) * O1_PLANE + (e >> 1) * 16)
                act += cycles(ad, 2, G32, 32); ideal += 2
res["conv1 ds_write_b16 x4/tile"] = (ideal, act)
# conv2 A reads (b128) + W2B reads
ideal = act = 0
for wave in range(4):
    for ks in range(18):
        for i in range(5):
            ad = []
            for l in range(64):
                n16, g4 = l & 15, l >> 4
                p = (5 * wave + i) * 16 + n16; oy, ox = p // W2o, p % W2o
                ad.append(OFF_O1 + o1_cell(g4 & 1, 2 * oy, 2 * ox) + (g4 >> 1) * O1_PLANE + ((ks // 3) * O1_XH + ks % 3) * 16)
            act += cycles(ad, 16, G128, 64); ideal += 4
res["conv2 A ds_read_b128"] = (ideal, act)
res["conv2 W ds_read_b128"] = (4 * 18 * 4, 4 * 18 * 4)
# conv2 epilogue: ds_write_b16 x 8 per tile
ideal = act = 0
for wave in range(4):
    for i in range(5):
        for nt in range(2):
            for e in range(4):
                ad = []
                for l in range(64):
                    n16, g4 = l & 15, l >> 4
                    pp = (5 * wave + i) * 16 + 4 * g4; y, x = pp // W2o, pp % W2o
                    ad.append(OFF_O2 + (n16 >> 3) * 2 * O2_PLANE + (n16 & 7) * 2 + o2_cell(0, y + 2, x + 2) + (nt * 4 + (e & 1)) * O2_PLANE + (e >> 1) * 16)
                act += cycles(ad, 2, G32, 32); ideal += 2
res["conv2 ds_write_b16 x8/tile"] = (ideal, act)
# conv3 A reads per slice (2 np waves read the same)
ideal = act = 0
for np_ in range(2):
    for tap in range(36):
        for i in range(5):
            ad = []
            for l in range(64):
                n16, g4 = l & 15, l >> 4
                q = i * 16 + n16; oy, ox = q // W3, q % W3
                ad.append(OFF_O2 + o2_cell(g4, 2 * oy, 2 * ox) + o2_cell(0, tap // 6, tap % 6))
            act += cycles(ad, 16, G128, 64); ideal += 4
res["conv3 A ds_read_b128"] = (ideal, act)
ti = ta = 0
for k, (a, b) in res.items():
    print(f"{k:32s} ideal {a:6d}  modelled {b:6d}  x{b / a:.2f}")
    ti += a; ta += b
print(f"{'total LDS-array cycles / slice':32s} ideal {ti:6d}  modelled {ta:6d}")

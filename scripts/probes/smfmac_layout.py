"""Recover the operand layout of v_smfmac_i32_32x32x64_i8 from the dump of scripts/probes/smfmac_probe.hip: tries candidate
mappings of (lane, byte) -> (row / column, k) for the compressed A, its 2-bit indices, and the dense B, against the device's D."""
import itertools, sys
import numpy as np
raw = np.fromfile(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/smfmac_dump.bin", dtype=np.int32)
A = raw[:256].view(np.int8).reshape(64, 16).astype(np.int64)          # [lane][value]
B = raw[256:768].view(np.int8).reshape(64, 32).astype(np.int64)       # [lane][byte]
IDX = raw[768:832].view(np.uint32)                                     # [lane]
D = raw[832:].reshape(64, 16).astype(np.int64)                         # [lane][reg]
Dm = np.zeros((32, 32), np.int64)
for l in range(64):
    for r in range(16):
        Dm[4 * (l >> 5) + (r & 3) + 8 * (r >> 2), l & 31] = D[l, r]    # the dense 32x32 C/D map (row = 8 (r / 4) + 4 (lane / 32) + r % 4)

def a_k(l, v, mode):
    h, g, pos = l >> 5, v >> 1, (int(IDX[l]) >> (2 * v)) & 3
    if mode == "half32":      # lane half h owns k in [32 h, 32 h + 32): 8 groups of 4, two values per group
        return 32 * h + 4 * g + pos
    if mode == "interleave16":  # values 0-7 -> k block 16 (2 * 0 + h)?, values 8-15 -> the next
        return 16 * (2 * (v >> 3) + h) + 4 * ((v >> 1) & 3) + pos
    if mode == "interleave8":
        return 8 * (2 * (v >> 2) + h) + 4 * ((v >> 1) & 1) + pos
def b_k(l, byte, mode):
    h = l >> 5
    if mode == "half32":
        return 32 * h + byte
    if mode == "interleave16":
        return 16 * (2 * (byte >> 4) + h) + (byte & 15)
    if mode == "interleave8":
        return 8 * (2 * (byte >> 3) + h) + (byte & 7)
for am, bm in itertools.product(("half32", "interleave16", "interleave8"), repeat=2):
    Ad = np.zeros((32, 64), np.int64)
    Bd = np.zeros((64, 32), np.int64)
    ok = True
    for l in range(64):
        for v in range(16):
            Ad[l & 31, a_k(l, v, am)] += A[l, v]
        for byte in range(32):
            Bd[b_k(l, byte, bm), l & 31] = B[l, byte]
    match = np.array_equal(Ad @ Bd, Dm)
    print(f"A {am:13s} B {bm:13s}: {'MATCH' if match else 'no'}  (mismatching entries {int((Ad @ Bd != Dm).sum())})")

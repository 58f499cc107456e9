#!/usr/bin/env python3
"""Where the waves of the two-pass kernel's workgroups ran.  Input: the per-workgroup stamps written with
ALACGPU_DEBUG_STAMPS=1 ALACGPU_DEBUG_STAMPS_FILE=<file> (slots 3, 5, 6 = XCC_ID << 32 | HW_ID of waves 0, 1, 2 of each
workgroup).  Prints, for the CUs that hold two workgroups, how far the second one's first wave sits from the first one's
in the dispatcher's SIMD round (0, 2, 1, 3), and on how many CUs two heavy waves would share a SIMD if the roles went by
wave index (entropy, output, FIR) -- the measurement behind the role assignment in ab_kernel_body ("Placement").
usage: python tools/placement_log.py <file>"""
import numpy as np, sys, collections
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 3] != 0]
def dec(x):
    hw = x & 0xFFFFFFFF; xcc = x >> 32
    return (hw >> 4) & 3, ((xcc & 0xF) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
s0, cu0 = dec(a[:, 3]); s1, cu1 = dec(a[:, 5]); s2, cu2 = dec(a[:, 6])
pos = {0: 0, 2: 1, 1: 2, 3: 3}
d = collections.defaultdict(list)
for i in range(len(a)): d[int(cu0[i])].append((i, int(s0[i]), int(s1[i]), int(s2[i])))
steps = collections.Counter(); share = collections.Counter()
for cu, l in d.items():
    l.sort()
    if len(l) == 2:
        steps[(pos[l[1][1]] - pos[l[0][1]]) & 3] += 1
        heavy = [l[0][1], l[0][3], l[1][1], l[1][3]]       # E0 F0 E1 F1
        share[len(set(heavy)) < 4] += 1
    else:
        steps[('n', len(l))] += 1
print("CUs", len(d), "start-position step between the two workgroups of a CU:", dict(steps), " CUs where two heavy waves share a SIMD:", share[True])

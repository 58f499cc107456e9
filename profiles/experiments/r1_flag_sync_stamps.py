"""Diagnostic: reads the per-workgroup stamps of the two-pass kernel (ALACGPU_DEBUG_STAMPS_FILE) -- how often each
wave found the wave before / after it not ready when it came to a chunk, and what the entropy wave's units were."""
import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
t = (a[:, 2] - a[:, 0]).astype(float)
print("wgs", len(a), "E total mean %.0f max %.0f" % (t.mean(), t.max()))
names = [(5, "E had to wait for F (queue full)"), (6, "F had to wait for E"), (7, "tier-3 units"), (3, "generic units")]
cols = []
for j, name in names:
    v = a[:, j].astype(float)
    cols.append(v)
    print("%-34s mean %.1f  max %.0f   corr with E total %.2f" % (name, v.mean(), v.max(), np.corrcoef(t, v)[0, 1]))
X = np.stack([np.ones_like(t), cols[2], cols[3], cols[0]], axis=1)
coef, *_ = np.linalg.lstsq(X, t, rcond=None)
print("fit: E total = %.0f + %.0f * tier3 + %.0f * generic + %.0f * waits-for-F;  residual rms %.0f" %
      (coef[0], coef[1], coef[2], coef[3], np.sqrt(np.mean((X @ coef - t) ** 2))))
o = np.argsort(t)[-8:]
print("slowest 8 WGs: total, waitsF, waitsE, tier3, generic")
for i in o: print("  %.0f %d %d %d %d" % (t[i], a[i, 5], a[i, 6], a[i, 7], a[i, 3]))
o = np.argsort(t)[:4]
print("fastest 4 WGs:")
for i in o: print("  %.0f %d %d %d %d" % (t[i], a[i, 5], a[i, 6], a[i, 7], a[i, 3]))

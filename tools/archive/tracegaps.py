"""Kernel durations and inter-kernel gaps over time from a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).
usage: python tools/tracegaps.py <dir or csv> [kernel-name-substring-of-the-anchor]
Prints, for the longest run of back-to-back launches (the bench's timed region): per kernel the mean duration in the first / middle /
last tenth, the mean gap before it, and the period of the anchor kernel."""
import csv, glob, os, re, sys
from collections import defaultdict

src = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'pair_fused_kernel'
paths = [src] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for p in paths:
    for r in csv.DictReader(open(p)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
# the timed region = the longest stretch without a gap > 30 us
best, cur = [], []
for i, r in enumerate(rows):
    if cur and r[0] - cur[-1][1] > 30_000:
        if len(cur) > len(best):
            best = cur
        cur = []
    cur.append(r)
if len(cur) > len(best):
    best = cur
print(f'{len(rows)} launches, longest back-to-back stretch {len(best)} launches, {(best[-1][1] - best[0][0]) / 1e6:.2f} ms')
def short(n):
    m = re.search(r'(\w+)\s*(<[^()]*>)?\(', n)
    return (m.group(1) + (m.group(2) or ''))[:42] if m else n[:42]
dur, gap = defaultdict(list), defaultdict(list)
for i, (s, e, n) in enumerate(best):
    dur[short(n)].append((s, e - s))
    if i:
        gap[short(n)].append(s - best[i - 1][1])
for n, v in dur.items():
    k = max(1, len(v) // 10)
    d = [x[1] for x in v]
    f = lambda a: sum(a) / len(a) / 1e3
    print(f'{n:42s} n={len(v):5d} dur us: first10% {f(d[:k]):7.1f}  mid {f(d[len(d)//2 - k//2: len(d)//2 + k//2 + 1]):7.1f}  last10% {f(d[-k:]):7.1f}  '
          f'mean {f(d):7.1f}  min {min(d)/1e3:7.1f} | gap before: mean {f(gap[n]) if gap[n] else 0:6.1f}')
a = [x[0] for n, v in dur.items() if anchor in n for x in v]
if len(a) > 2:
    print(f'period of {anchor}: {(a[-1] - a[0]) / (len(a) - 1) / 1e3:.1f} us')
if '--dump' in sys.argv:
    d = [x[1] / 1e3 for n, v in dur.items() if anchor in n for x in v]
    print('durations (us) of', anchor, 'launches 100..139:', ' '.join(f'{x:.0f}' for x in d[100:140]))
    ds = sorted(d)
    print('percentiles 5/25/50/75/95:', ' '.join(f'{ds[int(len(ds) * q)]:.1f}' for q in (0.05, 0.25, 0.5, 0.75, 0.95)))
    print('even launches mean %.1f  odd launches mean %.1f' % (sum(d[0::2]) / len(d[0::2]), sum(d[1::2]) / len(d[1::2])))

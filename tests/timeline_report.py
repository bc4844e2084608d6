"""Reads a rocprofv3 kernel_trace.csv of the two-stream flow (tests/gpu_timeline.sh) and prints, for the SHORTEST compress call in it,
one line per kernel launch (start relative to the call's first kernel, duration, stream/queue, grid) plus how much of the
call's span had 0 / 1 / 2+ kernels in flight.  python tests/timeline_report.py gpurun_out/timeline/tl_enwik_kernel_trace.csv [--all]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
K = []
for r in rows:
    K.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Queue_Id', '?'),
              (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * (int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y']))) * int(r['Grid_Size_Z']), int(r['Workgroup_Size_X'])))
K.sort()
# calls start with k0_tile_last / the K0 pre-pass: split on the first kernel name of the trace
first = [i for i, k in enumerate(K) if k[2] == K[0][2]]
# steps: a new step starts where the gap to the previous kernel's end exceeds 300 us
steps, cur, last_end = [], [], None
for k in K:
    if last_end is not None and k[2] == 'k0_tile_last' and cur:
        steps.append(cur); cur = []
    cur.append(k); last_end = max(last_end or 0, k[1])
if cur: steps.append(cur)
S = min((st for st in steps if any(k[2] == 'k1f_bsort' for k in st)), key=lambda st: max(k[1] for k in st) - st[0][0])   # the shortest of the traced calls (the tracer's own hiccups lengthen some)
ends = [i for i, k in enumerate(S) if k[2] == 'k5_end']       # the call ends with k5_end (+ the read-back of the stream state)
if ends: S = S[:min(len(S), ends[0] + 2)]
t0 = S[0][0]
span = max(k[1] for k in S) - t0
print('kernels %d span %.3f ms' % (len(S), span / 1e6))
ev = []
for k in S:
    ev.append((k[0], 1)); ev.append((k[1], -1))
ev.sort()
lvl, prev, occ = 0, t0, {}
for t, d in ev:
    occ[min(lvl, 3)] = occ.get(min(lvl, 3), 0) + (t - prev); prev = t; lvl += d
print('in flight: ' + '  '.join('%d%s: %.3f ms' % (l, '+' if l == 3 else '', v / 1e6) for l, v in sorted(occ.items())))
if '--all' in sys.argv:
    for k in S:
        print('%9.1f us  %8.1f us  q%-3s %6d x %-4d %s' % ((k[0] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[4], k[5], k[2]))

"""profiles/sinkhorn_roofline.json from the round's SQ counter summaries + kernel stats of the stand-alone Sinkhorn kernel
(tools/profile_r2.sh).  The kernel is VALU / transcendental issue bound: its roof is the SIMDs' VALU pipe being busy every
cycle, achieved = SQ_ACTIVE_INST_VALU (quad-cycles, MI355X_MICROARCH.md) * 4 / (SIMDs * kernel cycles).
usage: python tools/make_sinkhorn_roofline.py"""
import csv, json, re

SIMDS, CLOCK_GHZ = 1024, 2.4
out = {}
for shape, pairs, label in (('32x50000x8', 32 * 50000, 'config 3 shape'), ('128x8192x12', 128 * 8192, 'config 5 slice')):
    cnt, cur, disp = {}, None, 0
    for line in open(f'profiles/r02_sinkhorn_{shape}_sq_counters.txt'):
        if 'dispatches' in line:
            cur = 'sinkhorn_block_kernel' in line
            if cur:
                disp += int(line.split('dispatches')[1])
        elif cur:
            k, v = line.split()[0], float(line.split()[1])
            cnt[k] = cnt.get(k, 0.0) + v          # summed over the dispatches of ONE scoring call (the call runs in candidate
                                                  # chunks; a small last chunk may take another lanes-per-pair layout)
    total_ns = calls = 0
    names = []
    for r in csv.DictReader(open(f'profiles/r02_ot_l2max_{shape}_kernel_stats.csv')):
        if 'sinkhorn_block_kernel' in r['Name']:
            total_ns += float(r['TotalDurationNs'])
            calls += int(r['Calls'])
            names.append((float(r['TotalDurationNs']), re.search(r'sinkhorn_block_kernel<[^>]*>', r['Name']).group(0)))
    name = ' + '.join(n for _, n in sorted(names, reverse=True))
    reps = calls // disp                      # the stats run repeats the call
    kernel_us = total_ns / reps / 1e3         # all chunks of one call
    valu_cycles = cnt['SQ_ACTIVE_INST_VALU'] * 4
    busy = valu_cycles / (SIMDS * kernel_us * 1e-6 * CLOCK_GHZ * 1e9)
    out[shape] = {
        'what': label, 'kernel': name, 'pairs': pairs, 'kernel_us_per_call': kernel_us, 'ns_per_pair': kernel_us * 1e3 / pairs,
        'valu_wave_instructions_per_pair': cnt['SQ_INSTS_VALU'] / pairs,
        'valu_busy_cycles_per_simd': valu_cycles / SIMDS,
        'bound': 'valu-issue', 'achieved_frac': busy,
        'issue_floor_us': valu_cycles / SIMDS / (CLOCK_GHZ * 1e3),
        'counters': cnt,
    }
out['note'] = ('achieved_frac = VALU-busy cycles / (1024 SIMDs x kernel duration at 2.4 GHz): the fraction of the issue roof the kernel '
               'runs at.  Sources: profiles/r02_sinkhorn_*_sq_counters.txt (rocprofv3 --pmc, one scoring call) and '
               'profiles/r02_ot_l2max_*_kernel_stats.csv (rocprofv3 --kernel-trace --stats, five calls).')
json.dump(out, open('profiles/sinkhorn_roofline.json', 'w'), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x != 'counters'})

"""Rebuild profiles/traffic.json from the round's rocprofv3 PMC summaries (tools/profile_r2.sh -> tools/pmcsum.py).
usage: python tools/make_traffic.py profiles/r02_bench_20x1000_fetch_size.txt profiles/r02_bench_20x1000_write_size.txt \
                                    profiles/r02_bench_20x1000_kernel_stats.csv"""
import csv, json, re, sys

fetch_txt, write_txt, stats_csv = sys.argv[1:4]
K, NC, S, D = 20, 1000, 8, 768


def per_dispatch(path, counter, kernel):
    cur = None
    for line in open(path):
        if 'dispatches' in line:
            cur = line
        elif counter in line and cur and kernel in cur:
            return float(line.split('per dispatch')[1])
    raise KeyError((path, counter, kernel))


fetch_kb = per_dispatch(fetch_txt, 'FETCH_SIZE', 'pair_fused_kernel')
write_kb = per_dispatch(write_txt, 'WRITE_SIZE', 'pair_fused_kernel')
stats = {}
for r in csv.DictReader(open(stats_csv)):
    m = re.search(r'(pair_fused_kernel|topk_select_kernel|batch_prep_kernel)', r['Name'])
    if m and m.group(1) not in stats:
        stats[m.group(1)] = dict(avg_us=float(r['AverageNs']) / 1e3, min_us=float(r['MinNs']) / 1e3, calls=int(r['Calls']))
alg = K * (4 * D * (NC * S + S) + 4 * NC)
traffic = int(2 * fetch_kb * 1024 + write_kb * 1024)
json.dump({
    'round': 2,
    'command': 'rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python bench.py --steps 20 --warmup 5 '
               '--repeats 6 --streams 1 --no-cpu-baseline --no-probes   (tools/profile_r2.sh; summed per kernel by tools/pmcsum.py)',
    'workload': 'bench.py: 20 jobs x (1 query x 1000 candidates x 8 sents x 768 d) per aspire_ot_rank_batch_f32 call; the scoring launch = '
                'pair_fused_kernel<true, true, true> (costs + Sinkhorn solves, in-wave tables), rotating cold pools, one call at a time',
    'jobs_per_launch': K,
    'FETCH_SIZE_KB_per_launch': fetch_kb, 'WRITE_SIZE_KB_per_launch': write_kb,
    'correction': 'MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced (16 B/lane) '
                  'read stream -> doubled; WRITE_SIZE uncalibrated, taken as is; Infinity-Cache hits are counted too, but every launch reads '
                  '492 MB of pools that were last touched 492 MB ago (two alternating sets): nothing survives in the 256 MiB L3',
    'cost_kernel_hbm_bytes_per_launch': traffic,
    'algorithmic_bytes_per_launch': alg,
    'ratio': traffic / alg,
    'kernel_stats': {'source': stats_csv + ' (rocprofv3 --kernel-trace --stats on bench.py --steps 20 --warmup 5 --repeats 60 --streams 1)', **stats},
}, open('profiles/traffic.json', 'w'), indent=1)
print(traffic, alg, traffic / alg, stats)

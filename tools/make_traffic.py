"""Rebuild profiles/traffic.json + the kernel-stats copy from a gpurun_out profile set.
usage: python tools/make_traffic.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir>"""
import csv, collections, json, shutil, sys
stats_dir, fdir, wdir = sys.argv[1:4]
shutil.copy(f'{stats_dir}/bench_kernel_stats.csv', 'profiles/r01_bench_ot1x1000_kernel_stats.csv')
agg = collections.defaultdict(list)
for d in (fdir, wdir):
    for r in csv.DictReader(open(f'{d}/pmc_counter_collection.csv')):
        n = r['Kernel_Name']
        k = 'pair_cost' if 'pair_cost' in n else 'sinkhorn' if 'sinkhorn_kernel' in n else None
        if k: agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
m = {k: sum(v) / len(v) for k, v in agg.items()}
stats = {}
for r in csv.DictReader(open('profiles/r01_bench_ot1x1000_kernel_stats.csv')):
    n = r['Name']
    k = 'pair_cost' if 'pair_cost' in n else 'sinkhorn' if 'sinkhorn_kernel' in n else 'topk' if 'topk' in n else None  # topk_select_kernel or topk_pass_kernel
    if k: stats[k] = (float(r['AverageNs']), int(r['Calls']), n.split('(anonymous namespace)::')[1].split('(')[0] if '(anonymous namespace)::' in n else n)
fetch = {k: m[(k, 'FETCH_SIZE')] for k in ('pair_cost', 'sinkhorn')}
write = {k: m[(k, 'WRITE_SIZE')] for k in ('pair_cost', 'sinkhorn')}
traffic = int(sum(2 * fetch[k] * 1024 + write[k] * 1024 for k in fetch))
json.dump({
    'round': 1,
    'command': 'rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python bench.py --steps 50 --no-graph --no-cpu-baseline',
    'workload': '1 query x 1000 candidates x 8 sents x 768 d; one aspire_ot_sinkhorn_f32 call = cost kernel + sinkhorn_kernel<1>; hbm_bytes_per_launch = both kernels, cost_kernel_hbm_bytes_per_launch = the cost kernel alone',
    'FETCH_SIZE_mean_KB': fetch, 'WRITE_SIZE_mean_KB': write,
    'correction': 'MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced (16 B/lane) read stream -> doubled; WRITE_SIZE uncalibrated, taken as is',
    'hbm_bytes_per_launch': traffic, 'cost_kernel_hbm_bytes_per_launch': int(2 * fetch['pair_cost'] * 1024 + write['pair_cost'] * 1024),
    'algorithmic_bytes_per_launch': 24604576,
    'breakdown': {'source': 'profiles/r01_bench_ot1x1000_kernel_stats.csv (rocprofv3 --kernel-trace --stats on bench.py --steps 480 --streams 1)',
                  'cost_kernel': stats['pair_cost'][2], 'cost_kernel_us': stats['pair_cost'][0] / 1e3,
                  'sinkhorn_kernel_us': stats['sinkhorn'][0] / 1e3, 'topk_pass_kernel_us': stats['topk'][0] / 1e3,
                  'cost_kernel_GBs': 24604576 / stats['pair_cost'][0]},
}, open('profiles/traffic.json', 'w'), indent=1)
print(json.dumps(json.load(open('profiles/traffic.json'))['breakdown']), traffic / 24604576)

"""Copy the summaries of `bash tools/profile_r6.sh` (gpurun_out/prof6) into profiles/ under round-6 names and rebuild the derived
json files (profiles/traffic.json, profiles/sinkhorn_roofline.json, profiles/r06_summary.json).   python tools/collect_r6.py"""
import csv, glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'prof6')
DST = os.path.join(ROOT, 'profiles')
SIMDS, CLOCK_GHZ = 1024, 2.4


def stats_csv(name):
    f = glob.glob(os.path.join(SRC, name, '**', '*kernel_stats.csv'), recursive=True)
    return f[0] if f else None


def copy_stats(name, out):
    f = stats_csv(name)
    if f:
        shutil.copy(f, os.path.join(DST, out))
    return f


def counters(summary, kernel):
    """{counter: per-dispatch value} of the first kernel whose name contains `kernel` (tools/pmcsum.py output)"""
    out, cur = {}, False
    path = os.path.join(SRC, summary + '.summary.txt')
    if not os.path.exists(path):
        return out
    for line in open(path):
        if 'dispatches' in line:
            cur = kernel in line
        elif cur and 'per dispatch' in line:
            out[line.split()[0]] = float(line.split('per dispatch')[1])
    return out


def kernel_rows(path, pattern):
    rows = []
    if path:
        for r in csv.DictReader(open(path)):
            if re.search(pattern, r['Name']):
                m = re.search(r'(\w+_kernel)(<[^(]*>)?', r['Name'])
                rows.append(dict(name=(m.group(0) if m else r['Name'])[:80], calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3,
                                 min_us=float(r['MinNs']) / 1e3))
    return rows


summary = {}
# ---- 1. the bench workload ----------------------------------------------------------------------------------------------------
bench = copy_stats('bench_stats', 'r06_bench_20x1000_kernel_stats.csv')
copy_stats('bench_stats_s3', 'r06_bench_20x1000_3streams_kernel_stats.csv')
for s, o in (('pmc_fetch', 'r06_bench_20x1000_fetch_size.txt'), ('pmc_write', 'r06_bench_20x1000_write_size.txt'),
             ('pmc_sq_bench', 'r06_bench_20x1000_sq_counters.txt')):
    p = os.path.join(SRC, s + '.summary.txt')
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, o))
K, NC, S, D = 20, 1000, 8, 768
alg = K * (4 * D * (NC * S + S) + 4 * NC)
fetch, write = counters('pmc_fetch', 'pair_fused_kernel'), counters('pmc_write', 'pair_fused_kernel')
sq = counters('pmc_sq_bench', 'pair_fused_kernel')
fused = kernel_rows(bench, r'pair_fused_kernel<true, true, true')
traffic = {}
if fetch and write and fused:
    hbm = int(2 * fetch['FETCH_SIZE'] * 1024 + write['WRITE_SIZE'] * 1024)
    kern_us = fused[0]['avg_us']
    valu_busy = sq.get('SQ_ACTIVE_INST_VALU', 0) * 4 / (SIMDS * kern_us * 1e-6 * CLOCK_GHZ * 1e9)
    traffic = {
        'round': 6,
        'command': 'rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, --pmc WRITE_SIZE) --output-format csv -- python bench.py --steps 20 '
                   '--warmup 5 --repeats 6 --streams 1 --no-cpu-baseline --no-probes   (tools/profile_r6.sh; summed per kernel by tools/pmcsum.py)',
        'workload': 'bench.py: 20 jobs x (1 query x 1000 candidates x 8 sents x 768 d) per aspire_ot_rank_batch_f32 call; the scoring launch = '
                    'pair_fused_kernel<true, true, true> (costs + Sinkhorn solves, in-wave tables), rotating cold pools, one call at a time',
        'jobs_per_launch': K, 'FETCH_SIZE_KB_per_launch': fetch['FETCH_SIZE'], 'WRITE_SIZE_KB_per_launch': write['WRITE_SIZE'],
        'correction': 'MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced (16 B/lane) read '
                      'stream -> doubled; WRITE_SIZE taken as is',
        'cost_kernel_hbm_bytes_per_launch': hbm, 'algorithmic_bytes_per_launch': alg, 'ratio': hbm / alg,
        'kernel_stats': {'source': 'profiles/r06_bench_20x1000_kernel_stats.csv', 'pair_fused_kernel': fused[0],
                         'topk_select_kernel': (kernel_rows(bench, r'topk_select_kernel') or [None])[0]},
        'valu_busy': {'convention': 'SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x kernel duration x 2.4 GHz) -- the ONE convention of '
                                    'this repo (NOTES.md round-2 text that says 29 % divided by per-wave cycles)',
                      'value': valu_busy, 'SQ_ACTIVE_INST_VALU': sq.get('SQ_ACTIVE_INST_VALU'), 'SQ_INSTS_VALU': sq.get('SQ_INSTS_VALU'),
                      'SQ_WAIT_INST_ANY': sq.get('SQ_WAIT_INST_ANY'), 'SQ_WAIT_ANY': sq.get('SQ_WAIT_ANY'), 'SQ_WAVE_CYCLES': sq.get('SQ_WAVE_CYCLES')},
    }
    # ---- config 4 ---------------------------------------------------------------------------------------------------------------
    c4 = {}
    for m, kern in (('ot', 'pair_fused_kernel'), ('l2max', 'pair_fused_kernel')):
        st = copy_stats(f'csf_{m}_stats', f'r06_csf_50x125_{m}_kernel_stats.csv')
        for kind in ('fetch', 'write'):
            p = os.path.join(SRC, f'csf_{m}_{kind}.summary.txt')
            if os.path.exists(p):
                shutil.copy(p, os.path.join(DST, f'r06_csf_50x125_{m}_{kind}_size.txt'))
        f_, w_ = counters(f'csf_{m}_fetch', kern), counters(f'csf_{m}_write', kern)
        log = os.path.join(SRC, f'csf_{m}_stats.log')
        algb = None
        if os.path.exists(log):
            mm = re.search(r'algorithmic bytes per call (\d+)', open(log).read())
            algb = int(mm.group(1)) if mm else None
        rows = kernel_rows(st, r'aspire::')
        call_us = sum(r['avg_us'] for r in rows if r['calls'] >= 40)
        if f_ and w_ and algb:
            hb = int(2 * f_['FETCH_SIZE'] * 1024 + w_['WRITE_SIZE'] * 1024)
            c4[m] = {'kernels': rows, 'sum_of_kernel_avgs_us': call_us, 'scoring_kernel_hbm_bytes_per_launch': hb,
                     'algorithmic_bytes_per_call': algb, 'ratio': hb / algb,
                     'note': 'the 212 MiB of reps fit the 256 MiB Infinity Cache and the same data is read by every call of this run: FETCH_SIZE '
                             'counts the L2 misses whether the Infinity Cache or HBM serves them'}
    p = os.path.join(SRC, 'csf_ot_sq.summary.txt')
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, 'r06_csf_50x125_ot_sq_counters.txt'))
    traffic['config4'] = c4
    json.dump(traffic, open(os.path.join(DST, 'traffic.json'), 'w'), indent=1)
summary['bench'] = {'fused_kernel': fused[:1], 'traffic_ratio': traffic.get('ratio')}
summary['config4'] = traffic.get('config4')

# ---- 2. encoder / GEMM ------------------------------------------------------------------------------------------------------------
enc = copy_stats('enc_stats', 'r06_encoder_B32_L256_kernel_stats.csv')
copy_stats('gemm_stats', 'r06_gemm_8192x2304x768_kernel_stats.csv')
with open(os.path.join(DST, 'r06_gemm_8192x2304x768_sq_counters.txt'), 'w') as out:
    out.write('# rocprofv3 --pmc (three passes: two SQ sets + GRBM) -- python tools/gemmprof.py 8192 2304 768 5 planes : 5 launches of '
              'gemm_p_kernel<2, 128, false>; summed per kernel by tools/pmcsum.py\n')
    for s in ('gemm_pmc1', 'gemm_pmc2', 'gemm_pmc3'):
        p = os.path.join(SRC, s + '.summary.txt')
        if os.path.exists(p):
            out.write(open(p).read())
g1, g3 = counters('gemm_pmc1', 'gemm_p_kernel'), counters('gemm_pmc3', 'gemm_p_kernel')
gk = kernel_rows(stats_csv('gemm_stats'), r'gemm_p_kernel')
if g1 and gk:
    us = gk[0]['avg_us']
    busy = g1['SQ_VALU_MFMA_BUSY_CYCLES'] / SIMDS
    gui = g3.get('GRBM_GUI_ACTIVE')
    gui = gui / 8 if gui else None          # rocprofv3 sums the counter over the chip's 8 XCDs (one GRBM each)
    summary['gemm'] = {'kernel': gk[0], 'mfma_busy_cycles_per_simd': busy, 'mfma_busy_frac_at_2.4GHz': busy / (us * 1e-6 * CLOCK_GHZ * 1e9),
                       'GRBM_GUI_ACTIVE_per_launch_per_xcd': gui, 'effective_clock_GHz': gui / (us * 1e3) if gui else None,
                       'mfma_busy_frac_at_effective_clock': busy / gui if gui else None,
                       'note': 'SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against the launch duration at the nominal 2.4 GHz and against the cycles '
                               'the chip actually ran (GRBM_GUI_ACTIVE / 8 XCDs, a separate counter pass of the same five launches)',
                       'wave_cycle_split': {k: g1.get(k) for k in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU')}}
with open(os.path.join(DST, 'r06_attn_B32_L256_sq_counters.txt'), 'w') as out:
    out.write('# rocprofv3 --pmc (two SQ passes) -- python tools/encbench.py 32 256 (13 forwards of 12 layers); summed per kernel by tools/pmcsum.py; '
              'the flash_attn_p_kernel block of each pass (round 6: operands as planes from the QKV GEMM, LDS-DMA staging, transpose reads)\n')
    for s_ in ('attn_pmc1', 'attn_pmc2'):
        p_ = os.path.join(SRC, s_ + '.summary.txt')
        if os.path.exists(p_):
            keep = False
            for line in open(p_):
                if 'dispatches' in line:
                    keep = 'flash_attn' in line
                if keep:
                    out.write(line)
p_ = os.path.join(SRC, 'e2e_full_share.log')
if os.path.exists(p_):
    shutil.copy(p_, os.path.join(DST, 'r06_e2e_full_share_125000.txt'))
for name in ('gemmbench', 'poolbench'):
    p = os.path.join(SRC, name + '.log')
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f'r06_{name}.txt'))
summary['encoder'] = kernel_rows(enc, r'aspire::')
# ---- 3. end to end, pooling -------------------------------------------------------------------------------------------------------
e2e = copy_stats('e2e_stats', 'r06_e2e_kernel_stats.csv')
p = os.path.join(SRC, 'e2e_stats.log')
if os.path.exists(p):
    txt = open(p).read()
    i = txt.find('{')
    if i >= 0:
        obj, _ = json.JSONDecoder().raw_decode(txt[i:])
        json.dump(obj, open(os.path.join(DST, 'r06_e2e.json'), 'w'), indent=1)
        summary['e2e'] = obj
copy_stats('pool_stats', 'r06_pool_B256_L512_kernel_stats.csv')
# ---- 4. the stand-alone Sinkhorn kernel: VALU busy AND an algorithmic floor ---------------------------------------------------------
sink = {}
for shape, pairs, rows, label in (('32x50000x8', 32 * 50000, 8, 'config 3 shape'), ('128x8192x12', 128 * 8192, 12, 'config 5 slice')):
    copy_stats(f'ot_{shape}', f'r06_ot_l2max_{shape}_kernel_stats.csv')
    # (round 6, VERDICT r5 item 4: the Sinkhorn kernel's duration AND counters come from the plane-store call -- tools/planeprof.py ... planes ot,
    # what bench.py's config-5 block and the many-query calls run -- not from the fp32-row call of otprof.py)
    st = stats_csv(f'ot_planes_{shape}')
    p = os.path.join(SRC, f'pmc_sink_{shape}.summary.txt')
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f'r06_sinkhorn_{shape}_sq_counters.txt'))
    cnt, cur, disp = {}, None, 0
    if os.path.exists(p):
        for line in open(p):
            if 'dispatches' in line:
                cur = 'sinkhorn_block_kernel' in line
                if cur:
                    disp += int(line.split('dispatches')[1])
            elif cur and 'per dispatch' in line:
                k, v = line.split()[0], float(line.split()[1])
                cnt[k] = cnt.get(k, 0.0) + v
    total_ns = calls = 0
    names = []
    if st:
        for r in csv.DictReader(open(st)):
            if 'sinkhorn_block_kernel' in r['Name']:
                total_ns += float(r['TotalDurationNs'])
                calls += int(r['Calls'])
                names.append((float(r['TotalDurationNs']), re.search(r'sinkhorn_block_kernel<[^>]*>', r['Name']).group(0)))
    if not cnt or not calls:
        continue
    reps = max(1, calls // max(disp, 1))
    kernel_us = total_ns / reps / 1e3
    valu_cycles = cnt['SQ_ACTIVE_INST_VALU'] * 4
    # the algorithmic floor: one exponential per entry and one logarithm per row and per column and epsilon step are NECESSARY
    # (the schedule's length is the reference's: ~ln(diam / blur) / -ln(scaling) + 3 = ~77 steps on N(0, 1) data), a transcendental
    # issues every 8 cycles (measured, NOTES.md section 3); plus three multiply-adds per entry and step (exponent, row sum, column sum)
    # at the plain VALU rate of 4 cycles per wave-instruction.  64 lanes of work per wave-instruction.
    steps, ent = 77, rows * rows
    need_trans = pairs * steps * (ent + 2 * rows) / 64.0
    need_fma = pairs * steps * 3 * ent / 64.0
    floor_cycles = (need_trans * 8 + need_fma * 4) / SIMDS
    trans = cnt.get('SQ_INSTS_VALU_TRANS')
    sink[shape] = {
        'what': label, 'kernel': ' + '.join(n for _, n in sorted(names, reverse=True)), 'pairs': pairs, 'kernel_us_per_call': kernel_us,
        'ns_per_pair': kernel_us * 1e3 / pairs, 'valu_wave_instructions_per_pair': cnt['SQ_INSTS_VALU'] / pairs,
        'transcendental_wave_instructions_per_pair': trans / pairs if trans else None,
        'bound': 'valu-issue', 'achieved_frac': valu_cycles / (SIMDS * kernel_us * 1e-6 * CLOCK_GHZ * 1e9),
        'issue_floor_us': valu_cycles / SIMDS / (CLOCK_GHZ * 1e3),
        'algorithmic_floor': {'steps': steps, 'necessary_transcendental_wave_instructions_per_pair': need_trans / pairs,
                              'necessary_fma_wave_instructions_per_pair': need_fma / pairs, 'floor_us': floor_cycles / (CLOCK_GHZ * 1e3),
                              'floor_over_kernel': floor_cycles / (CLOCK_GHZ * 1e3) / kernel_us,
                              'what': 'per pair and step S^2 exponentials + 2 S logarithms (8-cycle issue interval) and 3 S^2 multiply-adds '
                                      '(4 cycles), 64 lanes per wave-instruction, 1024 SIMDs at 2.4 GHz: what ANY kernel that follows the '
                                      "reference's schedule has to issue"},
        'counters': cnt,
    }
if sink:
    sink['note'] = ('achieved_frac = VALU-busy cycles (SQ_ACTIVE_INST_VALU x 4) / (1024 SIMDs x kernel duration at 2.4 GHz): how busy the issue '
                    "port is with the kernel's OWN instruction stream; algorithmic_floor.floor_over_kernel = how much of the kernel's time the "
                    'necessary transcendentals and multiply-adds alone account for.  Sources (round 6: both from the PLANE-STORE otAspire call, '
                    'python tools/planeprof.py Q C S N planes ot): profiles/r06_sinkhorn_*_sq_counters.txt, '
                    'profiles/r06_ot_planes_*_kernel_stats.csv (tools/profile_r6.sh).')
    json.dump(sink, open(os.path.join(DST, 'sinkhorn_roofline.json'), 'w'), indent=1)
copy_stats('ot_1x20000x12', 'r06_ot_l2max_1x20000x12_kernel_stats.csv')
summary['sinkhorn'] = {k: {x: v[x] for x in ('kernel_us_per_call', 'achieved_frac', 'algorithmic_floor')} for k, v in sink.items() if isinstance(v, dict)}
# ---- 5. the fp16-plane cost tiles (gramp.hip): config 3's shape, the config-5 slice, one query x 20 000 x 12 -----------------------
planes = {}
for name, label, flop, nbytes in (('planes_32x50000x8', 'config 3: tsAspire 32 x 50 000 x 8', 2.0 * 64 * 768 * 32 * 50000, 4 * 768 * (50000 * 8 + 256)),
                                  ('planes_1x20000x12', 'tsAspire 1 x 20 000 x 12', 2.0 * 144 * 768 * 20000, 4 * 768 * (20000 * 12 + 12))):
    st = copy_stats(name + '_stats', f'r06_{name}_kernel_stats.csv')
    with open(os.path.join(DST, f'r06_{name}_sq_counters.txt'), 'w') as out:
        out.write(f'# rocprofv3 --pmc (three passes: two SQ sets, GRBM_GUI_ACTIVE + TCC hits / misses) -- python tools/planeprof.py ... : 5 launches of '
                  f'pair_gram_p_kernel; summed per kernel by tools/pmcsum.py\n')
        for k in ('pmc1', 'pmc2', 'pmc3'):
            p = os.path.join(SRC, f'{name}_{k}.summary.txt')
            if os.path.exists(p):
                out.write(open(p).read())
    c1, c3 = counters(f'{name}_pmc1', 'pair_gram_p_kernel'), counters(f'{name}_pmc3', 'pair_gram_p_kernel')
    rows = kernel_rows(st, r'pair_gram_p_kernel')
    if rows and c1:
        us = rows[0]['avg_us']
        busy = c1['SQ_VALU_MFMA_BUSY_CYCLES'] / SIMDS
        gui = c3.get('GRBM_GUI_ACTIVE')
        gui = gui / 8 if gui else None
        planes[name] = {'what': label, 'kernel': rows[0], 'algorithmic_tflops': flop / us / 1e6, 'executed_mfma_tflops': 3 * flop / us / 1e6 if '32x' in name else None,
                        'algorithmic_GBs': nbytes / us / 1e3, 'mfma_busy_cycles_per_simd': busy,
                        'mfma_busy_frac_at_2.4GHz': busy / (us * 1e-6 * CLOCK_GHZ * 1e9),
                        'cycles_the_chip_ran_per_launch': gui, 'mfma_busy_frac_at_the_clock_held': busy / gui if gui else None,
                        'TCC_HIT': c3.get('TCC_HIT_sum'), 'TCC_MISS': c3.get('TCC_MISS_sum'),
                        'wave_cycle_split': {k: c1.get(k) for k in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_INSTS_VALU')}}
summary['plane_tiles'] = planes
for name in ('mfmapeak', 'planebench', 'fusedclock'):
    p = os.path.join(SRC, name + '.log')
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, f'r06_{name}.txt'))
copy_stats('ot_planes_32x50000x8', 'r06_ot_planes_32x50000x8_kernel_stats.csv')
copy_stats('ot_planes_128x8192x12', 'r06_ot_planes_128x8192x12_kernel_stats.csv')
# ---- 6. round 6: config 3's HBM traffic pass, its per-call series, the attention A/B ---------------------------------------------
with open(os.path.join(DST, 'r06_planes_32x50000x8_fetch_write_size.txt'), 'w') as out:
    out.write('# rocprofv3 --pmc FETCH_SIZE and, in its own pass, --pmc WRITE_SIZE -- python tools/planeprof.py 32 50000 8 3 planes (tools/profile_r6.sh); '
              'KB per dispatch, summed per kernel by tools/pmcsum.py.  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)\n')
    for cname in ('FETCH_SIZE', 'WRITE_SIZE'):
        cc = counters(f'planes_32x50000x8_{cname}', 'pair_gram_p_kernel')
        if cc:
            out.write('%s pair_gram_p_kernel per dispatch KB %.1f\n' % (cname, cc[cname]))
for src, dst in (('c3trace.json', 'r06_config3_per_call_trace.json'), ('c3trace_prof.summary.txt', 'r06_config3_per_call_trace_under_profiler.txt')):
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(DST, dst))
copy_stats('enc_stats_64x256', 'r06_encoder_B64_L256_kernel_stats.csv')
copy_stats('enc_stats_attn_f16x2', 'r06_encoder_B64_L256_attn_f16x2_kernel_stats.csv')
for src, dst in (('e2e_attn_f16x2.log', 'r06_e2e_attn_f16x2.json'), ('e2e_attn_planes.log', 'r06_e2e_attn_planes.json')):
    p = os.path.join(SRC, src)
    if os.path.exists(p):
        txt = open(p).read()
        i = txt.find('{')
        if i >= 0:
            obj, _ = json.JSONDecoder().raw_decode(txt[i:])
            json.dump(obj, open(os.path.join(DST, dst), 'w'), indent=1)
            summary[dst[:-5]] = {k: obj.get(k) for k in ('docs_per_s', 'encode_s', 'split_ms')}
            summary[dst[:-5]]['encoder_frac'] = obj.get('encoder_roofline', {}).get('frac')
json.dump(summary, open(os.path.join(DST, 'r06_summary.json'), 'w'), indent=1)
print(json.dumps(summary, indent=1)[:6000])

#!/bin/bash
# The N = 8 control flow of bench.py on ONE GPU: eight ranks on cuda:0 over gloo -- the headline's sharded schedule
# (key-form rank, all-gather, merge kernel), config 4 sharded by job (7 7 6 6 6 6 6 6), config 5's per-rank encode -> shard -> merge.
# INVALID as a performance result (eight processes share one GPU); what it records is that the 8-rank code path runs and agrees.
#   bash tools/experiments/rehearse8.sh [out.json]
out=${1:-gpurun_out/r6_rehearse8.json}
export ASPIRE_BENCH_ONE_GPU=1 MASTER_ADDR=127.0.0.1 ASPIRE_BENCH_E2E_DOCS=${ASPIRE_BENCH_E2E_DOCS:-1024} HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --steps 20 --warmup 5 --repeats 6 \
    2> ${out%.json}.stderr | grep '^{' | tail -1 > $out
python - "$out" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
r, c4, e = j['rccl'], j['config4'], j['e2e']
print('ranks seen', r['ranks_seen'], 'backend', r['backend'], 'merged ranking agrees', r['merged_ranking_agrees_on_all_ranks'], 'shards in merged top-k', r['shards_in_merged_top_k'])
print('config 4 by job:', [x['jobs'] for x in c4['ranks']], 'same result on all ranks', c4['all_ranks_hold_the_same_result'], 'order = one GPU', c4['order_equals_one_gpu'],
      'max |score diff|', c4['max_abs_score_diff_vs_one_gpu'])
print('config 5 per rank: merged top-1 agrees', e.get('merged_top1_agrees'), 'shards in top-k', [x['shards_in_top_k'] for x in e.get('ranks', [])])
PY

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fused.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -6
for np in 1 0; do echo "== FUSED_NOPAIR=$np"; ASPIRE_HIP_FUSED_NOPAIR=$np python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probes 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value %.1f M  one_stream %.1f M  kernel %.1f us  frac %.3f' % (j['value']/1e6, j['one_stream']['value']/1e6, j['roofline']['kernel_ms']*1e3, j['roofline']['frac']))"; done

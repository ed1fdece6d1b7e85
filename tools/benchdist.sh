#!/bin/bash
# distribution of bench.py's value over repeated process runs: tools/benchdist.sh "<streams list>" <runs>
for s in $1; do
  for i in $(seq 1 $2); do
    timeout -s KILL 60 python bench.py --no-cpu-baseline --streams $s 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams', '$s', round(d['value']/1e6,1), 'M/s')"
  done
done

#!/bin/bash
# headline step against the number of side streams and the window batch (bench.py --legs none); usage: tools/sweep_streams.sh "1 3" "128 192 384"
for st in ${1:-1 2 3 4 6}; do for wb in ${2:-64 128 192}; do
  r=$(ALIVE_STREAMS=$st python bench.py --legs none --cpu-seconds 0 --no-nets-roofline --window-batch $wb --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'])")
  echo "streams $st window_batch $wb: $r"
done; done

# usage: bash tools/ab_variants.sh "<lib> <lib> ..." -- same-box A/B of library variants with bench.py and the trainer ("-" = in-tree build)
for rep in 1 2; do
for v in $1; do
  lib=$v; [ "$v" = "-" ] && lib=""
  MBRL_LIB=$lib timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu --no-extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'bench', round(d['value']/1e6,3), 'M J/s', round(d['ms_per_step'],4), 'ms/step, kernel', round(d['roofline']['kernel_ms'],4))"
  echo -n "$v trainer "; MBRL_LIB=$lib BATCHES=4096 timeout -k 10 300 python3 tools/train_throughput.py 2>&1 | grep "^B"
done; done

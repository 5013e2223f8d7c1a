# same-box A/B of library variants with bench.py (MBRL_LIB picks the library; "" = the in-tree build)
for rep in 1 2; do
for v in "" tests/_build/lib_fixup_kernel.so tests/_build/lib_noredo.so; do
  MBRL_LIB=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-cpu --no-extra 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${v:-in-tree}', round(d['value']/1e6,3), round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
done; done

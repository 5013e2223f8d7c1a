# usage: bash tools/ab_envs2.sh <Env> <N> "<lib> ..."
for v in $3; do echo -n "$v $1 $2: "; MBRL_LIB=$v timeout -k 10 200 python3 tools/coherence_probe.py $1 $2 2>/dev/null | grep "^bench batch"; done

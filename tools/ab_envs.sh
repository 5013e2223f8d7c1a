# usage: bash tools/ab_envs.sh "<lib> <lib> ..."  -- plan / planned-kernel ms of Hopper, Walker2d, HalfCheetah per library variant ("-" = in-tree)
for v in $1; do
  lib=$v; [ "$v" = "-" ] && lib=""
  for e in "HopperEnv 163840" "Walker2dEnv 65536" "HalfCheetahEnv 163840"; do
    echo -n "$v $e: "; MBRL_LIB=$lib timeout -k 10 200 python3 tools/coherence_probe.py $e 2>/dev/null | grep "^bench batch"
  done
done

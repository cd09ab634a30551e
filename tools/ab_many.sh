# same-box A/B of two libraries over several (model, batch) configs: tools/ab_many.sh libA libB "model batch" ...
A=$1; B=$2; shift 2
for cfg in "$@"; do set -- $cfg; m=$1; b=$2
  for r in 1 2; do for lib in $A $B; do
    v=$(SRK_LIB_PATH=$PWD/$lib python bench.py --model $m --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "$m b$b $(basename $lib) $v"
  done; done
done

# same-box A/B of bench lines between libraries: tools/ab_libs.sh "<bench args>" lib1 lib2 ...
args="$1"; shift
for r in 1 2; do
for lib in "$@"; do
SRK_LIB_PATH=$PWD/$lib python bench.py $args --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', d['value'], (r.get('in_step') or {}).get('graph_us'), r.get('isolated', r).get('variants_us'))"
done; done
